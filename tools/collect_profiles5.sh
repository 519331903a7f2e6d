#!/bin/bash
# usage: tools/collect_profiles5.sh <tag>   - copies the judged summaries of gpurun_out/<tag>/ (tools/profile_round5.sh) into profiles/round5_*
tag=$1; cd "$(dirname "$0")/.."; O=gpurun_out/$tag
for n in 1e6 1e6_clear 1e6_mcica5 5e5_aer137 1e4_clear 125000_rank_proxy 1e6_cloudy_towers 1e6_cloudy_scatter 1e6_cloudy_deep 1e6_cloudy_orography 1e6_torchrun1; do [ -s $O/bench_$n.json ] && tail -1 $O/bench_$n.json > profiles/round5_bench_$n.json; done
cp $O/kernel_stats.csv profiles/round5_kernel_stats.csv
for k in cloudy_L72 clear_L72 cloudy_L72_mcica5 aer_idrv_L137 cloudy_deep_L72 cloudy_orography_L72; do [ -f $O/pmc_$k.md ] && cp $O/pmc_$k.md profiles/round5_pmc_$k.md; done
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cp $O/pmc_compute.json profiles/pmc_compute.json
