#!/bin/bash
# usage: tools/collect_profiles.sh <tag>   - copies the judged summaries of gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/round2_*
tag=$1; cd "$(dirname "$0")/.."; O=gpurun_out/$tag
for n in 1e6 1e6_clear 1e6_mcica5 5e5_aer137 1e4_clear 125000_rank_proxy; do tail -1 $O/bench_$n.json > profiles/round2_bench_$n.json; done
cp $O/stats/*/*_kernel_stats.csv profiles/round2_kernel_stats.csv
cp $O/pmc_cloudy.md profiles/round2_pmc_cloudy.md
cp $O/pmc_clear.md profiles/round2_pmc_clear.md
[ -f $O/pmc_mcica.md ] && cp $O/pmc_mcica.md profiles/round2_pmc_mcica.md
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python3 tools/pmc_to_compute.py $O/pmc_cloudy.md cloudy_L72 250000
python3 tools/pmc_to_compute.py $O/pmc_clear.md clear_L72 250000
