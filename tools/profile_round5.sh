# Round-end measurement on the GPU box (round 5): PMC passes per configuration first (HBM bytes in separate passes, instruction / stall / LDS
# counters) -> profiles/pmc_traffic.json / pmc_compute.json of THIS run, then the bench lines of the BASELINE configurations and the cloud-field
# variants (their `traffic` / `compute` objects read those files), the rocprofv3 kernel stats of the bench command, the one-rank torchrun
# rehearsal of the sharded step.     usage: bash tools/profile_round5.sh <tag> [quick]     (writes gpurun_out/<tag>/...)
TAG=${1:-round5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export RRTMG_LW_ALLOW_STANDIN=1
B="--no-cpu-baseline --host-cols 0"
pmc() {   # pmc <key> <columns> <pmc_run.py flags...>
  key=$1; cols=$2; shift 2
  P=$O/pmc_$key; mkdir -p $P
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -f csv -- python3 tools/pmc_run.py --ncol $cols "$@" > $P/fetch.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -f csv -- python3 tools/pmc_run.py --ncol $cols "$@" > $P/write.log 2>&1 || return 1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $P/sq1 -f csv -- python3 tools/pmc_run.py --ncol $cols "$@" > $P/sq1.log 2>&1 || return 1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $P/sq2 -f csv -- python3 tools/pmc_run.py --ncol $cols "$@" > $P/sq2.log 2>&1 || return 1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $P/sq3 -f csv -- python3 tools/pmc_run.py --ncol $cols "$@" > $P/sq3.log 2>&1 || return 1
  python3 tools/pmc_summarize.py $P --md $O/pmc_$key.md --json $O/pmc_traffic.json --key $key --columns $cols > $P/summarize.log 2>&1 || return 1
  python3 tools/pmc_to_compute.py $O/pmc_$key.md $key $cols --out $O/pmc_compute.json > $P/compute.log 2>&1
  find $P -name "*counter_collection.csv" -size +4M -delete
  echo "pmc $key done"
}
pmc cloudy_L72 250000 || exit 1
if [ "$2" != "quick" ]; then
  pmc clear_L72 250000 --config clear || exit 1
  pmc cloudy_L72_mcica5 250000 --mcica 5 || exit 1
  pmc aer_idrv_L137 125000 --config aer_idrv --nlay 137 || exit 1
  pmc cloudy_deep_L72 250000 --config cloudy_deep || exit 1
  pmc cloudy_orography_L72 250000 --config cloudy_orography || exit 1
fi
# the bench lines below take `traffic` and `compute` from the passes just made
cp $O/pmc_traffic.json profiles/pmc_traffic.json; cp $O/pmc_compute.json profiles/pmc_compute.json
python3 bench.py --steps 10 --warmup 2 > $O/bench_1e6.json 2> $O/bench_1e6.err || exit 1
echo "bench 1e6 done"; tail -c 300 $O/bench_1e6.json
python3 bench.py --steps 5 --warmup 1 --mcica 5 $B > $O/bench_1e6_mcica5.json 2> $O/bench_1e6_mcica5.err || exit 1
python3 bench.py --steps 5 --warmup 1 --config aer_idrv --nlay 137 --ncol 500000 $B > $O/bench_5e5_aer137.json 2> $O/bench_aer137.err || exit 1
python3 bench.py --steps 20 --warmup 5 --config clear --ncol 10000 $B > $O/bench_1e4_clear.json 2> $O/bench_1e4_clear.err || exit 1
python3 bench.py --steps 5 --warmup 1 --config clear $B > $O/bench_1e6_clear.json 2> $O/bench_1e6_clear.err || exit 1
python3 bench.py --steps 10 --warmup 2 --ncol 125000 $B > $O/bench_125000_rank_proxy.json 2> $O/bench_125000.err || exit 1
for cfg in cloudy_towers cloudy_scatter cloudy_deep cloudy_orography; do python3 bench.py --config $cfg --steps 5 --warmup 1 $B > $O/bench_1e6_$cfg.json 2> $O/bench_$cfg.err || exit 1; done
echo "benches done"
# the driver's multi-GPU command with one rank: RCCL initialised, the all-gather of the packed outputs in the timed region
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 --force-gather $B > $O/bench_1e6_torchrun1.json 2> $O/bench_torchrun1.err || echo "torchrun rehearsal failed"
rocprofv3 --kernel-trace --stats -d $O/stats -f csv -- python3 bench.py --steps 5 --warmup 2 $B > $O/stats.log 2>&1 || exit 1
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv
echo "stats done"
ls $O | head -60
