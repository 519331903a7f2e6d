# Round-end measurement on the GPU box: bench lines of the BASELINE configurations, rocprofv3 kernel stats of the bench command, PMC passes
# (HBM bytes in separate passes, instruction / stall / LDS counters) for the cloudy, clear and McICA shapes.
# usage: bash tools/profile_round.sh <tag>     (writes gpurun_out/<tag>/...)
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export RRTMG_LW_ALLOW_STANDIN=1
python3 bench.py --steps 10 --warmup 2 > $O/bench_1e6.json 2> $O/bench_1e6.err
echo "bench 1e6 done" ; tail -c 300 $O/bench_1e6.json
python3 bench.py --steps 5 --warmup 1 --mcica 5 --no-cpu-baseline --host-cols 0 > $O/bench_1e6_mcica5.json 2> $O/bench_1e6_mcica5.err
python3 bench.py --steps 5 --warmup 1 --config aer_idrv --nlay 137 --ncol 500000 --no-cpu-baseline --host-cols 0 > $O/bench_5e5_aer137.json 2> $O/bench_aer137.err
python3 bench.py --steps 20 --warmup 5 --config clear --ncol 10000 --no-cpu-baseline --host-cols 0 > $O/bench_1e4_clear.json 2> $O/bench_1e4_clear.err
python3 bench.py --steps 5 --warmup 1 --config clear --no-cpu-baseline --host-cols 0 > $O/bench_1e6_clear.json 2> $O/bench_1e6_clear.err
python3 bench.py --steps 10 --warmup 2 --ncol 125000 --no-cpu-baseline --host-cols 0 > $O/bench_125000_rank_proxy.json 2> $O/bench_125000.err
echo "benches done"
rocprofv3 --kernel-trace --stats -d $O/stats -f csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --host-cols 0 > $O/stats.log 2>&1
echo "stats done"
P=$O/pmc
mkdir -p $P
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -f csv -- python3 tools/pmc_run.py > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -f csv -- python3 tools/pmc_run.py > $P/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $P/sq1 -f csv -- python3 tools/pmc_run.py > $P/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $P/sq2 -f csv -- python3 tools/pmc_run.py > $P/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $P/sq3 -f csv -- python3 tools/pmc_run.py > $P/sq3.log 2>&1
python3 tools/pmc_summarize.py $P --md $O/pmc_cloudy.md --json $O/pmc_traffic.json > $O/pmc_summarize.log 2>&1
echo "pmc cloudy done"
for cfg in clear; do
  Q=$O/pmc_$cfg; mkdir -p $Q
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $Q/sq1 -f csv -- python3 tools/pmc_run.py --config $cfg > $Q/sq1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $Q/sq2 -f csv -- python3 tools/pmc_run.py --config $cfg > $Q/sq2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $Q/sq3 -f csv -- python3 tools/pmc_run.py --config $cfg > $Q/sq3.log 2>&1
  python3 tools/pmc_summarize.py $Q --md $O/pmc_$cfg.md > $Q/summarize.log 2>&1
done
echo "pmc clear done"
# McICA shape (generator + cldprmc + rtrnmc): HBM bytes and the instruction counters
Q=$O/pmc_mcica; mkdir -p $Q
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $Q/fetch -f csv -- python3 tools/pmc_run.py --mcica 5 > $Q/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $Q/write -f csv -- python3 tools/pmc_run.py --mcica 5 > $Q/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $Q/sq1 -f csv -- python3 tools/pmc_run.py --mcica 5 > $Q/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $Q/sq2 -f csv -- python3 tools/pmc_run.py --mcica 5 > $Q/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $Q/sq3 -f csv -- python3 tools/pmc_run.py --mcica 5 > $Q/sq3.log 2>&1
python3 tools/pmc_summarize.py $Q --md $O/pmc_mcica.md > $Q/summarize.log 2>&1
echo "pmc mcica done"
# keep what is merged back small: the raw counter CSVs are large
find $O -name "*counter_collection.csv" -size +4M -delete
ls $O | head -40
