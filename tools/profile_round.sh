# Round-end measurement on the GPU box: bench lines, rocprofv3 kernel stats of the same command, PMC passes.
# usage: bash tools/profile_round.sh <tag>     (writes gpurun_out/<tag>/...)
TAG=${1:-round}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 bench.py --steps 5 --warmup 2 > $O/bench_1e6.json 2> $O/bench_1e6.err
python3 bench.py --steps 3 --warmup 1 --mcica 5 --no-cpu-baseline > $O/bench_1e6_mcica5.json 2> $O/bench_1e6_mcica5.err
python3 bench.py --steps 3 --warmup 1 --config aer_idrv --nlay 137 --ncol 500000 --no-cpu-baseline > $O/bench_5e5_aer137.json 2> $O/bench_aer137.err
python3 bench.py --steps 5 --warmup 2 --config clear --ncol 10000 --no-cpu-baseline > $O/bench_1e4_clear.json 2> $O/bench_1e4_clear.err
rocprofv3 --kernel-trace --stats -d $O/stats -f csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --host-cols 0 > $O/stats.log 2>&1
P=$O/pmc
mkdir -p $P
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -f csv -- python3 tools/pmc_run.py > $P/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -f csv -- python3 tools/pmc_run.py > $P/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $P/sq1 -f csv -- python3 tools/pmc_run.py > $P/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $P/sq2 -f csv -- python3 tools/pmc_run.py > $P/sq2.log 2>&1
ls $O $O/stats/* | head -30
