cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ic
mkdir -p $O
cd $R
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/ic -f csv -- python3 tools/pmc_run.py --ncol 65536 > $O/ic.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQC_DCACHE_MISSES SQC_DCACHE_REQ --kernel-trace -d $O/lvl -f csv -- python3 tools/pmc_run.py --ncol 65536 > $O/lvl.log 2>&1
tail -n 2 $O/*.log
