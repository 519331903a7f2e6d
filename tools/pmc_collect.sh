cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -f csv -- python3 tools/pmc_run.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -f csv -- python3 tools/pmc_run.py > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/sq1 -f csv -- python3 tools/pmc_run.py > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/sq2 -f csv -- python3 tools/pmc_run.py > $O/sq2.log 2>&1
find $O -name "*.csv" | head -20; tail -3 $O/*.log
