# instruction / stall counters only (no HBM byte passes): tools/pmc_quick.sh [outdir-name]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmcq}
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/sq1 -f csv -- python3 tools/pmc_run.py --ncol 131072 $PMC_ARGS > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/sq2 -f csv -- python3 tools/pmc_run.py --ncol 131072 $PMC_ARGS > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace -d $O/sq3 -f csv -- python3 tools/pmc_run.py --ncol 131072 $PMC_ARGS > $O/sq3.log 2>&1
python3 tools/pmc_summarize.py $O --md $O/summary.md > $O/summarize.log 2>&1
tail -2 $O/*.log
