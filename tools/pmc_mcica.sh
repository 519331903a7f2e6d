cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mc0; mkdir -p $O; cd $R
export RRTMG_LW_ALLOW_STANDIN=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -f csv -- python3 tools/pmc_run.py --mcica 5 --ncol 262144 > $O/fetch.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/sq1 -f csv -- python3 tools/pmc_run.py --mcica 5 --ncol 262144 > $O/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/sq2 -f csv -- python3 tools/pmc_run.py --mcica 5 --ncol 262144 > $O/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/mc0"
for d in ("fetch","sq1","sq2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("FETCH_SIZE","SQ_WAVES","SQ_WAIT_ANY"): n[k]+=1
    for k in acc:
        if "subcol" in k or "layer" in k or "cloudmc" in k:
            print(d, k, n[k], {c: v/max(n[k],1) for c,v in acc[k].items()})
PY
find $O -name "*counter_collection.csv" -size +4M -delete
