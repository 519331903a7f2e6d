# usage: bash tools/kernel_alone.sh <tag> [pmc_run.py flags]   - rocprofv3 per-kernel averages of one call of the hot path (kernels serialised by the profiler)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export RRTMG_LW_ALLOW_STANDIN=1
rocprofv3 --kernel-trace --stats -d $O/stats -f csv -- python3 tools/pmc_run.py "$@" > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rrlw" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
