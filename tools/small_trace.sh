#!/bin/bash
# usage (GPU box): bash tools/small_trace.sh <tag> <case e.g. cloudy:1024> [lib]  - kernel timeline of the last plain call and the last graph replay of a small device-resident call
TAG=$1; CASE=$2; LIB=${3:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export RRTMG_LW_ALLOW_STANDIN=1
[ -n "$LIB" ] && export RRTMG_LW_HIP_LIB=$R/$LIB
rocprofv3 --kernel-trace -d $O/trace -f csv -- python3 tools/small_graph.py --cases $CASE --reps 10 $SMALL_ARGS > $O/run.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rrlw" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void rrlw::", "").split("(")[0]))
rows.sort()
# calls = runs of kernels starting with k_colprep
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_colprep")]
def show(i0, i1, title):
    t0 = rows[i0][0]
    print(title)
    prev_end = t0
    for a, b, n in rows[i0:i1]:
        print(f"  {n[:34]:34s} start {(a - t0) / 1e3:8.1f} us  dur {(b - a) / 1e3:7.1f}  gap after previous end {(a - prev_end) / 1e3:6.1f}")
        prev_end = max(prev_end, b)
    print(f"  chain: {(max(r[1] for r in rows[i0:i1]) - t0) / 1e3:.1f} us")
n = len(starts)
half = n // 2
show(starts[half - 1], starts[half], "last call with plain launches")
show(starts[-1], len(rows), "last call replayed as a graph")
PY
