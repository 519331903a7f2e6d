# usage: bash tools/call_timeline.sh <tag> [pmc_run.py flags]  - kernel timeline of the last call of tools/pmc_run.py (start offsets, durations, gaps)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export RRTMG_LW_ALLOW_STANDIN=1
rocprofv3 --kernel-trace -d $O/trace -f csv -- python3 tools/pmc_run.py "$@" > $O/trace.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rrlw" in r["Kernel_Name"] and "calibrate" not in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rrlw::","")))
rows.sort()
# the last call = the last half of the kernels
n=len(rows)//2; rows=rows[n:]
t0=rows[0][0]; busy=0; last_end=t0
for a,b,k in rows:
    print(f"{(a-t0)/1e3:9.1f} us  +{(b-a)/1e3:8.1f} us  gap {(a-last_end)/1e3:7.1f}  {k[:50]}")
    last_end=max(last_end,b); busy+=b-a
print(f"span {(last_end-t0)/1e3:.1f} us, sum of kernel durations {busy/1e3:.1f} us, {len(rows)} kernels")
PY
