#!/bin/bash
# Step time against the internal batch size (rrtmg_lw_hip_set_batch) over the benchmark configurations and cloud / pressure fields:
# which is the smallest batch within 1 % of the best.   usage (GPU box): bash tools/batch_sweep.sh <tag>
TAG=${1:-batch_sweep}
O=gpurun_out/$TAG; mkdir -p $O
export RRTMG_LW_ALLOW_STANDIN=1
: > $O/table.txt
run() {   # name, bench args
  name=$1; shift
  for b in 32768 65536 131072 262144; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --host-cols 0 --steps 4 --warmup 1 --batch $b "$@" > $O/${name}_$b.json 2> $O/${name}_$b.err || exit 1
    python3 - $name $b $O/${name}_$b.json <<'PY' | tee -a $O/table.txt
import sys, json, ctypes
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], 'ms/step', d['ms_per_step'], 'workspace_GB', d.get('workspace_GB'))
PY
  done
}
run cloudy --config cloudy
run clear --config clear
run mcica5 --config cloudy --mcica 5
run aer137 --config aer_idrv --nlay 137 --ncol 500000
run cloudy_deep --config cloudy_deep
run cloudy_scatter --config cloudy_scatter
run cloudy_orography --config cloudy_orography
