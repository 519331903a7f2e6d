#!/bin/bash
# the five BASELINE.json configurations (plus the overlap switch) with the shipped library; one JSON line each into gpurun_out/<tag>_*.json
tag=${1:-r2}
run() { name=$1; shift; RRTMG_LW_ALLOW_STANDIN=1 timeout -k 10 400 python bench.py --no-cpu-baseline --host-cols 0 --steps 5 --warmup 1 "$@" 2>/dev/null | tail -1 > gpurun_out/${tag}_$name.json
  python - "$name" gpurun_out/${tag}_$name.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[2]).read())
print(sys.argv[1], 'ms/step', d['ms_per_step'], 'Mcol/s', round(d['value'] / 1e6, 2), d['path']['families'])
PY
}
run cloudy_1e6
run cloudy_1e6_overlap --overlap
run clear_1e6 --config clear
run clear_1e4 --config clear --ncol 10000 --steps 50 --warmup 5
run mcica5_1e6 --mcica 5
run aer137_5e5 --config aer_idrv --nlay 137 --ncol 500000
