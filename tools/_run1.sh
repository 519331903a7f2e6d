#!/bin/bash
# scratch: run through gpurun
set -o pipefail
for n in 4096 8192 16384 32768 65536 131072; do
  for cfg in cloudy clear; do
    timeout -k 10 200 python bench.py --ncol $n --config $cfg --steps 50 --warmup 5 --no-cpu-baseline --host-cols 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', $n, 'ms/step', d['ms_per_step'], 'Mcol/s', round(d['value']/1e6,2), {k:round(v,2) for k,v in d['path']['families'].items()})
" || exit 1
  done
done
