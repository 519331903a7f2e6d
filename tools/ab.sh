#!/bin/bash
# usage (on the GPU box): bash tools/ab.sh tag lib1.so lib2.so ...   - bench (cloudy 1e6, --check) per tuning library, round-robin twice; results -> gpurun_out/<tag>.txt
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/$tag.txt
: > $out
for rep in 1 2; do
for lib in "$@"; do
    echo "== $lib rep $rep $BENCH_ARGS" | tee -a $out
    RRTMG_LW_HIP_LIB=$PWD/$lib RRTMG_LW_ALLOW_STANDIN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --host-cols 0 --steps ${STEPS:-6} --warmup 2 --check $BENCH_ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('# check'): print(l.strip())
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'],'value',d['value']); print({k:round(v,2) for k,v in d['path']['families'].items()}); print(d['path']['kernels'])
" | tee -a $out || exit 1
done
done
