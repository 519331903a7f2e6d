#!/bin/bash
# usage: tools/run_variants.sh lib1.so lib2.so ...   (paths relative to repo root) - runs the bench per library, prints ms/step and per-kernel ms
for lib in "$@"; do
    echo "== $lib $BENCH_ARGS"
    RRTMG_LW_HIP_LIB=$PWD/$lib RRTMG_LW_ALLOW_STANDIN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --host-cols 0 --steps ${STEPS:-5} --warmup 1 $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'],'value',d['value']); print({k:round(v,1) for k,v in d['path']['families'].items()}); print(d['path']['kernels'])
" || exit 1
done
