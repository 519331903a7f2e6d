#!/bin/bash
# usage: tools/run_variants.sh lib1.so lib2.so ...   (paths relative to repo root) - runs the bench (no overlap + overlap) per library
for lib in "$@"; do
  for extra in "" "--overlap"; do
    echo "== $lib $extra"
    RRTMG_LW_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --host-cols 0 --steps 3 --warmup 1 $extra $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'],'value',d['value']); print({k:round(v,1) for k,v in d['path']['families'].items()})
" || exit 1
  done
done
