#!/bin/bash
# the four BASELINE configurations that fit one GPU (device-resident rate), one line each: tools/bench_configs.sh [extra bench flags]
run() { python bench.py --no-cpu-baseline --host-cols 0 --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-70s %8.2f ms  %6.2f M col/s' % (d['config']['workload'][:70], d['ms_per_step'], d['value']/1e6)); print('   ', {k:round(v,1) for k,v in d['path']['families'].items() if v})
"; }
run --ncol 10000 --config clear "$@"
run --ncol 1000000 --config clear "$@"
run --ncol 1000000 --config cloudy "$@"
run --ncol 1000000 --config cloudy --mcica 5 "$@"
run --ncol 500000 --nlay 137 --config aer_idrv "$@"
