export RRTMG_LW_ALLOW_STANDIN=1
mkdir -p gpurun_out/r5_quick
for args in "--config clear" "--config cloudy" "--config cloudy --mcica 5" "--config aer_idrv --nlay 137 --ncol 500000" "--config clear --ncol 10000"; do
  python3 bench.py --no-cpu-baseline --host-cols 0 --steps 5 --warmup 2 $args 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$args', 'ms/step',d['ms_per_step'], {k:round(v,2) for k,v in d['path']['families'].items()})
"
done
