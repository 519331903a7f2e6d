export RRTMG_LW_ALLOW_STANDIN=1
for l in cur fan cur fan cur fan; do echo "== $l"; RRTMG_LW_HIP_LIB=$PWD/exp/lib_$l.so timeout -k 10 300 python bench.py --no-cpu-baseline --host-cols 0 --steps 10 --warmup 2 2> gpurun_out/_err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'])
"; done
for cfg in cloudy_deep; do for l in cur fan; do echo "== $l $cfg"; RRTMG_LW_HIP_LIB=$PWD/exp/lib_$l.so timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --host-cols 0 --steps 5 --warmup 1 2> gpurun_out/_err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'])
"; done; done
