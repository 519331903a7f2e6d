export RRTMG_LW_ALLOW_STANDIN=1
for l in lbot p1split lbot p1split; do echo "== $l"; RRTMG_LW_HIP_LIB=$PWD/exp/lib_$l.so timeout -k 10 300 python bench.py --check --no-cpu-baseline --host-cols 0 --steps 8 --warmup 2 2> gpurun_out/_err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'], {k:v for k,v in d['path']['kernels'].items() if 'sweepc<4' in k})
"; grep "check vs" gpurun_out/_err.txt; done
