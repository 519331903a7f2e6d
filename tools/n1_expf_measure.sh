#!/bin/bash
# Measurement of the two kernel variants BASELINE.json's north_star names:
#   - north-star mapping prototype k_n1 (bench.py --n1-prototype = rrtmg_lw_hip_set_n1_prototype, cloud-free calls): correctness by its GPU test, then ms per 1e6 clear columns
#   - exp2f / rcp transmittance instead of the table in LDS (exp/lib_expf*.so): ms per 1e6 columns and max |d| vs the oracle (bench --check)
O=gpurun_out/${1:-r2_n1expf}.log
: > $O
export RRTMG_LW_ALLOW_STANDIN=1
echo "== k_n1 correctness" >> $O
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k north_star >> $O 2>&1
b() { timeout -k 10 300 python bench.py --no-cpu-baseline --host-cols 0 --steps 5 --warmup 1 --check "$@" 2>> $O | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',d['ms_per_step'],'Mcol/s',round(d['value']/1e6,2),d['path']['kernels'])
" >> $O; }
echo "== clear 1e6, production mapping" >> $O; b --config clear
echo "== clear 1e6, k_n1 (one column per wavefront, table in LDS)" >> $O; b --config clear --n1-prototype
for v in expf2 expf3; do
  echo "== $v: clear 1e6" >> $O; RRTMG_LW_HIP_LIB=$PWD/exp/lib_$v.so b --config clear
  echo "== $v: cloudy 1e6" >> $O; RRTMG_LW_HIP_LIB=$PWD/exp/lib_$v.so b
  echo "== $v: clear 1e6 k_n1 + exp2f" >> $O; RRTMG_LW_HIP_LIB=$PWD/exp/lib_$v.so b --config clear --n1-prototype
done
echo "== cloudy 1e6, production" >> $O; b
cat $O
