#!/bin/bash
# scratch: run through gpurun
set -o pipefail
mkdir -p gpurun_out/stage
timeout -k 10 900 python -m pytest tests/test_fortran_shim.py tests/test_hip_parity.py tests/test_hip_mcica.py -m gpu -x -q > gpurun_out/stage/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/stage/pytest.log
[ $rc -ne 0 ] && exit $rc
for cfg in cloudy aer_idrv; do
  timeout -k 10 200 python tools/e2e_timing.py $cfg 131072 72 || exit 1
  timeout -k 10 200 python tools/e2e_timing.py $cfg 524288 72 || exit 1
done
timeout -k 10 200 python tools/e2e_timing.py aer_idrv 262144 137 || exit 1
