#!/bin/bash
# scratch: run through gpurun
set -o pipefail
mkdir -p gpurun_out/stage
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/stage/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/stage/pytest.log
[ $rc -ne 0 ] && exit $rc
for n in 131072 524288; do timeout -k 10 200 python tools/e2e_timing.py cloudy $n 72 || exit 1; done
timeout -k 10 200 python tools/e2e_timing.py aer_idrv 131072 72 || exit 1
RRTMG_LW_STAGE_TIMING=1 timeout -k 10 200 python tools/e2e_timing.py cloudy 524288 72 2>&1 | tail -3
