#!/bin/bash
# scratch: run through gpurun
set -o pipefail
mkdir -p gpurun_out/stage
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/stage/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/stage/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/fortran_queue_timing.py > gpurun_out/stage/fortran_queue.md 2> gpurun_out/stage/fortran_queue.err || exit 1
cat gpurun_out/stage/fortran_queue.md
for n in 131072 524288; do timeout -k 10 200 python tools/e2e_timing.py cloudy $n 72 || exit 1; done
timeout -k 10 200 python tools/e2e_timing.py aer_idrv 131072 72 || exit 1
