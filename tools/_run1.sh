#!/bin/bash
# scratch: run through gpurun
set -o pipefail
mkdir -p gpurun_out/stage
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/stage/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/stage/pytest.log
exit $rc
