#!/bin/bash
# scratch: run through gpurun
set -o pipefail
for v in 1 0 1 0; do
  echo "== flux aside $v"
  RRTMG_LW_FLUX_ASIDE=$v BENCH_ARGS="--check" bash tools/run_variants.sh exp/lib_fx.so 2>&1 | cut -c1-250 || exit 1
done
RRTMG_LW_FLUX_ASIDE=1 BENCH_ARGS="--config cloudy_deep" bash tools/run_variants.sh exp/lib_fx.so 2>&1 | cut -c1-120
RRTMG_LW_FLUX_ASIDE=0 BENCH_ARGS="--config cloudy_deep" bash tools/run_variants.sh exp/lib_fx.so 2>&1 | cut -c1-120
