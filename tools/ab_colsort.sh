# usage: bash tools/ab_colsort.sh   (GPU box) - the column order of k_colsort on / off / by threshold, four cloud fields
LIB=${LIB:-exp/lib_sort.so}
for cfg in cloudy cloudy_scatter cloudy_deep cloudy_towers; do
  echo "#### $cfg COLSORT=0"
  RRTMG_LW_COLSORT=0 BENCH_ARGS="--config $cfg" STEPS=5 bash tools/run_variants.sh $LIB || exit 1
  for min in ${MINS:-0 20 40 60}; do
    echo "#### $cfg COLSORT_MIN=$min"
    RRTMG_LW_COLSORT=1 RRTMG_LW_COLSORT_MIN=$min BENCH_ARGS="--config $cfg" STEPS=5 bash tools/run_variants.sh $LIB || exit 1
  done
done
echo "#### check"
for cfg in cloudy cloudy_deep cloudy_scatter; do
RRTMG_LW_COLSORT=1 RRTMG_LW_COLSORT_MIN=0 RRTMG_LW_HIP_LIB=$PWD/$LIB RRTMG_LW_ALLOW_STANDIN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --host-cols 0 --steps 2 --warmup 1 --check --config $cfg 2>&1 | grep "check vs"
done
