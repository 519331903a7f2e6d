export RRTMG_LW_ALLOW_STANDIN=1
STEPS=5 bash tools/run_variants.sh exp/lib_tune_ldsbar.so exp/lib_k6.so exp/lib_k16.so 2>&1 | grep -E "^==|ms/step|k_layer<"  | sed -e "s/'k_sweepz.*//"
