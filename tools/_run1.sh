export RRTMG_LW_ALLOW_STANDIN=1
O=gpurun_out/code_bits.md; : > $O
echo "| bits per cell code | case | max abs d flux, W m-2 | max abs d heating rate, K d-1 | max d heating rate / max(abs rate, 1) |" >> $O
echo "|---|---|---|---|---|" >> $O
python tools/code_bits_table.py 32 >> $O 2>gpurun_out/_e32.txt
RRTMG_LW_HIP_LIB=$PWD/exp/lib_full_code24.so python tools/code_bits_table.py 24 >> $O 2>gpurun_out/_e24.txt
RRTMG_LW_HIP_LIB=$PWD/exp/lib_full_code16.so python tools/code_bits_table.py 16 >> $O 2>gpurun_out/_e16.txt
echo >> $O; echo "| bits | configuration | ms per step | k_layer | k_sweepc | k_sweepz |" >> $O; echo "|---|---|---|---|---|---|" >> $O
for b in 32 24 16; do
  L=$PWD/exp/lib_full_code$b.so; [ $b = 32 ] && L=$PWD/rrtmg_lw_amd/librrtmg_lw_hip.so
  for c in "cloudy:" "clear:--config clear" "mcica5:--mcica 5" "aer137_5e5:--config aer_idrv --nlay 137 --ncol 500000" "clear_1e4:--config clear --ncol 10000 --steps 50"; do
    n=${c%%:*}; f=${c#*:}
    RRTMG_LW_HIP_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --host-cols 0 --steps 5 --warmup 1 $f 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); f=d['path']['families']; print('| $b | $n |', d['ms_per_step'], '|', f['k_layer'], '|', f['k_sweepc'], '|', f['k_sweepz'], '|')
" >> $O
  done
done
cat $O
