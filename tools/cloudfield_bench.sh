# Step time against the vertical extent of the cloud field (synth.py: cloudy / cloudy_towers / cloudy_scatter / cloudy_deep), 1e6 columns x 72 layers:
# one bench line each, then rocprofv3 kernel stats of the towers and the deep shape.   usage: bash tools/cloudfield_bench.sh <tag>
TAG=${1:-cloudfield}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export RRTMG_LW_ALLOW_STANDIN=1
for cfg in cloudy cloudy_towers cloudy_scatter cloudy_deep; do
  timeout -k 10 400 python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --host-cols 0 > $O/bench_$cfg.json 2> $O/bench_$cfg.err || exit 1
  python3 - $cfg $O/bench_$cfg.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', d['ms_per_step'], 'Mcol/s', round(d['value'] / 1e6, 2), d['path']['families'])
PY
done
for cfg in cloudy_towers cloudy_deep; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$cfg -f csv -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --host-cols 0 > $O/stats_$cfg.log 2>&1 || exit 1
  cp $O/stats_$cfg/*/*_kernel_stats.csv $O/kernel_stats_$cfg.csv
done
ls $O
