# usage: bash tools/bench_legs.sh tag [configs]  (GPU box) -- bench, prints headline + per-launch legs
for cfg in ${2:-c2_1m c4_10m}; do
  timeout -k 10 200 python bench.py --config $cfg --steps 3 --no-cpu-baseline --quiet > gpurun_out/legs_$1_$cfg.json 2> gpurun_out/legs_$1_$cfg.err || echo fail
  python -c "
import json
d=json.loads(open('gpurun_out/legs_$1_$cfg.json').read().strip().splitlines()[-1])
print('$1 $cfg', '%.3e'%d['value'], '%.1f ms/step'%d['ms_per_step'], 'its/newton %.1f'%d['config']['krylov_its_per_newton'])
print('   ', {k:round(v['avg_launch_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()}, {k:round(v,1) for k,v in d['roofline']['phase_ms'].items()})" | tee -a gpurun_out/legs.log
done
