# usage: bash tools/sweep_env.sh "VAR=a VAR2=b" "VAR=c" ...  (GPU box): bench both sizes under each environment
for envs in "$@"; do
  for cfg in c2_1m c4_10m; do
    env $envs timeout -k 10 120 python bench.py --config $cfg --steps 3 --no-cpu-baseline --no-roofline --quiet > gpurun_out/sw.json 2>gpurun_out/sw.err || echo fail
    python -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$envs | $cfg', '%.3e'%d['value'], '%.1f'%d['ms_per_step'], d['config']['newton_its'], '%.1f'%d['config']['krylov_its_per_newton'], flush=True)" | tee -a gpurun_out/sweep_env.log
  done
done
