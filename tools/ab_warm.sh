# usage (GPU box): bash tools/ab_warm.sh "ENV=.. ENV2=.." ...   -- the default bench at 10M and 1M DOF under each environment
for envs in "$@"; do for cfg in c4_10m c2_1m; do
env $envs python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --steady-max 0 --strict-steps 0 2> gpurun_out/warm_last.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$envs | $cfg %.3e %.1f ms/step newton %d krylov %d' % (d['value'], d['ms_per_step'], d['config']['newton_its'], d['config']['krylov_its']), flush=True)" | tee -a gpurun_out/warm_ab.log
done; done
