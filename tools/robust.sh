# usage: bash tools/robust.sh "ENV=.. | bench args" ...   (GPU box)
for spec in "$@"; do
  envs="${spec%%|*}"; a="${spec#*|}"
  env $envs timeout -k 10 400 python bench.py $a --no-cpu-baseline --no-roofline > gpurun_out/rb.json 2> gpurun_out/rb.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "FAIL [$envs] $a"; grep -E "timed step|warmup|Error" gpurun_out/rb.err | tail -4; else python -c "
import json
d=json.loads(open('gpurun_out/rb.json').read().strip().splitlines()[-1]); print('[$envs] $a', '%.3e'%d['value'], '%.1f'%d['ms_per_step'], d['config']['newton_its'], '%.1f'%d['config']['krylov_its_per_newton'])"; fi
done
