# usage: bash tools/sweep_smoother.sh "w1 w2 alpha" ...   (GPU box; appends to gpurun_out/sweep_w.log)
for w in "$@"; do
  set -- $w
  for cfg in c2_1m c4_10m; do
    SHK_AMG_W1=$1 SHK_AMG_W2=$2 SHK_AMG_ALPHA=$3 timeout -k 10 120 python bench.py --config $cfg --steps 3 --no-cpu-baseline --no-roofline --quiet > gpurun_out/sw.json 2>gpurun_out/sw.err || echo fail
    python -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$1 $2 $3 $cfg', '%.3e'%d['value'], '%.1f'%d['ms_per_step'], d['config']['newton_its'], '%.1f'%d['config']['krylov_its_per_newton'], flush=True)" | tee -a gpurun_out/sweep_w.log
  done
done
