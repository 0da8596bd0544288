# usage: bash tools/rehearse_dd.sh NRANKS CONFIG "ENV=..." ...  -- several subdomains on ONE GPU through the gloo transport
n=$1; cfg=$2; shift 2
for envs in "$@"; do
  env $envs timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $n --config $cfg --transport gloo --steps 3 --warmup 1 --no-roofline --quiet > gpurun_out/dd.json 2> gpurun_out/dd.err || { echo "FAIL $envs"; tail -5 gpurun_out/dd.err; }
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/dd.json') if l.startswith('{')][-1]); print('$n ranks $cfg [$envs]', '%.3e'%d['value'], '%.1f ms/step'%d['ms_per_step'], d['config']['newton_its'], '%.1f its/newton'%d['config']['krylov_its_per_newton'])" | tee -a gpurun_out/rehearse_dd.log
done
