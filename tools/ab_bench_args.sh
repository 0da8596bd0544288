# usage: bash tools/ab_bench_args.sh CONFIG "ARGS A" "ARGS B" ...  -- bench (20 steps after 5) once per argument string
# (e.g.  c4_10m "--forcing 0" "--forcing 0.1"): throughput, iteration counts and phase times side by side
cfg=$1; shift
mkdir -p gpurun_out
i=0
for a in "$@"; do i=$((i+1)); timeout -k 10 400 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --steady-max 0 --strict-steps 0 $a > gpurun_out/ab_args_$i.json 2> gpurun_out/ab_args_$i.err || tail -3 gpurun_out/ab_args_$i.err; python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[3], '[', sys.argv[2], ']', '%.4g DOF-updates/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], 'krylov', d['config']['krylov_its'], 'newton', d['config']['newton_its'], d['assembly_passes'])
w=d.get('windows') or {}
print('   per step (newton, krylov):', [(s['newton_its'], s['krylov_its']) for s in w.get('per_step', [])])
" gpurun_out/ab_args_$i.json "$a" $cfg; done
