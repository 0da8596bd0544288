# usage: bash tools/ab_fused_sweeps.sh CONFIG ROWS...  -- bench (20 steps after 5) with the fused four-sweep smoother on levels of at most ROWS rows (0 = off)
cfg=$1; shift
for fr in "$@"; do SHK_AMG_FUSED_ROWS=$fr timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --steady-max 0 --strict-steps 0 > gpurun_out/b_tb_$fr.json 2> gpurun_out/b_tb_$fr.err || tail -3 gpurun_out/b_tb_$fr.err; python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[3], 'fused rows <=', sys.argv[2], '%.4g DOF-updates/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], 'krylov', d['config']['krylov_its'], 'newton', d['config']['newton_its'])
p=d['roofline']['phase_ms']; print('   profiled step', d['roofline']['profiled_step'], {k:round(v,2) for k,v in p.items() if v})
" gpurun_out/b_tb_$fr.json $fr $cfg; done
