# usage: bash tools/ab_switch.sh CONFIG SWITCH VALUE...  -- bench (20 steps after 5) once per VALUE of the library switch SWITCH
# (csrc/shk_tunables.h; e.g.  c4_10m SHK_AMG_BF16_ROWS 0 500000): throughput, iteration counts and phase times side by side
cfg=$1; sw=$2; shift; shift
mkdir -p gpurun_out
for v in "$@"; do env $sw=$v timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --steady-max 0 --strict-steps 0 > gpurun_out/ab_${sw}_$v.json 2> gpurun_out/ab_${sw}_$v.err || tail -3 gpurun_out/ab_${sw}_$v.err; python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[3], sys.argv[4], '=', sys.argv[2], '%.4g DOF-updates/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], 'krylov', d['config']['krylov_its'], 'newton', d['config']['newton_its'])
p=d['roofline']['phase_ms']; print('   profiled step', d['roofline']['profiled_step'], {k:round(v,2) for k,v in p.items() if v})
g=d['roofline']['phase_gbs']; print('   GB/s', {k:round(v) for k,v in g.items() if v})
" gpurun_out/ab_${sw}_$v.json $v $cfg $sw; done
