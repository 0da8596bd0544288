# GPU box: bash tools/ab_galerkin.sh "ENV.." ...  -- durations of the Galerkin kernels of one refresh under each environment
for v in "$@"; do
  env $v TAG="$v" bash tools/probe_galerkin.sh | tail -1 | tee -a gpurun_out/galerkin_variants.log
done
