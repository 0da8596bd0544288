set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gaps; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python3 bench.py --config ${CONFIG:-c4_10m} --no-cpu-baseline --no-roofline --steady-max 0 --strict-steps 0 --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $out/bench.json 2> $out/bench.err
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $f k_update_b:${WARMUP:-5} > $out/gaps_timed.txt
cat $out/gaps_timed.txt
gzip -c $f > $out/kernel_trace.csv.gz
tail -2 $out/bench.err
rm -rf $out/prof
