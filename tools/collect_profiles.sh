#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything profiles/ holds for the default bench.
#   bash tools/collect_profiles.sh pmc     -> the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) + JSON
#   bash tools/collect_profiles.sh bench   -> bench line, rocprofv3 --kernel-trace --stats summary, by-grid table
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
mkdir -p $out
if [ "$1" = "pmc" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- python3 tools/pmc_probe.py c4_10m > $out/pmc_$ctr.log 2>&1
    echo "pass $ctr done"
  done
  # third pass: what binds the assembly kernel (fp64 VALU issue): wave-instructions and the share of cycles its waves wait
  rm -rf $out/pmc_VALU
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_VALU -- python3 tools/pmc_probe.py c4_10m > $out/pmc_VALU.log 2>&1
  echo "pass VALU done"
  slots=$(grep -o "slots [0-9]*" $out/pmc_FETCH_SIZE.log | head -1 | cut -d" " -f2)
  f=$(ls $out/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1)
  w=$(ls $out/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1)
  v=$(ls $out/pmc_VALU/*/*counter_collection.csv | head -1)
  python tools/pmc_to_json.py $f $w $((8 * slots)) $out/pmc_traffic_c4_10m.json $v > $out/pmc_to_json.log
  cp $out/pmc_traffic_c4_10m.json profiles/pmc_traffic_c4_10m.json   # so that a bench leg of the same call finds it
  rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_VALU
  tail -5 $out/pmc_to_json.log
else
  python bench.py > $out/bench_default.json 2> $out/bench_default.err
  echo "bench done"
  python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench_steps20_warmup5.err
  echo "bench 20/5 done"
  rm -rf $out/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
  cp $(ls $out/prof/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
  python tools/trace_by_grid.py $(ls $out/prof/*/*kernel_trace.csv | head -1) > $out/kernel_trace_by_grid.txt
  rm -rf $out/prof
  head -12 $out/kernel_trace_by_grid.txt
fi
