# GPU box: the driver's bench command twice in a row; the per-step Krylov counts must be identical (fixed summation orders, no atomics)
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --steady-max 0 --strict-steps 0 2>&1 >/dev/null | grep -E "step" | sed 's/.*step/step/' > gpurun_out/det_$i.txt; done
diff gpurun_out/det_1.txt gpurun_out/det_2.txt && echo IDENTICAL
