# GPU box: durations of the Galerkin refresh kernels of one hierarchy refresh at 10M DOF (from a rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/galerkin; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-roofline --steady-max 0 --strict-steps 0 --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err
f=$(ls $out/prof/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "k_galerkin" in n:
        rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3, int(r.get("Grid_Size_X", r.get("Grid_Size",0)))))
rows.sort()
seq=rows[-12:]
import os; print(os.environ.get("TAG",""), "last refresh:", " ".join(f"{d:.0f}" for _,d,_ in seq), "us; sum %.0f us" % sum(d for _,d,_ in seq))
PY
rm -rf $out/prof
