"""bench.py's output contract on the GPU (small configuration): ONE JSON line on stdout with the metric of BASELINE.json,
the `roofline` and `cpu_baseline` objects, the transport field and the strict-linear-solve leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_every_contract_field():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1_12k", "--steps", "3", "--warmup", "1",
                        "--cpu-sample", "71x71", "--steady-max", "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # exactly one JSON line
    d = json.loads(lines[0])
    assert d["metric"] == "DOF-updates/s per Newton iter (assembly+linear solve)" and d["unit"] == "DOF-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - d["config"]["dofs"] * d["config"]["newton_its"] / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["transport"].startswith("none")
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["traffic"] is None and "traffic_source" in roof      # no counter file for this configuration
    whole = roof["whole_iterations"]                                 # whole Krylov / Newton iterations
    assert whole["krylov_iteration"]["algorithmic_bytes"] == 416 * d["config"]["dofs"]
    assert 0 < whole["krylov_iteration"]["frac"] < 1 and 0 < whole["newton_iteration"]["frac"] < 1
    assert d["config"]["krylov_warm_start"] == 4 and d["config"]["krylov_newton_eta"] == 0.1
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and set(cpu["legs"]) >= {"sample_lu", "c1_5k_lu", "c1_5k_bicgstab"}
    assert d["strict_linear_solves"]["krylov_its_per_newton"] >= d["config"]["krylov_its_per_newton"]
    assert d["steady_state"]["ended_by"] in ("tolerance", "step limit")
