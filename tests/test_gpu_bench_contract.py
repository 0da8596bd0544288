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
    # round 3: needed-bytes pricing with the 8d figure beside it, whole-step utilisation, per-level times, fixed windows
    assert roof["algorithmic"]["bytes_per_launch"] >= roof["bytes_per_launch"] > 0
    assert 0 < roof["step"]["frac"] < 1 and roof["step"]["bytes"] > 0 and "coarse_levels" in roof
    assert roof["kernels"]["assemble"]["bound"] == "fp64 valu"
    assert set(d["windows"]) >= {"steps_1_8", "steps_5_24", "per_step"} and d["windows"]["steps_1_8"]["newton_its"] > 0
    assert d["env_overrides"] is None and d["assembly_passes"]["assemblies_full"] > 0
    # ... and the storage figures VERDICT r02 item 7 asked for beside the kernel times, the inexact-Newton rule in the config
    st = d["storage"]
    assert 0 <= st["sell_padding"] < 0.5 and 0.5 < st["col16_coverage"] <= 1 and st["sell_slots"] >= st["nnz"] > 0
    assert d["config"]["krylov_forcing"] == 0.1 and "linear_solves_forced" in d["assembly_passes"]


def test_bench_line_of_a_two_rank_launch_reports_every_rank():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` on one GPU (host-staged transport): the N > 1
    path of the contract -- mesh built on rank 0 only, start-up check, one JSON line from rank 0 with the wiring and the
    message rounds of EVERY rank (what the first real multi-GPU run prints)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c2_1m", "--transport", "gloo",
           "--steps", "3", "--warmup", "1", "--strict-steps", "0", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["transport"] == "gloo" and d["config"]["parallelism"] == "dd2" and d["value"] > 0
    assert [w["rank"] for w in d["ranks"]] == [0, 1] and all(w["n_ghost"] > 0 and w["neighbours"] for w in d["ranks"])
    assert sum(w["n_own"] for w in d["ranks"]) == d["config"]["dofs"]
    rounds = d["message_rounds_by_rank"]
    assert len(rounds) == 2 and rounds[0]["exchanges"] == rounds[1]["exchanges"] > 0
    # the padded all-reduces of rounds 1-2 are gone: scalars only; the replicated level travels by all-gathers
    assert rounds[0]["bytes_allreduced"] < 1e3 and rounds[0]["allgathers"] > 0
    assert "rank 1: device" in r.stderr and "rank 1: per Krylov iteration" in r.stderr
