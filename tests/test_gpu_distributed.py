"""Domain-decomposed solve == undecomposed solve, through the real subdomain contexts (pytest -m gpu).

The one-GPU box cannot host several RCCL ranks, so the multi-rank case runs 2 and 3 processes on the
same GPU with the host-staged gloo transport: identical kernels, halo plan, reduction layout and solver
control flow as the RCCL path; only the byte transport differs.  RCCL itself is exercised with a
one-rank communicator (create id, init, all-reduce inside a solve)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, transport, port, extra=(), env_extra=None):
    # The library's DEFAULT schedule (serialised exchanges, interior / boundary overlap off) is what the first real
    # multi-GPU run takes, so it is what every decomposition below runs; the overlapped schedule (SHK_OVERLAP=1) is an
    # extra parametrisation of the tests that name it.
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SHK_OVERLAP", None)
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"),
           "--transport", transport, *extra]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("nproc,precond", [(2, "jacobi"), (3, "jacobi"), (3, "amg"), (2, "amg_local")])
def test_partitioned_matches_single_gloo(nproc, precond):
    r = _launch(nproc, "gloo", 29511 + nproc + {"jacobi": 0, "amg": 10, "amg_local": 20}[precond], ("--precond", precond))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"] and rep["ghost_mismatch"] == 0.0
    assert max(rep["errs"].values()) < 1e-7


def test_partitioned_matches_single_at_1m_dof():
    """BASELINE.json's 1M-DOF mesh (100 km x 20 km) split over 2 subdomains: fields after 3 steps equal the
    undecomposed run's to 1e-7, same Newton iteration counts, Krylov counts within the distributed hierarchy's
    margin, ghost copies identical to their owners."""
    r = _launch(2, "gloo", 29561, ("--precond", "amg", "--nx", "2236", "--ny", "447", "--lx", "100e3", "--ly", "20e3"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"] and rep["ghost_mismatch"] == 0.0 and max(rep["errs"].values()) < 1e-7


def test_interior_boundary_overlap_changes_nothing_but_the_schedule():
    """The interior / boundary split of the finest level's sweeps (shk_comm_overlap, SHK_OVERLAP=1) against the serialised exchange (SHK_OVERLAP=0) on a 125k-DOF mesh over 3 subdomains: both match the
    undecomposed run to 1e-7 with its Newton counts, and their Krylov counts agree to within a few iterations (the split only
    changes the order in which partial sums of the dot products are added)."""
    reps = []
    for k, flag in enumerate(("1", "0")):
        r = _launch(3, "gloo", 29591 + k, ("--precond", "amg", "--nx", "500", "--ny", "250"), {"SHK_OVERLAP": flag})
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        reps.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
        assert reps[-1]["ok"] and reps[-1]["ghost_mismatch"] == 0.0 and max(reps[-1]["errs"].values()) < 1e-7
    on, off = reps
    assert on["overlap"]["active"] and not off["overlap"]["active"]
    assert on["overlap"]["boundary_slices"] < 0.25 * on["overlap"]["slices"]
    assert [x[0] for x in on["infos"]] == [x[0] for x in off["infos"]]
    # (the summation order of the dot products differs; BiCGStab amplifies that to a few iterations either way)
    assert all(abs(x[1] - y[1]) <= max(4, 0.25 * y[1]) for x, y in zip(on["infos"], off["infos"])), (on["infos"], off["infos"])


@pytest.mark.parametrize("rep_rows,overlap,bf16_rows", [("30000", "0", None), ("8000", "0", None), ("30000", "1", None),
                                                        ("8000", "0", "1"), ("30000", "1", "1")])
def test_replicated_coarse_levels_four_subdomains(rep_rows, overlap, bf16_rows):
    """The gathered-and-replicated coarse part of the distributed multigrid on an 80k-DOF mesh
    split four ways, taking over at level 1 (20k global rows) and at level 2 (5k): same fields as the undecomposed run,
    same Newton counts, Krylov counts at the one-subdomain level.  (Four subdomains make the replicated level larger
    than a subdomain's own share of it, the case that needs the reduction partials cleared.)
    bf16_rows = "1": every sparse level of the decomposed part AND of the replicated part smooths on its packed bfloat16
    copy (DevSell::pk; by default only levels of >= 500k rows per subdomain do): the frozen-ghost first sweep, the
    interior / boundary passes and the transfer onto the global level (whose rows' own slot is a GLOBAL column) in their
    packed instances."""
    env = {"SHK_AMG_REP_ROWS": rep_rows, "SHK_OVERLAP": overlap}
    if bf16_rows:
        env["SHK_AMG_BF16_ROWS"] = bf16_rows
    r = _launch(4, "gloo", 29565 + int(rep_rows) // 8000 + 10 * int(overlap) + (20 if bf16_rows else 0),
                ("--precond", "amg", "--nx", "400", "--ny", "200"), env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"] and rep["ghost_mismatch"] == 0.0 and max(rep["errs"].values()) < 1e-7


def test_partitioned_matches_single_at_1m_dof_four_subdomains():
    """Four subdomains on the 1M-DOF mesh (the GPU box admits 6 processes on its card: the test runner, four ranks and
    a margin; the 8-way split of BASELINE config 4 itself needs an 8-GPU node, and tests/test_partition.py runs the
    8-rank plan and transport on the CPU): the distributed hierarchy with its replicated coarse part, 2-D blocks of the
    bisection with several neighbours, the all-reduced scalars."""
    r = _launch(4, "gloo", 29571, ("--precond", "amg", "--nx", "2236", "--ny", "447", "--lx", "100e3", "--ly", "20e3",
                                   "--steps", "2"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"] and rep["ghost_mismatch"] == 0.0 and max(rep["errs"].values()) < 1e-7


@pytest.mark.parametrize("precond,overlap", [("amg", "0"), ("jacobi", "0"), ("amg", "1")])
def test_partitioned_matches_single_on_the_unstructured_basin_mesh(precond, overlap):
    """Recursive bisection of the Delaunay basin mesh (hole, curved outlet, valence-13 vertices, random vertex order):
    irregular neighbour sets and halo lists, the distributed hierarchy on irregular aggregates; three subdomains."""
    r = _launch(3, "gloo", (29585 if precond == "amg" else 29586) + 2 * int(overlap), ("--precond", precond, "--basin", "30000"),
                {"SHK_OVERLAP": overlap})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"] and rep["ghost_mismatch"] == 0.0 and max(rep["errs"].values()) < 1e-7


@pytest.mark.parametrize("precond", ["amg", "jacobi"])
def test_md_solve_on_two_ranks_writes_the_same_files_as_on_one(tmp_path, precond):
    """`torch.distributed.run --nproc-per-node 2 ... setup.initialize(TorchComm()); md.solve()`: partition, per-rank
    upload, gathers of the result rows and the results directory guard of solvers.solve's multi-rank branch."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", SHK_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29575" if precond == "amg" else "29576", os.path.join(ROOT, "tests", "md_solve_worker.py"),
           str(tmp_path), precond]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"], rep


def test_rccl_two_rank_parity():
    """The RCCL data path proper (grouped ncclSend / ncclRecv halos, scalar all-reduces) needs one GPU per rank: runs
    wherever two GPUs are visible, skips on the one-GPU boxes of this build (where it has therefore NEVER run: DESIGN.md
    section 5)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    r = _launch(2, "rccl", 29581, ("--precond", "amg", "--nx", "400", "--ny", "200"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ok"] and rep["ghost_mismatch"] == 0.0 and max(rep["errs"].values()) < 1e-7


def test_rccl_loopback_selftest_of_the_data_path_calls():
    """The grouped ncclSend / ncclRecv, ncclAllReduce and async-error calls of csrc/shk_comm.hip, executed for real: a
    one-rank communicator exchanging with itself (the closest a one-GPU box gets to the multi-rank data path)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_selftest_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SELFTEST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_rccl_single_rank_communicator():
    r = _launch(1, "rccl", 29521)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_rccl_refusal_is_collective_and_falls_back_to_the_host_staged_transport():
    """Two ranks on ONE GPU ask for an RCCL communicator.  RCCL refuses duplicate devices, which exercises a real
    multi-rank bootstrap through the dlopen()ed library (id broadcast, ncclCommInitRank on both ranks), the
    collective error report of distributed.make_context (make_runner raises on BOTH ranks) and, with
    --allow-host-staged, the fallback of runner.make_runner: both ranks end up on the gloo transport and take the same
    solver decisions."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tests", "rccl_shared_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stderr.count("RCCL communicator could not be created") >= 2
    assert r.stdout.count("refused True") == 2          # no silent fallback: it takes --allow-host-staged
    import re
    lines = re.findall(r"rank (\d): transport gloo, step 0 newton (\d+) krylov (\d+)", r.stdout)
    assert sorted(l[0] for l in lines) == ["0", "1"] and lines[0][1:] == lines[1][1:]
