"""Golden fixture tests/golden/c1_5k_oracle.npz (made by tests/golden/make_golden.py with the oracle).
CPU: the oracle still reproduces it.  GPU (-m gpu): the HIP path reproduces the 10-step trajectory."""
import os

import numpy as np
import pytest

import shakti_oracle as O
from cases import c1_case, rel_l2, upload

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_5k_oracle.npz")


def test_oracle_reproduces_golden_first_step():
    g = np.load(GOLD)
    dom, f, bc, val = c1_case()
    assert np.allclose([dom.xy.sum(), float(dom.cells.astype(np.int64).sum())], g["xy_checksum"], rtol=1e-14)
    ts = np.arange(11) * 3600.0
    fo, log = O.run(dom.xy, dom.cells, f, ts, O.Params(), bc, val, nsteps=1)
    assert log[0]["niter"] == g["newton_its"][0]
    assert rel_l2(fo.N, g["step1_N"]) < 1e-10 and rel_l2(fo.b, g["step1_b"]) < 1e-12
    assert rel_l2(fo.q, g["step1_q"]) < 1e-10 and rel_l2(fo.melt_n, g["step1_melt_n"]) < 1e-10


@pytest.mark.gpu
def test_hip_path_reproduces_golden_trajectory():
    from shakti_fenics_amd import _lib
    g = np.load(GOLD)
    dom, f, bc, val = c1_case()
    ctx = _lib.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, bc, val)
    its = []
    for i in range(10):
        info = ctx.step(360.0 if i == 0 else 3600.0)
        assert info.converged
        its.append(info.newton_its)
        if i in (0, 9):
            k = f"step{i + 1}"
            # BASELINE.json's bar is 1e-6 rel-L2 on the head / N field; the same-algorithm agreement is far tighter
            assert rel_l2(ctx.get_field("N"), g[k + "_N"]) < 1e-7
            assert rel_l2(ctx.get_field("b"), g[k + "_b"]) < 1e-7
            assert rel_l2(ctx.get_field("q"), g[k + "_q"]) < 1e-6
            assert rel_l2(ctx.get_field("melt_n"), g[k + "_melt_n"]) < 1e-6
    assert its == list(g["newton_its"])
    ctx.close()
