#!/usr/bin/env python3
"""How far can the DOLFINx semantics this build could not verify move the answer?  

Two restated semantics are common-mode between the oracle and the HIP path, so no parity test here can see a wrong
guess: (a) Basix's degree-7 quadrature TABLE (the build derives its own symmetric 15-point rule), (b) the CELL ORDER
behind `Function.interpolate(Expression)`'s last-cell-wins (solvers.py:186-192).  Both are injectable: this probe runs
the HIP path (and, on C1, the LU oracle) with
  * a different exact degree-7 rule -- the 16-point conical Gauss-Jacobi product (tests/cases.py: conical_rule),
  * the cell list reversed, and randomly permuted,
and reports rel-L2 of N, b, q against the built-in rule / the mesh generator's cell order after the 10 steps of C1 and
after 3 steps on the 62k-DOF mesh with moulins (1 + omega Re active).  tests/test_gpu_sensitivity.py pins the numbers.

    python tools/probe_sensitivity.py [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

DT = 3600.0


def run_hip(dom, f, bc, g, steps, quad=None, cells=None):
    from cases import upload
    from shakti_fenics_amd import _lib
    ctx = _lib.ShaktiHip(dom.xy, dom.cells if cells is None else cells)
    ctx.set_params(precond=_lib.PRECOND["amg"])
    if quad is not None:
        ctx.set_quadrature(quad)
    upload(ctx, f, bc, g)
    its = []
    for i in range(steps):
        info = ctx.step(0.1 * DT if i == 0 else DT)
        assert info.converged and not info.krylov_failed, (i, info.residual)
        its.append(info.newton_its)
    out = {k: ctx.get_field(k) for k in ("N", "b", "q", "melt_n")}
    ctx.close()
    return out, its


def run_oracle(dom, f, bc, g, steps, quad=None, cells=None):
    import shakti_oracle as O
    fo, log = O.run(dom.xy, dom.cells if cells is None else cells, f.copy(), np.arange(steps + 1) * DT, O.Params(), bc, g,
                    nsteps=steps, quad=quad)
    return {"N": fo.N, "b": fo.b, "q": fo.q, "melt_n": fo.melt_n}, [l["niter"] for l in log]


def sensitivities(case: str, with_oracle: bool) -> dict:
    from cases import c1_case, conical_rule, moulin_case_62k, rel_l2
    dom, f, bc, g = c1_case() if case == "c1" else moulin_case_62k()
    steps = 10 if case == "c1" else 3
    quad = conical_rule(4)
    ne = dom.cells.shape[0]
    variants = {"conical_quadrature": dict(quad=quad),
                "cells_reversed": dict(cells=np.ascontiguousarray(dom.cells[::-1])),
                "cells_permuted": dict(cells=np.ascontiguousarray(dom.cells[np.random.default_rng(11).permutation(ne)]))}
    base, its0 = run_hip(dom, f, bc, g, steps)
    out = {"case": case, "dofs": dom.num_vertices, "steps": steps, "newton_its": its0, "variants": {}}
    if with_oracle:
        ob, oits = run_oracle(dom, f, bc, g, steps)
        out["hip_vs_oracle_builtin"] = {k: rel_l2(base[k], ob[k]) for k in ("N", "b", "q")}
    for name, kw in variants.items():
        v, its = run_hip(dom, f, bc, g, steps, **kw)
        rec = {"rel_l2_vs_builtin": {k: rel_l2(v[k], base[k]) for k in ("N", "b", "q", "melt_n")}, "newton_its": its}
        if with_oracle:
            ov, oits = run_oracle(dom, f, bc, g, steps, **kw)
            rec["hip_vs_oracle_same_variant"] = {k: rel_l2(v[k], ov[k]) for k in ("N", "b", "q")}
            rec["oracle_newton_its"] = oits
        out["variants"][name] = rec
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    res = [sensitivities("c1", True), sensitivities("62k_moulins", False)]
    for r in res:
        print(f"== {r['case']}: {r['dofs']} DOF, {r['steps']} steps, Newton {r['newton_its']}")
        if "hip_vs_oracle_builtin" in r:
            print("   HIP vs LU oracle, built-in rule and order:", {k: f"{v:.2e}" for k, v in r["hip_vs_oracle_builtin"].items()})
        for name, rec in r["variants"].items():
            print(f"   {name:20s} vs built-in:", {k: f"{v:.2e}" for k, v in rec["rel_l2_vs_builtin"].items()}, "Newton", rec["newton_its"])
            if "hip_vs_oracle_same_variant" in rec:
                print(f"   {'':20s} HIP vs oracle under the same injection:", {k: f"{v:.2e}" for k, v in rec["hip_vs_oracle_same_variant"].items()})
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
