#!/usr/bin/env python3
"""Transient run of a synthetic configuration on the GPU with a per-step log (BASELINE.json configs 3 and 5):
Newton / Krylov iterations, ||F|| before and after the Newton solve, wall time -- and, with --floor, the
evaluation floor of ||F||_2 in fp64: residual entries of sampled rows against the SAME element formulas summed
in extended precision (np.longdouble, 64-bit mantissa) on the sub-mesh around them, scaled to the whole mesh.

    python tools/run_transient.py c4_10m --steps 24 --storage 1 --moulins 100 --floor 6,12,18,23 > log.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def floor_probe(r, dt, nsample=3000, seed=3):
    """sqrt(nv / ns) * || F_gpu[sample] - F_ext[sample] ||_2 : the fp64 evaluation error of the residual norm."""
    import shakti_oracle as O
    from shakti_fenics_amd.synthetic import N_BDRY
    c, dom = r.ctx, r.dom
    nv = dom.num_vertices
    c.assemble(dt)
    F = c.residual()
    names = ("N", "N_n", "b", "melt_n", "z_b", "z_s", "G", "storage", "inputs")
    host = {k: c.get_field(k) for k in names}
    host["q"] = c.get_field("q")
    rng = np.random.default_rng(seed)
    sample = rng.choice(nv, min(nsample, nv), replace=False)
    isbc = np.zeros(nv, dtype=bool)
    isbc[r.bc] = True
    sample = sample[~isbc[sample]]
    ins = np.zeros(nv, dtype=bool)
    ins[sample] = True
    csel = np.nonzero(ins[dom.cells].any(axis=1))[0]
    verts = np.unique(dom.cells[csel])
    loc = np.full(nv, -1, dtype=np.int64)
    loc[verts] = np.arange(verts.size)
    ld = np.longdouble
    sub = O.Fields(**{k: host[k][verts].astype(ld) for k in host})
    bc_sub = np.nonzero(isbc[verts])[0]
    quad = O.load_quadrature().astype(ld)
    xy = dom.xy[verts].astype(ld)
    Fe, _ = O.element_tensors(xy, loc[dom.cells[csel]], sub, ld(dt), O.Params(), quad, want_jacobian=bool(bc_sub.size))
    cc = loc[dom.cells[csel]]
    if bc_sub.size:   # lifting (zero when the Dirichlet values are satisfied, as they are after the first solve)
        _, Ke = O.element_tensors(xy, cc, sub, ld(dt), O.Params(), quad, True)
        cb = np.zeros(verts.size, dtype=bool)
        cb[bc_sub] = True
        Fe = Fe + np.einsum("eij,ej->ei", Ke, np.where(cb[cc], ld(N_BDRY) - sub.N[cc], ld(0)))
    Fx = np.zeros(verts.size, dtype=ld)
    np.add.at(Fx, cc.ravel(), Fe.ravel())
    d = (F[sample].astype(ld) - Fx[loc[sample]]).astype(np.float64)
    # Sensitivity of F to the fp64 REPRESENTATION of the unknown: move every free N by one ulp (random sign) and
    # re-assemble.  Rounding N - dx to fp64 perturbs N by <= 0.5 ulp (rms ulp / sqrt(12)), so no solver that stores
    # N in fp64 -- DOLFINx/PETSc included -- can drive ||F|| below about  sens / sqrt(12).
    N0 = host["N"]
    sgn = np.random.default_rng(seed + 1).choice([-1.0, 1.0], nv)
    sgn[isbc] = 0.0
    c.set_field("N", N0 + np.spacing(N0) * sgn)
    c.assemble(dt)
    sens = float(np.linalg.norm(c.residual() - F))
    c.set_field("N", N0)
    return dict(floor=float(np.sqrt(nv / sample.size) * np.linalg.norm(d)), worst_entry=float(np.abs(d).max()),
                sampled_rows=int(sample.size), norm_F=float(np.linalg.norm(F)), one_ulp_N_sensitivity=sens,
                representation_floor=sens / np.sqrt(12.0), b_max=float(host["b"].max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--storage", type=int, default=1)
    ap.add_argument("--moulins", type=int, default=0)
    ap.add_argument("--dt", type=float, default=3600.0)
    ap.add_argument("--precond", default="amg")
    ap.add_argument("--floor", default="", help="comma-separated steps after which the fp64 floor is probed")
    ap.add_argument("--shape", default="", help="nx,ny,Lx,Ly instead of a named config")
    a = ap.parse_args()
    from shakti_fenics_amd.runner import SingleRunner
    shape = tuple(float(v) if i > 1 else int(v) for i, v in enumerate(a.shape.split(","))) if a.shape else None
    t0 = time.perf_counter()
    r = SingleRunner(a.config, storage=bool(a.storage), moulins=a.moulins, dt=a.dt, precond=a.precond, shape=shape)
    print(json.dumps(dict(event="setup", seconds=time.perf_counter() - t0, workload=r.describe(), dofs=r.nv_global)), flush=True)
    probes = {int(s) for s in a.floor.split(",") if s}
    for i in range(a.steps):
        t = time.perf_counter()
        dt = 0.1 * a.dt if i == 0 else a.dt
        info = r.ctx.step(dt)
        r.sync()
        rec = dict(step=i, newton=info.newton_its, krylov=info.krylov_its, converged=bool(info.converged),
                   krylov_failed=bool(info.krylov_failed), residual0=info.residual0, residual=info.residual,
                   seconds=time.perf_counter() - t)
        print(json.dumps(rec), flush=True)
        if not info.converged:
            print(json.dumps(dict(event="newton_not_converged", step=i, floor=floor_probe(r, dt))), flush=True)
            break
        r.next_step = i + 1
        if i in probes:
            print(json.dumps(dict(event="floor", step=i, **floor_probe(r, dt))), flush=True)
    r.close()


if __name__ == "__main__":
    main()
