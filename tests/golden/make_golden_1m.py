#!/usr/bin/env python3
"""Generate tests/golden/c2_1m_oracle_sample.npz: the LU oracle (NOT the reference -- FEniCSx is not installable here,
DESIGN.md section 1) on BASELINE.json's 1M-DOF mesh (2236 x 447 on 100 km x 20 km, morton order), lake storage on,
20 moulins, 3 time steps (360 s + 2 x 3600 s).  A solve at this size takes the build container tens of minutes
(SuperLU, one thread), so the GPU test compares against this committed SAMPLE instead of running the oracle: values
of N, b, q, melt_n at 4000 seeded vertices after every step, the fields' l2 norms, Newton counts and residuals.

    python tests/golden/make_golden_1m.py            (about half an hour, ~6 GB of memory)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import shakti_oracle as O  # noqa: E402
from shakti_fenics_amd.synthetic import N_BDRY, config_mesh, outflow_predicate, synthetic_fields  # noqa: E402

CONFIG, MOULINS, NSAMPLE, SEED, DT = "c2_1m", 20, 4000, 42, 3600.0


def main():
    dom = config_mesh(CONFIG)
    sf = synthetic_fields(dom, storage_on=True, moulins=MOULINS)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"], inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    prm = O.Params()
    last, _ = O.last_cell_of_vertex(nv, dom.cells)
    sample = np.sort(np.random.default_rng(SEED).choice(nv, NSAMPLE, replace=False))
    # always include the moulin centres' neighbourhoods and some Dirichlet dofs
    sample = np.unique(np.concatenate((sample, np.nonzero(sf["inputs"] > 0)[0][::7], bc[::40])))
    out = dict(sample=sample.astype(np.int64), xy_checksum=np.array([dom.xy.sum(), float(dom.cells.astype(np.int64).sum())]),
               moulins=np.array(MOULINS), dt=np.array(DT))
    its, res = [], []
    for i in range(3):
        dt = 0.1 * DT if i == 0 else DT
        t = time.time()
        n, conv, info = O.newton_solve(dom.xy, dom.cells, f, dt, prm, bc, N_BDRY)
        assert conv
        O.update_explicit(dom.xy, dom.cells, f, dt, prm, last)
        its.append(n)
        res.append(info["residuals"] + [np.nan] * (12 - len(info["residuals"])))
        k = f"step{i + 1}"
        out[k + "_N"], out[k + "_b"], out[k + "_q"], out[k + "_melt_n"] = f.N[sample], f.b[sample], f.q[sample], f.melt_n[sample]
        out[k + "_norms"] = np.array([np.linalg.norm(f.N), np.linalg.norm(f.b), np.linalg.norm(f.q), np.linalg.norm(f.melt_n)])
        print(f"step {i}: newton {n} residuals {info['residuals']} {time.time() - t:.0f} s", flush=True)
        np.savez_compressed(os.path.join(HERE, "c2_1m_oracle_sample.partial.npz"), newton_its=np.array(its), residuals=np.array(res), **out)
    out["newton_its"], out["residuals"] = np.array(its), np.array(res)
    np.savez_compressed(os.path.join(HERE, "c2_1m_oracle_sample.npz"), **out)
    os.remove(os.path.join(HERE, "c2_1m_oracle_sample.partial.npz"))


if __name__ == "__main__":
    main()
