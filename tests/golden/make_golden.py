#!/usr/bin/env python3
"""Generate tests/golden/c1_5k_oracle.npz with the CPU oracle (NOT with the reference: FEniCSx is not
installable here, see DESIGN.md "Oracle").  The fixture pins the oracle against regressions and is the
full-trajectory parity target for the HIP path on configuration C1: 71x71 jittered mesh of
a 100 km box, |0.001 + N(0, 0.005)| initial gap (rng seed 0), lake storage on, 10 steps of 3600 s (first
360 s), Dirichlet N = 3.7e5 Pa on x = 0.  Inputs are regenerated from seeds by tests/cases.py::c1_case."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import shakti_oracle as O  # noqa: E402
from cases import c1_case  # noqa: E402


def main():
    dom, f, bc, g = c1_case()
    ts = np.arange(11) * 3600.0
    snaps = {}

    def cb(i, ff):
        if i in (0, 9):
            k = f"step{i + 1}"
            snaps[k + "_N"], snaps[k + "_b"] = ff.N.copy(), ff.b.copy()
            snaps[k + "_q"], snaps[k + "_melt_n"] = ff.q.copy(), ff.melt_n.copy()

    fo, log = O.run(dom.xy, dom.cells, f, ts, O.Params(), bc, g, nsteps=10, callback=cb)
    snaps["newton_its"] = np.array([l["niter"] for l in log])
    snaps["xy_checksum"] = np.array([dom.xy.sum(), float(dom.cells.astype(np.int64).sum())])
    np.savez_compressed(os.path.join(HERE, "c1_5k_oracle.npz"), **snaps)
    print("newton its", snaps["newton_its"], "N range", fo.N.min(), fo.N.max())


if __name__ == "__main__":
    main()
