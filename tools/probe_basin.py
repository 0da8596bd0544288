#!/usr/bin/env python3
"""The unstructured Delaunay basin mesh (hole, curved outlet, graded spacing, random vertex order) at scale: N vertices,
STEPS time steps with storage and moulins, with and without the warm-started linear solves.  GPU box.
usage: probe_basin.py [n_vertices=1000000] [steps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shakti_fenics_amd.runner import SingleRunner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fields = {}
for warm in (4, 0):
    t = time.time()
    r = SingleRunner(basin=n, storage=True, moulins=10)
    r.ctx.set_params(krylov_warm_start=warm)
    print(f"warm {warm}: {r.describe()}; setup {time.time() - t:.1f} s; max row {r.stats['max_row_len']}", flush=True)
    newton, krylov = [], []
    r.sync(); t = time.time()
    for i in range(steps):
        info = r.step(i)
        newton.append(info.newton_its); krylov.append(info.krylov_its)
    r.sync()
    print(f"warm {warm}: {steps} steps in {time.time() - t:.2f} s, newton {newton}, krylov {krylov} (sum {sum(krylov)})", flush=True)
    fields[warm] = r.ctx.get_field("N")
    r.close()
d = np.linalg.norm(fields[4] - fields[0]) / np.linalg.norm(fields[0])
print(f"rel-L2 difference of N after {steps} steps, warm start 4 against 0: {d:.2e}")
