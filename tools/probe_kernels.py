#!/usr/bin/env python3
"""Time the assembly and SpMV kernels alone (hipEvents on the library stream) on one config."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shakti_fenics_amd.runner import SingleRunner

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2_1m"
order = sys.argv[2] if len(sys.argv) > 2 else "morton"
t = time.time()
r = SingleRunner(cfg, order=order)
print(f"setup {time.time()-t:.1f}s", r.stats, flush=True)
c = r.ctx
c.assemble(360.0)
c.sync()
nv, ne, nnz = r.nv_global, r.ne_global, r.nnz_global
ms = c.time_kernel("assemble", 10, 360.0)
b = 12 * ne + 16 * nv + 88 * nv + 8 * nv + 8 * nnz
print(f"assemble {ms*1e3:.1f} us  {b/ms/1e6:.0f} GB/s algorithmic", flush=True)
for _ in range(3):
    ms = c.time_kernel("spmv", 50)
    b = 12 * nnz + 4 * (nv + 1) + 16 * nv
    print(f"spmv {ms*1e3:.1f} us  {b/ms/1e6:.0f} GB/s algorithmic", flush=True)
