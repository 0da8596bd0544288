#!/usr/bin/env python3
"""One Jacobi-BiCGStab solve on the first Newton system of a config: iterations, recursive vs true residual."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from shakti_fenics_amd.runner import SingleRunner

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2_1m"
pc = sys.argv[2] if len(sys.argv) > 2 else "jacobi"
cpu = False
from shakti_fenics_amd import _lib
r = SingleRunner(cfg)
c = r.ctx
c.set_params(precond=_lib.PRECOND[pc], krylov_max_it=int(sys.argv[3]) if len(sys.argv) > 3 else 20000)
for step in range(2):
    dt = 360.0 if step == 0 else 3600.0
    c.assemble(dt)
    t = time.time()
    its, conv, rr = c.linear_solve()
    el = time.time() - t
    rp, ci, va = c.csr()
    n = r.nv_global
    J = sp.csr_matrix((va, ci, rp), shape=(n, n))
    F = c.residual()
    dx = c.get_field("dx")
    true = np.linalg.norm(J @ dx - F) / np.linalg.norm(F)
    print(f"step {step}: gpu its {its} conv {conv} recursive relres {rr:.3e} true relres {true:.3e}  {el:.2f}s "
          f"({el/max(its,1)*1e6:.1f} us/it)", flush=True)
    if cpu:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import shakti_oracle as O
        t = time.time()
        x, it_cpu, rn = O.jacobi_bicgstab(J, F, 1e-10, 1e-50, 60000)
        print(f"   cpu classical its {it_cpu} true relres {np.linalg.norm(J@x-F)/np.linalg.norm(F):.3e} {time.time()-t:.1f}s", flush=True)
    info = c.step(dt)
    print(f"   step: newton {info.newton_its} krylov {info.krylov_its} conv {info.converged} res0 {info.residual0:.3e} res {info.residual:.3e}", flush=True)
