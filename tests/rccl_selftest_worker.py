"""One process, one GPU: a one-rank RCCL communicator through the library's dlopen()ed librccl, then the loop-back
self-test of the data path's call sequence (shk_comm_selftest) and a solve on the context that owns the communicator."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from cases import make_case, upload  # noqa: E402
from shakti_fenics_amd import _lib  # noqa: E402

dom, f, bc, g = make_case(nx=41, ny=31)
ctx = _lib.ShaktiHip(dom.xy, dom.cells)
ctx.set_halo(np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(1, np.int64))
ctx.comm_init_rccl(0, 1, _lib.rccl_unique_id())
ctx.comm_selftest()
ctx.set_params(precond=_lib.PRECOND["amg"])
upload(ctx, f, bc, g)
info = ctx.step(360.0)
assert info.converged
st = ctx.comm_stats()
print(f"RCCL_SELFTEST_OK newton {info.newton_its} krylov {info.krylov_its} stats {st}", flush=True)
ctx.close()
