#!/usr/bin/env python3
"""Software floor of an RCCL message round on this GPU: a one-rank communicator (the peer is the rank itself), the three
kinds of round the decomposed solver issues, at the sizes of the 8-way split of the 10M-DOF mesh, back to back on the
context's stream.  No link is crossed, so the numbers are LOWER bounds of the per-round cost the strong-scaling model
(tools/scaling_model.py, profiles/r03_scaling_model.md) leaves as a free parameter.  With several GPUs visible, launch one
process per GPU through torch.distributed.run and kind "sendrecv" talks to rank ^ 1 over xGMI."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from shakti_fenics_amd import _lib  # noqa: E402
from shakti_fenics_amd.mesh import rectangle_mesh  # noqa: E402

dom = rectangle_mesh(41, 31, 10e3, 8e3)
ctx = _lib.ShaktiHip(dom.xy, dom.cells)
ctx.set_halo(np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(1, np.int64))
ctx.comm_init_rccl(0, 1, _lib.rccl_unique_id())
out = {}
for kind, n, what in (("sendrecv", 1500, "ghost exchange of the finest level, 1500 doubles (12 KB)"),
                      ("sendrecv", 200, "ghost exchange of a coarse level, 200 doubles"),
                      ("allreduce", 4, "the Krylov scalars, 4 doubles"),
                      ("allgather", 80_000, "replicated level's right-hand side, 80 KB per rank"),
                      ("allgather", 700_000, "replicated level's operator values, 0.7 MB per rank")):
    us = ctx.comm_time_round(kind, n, 300)
    out[f"{kind}_{n}"] = {"us_per_round": us, "what": what}
    print(f"{kind:10s} n = {n:7d}: {us:7.2f} us per round   ({what})", flush=True)
print(json.dumps(out))
ctx.close()
