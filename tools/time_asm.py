#!/usr/bin/env python3
"""Both instances of the assembly kernel timed alone (10 launches between one hipEvent pair, three times).
    python tools/time_asm.py [CONFIG]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shakti_fenics_amd.runner import SingleRunner

r = SingleRunner(sys.argv[1] if len(sys.argv) > 1 else "c4_10m")
c = r.ctx
c.assemble(360.0)
c.sync()
for _ in range(3):
    print("full %.1f us   residual-only %.1f us" % (c.time_kernel("assemble", 10, 360.0) * 1e3,
                                                    c.time_assemble_residual(10, 360.0) * 1e3), flush=True)
