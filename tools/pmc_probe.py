#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs): a few launches of the
calibration read, the plain SpMV and the assembly at one config, then one multigrid-preconditioned solve."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shakti_fenics_amd import _lib
from shakti_fenics_amd.runner import SingleRunner

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4_10m"
r = SingleRunner(cfg, precond="amg")
c = r.ctx
c.assemble(360.0)
st = r.stats
print("nv", st["nv"], "nnz", st["nnz"], "slots", st["sell_slots"], "ne", st["ne"], flush=True)
print("stream-read ms", c.time_kernel("other", 5), "bytes", 8 * st["sell_slots"], flush=True)
print("spmv ms", c.time_kernel("spmv", 5), flush=True)
print("assemble ms", c.time_kernel("assemble", 3, 360.0), flush=True)
print("assemble (residual only) ms", c.time_assemble_residual(3, 360.0), flush=True)
c.assemble(360.0)
print("linear solve", c.linear_solve(), flush=True)
