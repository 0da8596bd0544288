#!/usr/bin/env python3
"""A few launches of k_assemble alone on one config (workload of rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shakti_fenics_amd.runner import SingleRunner
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4_10m"
r = SingleRunner(cfg)
print("stats", r.stats, flush=True)
print("assemble ms", r.ctx.time_kernel("assemble", 3, 360.0), flush=True)
r.close()
