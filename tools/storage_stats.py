#!/usr/bin/env python3
"""SELL-64 padding, 16-bit-column coverage and packed multigrid levels of a configuration (no solve).
    python tools/storage_stats.py CONFIG [CONFIG ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shakti_fenics_amd.runner import SingleRunner

for cfg in sys.argv[1:] or ["c4_10m"]:
    r = SingleRunner(cfg)
    st = r.ctx.storage_stats()
    ps = r.ctx.plan_stats()
    print(cfg, json.dumps(dict(st, max_row_len=ps["max_row_len"], amg_levels=ps["amg_levels"])), flush=True)
    r.ctx.close()
