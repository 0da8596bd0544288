#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv by (kernel, grid size): separates the multigrid levels."""
import collections, csv, sys
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        if dur < 1.5:   # early-exit launches after the stop flag
            continue
        name = row["Kernel_Name"].split("(")[0][-28:]
        a = agg[(name, int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row["Grid_Size"]))]
        a[0] += 1; a[1] += dur
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.1f} ms")
for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{name:30s} grid {grid:9d} calls {n:6d} avg {us/n:9.1f} us  total {us/1e3:9.1f} ms  {100*us/tot:5.1f}%")
