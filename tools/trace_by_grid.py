#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv by (kernel, grid size): separates the multigrid levels.
Kernels are grouped by their FULL name (template arguments included); the label abbreviates it."""
import collections, csv, gzip, re, sys


def label(name: str) -> str:
    n = name.split("(")[0].replace("void ", "").replace("shk::", "")
    head, sep, targs = n.partition("<")   # abbreviate the template arguments only
    for a, b in (("double", "d"), ("float", "f"), ("true", "T"), ("false", "F"), (", ", ",")):
        targs = targs.replace(a, b)
    return (head + sep + targs)[:44]


agg = collections.defaultdict(lambda: [0, 0.0])
early = 0
opener = gzip.open if sys.argv[1].endswith(".gz") else open
with opener(sys.argv[1], "rt") as fh:
    for row in csv.DictReader(fh):
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        if dur < 3.2 and re.search(r"k_(amg|spmv|bicg|dense)", row["Kernel_Name"]):   # launches behind the stop flag return at once
            early += 1
            continue
        a = agg[(label(row["Kernel_Name"]), int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row["Grid_Size"]))]
        a[0] += 1; a[1] += dur
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.1f} ms ({early} launches that returned at once behind the Krylov stop flag not counted)")
for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:44]:
    print(f"{name:44s} grid {grid:9d} calls {n:6d} avg {us/n:9.1f} us  total {us/1e3:9.1f} ms  {100*us/tot:5.1f}%")
