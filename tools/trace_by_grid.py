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


# Launches queued behind a Krylov stop flag return at once: 2-5 us whatever the kernel (the figure varies from box to box).
# A kernel whose real launches take > 25 us (median) sheds everything below 8 us; the small multigrid levels' kernels
# (6-17 us) cannot be told apart from such launches by duration alone: they keep a 3.2 us cut and their averages carry the
# remainder (the library's own profile drops those launches by queue position: shk_api.hip, krylov_inner).
groups = collections.defaultdict(list)
opener = gzip.open if sys.argv[1].endswith(".gz") else open
with opener(sys.argv[1], "rt") as fh:
    for row in csv.DictReader(fh):
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        grid = int(row["Grid_Size_X"]) if "Grid_Size_X" in row else int(row["Grid_Size"])
        groups[(row["Kernel_Name"], grid)].append(dur)
agg, early = {}, 0
for (name, grid), durs in groups.items():
    durs.sort()
    solver = re.search(r"k_(amg|spmv|bicg|dense|krylov)", name) is not None
    cut = 0.0
    if solver:
        big = [d for d in durs if d >= 8.0]
        cut = 8.0 if big and big[len(big) // 2] > 25.0 else 3.2
    kept = [d for d in durs if d >= cut]
    early += len(durs) - len(kept)
    if kept:
        a = agg.setdefault((label(name), grid), [0, 0.0])
        a[0] += len(kept); a[1] += sum(kept)
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.1f} ms ({early} launches that returned at once behind the Krylov stop flag not counted)")
for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:44]:
    print(f"{name:44s} grid {grid:9d} calls {n:6d} avg {us/n:9.1f} us  total {us/1e3:9.1f} ms  {100*us/tot:5.1f}%")
