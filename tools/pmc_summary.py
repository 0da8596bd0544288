#!/usr/bin/env python3
"""Per-kernel sums of one rocprofv3 --pmc counter from its counter_collection.csv."""
import collections, csv, sys
f, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != ctr:
            continue
        a = agg[row["Kernel_Name"][:70]]
        v = float(row["Counter_Value"])
        a[0] += 1; a[1] += v; a[2] = max(a[2], v)
print(f"# {ctr}: kernel, calls, sum, max per dispatch")
for k, (n, s, m) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{ctr},{k},{n},{s:.0f},{m:.0f}")
