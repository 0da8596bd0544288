#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv:  pmc_table.py file.csv [kernel substring]"""
import collections, csv, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for row in csv.DictReader(open(sys.argv[1])):
    if pat in row["Kernel_Name"]:
        d[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in d.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
