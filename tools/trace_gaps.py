#!/usr/bin/env python3
"""Idle time of the GPU in a rocprofv3 kernel_trace.csv: gaps between one kernel's end and the next one's start, grouped
by the pair (previous kernel, next kernel).  Shows where a step's wall time goes beyond the kernels themselves (host
polls, synchronous copies, dispatch latency).   usage: trace_gaps.py kernel_trace.csv [KERNEL:N]
With KERNEL:N the analysis starts behind the N-th launch of KERNEL (k_update_b:2 = behind two warm-up steps)."""
import collections, csv, sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void shk::", "").replace("shk::", "")[:44]))
rows.sort()
if len(sys.argv) > 2:
    kname, nth = sys.argv[2].split(":")
    hits = [r for r in rows if r[2].startswith(kname)]
    t0 = hits[int(nth) - 1][1]
    rows = [r for r in rows if r[0] >= t0]
span = (rows[-1][1] - rows[0][0]) * 1e-6
busy = sum(e - s for s, e, _ in rows) * 1e-6
early = sum(1 for s, e, _ in rows if e - s < 1500)
print(f"span {span:.1f} ms, kernels {busy:.1f} ms ({100 * busy / span:.1f} %), idle {span - busy:.1f} ms; {len(rows)} launches, {early} of them early exits (< 1.5 us)")
pair = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
end = rows[0][1]
prev = rows[0][2]
for s, e, name in rows[1:]:
    gap = (s - end) * 1e-3
    if gap > 0:
        p = pair[(prev, name)]
        p[0] += 1; p[1] += gap
        hist["<2us" if gap < 2 else "2-5us" if gap < 5 else "5-20us" if gap < 20 else "20-100us" if gap < 100 else ">100us"] += gap
    if e > end:
        end, prev = e, name
print("idle by gap length (ms):", {k: round(v * 1e-3, 1) for k, v in sorted(hist.items())})
for (a, b), (n, us) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{us * 1e-3:8.1f} ms  {n:6d} x {us / n:8.1f} us   {a}  ->  {b}")
