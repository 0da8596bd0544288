#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE counter_collection.csv) into
profiles/<name>.json: HBM-side bytes per launch of the kernels bench.py reports.

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts exactly half of the bytes of a streaming read -- verified in the same pass on `k_norm2` over the value
array (a pure 8-B-per-lane streaming read of known size, tools/pmc_probe.py) -- so reads are doubled; WRITE_SIZE
is exact for streaming stores.  Per-kernel value = median over the launches that did real work."""
import collections, csv, json, os, statistics, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shakti_fenics_amd.runner import kernel_source_hash

fetch_csv, write_csv, calib_bytes, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
valu_csv = sys.argv[5] if len(sys.argv) > 5 else None


def per_kernel(path, ctr):
    d = collections.defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == ctr:
                d[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return d


F, W = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
norm = max(F.get("shk::k_norm2(long, double const*, double*)", [0.0]))
factor = calib_bytes / (norm * 1024.0) if norm else 2.0
res = {"kernel_source_sha256": kernel_source_hash(), "fetch_size_correction": factor, "calibration": {"kernel": "k_norm2 over the SELL value array",
       "known_bytes": calib_bytes, "fetch_size_kib": norm}, "kernels": {}}
want = {"k_amg_post<true": "amg_fine", "k_amg_first<true": "amg_first", "k_spmv<0": "spmv_plain", "k_spmv<1": "spmv1", "k_spmv<2": "spmv2",
        "k_assemble<256, 15, 7, true>": "assemble", "k_assemble<256, 15, 7, false>": "assemble_residual_only",
        "k_bicg_u": "bicg_u", "k_bicg_s": "bicg_s", "k_amg_sweeps": "amg_sweeps"}
for name in set(F) | set(W):
    for pat, key in want.items():
        if pat in name:
            big_f = [v for v in F.get(name, []) if v > 0.5 * max(F[name])] if F.get(name) else [0.0]
            big_w = [v for v in W.get(name, []) if v > 0.5 * max(W[name])] if W.get(name) else [0.0]
            fr, wr = statistics.median(big_f) * 1024 * factor, statistics.median(big_w) * 1024
            res["kernels"][key] = {"kernel": name.split("(")[0], "read_bytes": fr, "write_bytes": wr,
                                   "hbm_bytes": fr + wr, "launches_counted": len(big_f)}
if valu_csv:
    # the assembly kernel's own ceiling: fp64 wave-instructions (4 issue cycles each on 1024 SIMDs) and how long its waves wait
    V = {c: per_kernel(valu_csv, c) for c in ("SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")}
    for name, vals in V["SQ_INSTS_VALU"].items():
        for pat, key in (("k_assemble<256, 15, 7, true>", "assemble"), ("k_assemble<256, 15, 7, false>", "assemble_residual_only")):
            if pat in name:
                big = [v for v in vals if v > 0.5 * max(vals)]
                wait, cyc = V["SQ_WAIT_ANY"].get(name, [0.0]), V["SQ_WAVE_CYCLES"].get(name, [0.0])
                rec = res["kernels"].setdefault(key, {"kernel": name.split("(")[0]})
                rec["valu_wave_insts"] = statistics.median(big)
                rec["valu_issue_floor_ms"] = rec["valu_wave_insts"] * 4.0 / (1024 * 2.4e9) * 1e3
                rec["wait_any_frac"] = (statistics.median(wait) / statistics.median(cyc)) if statistics.median(cyc) > 0 else None
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
