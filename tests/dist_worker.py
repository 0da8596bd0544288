"""Worker for the multi-process domain-decomposition tests (launched by torch.distributed.run).

Every rank owns one subdomain context; with --transport gloo all ranks may share one GPU (host-staged
halo), which is how the partitioned solver code path is exercised on a one-GPU box.  Rank 0 also runs
the undecomposed problem and checks that the decomposed fields match it.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", default="gloo")
    ap.add_argument("--nx", type=int, default=61)
    ap.add_argument("--ny", type=int, default=31)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--precond", default="jacobi")
    ap.add_argument("--lx", type=float, default=20e3)
    ap.add_argument("--ly", type=float, default=10e3)
    ap.add_argument("--basin", type=int, default=0, help="unstructured Delaunay basin mesh of about this many vertices")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.transport == "rccl":
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dev = 0
        dist.init_process_group("gloo")
    from shakti_fenics_amd.runner import PartitionedRunner, SingleRunner

    shape = (a.nx, a.ny, a.lx, a.ly)
    run = PartitionedRunner(rank, world, dev, shape=shape, storage=True, transport=a.transport, precond=a.precond,
                            basin=a.basin)
    infos = []
    for i in range(a.steps):
        info = run.step(i)
        infos.append((info.newton_its, info.krylov_its))
    sub = run.sub
    mine = {k: run.ctx.get_field(k)[: sub.n_own] for k in ("N", "b", "qx", "qy", "melt_n", "N_n")}
    # ghost copies must equal the owners' values after a step
    ghost_N = run.ctx.get_field("b")[sub.n_own:]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((sub.gid[: sub.n_own], mine, sub.gid[sub.n_own:], ghost_N), gathered, dst=0)
    ok = True
    report = {"world": world, "transport": a.transport, "infos": infos, "overlap": run.ctx.comm_overlap()}
    if rank == 0:
        ref = SingleRunner(shape=shape, storage=True, device=dev, precond=a.precond, basin=a.basin)
        ref_infos = []
        for i in range(a.steps):
            info = ref.step(i)
            ref_infos.append((info.newton_its, info.krylov_its))
        nv = ref.nv_global
        errs = {}
        for k in mine:
            full = np.full(nv, np.nan)
            for gid, vals, _, _ in gathered:
                full[gid] = vals[k]
            r = ref.ctx.get_field(k)
            errs[k] = float(np.linalg.norm(full - r) / max(np.linalg.norm(r), 1e-300))
        bfull = np.full(nv, np.nan)
        for gid, vals, _, _ in gathered:
            bfull[gid] = vals["b"]
        gerr = max(float(np.max(np.abs(bfull[gg] - gv))) if gg.size else 0.0 for _, _, gg, gv in gathered)
        report.update(errs=errs, ref_infos=ref_infos, ghost_mismatch=gerr)
        ok = (all(e < 1e-7 for e in errs.values()) and gerr == 0.0
              and [x[0] for x in infos] == [x[0] for x in ref_infos])
        if a.precond == "amg":
            # the distributed hierarchy keeps every cross-subdomain coupling (only the aggregates differ: they follow
            # each subdomain's own k-d order, and ghost columns are frozen at the prolongated coarse correction inside a
            # level's sweeps), so Krylov iteration counts must stay at the one-subdomain level.  The margin is a
            # single-versus-decomposed one, not run-to-run noise (both runs are deterministic): the two hierarchies are
            # different preconditioners, and the warm start of the later steps amplifies a difference of a few
            # iterations in one solve into the next solves' starting points
            ok = ok and all(x[1] <= 1.5 * y[1] + 3 for x, y in zip(infos, ref_infos))
        ov = report["overlap"]
        # off unless SHK_OVERLAP=1
        if world > 1 and os.environ.get("SHK_OVERLAP", "0") != "0":
            # the level-0 exchanges travelled on the second stream behind an interior pass, over a genuine split
            ok = ok and ov["active"] and ov["overlapped_exchanges"] > 0 and 0 < ov["boundary_slices"] < ov["slices"]
        else:
            ok = ok and not ov["active"] and ov["overlapped_exchanges"] == 0
        report["ok"] = ok
        print(json.dumps(report), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(report, f)
        ref.close()
    run.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
