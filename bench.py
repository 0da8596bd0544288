#!/usr/bin/env python3
"""Headline benchmark: DOF-updates/s per Newton iteration (assembly + linear solve) on the
synthetic rectangular ice-sheet mesh of BASELINE.json / SURVEY.md section 8(d).

    python bench.py --gpus 1 --steps K --warmup W            (one process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

A "step" is one time step of the solve loop (`/root/reference/source/solvers.py:168-229`): Newton
solve for N (fused P1 assembly + Jacobi-BiCGStab per Newton iteration) followed by the fused
flux / melt / gap-height updates.  value = Nv * (Newton iterations in the timed steps) / wall time.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8,
                    help="timed steps; 8 = one rebuild period of the multigrid's dense coarsest inverse, so the default "
                         "run pays for exactly one rebuild inside the timed region")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c4_10m", help="c1_5k | c1_12k | c2_1m | c4_10m (default: the 10M-DOF mesh)")
    ap.add_argument("--order", default="morton")
    ap.add_argument("--dt", type=float, default=3600.0)
    ap.add_argument("--storage", type=int, default=0, help="lake storage term on (C3) or off (steady-state march)")
    ap.add_argument("--moulins", type=int, default=0)
    ap.add_argument("--krylov-rtol", type=float, default=1e-10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="560x112", help="nx x ny of the CPU-baseline sample mesh (same geometry)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--transport", default="rccl", help="rccl (xGMI) | gloo (host-staged, tests)")
    ap.add_argument("--precond", default="amg", help="amg (aggregation multigrid, default) | jacobi (north_star's solver; diverges at 10M DOF)")
    ap.add_argument("--amg-steps", type=int, default=2, help="extra steps timed with the multigrid preconditioner (N=1)")
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args()


def cpu_baseline(args) -> dict:
    """The oracle in the reference's algorithmic configuration (P1 assembly, 15-point quadrature,
    exact sparse LU per Newton iteration, DOLFINx Newton defaults), one thread, on a bounded sample
    of the same geometry.  This is the only place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import shakti_oracle as O
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import CONFIGS, N_BDRY, outflow_predicate, synthetic_fields

    nx, ny = (int(v) for v in args.cpu_sample.split("x"))
    _, _, Lx, Ly = CONFIGS[args.config]
    dom = rectangle_mesh(nx, ny, Lx, Ly, order=args.order)
    sf = synthetic_fields(dom, storage_on=bool(args.storage), moulins=0)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"],
                 inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    prm = O.Params()
    last, _ = O.last_cell_of_vertex(nv, dom.cells)
    steps = 5   # ~12-15 s of single-thread work on the GPU box's host
    t0 = time.perf_counter()
    its = 0
    for i in range(steps):
        dt = 0.1 * args.dt if i == 0 else args.dt
        n, conv, _ = O.newton_solve(dom.xy, dom.cells, f, dt, prm, bc, N_BDRY, linear="lu")
        its += n
        O.update_explicit(dom.xy, dom.cells, f, dt, prm, last)
    wall = time.perf_counter() - t0
    return dict(value=nv * its / wall, unit="DOF-updates/s", cores=1, kind="port",
                sample=f"oracle (NumPy assembly + SciPy SuperLU, reference configuration) on a {nx}x{ny} = {nv}-DOF "
                       f"mesh of the same {Lx/1e3:.0f}x{Ly/1e3:.0f} km geometry, {steps} steps, {its} Newton "
                       f"iterations, {wall:.1f} s on 1 of {os.cpu_count()} host cores")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.transport == "gloo":   # rehearsal: several ranks on whatever GPUs exist, host-staged halo
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    from shakti_fenics_amd import _lib
    from shakti_fenics_amd.runner import make_runner

    def say(msg):
        if rank == 0 and not args.quiet:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    run = make_runner(args, rank, world, local_rank)
    say(f"setup done: {run.describe()}")

    def barrier():
        run.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        info = run.step(i)
        say(f"warmup step {i}: newton {info.newton_its} krylov {info.krylov_its}")
    barrier()
    t0 = time.perf_counter()
    newton = krylov = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = run.step(i)
        newton += info.newton_its
        krylov += info.krylov_its
        say(f"timed step {i}: newton {info.newton_its} krylov {info.krylov_its}")
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device="cpu" if args.transport == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    nv = run.nv_global
    out = {
        "metric": "DOF-updates/s per Newton iter (assembly+linear solve)",
        "value": nv * newton / wall if newton else 0.0,
        "unit": "DOF-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {run.describe()}",
            "dofs": nv, "cells": run.ne_global, "nnz": run.nnz_global,
            "newton_its": newton, "krylov_its": krylov,
            "krylov_its_per_newton": krylov / max(newton, 1),
            "krylov": f"BiCGStab, right preconditioner {args.precond}, true-residual rtol {args.krylov_rtol:g}",
            "parallelism": f"dd{world}" if world > 1 else "single",
        },
    }
    if not args.no_roofline:
        roof = run.roofline(HBM_PEAK_GBS)  # one more (collective) step with per-launch hipEvents
        if rank == 0:
            out["roofline"] = roof
        say("roofline step done")
    if world == 1 and args.precond == "jacobi" and args.amg_steps > 0:
        # same state, same metric, with the multigrid preconditioner of DESIGN.md section 9
        run.set_precond("amg")
        run.sync()
        t1 = time.perf_counter()
        nn = kk = 0
        for _ in range(args.amg_steps):
            info = run.step()
            nn += info.newton_its
            kk += info.krylov_its
        run.sync()
        w = time.perf_counter() - t1
        out["multigrid"] = {"value": nv * nn / w, "unit": "DOF-updates/s", "steps": args.amg_steps,
                            "ms_per_step": 1e3 * w / args.amg_steps, "newton_its": nn, "krylov_its": kk,
                            "krylov_its_per_newton": kk / max(nn, 1),
                            "note": "same workload continued with precond=amg (aggregation multigrid V(1,1)); "
                                    "not the headline: north_star names Jacobi-BiCGStab"}
        say(f"multigrid leg done: {nn} newton, {kk} krylov, {w:.2f} s")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
