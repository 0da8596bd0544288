#!/usr/bin/env python3
"""Headline benchmark: DOF-updates/s per Newton iteration (assembly + linear solve) on the
synthetic rectangular ice-sheet mesh of BASELINE.json / SURVEY.md section 8(d).

    python bench.py --gpus 1 --steps K --warmup W            (one process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

A "step" is one time step of the solve loop (`/root/reference/source/solvers.py:168-229`): Newton
solve for N (fused P1 assembly + one BiCGStab solve per Newton iteration, right-preconditioned by the
aggregation multigrid of DESIGN.md 4b; `--precond jacobi` selects north_star's Jacobi, which diverges at
10M DOF) followed by the fused flux / melt / gap-height updates.
value = Nv * (Newton iterations in the timed steps) / wall time.  Rank 0 prints ONE JSON line.
Every step assembles, refreshes the preconditioner and solves every Newton system to the same true-residual rule; the
linear solves start from a least-squares combination of the previous steps' solutions (shk_params.krylov_warm_start,
DESIGN.md section 4) and stop at Newton's own threshold (krylov_newton_eta).  `strict_linear_solves` in the line is
the same workload with both switched off (round 1's rule), `--warm-start 0` switches the first off for the whole run.

After the timed region (never inside it): one profiled step for the roofline legs; the steady-state march
of configurations C2 / C4 (until ||dN|| / ||N|| < 1e-8 or 50 steps, reporting which); the CPU
baselines of BASELINE.md section 2 (oracle LU and oracle Jacobi-BiCGStab, C1 and a bounded sample).
With several ranks the halo / reduction transport is RCCL; if it cannot be used the run FAILS (exit code != 0)
unless --allow-host-staged is given, and the line's "transport" field says which one ran.
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
    ap.add_argument("--config", default="c4_10m", help="c1_5k | c1_12k | c2_1m | c4_10m (default: the 10M-DOF jittered rectangle) | "
                    "basin_1m | basin_10m (genuinely unstructured Delaunay basin mesh on the same footprint)")
    ap.add_argument("--window-steps", type=int, default=25, help="N=1: steps run in all (plain steps are appended after the timed "
                    "region) so that the line can also report the windows 'steps 1-8' and 'steps 5-24' whatever --steps/--warmup say")
    ap.add_argument("--order", default="morton")
    ap.add_argument("--dt", type=float, default=3600.0)
    ap.add_argument("--storage", type=int, default=0, help="lake storage term on (C3) or off (steady-state march)")
    ap.add_argument("--moulins", type=int, default=0)
    ap.add_argument("--krylov-rtol", type=float, default=1e-10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="560x112", help="nx x ny of the CPU-baseline sample mesh (same geometry)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--transport", default="rccl", help="rccl (xGMI) | gloo (host-staged, tests)")
    ap.add_argument("--allow-host-staged", action="store_true",
                    help="if RCCL cannot be used, run the same solver over the host-staged gloo transport instead of failing")
    ap.add_argument("--strict-steps", type=int, default=2,
                    help="extra steps (after the timed region) with krylov_newton_eta = 0, krylov_warm_start = 0, krylov_forcing = 0: round 1's rule")
    ap.add_argument("--forcing", type=float, default=0.1,
                    help="shk_params.krylov_forcing (inexact Newton: non-final iterations stop at forcing x the residual the "
                         "iteration is expected to leave behind; 0: every linear solve to the Newton-aware floor, as in round 2)")
    ap.add_argument("--warm-start", type=int, default=4, help="shk_params.krylov_warm_start (0: every linear solve starts from zero)")
    ap.add_argument("--steady-max", type=int, default=50, help="N=1: total steps of the steady-state march after the timed "
                    "region (0: skip); stops early once ||dN||/||N|| < 1e-8")
    ap.add_argument("--precond", default="amg", help="amg (aggregation multigrid, default) | jacobi (north_star's solver; diverges at 10M DOF)")
    ap.add_argument("--amg-steps", type=int, default=2, help="extra steps timed with the multigrid preconditioner (N=1)")
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args()


def _oracle_case(dom, storage, moulins):
    import shakti_oracle as O
    from shakti_fenics_amd.synthetic import outflow_predicate, synthetic_fields
    sf = synthetic_fields(dom, storage_on=storage, moulins=moulins)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"],
                 inputs=sf["inputs"])
    return f, O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))


def _oracle_leg(dom, storage, steps, dt, linear, budget_s=60.0):
    """`steps` time steps of the oracle; returns DOF-updates/s with its split into assembly and linear solve."""
    import shakti_oracle as O
    from shakti_fenics_amd.synthetic import N_BDRY
    f, bc = _oracle_case(dom, storage, 0)
    nv = dom.num_vertices
    prm = O.Params()
    last, _ = O.last_cell_of_vertex(nv, dom.cells)
    its = kits = done = 0
    t0 = time.perf_counter()
    for i in range(steps):
        n, conv, info = O.newton_solve(dom.xy, dom.cells, f, 0.1 * dt if i == 0 else dt, prm, bc, N_BDRY, linear=linear,
                                       krylov_rtol=1e-10)
        if not conv:
            break
        O.update_explicit(dom.xy, dom.cells, f, 0.1 * dt if i == 0 else dt, prm, last)
        its += n
        kits += sum(info["krylov_its"])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    wall = time.perf_counter() - t0
    return dict(value=nv * its / wall if wall > 0 else 0.0, unit="DOF-updates/s", dofs=nv, steps=done, newton_its=its,
                krylov_its_per_newton=(kits / its if its and linear != "lu" else None), seconds=wall,
                linear="SciPy SuperLU (the reference's ksp preonly + pc lu)" if linear == "lu"
                else "Jacobi-BiCGStab, SciPy CSR SpMV, rtol 1e-10 (the GPU path's north_star solver)")


def cpu_baseline(args) -> dict:
    """BASELINE.md section 2 on the box's host cores, one thread each (NumPy assembly; SuperLU and SciPy's CSR product are
    single-threaded).  Headline = the oracle in the reference's algorithmic configuration (P1 assembly, 15-point
    quadrature, exact sparse LU per Newton iteration, DOLFINx Newton defaults) on a bounded sample of the bench
    geometry; `legs` add the same algorithm on C1 (5 041 and 12 321 DOF, 10 steps) and the CPU twin of the Jacobi-
    BiCGStab solver on C1 (at 62k DOF it already needs ~1900 iterations per Newton iteration and minutes per step).  This is the only place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import BASIN_CONFIGS, CONFIGS, config_mesh

    nx, ny = (int(v) for v in args.cpu_sample.split("x"))
    Lx, Ly = (BASIN_CONFIGS[args.config][1:] if args.config in BASIN_CONFIGS else CONFIGS[args.config][2:])
    sample = rectangle_mesh(nx, ny, Lx, Ly, order=args.order)
    head = _oracle_leg(sample, bool(args.storage), 5, args.dt, "lu")
    legs = {"sample_lu": head}
    legs["c1_5k_lu"] = _oracle_leg(config_mesh("c1_5k"), True, 10, args.dt, "lu", 20.0)
    legs["c1_12k_lu"] = _oracle_leg(config_mesh("c1_12k"), True, 10, args.dt, "lu", 20.0)
    legs["c1_5k_bicgstab"] = _oracle_leg(config_mesh("c1_5k"), True, 10, args.dt, "bicgstab", 20.0)
    legs["c1_12k_bicgstab"] = _oracle_leg(config_mesh("c1_12k"), True, 10, args.dt, "bicgstab", 20.0)
    return dict(value=head["value"], unit="DOF-updates/s", cores=1, kind="port",
                sample=f"oracle (NumPy assembly + SciPy SuperLU, reference configuration) on a {nx}x{ny} = {head['dofs']}-DOF "
                       f"mesh of the same {Lx/1e3:.0f}x{Ly/1e3:.0f} km geometry, {head['steps']} steps, {head['newton_its']} Newton "
                       f"iterations, {head['seconds']:.1f} s on 1 of {os.cpu_count()} host cores",
                legs=legs,
                published_reference="2.32e5 DOF-timesteps/s on 8 MPI ranks of an unstated CPU, real-data setup "
                                    "(/root/reference/notebooks/example.ipynb:54-55): different hardware, setup and unit")


def steady_march(run, args, say) -> dict:
    """march (storage off) until ||N_k - N_(k-1)|| / ||N_k|| < 1e-8 or 50 steps in all, and
    say which ended it.  Outside the timed region; fields cross PCIe once per step for the norm."""
    tol = 1e-8
    prev = run.ctx.get_field("N")
    drift = None
    while run.next_step < args.steady_max:
        run.step()
        cur = run.ctx.get_field("N")
        drift = float(np.linalg.norm(cur - prev) / np.linalg.norm(cur))
        prev = cur
        if drift < tol:
            break
    reached = drift is not None and drift < tol
    say(f"steady-state march: {run.next_step} steps, drift {drift}")
    return dict(rule=f"||dN||/||N|| < {tol:g} or {args.steady_max} steps", steps_run=run.next_step, drift_last_step=drift,
                ended_by="tolerance" if reached else "step limit",
                note=None if args.storage == 0 else "storage on: a transient configuration, the march is informative only")


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
    try:
        run = make_runner(args, rank, world, local_rank)
    except _lib.ShaktiCommStall as exc:
        _lib.exit_on_stall(exc)
    run.ctx.set_params(krylov_warm_start=args.warm_start, krylov_forcing=args.forcing)
    say(f"setup done: {run.describe()}")
    switches = _lib.env_overrides()
    if switches:
        say(f"SHK_* experiment switches set in the environment: {switches}")
    wiring = None
    if world > 1:
        # every rank's wiring (device, neighbours, ghost counts) after the start-up exchange + all-reduce check
        wiring = [None] * world
        dist.all_gather_object(wiring, run.wiring)
        for w in wiring:
            say(f"rank {w['rank']}: device {w['device']}, {w['n_own']} owned + {w['n_ghost']} ghost vertices, neighbours {w['neighbours']}, "
                f"ghosts per neighbour {w['ghosts_per_neighbour']}")

    def barrier():
        run.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # host time stamps after every step (no extra synchronisation: a step returns after its last residual norm, its update
    # kernels -- ~0.5 ms -- run into the next step): lets the line report fixed windows of steps beside the timed region
    stamps, counts = [time.perf_counter()], []

    def one_step(i):
        try:
            info = run.step(i)
        except _lib.ShaktiCommStall as exc:
            _lib.exit_on_stall(exc)
        stamps.append(time.perf_counter())
        counts.append((info.newton_its, info.krylov_its))
        return info

    for i in range(args.warmup):
        info = one_step(i)
        say(f"warmup step {i}: newton {info.newton_its} krylov {info.krylov_its}")
    barrier()
    stamps[-1] = time.perf_counter()
    st0 = run.ctx.comm_stats() if world > 1 else None
    t0 = time.perf_counter()
    newton = krylov = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_its
        krylov += info.krylov_its
        say(f"timed step {i}: newton {info.newton_its} krylov {info.krylov_its}")
    barrier()
    wall = time.perf_counter() - t0
    stamps[-1] = time.perf_counter()
    st1 = run.ctx.comm_stats() if world > 1 else None
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
        "transport": getattr(run, "transport", "none") if world > 1 else "none (one GPU)",
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
            "krylov_warm_start": run.ctx.get_params().krylov_warm_start, "krylov_newton_eta": run.ctx.get_params().krylov_newton_eta,
            "krylov_forcing": run.ctx.get_params().krylov_forcing,
            "krylov": f"BiCGStab, right preconditioner {args.precond}; every linear solve runs until its TRUE residual is below "
                      f"max({args.krylov_rtol:g} ||F_k||, 0.1 x Newton's own stopping threshold max(1e-10, 1e-9 ||F_0||)) "
                      "(shk_params.krylov_newton_eta); "
                      + (f"a Newton iteration that the previous step's history says cannot be the last stops its linear solve at "
                         f"{args.forcing:g} x the residual it is expected to leave behind (shk_params.krylov_forcing: inexact Newton; "
                         "the iteration expected to end the solve is solved as before); " if args.forcing > 0 else "")
                      + (f"the solve of Newton iteration k starts from the least-squares combination of the solutions of iteration k "
                         f"of the previous {args.warm_start} steps (shk_params.krylov_warm_start) " if args.warm_start > 0
                         else "every solve starts from zero ")
                      + "-- Newton counts equal the LU oracle's in every parity test",
            "first_step": "0.1 dt (solvers.py:81); |b_init| instead of the reference's signed draw (DESIGN.md section 1)",
            "parallelism": f"dd{world}" if world > 1 else "single",
        },
    }
    # SELL-64 padding, share of the slots whose columns are 16-bit offsets, multigrid levels on the packed copy (rank 0's context)
    out["storage"] = run.ctx.storage_stats()
    out["env_overrides"] = switches or None
    out["assembly_passes"] = dict(run.ctx.solver_stats(), note="since context creation (warm-up included): full = residual + Jacobian; "
                                  "residual_only = the pass after the update predicted to be a Newton solve's last (its Jacobian "
                                  "would be thrown away); redone = full passes repeated because that prediction was wrong")
    if world == 1:
        # the same transient read through fixed windows: the timed region above is whatever --steps / --warmup say (the
        # driver's command: 20 after 5; the default: 8 after 1), and the metric counts Newton iterations, which fall from
        # 5 to 2 per step over the first hours of the transient -- so both windows are always reported
        while len(counts) < args.window_steps:
            one_step(len(counts))
        run.sync()
        stamps[-1] = time.perf_counter()

        def window(a, b):   # steps a .. b inclusive
            if b >= len(counts):
                return None
            w = stamps[b + 1] - stamps[a]
            nn, kk = sum(c[0] for c in counts[a:b + 1]), sum(c[1] for c in counts[a:b + 1])
            return {"steps": f"{a}-{b}", "value": nv * nn / w, "unit": "DOF-updates/s", "ms_per_step": 1e3 * w / (b - a + 1),
                    "newton_its": nn, "krylov_its": kk}
        out["windows"] = {"steps_1_8": window(1, 8), "steps_5_24": window(5, 24),
                          "note": "host time stamps per step, no synchronisation added; `value` above is the timed region "
                                  f"(steps {args.warmup}-{args.warmup + args.steps - 1})",
                          "per_step": [{"newton_its": c[0], "krylov_its": c[1], "ms": 1e3 * (stamps[i + 1] - stamps[i])}
                                       for i, c in enumerate(counts)]}
    if world > 1:
        out["ranks"] = wiring
        st = {k: v - st0[k] for k, v in st1.items()}
        nk = max(krylov, 1)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {k: v / nk for k, v in st.items()})
        for r, m in enumerate(per_rank):
            say(f"rank {r}: per Krylov iteration {m['exchanges']:.1f} ghost exchanges, {m['allreduces']:.1f} all-reduces, "
                f"{m['allgathers']:.1f} all-gathers; {m['bytes_exchanged'] / 1e3:.1f} KB exchanged, {m['bytes_allreduced'] / 1e3:.2f} KB "
                f"all-reduced, {m['bytes_allgathered'] / 1e3:.1f} KB all-gathered")
        out["message_rounds_by_rank"] = per_rank
        out["message_rounds"] = {"rank": 0, "ghost_exchanges_per_krylov_it": st["exchanges"] / nk,
                                 "allreduces_per_krylov_it": st["allreduces"] / nk,
                                 "allgathers_per_krylov_it": st["allgathers"] / nk,
                                 "bytes_allgathered_per_krylov_it": st["bytes_allgathered"] / nk,
                                 "bytes_exchanged_per_krylov_it": st["bytes_exchanged"] / nk,
                                 "bytes_allreduced_per_krylov_it": st["bytes_allreduced"] / nk,
                                 "note": "rounds rank 0 issued inside the timed region (Newton iterations' ghost updates, "
                                         "hierarchy refreshes and the replicated level's gathers included) per Krylov iteration"}
    if args.strict_steps > 0:
        # the same workload continued with every linear solve started from zero and driven to krylov_rtol ||F_k||
        # (krylov_newton_eta = 0, krylov_warm_start = 0): what the Newton-aware stopping floor and the warm start save,
        # reported beside the headline, never instead of it
        run.ctx.set_params(krylov_newton_eta=0.0, krylov_warm_start=0, krylov_forcing=0.0)
        barrier()
        t1 = time.perf_counter()
        nn = kk = 0
        for _ in range(args.strict_steps):
            info = run.step()
            nn += info.newton_its
            kk += info.krylov_its
        barrier()
        w = time.perf_counter() - t1
        run.ctx.set_params(krylov_newton_eta=0.1, krylov_warm_start=args.warm_start, krylov_forcing=args.forcing)
        out["strict_linear_solves"] = {"value": nv * nn / w if nn else 0.0, "unit": "DOF-updates/s", "steps": args.strict_steps,
                                       "ms_per_step": 1e3 * w / args.strict_steps, "newton_its": nn, "krylov_its": kk,
                                       "krylov_its_per_newton": kk / max(nn, 1),
                                       "note": "krylov_newton_eta = 0, krylov_warm_start = 0, krylov_forcing = 0: every linear solve from zero to 1e-10 ||F_k||, as in round 1"}
        say(f"strict leg done: {nn} newton, {kk} krylov, {w:.2f} s")
    if not args.no_roofline:
        roof = run.roofline(HBM_PEAK_GBS)  # one more (collective) step with per-launch hipEvents
        if rank == 0:
            out["roofline"] = roof
        say("roofline step done")
    if world == 1 and args.steady_max > 0 and args.precond != "jacobi" and args.config not in ("basin_10m", "basin_1m"):
        out["steady_state"] = steady_march(run, args, say)
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
                            "note": "same workload continued with precond=amg (aggregation multigrid V(0,2), DESIGN.md 4b); "
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
