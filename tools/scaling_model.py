#!/usr/bin/env python3
"""Strong-scaling model of the decomposed solver from pieces measured on ONE GPU.

No 8-GPU node is available to this build, so a scaling curve cannot be measured.  What CAN be measured on one GPU is
the compute side of every term: for P = 2, 4, 8 this tool builds the P-way RCB decomposition of the bench mesh, sets all
P subdomain contexts up in ONE process (one thread per subdomain, a real in-process transport: the collective multigrid
setup and a few warm-up steps run for real, so the hierarchy, the dampings and the state are those of a P-GPU run),
then switches rank 0 to the timing-only transport (shk_comm_set_timing_only: messages skipped, results wrong, durations
right) and times ONE Newton iteration with a fixed number of Krylov iterations on the otherwise idle GPU: per-phase
kernel time (hipEvents riding on the dispatch packets), wall time, launches and message rounds per Krylov iteration.
The same measurement on the undecomposed context gives T(1).  The model then adds a message cost per round:

    T_it(P, m) = wall time per Krylov iteration of rank 0's subdomain + rounds per iteration x m,   m = 10 / 20 / 30 us
    T_step(P, m) = n_asm x assembly + n_newton x (hierarchy refresh + Newton-level kernels) + K x T_it + update

with n_asm, n_newton, K taken from the undecomposed run's own steps.  Output: a Markdown report (--out) + JSON.

    python tools/scaling_model.py --config c4_10m --parts 2,4,8 --out profiles/r03_scaling_model.md
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from shakti_fenics_amd import _lib  # noqa: E402
from shakti_fenics_amd.bc import locate_boundary_dofs  # noqa: E402
from shakti_fenics_amd.mesh import rectangle_mesh  # noqa: E402
from shakti_fenics_amd.partition import build_subdomain, rcb_partition  # noqa: E402
from shakti_fenics_amd.synthetic import CONFIGS, N_BDRY, outflow_predicate, synthetic_fields  # noqa: E402

KRYLOV_PHASES = ("spmv", "vector", "halo", "amg_fine", "amg_first", "amg_coarse")


class ThreadTransport:
    """In-process stand-in for the host-staged transport: P threads, one per subdomain, meet at a barrier."""

    def __init__(self, nranks: int, timeout: float = 900.0):
        self.P = nranks
        self.barrier = threading.Barrier(nranks, timeout=timeout)
        self.mail: dict = {}
        self.red = [None] * nranks

    def callbacks(self, rank: int):
        def exchange(nbr, send, sp, recv, rp):
            for k, r in enumerate(nbr):
                self.mail[(rank, int(r))] = send[sp[k]:sp[k + 1]].copy()
            self.barrier.wait()
            for k, r in enumerate(nbr):
                recv[rp[k]:rp[k + 1]] = self.mail[(int(r), rank)]
            self.barrier.wait()

        def allreduce(buf):
            self.red[rank] = buf.copy()
            self.barrier.wait()
            acc = self.red[0].copy()
            for r in range(1, self.P):   # fixed order: every rank ends up with the same bits
                acc += self.red[r]
            buf[:] = acc
            self.barrier.wait()

        return exchange, allreduce


def fill(ctx, sf, g, bc_local):
    pick = (lambda a: a) if g is None else (lambda a: a[g])
    ctx.set_field("z_b", pick(sf["z_b"])); ctx.set_field("z_s", pick(sf["z_s"])); ctx.set_field("G", pick(sf["G"]))
    ctx.set_field("inputs", pick(sf["inputs"])); ctx.set_field("storage", pick(sf["lake_bdry"]))
    ctx.set_field("b", np.abs(pick(sf["b_init"])))
    ctx.set_field("N_n", pick(sf["N_init"])); ctx.set_field("N", pick(sf["N_init"]))
    ctx.set_field("q", pick(sf["q_init"])); ctx.set_field("melt_n", np.zeros(ctx.nv))
    ctx.set_dirichlet(bc_local, N_BDRY)


def timed_newton(ctx, dt, k_fixed):
    """ONE Newton iteration with exactly k_fixed Krylov iterations (rtol unreachable), profiled; returns phases, wall."""
    p0 = ctx.get_params()
    saved = dict(krylov_rtol=p0.krylov_rtol, newton_max_it=p0.newton_max_it, krylov_max_it=p0.krylov_max_it,
                 krylov_warm_start=p0.krylov_warm_start, krylov_newton_eta=p0.krylov_newton_eta)
    ctx.set_params(krylov_rtol=1e-300, newton_max_it=1, krylov_max_it=k_fixed, krylov_warm_start=0, krylov_newton_eta=0.0)
    ctx.sync()
    n_start = ctx.get_field("N")
    # unprofiled pass first: wall time with the queue running free
    t0 = time.perf_counter()
    info = ctx.newton_solve(dt)
    ctx.sync()
    wall = time.perf_counter() - t0
    ctx.set_field("N", n_start)   # the profiled pass starts from the same iterate
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    info2 = ctx.newton_solve(dt)
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    ctx.set_params(**saved)
    return prof, wall, (info.krylov_its, info2.krylov_its)


def per_iteration(prof, k):
    d = {ph: prof[ph]["ms"] / k for ph in KRYLOV_PHASES}
    for key in _lib.COARSE_PHASES:
        d[key] = prof[key]["ms"] / k
    d["launches"] = sum(prof[ph]["launches"] for ph in ("spmv", "vector", "halo", "amg_fine", "amg_first")) / k
    d["kernel_ms"] = sum(d[ph] for ph in KRYLOV_PHASES)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4_10m")
    ap.add_argument("--parts", default="2,4,8")
    ap.add_argument("--warm", type=int, default=4, help="real warm-up steps before the timed iteration (all subdomains)")
    ap.add_argument("--k", type=int, default=30, help="Krylov iterations of the timed Newton iteration")
    ap.add_argument("--dt", type=float, default=3600.0)
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    nx, ny, Lx, Ly = CONFIGS[a.config]
    t00 = time.perf_counter()

    def say(msg):
        print(f"[model +{time.perf_counter() - t00:6.1f}s] {msg}", file=sys.stderr, flush=True)

    dom = rectangle_mesh(nx, ny, Lx, Ly)
    sf = synthetic_fields(dom, storage_on=False, moulins=0)
    bc = locate_boundary_dofs(dom, outflow_predicate(dom))
    nv = dom.num_vertices
    say(f"mesh {a.config}: {nv} vertices")
    res = {"config": a.config, "dofs": nv, "k_fixed": a.k, "warm_steps": a.warm, "runs": {}}

    # ---- P = 1
    ctx = _lib.ShaktiHip(dom.xy, dom.cells, device=0)
    ctx.set_params(precond=_lib.PRECOND["amg"])
    fill(ctx, sf, None, bc)
    hist = []
    for i in range(a.warm):
        info = ctx.step(0.1 * a.dt if i == 0 else a.dt)
        hist.append((info.newton_its, info.krylov_its))
    prof, wall, kk = timed_newton(ctx, a.dt, a.k)
    one = per_iteration(prof, a.k)
    one.update(wall_ms=1e3 * wall, asm_ms=prof["assemble"]["ms"] / max(prof["assemble"]["launches"], 1),
               other_ms=prof["other"]["ms"], krylov_its=kk, steps=hist, rounds=0.0, allreduces=0.0, allgathers=0.0)
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    ctx.update_explicit(a.dt)
    one["update_ms"] = ctx.profile_read(reset=True)["update"]["ms"]
    ctx.profile_enable(False)
    res["runs"]["1"] = one
    say(f"P=1: {one['kernel_ms']:.3f} ms of kernels per Krylov iteration, wall {1e3 * wall / a.k:.3f} ms per iteration incl. Newton overhead")
    ctx.close()

    # ---- P > 1
    for P in [int(v) for v in a.parts.split(",") if v]:
        owner = rcb_partition(dom.xy, P)
        subs = [build_subdomain(dom, owner, r, P) for r in range(P)]
        say(f"P={P}: subdomains built ({subs[0].n_own} owned + {subs[0].n_ghost} ghosts on rank 0, {len(subs[0].nbr)} neighbours)")
        tr = ThreadTransport(P)
        ctxs = [None] * P
        errs = [None] * P
        steps = [[] for _ in range(P)]
        stats_warm = [None] * P

        def work(r):
            try:
                sub = subs[r]
                c = _lib.ShaktiHip(sub.xy, sub.cells, device=0, n_own=sub.n_own)
                ctxs[r] = c
                c.set_halo(sub.nbr, sub.send_ptr, sub.send_idx, sub.recv_ptr)
                ex, ar = tr.callbacks(r)
                c.comm_init_callbacks(r, P, ex, ar)
                c.set_params(precond=_lib.PRECOND["amg"])   # collective: distributed hierarchy
                g2l = np.full(nv, -1, dtype=np.int64)
                g2l[sub.gid] = np.arange(sub.gid.size)
                loc = g2l[bc]
                fill(c, sf, sub.gid, loc[loc >= 0].astype(np.int32))
                s0 = c.comm_stats()
                for i in range(a.warm):
                    info = c.step(0.1 * a.dt if i == 0 else a.dt)
                    steps[r].append((info.newton_its, info.krylov_its))
                s1 = c.comm_stats()
                stats_warm[r] = {k: s1[k] - s0[k] for k in s1}
            except Exception as exc:   # noqa: BLE001
                errs[r] = repr(exc)
                tr.barrier.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if any(errs):
            raise SystemExit(f"P={P}: {errs}")
        kr = sum(k for _, k in steps[0])
        say(f"P={P}: {a.warm} real steps done: {steps[0]}")
        c0 = ctxs[0]
        c0.comm_set_timing_only(True)
        s0 = c0.comm_stats()
        prof, wall, kk = timed_newton(c0, a.dt, a.k)
        s1 = c0.comm_stats()
        run = per_iteration(prof, a.k)
        # two solves ran between the two stats reads (unprofiled + profiled)
        nk = max(kk[0] + kk[1], 1)
        run.update(wall_ms=1e3 * wall, asm_ms=prof["assemble"]["ms"] / max(prof["assemble"]["launches"], 1),
                   other_ms=prof["other"]["ms"], krylov_its=kk, steps=steps[0],
                   rounds=(s1["exchanges"] - s0["exchanges"]) / nk, allreduces=(s1["allreduces"] - s0["allreduces"]) / nk,
                   allgathers=(s1["allgathers"] - s0["allgathers"]) / nk,
                   bytes_allgathered=(s1["bytes_allgathered"] - s0["bytes_allgathered"]) / nk,
                   bytes_exchanged=(s1["bytes_exchanged"] - s0["bytes_exchanged"]) / nk,
                   bytes_allreduced=(s1["bytes_allreduced"] - s0["bytes_allreduced"]) / nk,
                   real_rounds_per_it={k: v / max(kr, 1) for k, v in stats_warm[0].items()},
                   n_own=subs[0].n_own, n_ghost=subs[0].n_ghost, neighbours=int(len(subs[0].nbr)),
                   plan=c0.plan_stats())
        c0.profile_enable(True); c0.profile_read(reset=True)
        c0.update_explicit(a.dt)
        run["update_ms"] = c0.profile_read(reset=True)["update"]["ms"]
        c0.profile_enable(False)
        res["runs"][str(P)] = run
        say(f"P={P}: rank 0 alone: {run['kernel_ms']:.3f} ms of kernels per Krylov iteration, {run['rounds']:.1f} exchanges + "
            f"{run['allreduces']:.1f} all-reduces + {run['allgathers']:.1f} all-gathers per iteration")
        for c in ctxs:
            c.close()
        del ctxs, subs

    # ---- model
    one = res["runs"]["1"]
    late = one["steps"][-2:] if len(one["steps"]) >= 2 else one["steps"]
    n_newton = float(np.mean([n for n, _ in late]))
    K = float(np.mean([k for _, k in late]))
    res["step_shape"] = {"newton_its": n_newton, "krylov_its": K, "assemblies": n_newton + 1}

    def t_it(run, m_us):   # ms per Krylov iteration
        newton_over = 2 * run["asm_ms"] + run["other_ms"]
        wall_it = (run["wall_ms"] - newton_over) / a.k
        return max(wall_it, run["kernel_ms"]) + (run["rounds"] + run["allreduces"] + run["allgathers"]) * m_us * 1e-3

    def t_step(run, m_us):
        return (n_newton + 1) * run["asm_ms"] + n_newton * run["other_ms"] + K * t_it(run, m_us) + run["update_ms"]

    model = {}
    for m_us in (0, 10, 20, 30):
        base = t_step(one, 0)
        model[str(m_us)] = {P: {"ms_per_krylov_it": t_it(r, m_us), "ms_per_step": t_step(r, m_us), "speedup": base / t_step(r, m_us)}
                            for P, r in res["runs"].items()}
    res["model"] = model
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    lines = report(res, a)
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    print(text)


def report(res, a):
    runs = res["runs"]
    Ps = sorted(runs, key=int)
    L = [f"# Strong-scaling model of `{res['config']}` ({res['dofs']} DOF) from pieces measured on ONE MI355X", "",
         "No multi-GPU node is available to this build: this is a MODEL, not a measured scaling curve.  Every compute term below",
         "is measured (rank 0's subdomain of the P-way RCB split, alone on the GPU, timing-only transport: messages skipped,",
         f"results wrong, durations right; one Newton iteration with exactly {res['k_fixed']} Krylov iterations after {res['warm_steps']} real steps of all P",
         "subdomains); the cost of a message round is the model's free parameter.  `tools/scaling_model.py`.", "",
         "## Measured: ms per Krylov iteration on rank 0's subdomain (hipEvent kernel time per phase)", "",
         "| P | owned rows | fine level (smoother 2 + first sweep 2) | Krylov products | vector kernels | decomposed coarse levels | replicated levels | restrict + dense + other | pack / unpack | kernels total | wall per iteration | launches | exchanges | all-reduces | all-gathers |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for P in Ps:
        r = runs[P]
        dec = sum(r[f"amg_l{l}"] for l in range(1, 9))
        rest = r["amg_restrict"] + r["amg_dense"] + r["amg_other"]
        newton_over = 2 * r["asm_ms"] + r["other_ms"]
        wall_it = (r["wall_ms"] - newton_over) / res["k_fixed"]
        L.append(f"| {P} | {r.get('n_own', res['dofs'])} | {r['amg_fine'] + r['amg_first']:.3f} | {r['spmv']:.3f} | {r['vector']:.3f} | {dec:.3f} | "
                 f"{r['amg_rep']:.3f} | {rest:.3f} | {r['halo']:.3f} | {r['kernel_ms']:.3f} | {wall_it:.3f} | {r['launches']:.0f}+coarse | "
                 f"{r['rounds']:.1f} | {r['allreduces']:.1f} | {r['allgathers']:.1f} |")
    L += ["", "Per-level sweeps of the decomposed hierarchy (ms per Krylov iteration = two cycles):", "",
          "| P | " + " | ".join(f"level {l}" for l in range(1, 9)) + " | restrictions | dense solve | replicated hierarchy |", "|---|" + "---|" * 11]
    for P in Ps:
        r = runs[P]
        L.append(f"| {P} | " + " | ".join(f"{r[f'amg_l{l}']:.3f}" for l in range(1, 9)) + f" | {r['amg_restrict']:.3f} | {r['amg_dense']:.3f} | {r['amg_rep']:.3f} |")
    L += ["", "Newton-level terms (ms): assembly pass / hierarchy refresh + Newton kernels per Newton iteration / explicit update:", ""]
    for P in Ps:
        r = runs[P]
        L.append(f"* P = {P}: {r['asm_ms']:.3f} / {r['other_ms']:.3f} / {r['update_ms']:.3f}; real steps before the timed iteration "
                 f"(Newton, Krylov): {r['steps']}" + (f"; message volume per Krylov iteration {r['bytes_exchanged'] / 1e3:.1f} KB exchanged, "
                                                      f"{r['bytes_allreduced'] / 1e3:.1f} KB all-reduced, {r['bytes_allgathered'] / 1e3:.1f} KB all-gathered" if P != '1' else ""))
    s = res["step_shape"]
    L += ["", f"## Model: one time step = {s['assemblies']:.0f} assemblies + {s['newton_its']:.0f} Newton iterations + {s['krylov_its']:.1f} Krylov iterations + update", "",
          "| message cost per round | " + " | ".join(f"P = {P}: ms/step (speedup)" for P in Ps) + " |", "|---|" + "---|" * len(Ps)]
    for m_us, row in res["model"].items():
        L.append(f"| {m_us} us | " + " | ".join(f"{row[P]['ms_per_step']:.1f} ({row[P]['speedup']:.2f}x)" for P in Ps) + " |")
    # what bounds the curve: the terms of a Krylov iteration that do NOT shrink with P
    one, last = runs["1"], runs[Ps[-1]]
    shrink = last["amg_fine"] + last["amg_first"] + last["spmv"] + last["vector"] + sum(last[f"amg_l{l}"] for l in range(1, 9))
    fixed = last["amg_rep"] + last["amg_restrict"] + last["amg_dense"] + last["amg_other"] + last["halo"]
    rounds = last["rounds"] + last["allreduces"] + last["allgathers"]
    L += ["", f"## What binds it at P = {Ps[-1]} (ms per Krylov iteration; one GPU: {one['kernel_ms']:.3f})", "",
          f"* work that shrinks with P (finest level, products, vector kernels, decomposed coarse levels): **{shrink:.3f}** "
          f"= {one['kernel_ms'] / shrink:.1f}x below one GPU's iteration if it were all there is;",
          f"* work that does not (replicated levels {last['amg_rep']:.3f}, restrictions + dense solve "
          f"{last['amg_restrict'] + last['amg_dense'] + last['amg_other']:.3f}, pack / unpack kernels {last['halo']:.3f}): **{fixed:.3f}**;",
          f"* {rounds:.1f} message rounds: **{rounds * 0.010:.3f} / {rounds * 0.020:.3f} / {rounds * 0.030:.3f}** at 10 / 20 / 30 us each "
          "(measured floor of a grouped send + recv round on one GPU, no link crossed: 4.8 us + its pack kernel -- "
          "`profiles/r03_rccl_round_floor.txt`).",
          f"* Not in the table: an all-gather is a RING of P - 1 hops, not one hop.  The {last['allgathers']:.1f} all-gathers of an iteration "
          f"bring {last['bytes_allgathered'] / 1e3:.0f} KB to every GPU ({last['bytes_allgathered'] / 1e3 / max(last['allgathers'], 1e-9):.0f} KB each: the replicated level's "
          f"right-hand side); at ~100 GB/s into a GPU that is {last['bytes_allgathered'] / 100e9 * 1e3:.3f} ms per iteration of bandwidth, and each "
          "is likelier to cost 2-3 of the table's rounds than one.  `SHK_AMG_REP_ROWS` trades its size against one more exchanging level.",
          f"* A 6x step would need {one['kernel_ms'] / 6:.3f} ms per iteration: less than the shrinking work plus the replicated levels "
          f"({shrink + last['amg_rep']:.3f}) before a single message is sent."]
    return L


if __name__ == "__main__":
    main()
