"""Benchmark / smoke driver around one device context: builds a synthetic configuration
(SURVEY.md section 8d), keeps every field resident in HBM and advances the solve loop of
`/root/reference/source/solvers.py:168-229` one time step per call."""
from __future__ import annotations

import numpy as np

from . import _lib
from .bc import locate_boundary_dofs
from .mesh import basin_mesh, rectangle_mesh
from .synthetic import CONFIGS, N_BDRY, outflow_predicate, synthetic_fields


KERNEL_SOURCES = ("shk_device.h", "shk_kernels.hip", "shk_assemble.hip", "shk_amg.hip", "shk_plan.cpp", "shk_plan.h")


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: ties a committed counter measurement to the code it was taken on."""
    import hashlib
    import os
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for name in KERNEL_SOURCES:
        with open(os.path.join(csrc, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _pmc_traffic(config, nv):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950
    calibration + WRITE_SIZE, separate passes; tools/pmc_probe.py, tools/pmc_to_json.py).  Counters cannot be
    collected from inside bench.py, so the numbers of the matching configuration are read from profiles/ -- and
    only trusted when that file was measured on THESE kernel sources (its kernel_source_sha256); otherwise the
    traffic entries are null and `traffic_source` says why."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                        f"pmc_traffic_{config}.json")
    if not config or not os.path.exists(path):
        return {"source": "no committed counter file for this configuration"}
    doc = json.load(open(path))
    have, now = doc.get("kernel_source_sha256"), kernel_source_hash()
    if have != now:
        return {"source": f"profiles/pmc_traffic_{config}.json was measured on other kernel sources "
                          f"(sha256 {str(have)[:12]} != {now[:12]}): traffic dropped; rerun tools/collect_profiles.sh pmc"}
    k = doc["kernels"]
    out = {"source": f"profiles/pmc_traffic_{config}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                     f"kernel sources sha256 {now[:12]})"}
    for key in ("amg_fine", "amg_first"):
        if key in k:
            out[key] = k[key]["hbm_bytes"]
    if "spmv1" in k and "spmv2" in k:
        out["spmv"] = 0.5 * (k["spmv1"]["hbm_bytes"] + k["spmv2"]["hbm_bytes"])
    if "assemble" in k:
        out["assemble"] = k["assemble"]["hbm_bytes"]
    return out


class SingleRunner:
    def __init__(self, config="c4_10m", order="morton", dt=3600.0, storage=False, moulins=0, device=0,
                 krylov_rtol=1e-10, shape=None, precond="amg", basin=0):
        nx, ny, Lx, Ly = CONFIGS[config] if shape is None else shape
        self.config_name = config if shape is None and not basin else None
        # basin=n: the unstructured Delaunay basin mesh of ~n vertices (mesh.basin_mesh) instead of the rectangle
        self.dom = basin_mesh(basin, order="random") if basin else rectangle_mesh(nx, ny, Lx, Ly, order=order)
        self._outflow = (lambda X: X[0] < 1e-9) if basin else outflow_predicate(self.dom)
        self.dt = dt
        sf = synthetic_fields(self.dom, storage_on=storage, moulins=moulins)
        self.ctx = _lib.ShaktiHip(self.dom.xy, self.dom.cells, device=device)
        self.ctx.set_params(krylov_rtol=krylov_rtol, precond=_lib.PRECOND[precond])
        c = self.ctx
        c.set_field("z_b", sf["z_b"]); c.set_field("z_s", sf["z_s"]); c.set_field("G", sf["G"])
        c.set_field("inputs", sf["inputs"]); c.set_field("storage", sf["lake_bdry"])
        # |b_init|: the reference's 0.001 + N(0, 0.005) draw (setup_cooke2.py:66) enters its flux law only
        # through |b|^3 (constitutive.py:14); see DESIGN.md "synthetic initial gap height"
        c.set_field("b", np.abs(sf["b_init"]))
        c.set_field("N_n", sf["N_init"]); c.set_field("N", sf["N_init"])  # initial guess, solvers.py:48
        c.set_field("q", sf["q_init"]); c.set_field("melt_n", np.zeros(self.dom.num_vertices))
        self.bc = locate_boundary_dofs(self.dom, self._outflow)
        c.set_dirichlet(self.bc, N_BDRY)
        self.transport = "none"
        st = c.plan_stats()
        self.nv_global, self.ne_global, self.nnz_global = st["nv"], st["ne"], st["nnz"]
        self.stats = st
        mesh_txt = (f"Delaunay basin mesh of {self.dom.num_vertices} vertices (hole, curved outlet, random order)" if basin else
                    f"{Lx/1e3:.0f} km x {Ly/1e3:.0f} km rectangle, {nx}x{ny} jittered P1 mesh ({order} order)")
        self._desc = (f"{mesh_txt}, dt {dt:g} s (first step 0.1 dt), storage {'on' if storage else 'off'}, {moulins} moulins, "
                      f"Dirichlet N = {N_BDRY:g} Pa on x = 0; |b_init| for the reference's signed draw (DESIGN.md section 1)")
        self.next_step = 0

    def describe(self):
        return self._desc

    def set_precond(self, name: str):
        self.ctx.set_params(precond=_lib.PRECOND[name])

    def step(self, i=None):
        i = self.next_step if i is None else i
        dt = 0.1 * self.dt if i == 0 else self.dt  # solvers.py:81,174-176
        info = self.ctx.step(dt)
        if info.krylov_failed:
            raise RuntimeError(f"linear solve failed at step {i}: true relative residual {info.krylov_relres:.3e} after "
                               f"{info.krylov_its} BiCGStab iterations")
        if not info.converged:
            raise RuntimeError(f"Newton did not converge at step {i}: residual {info.residual:g} "
                               f"after {info.newton_its} iterations")  # error_on_nonconvergence=True
        self.next_step = i + 1
        return info

    def sync(self):
        self.ctx.sync()

    def roofline(self, peak_gbs: float) -> dict:
        """Per-launch hipEvent timing of one more step on the library's stream.  The dominant kernel is the
        SELL-64 SpMV in its two guises: `k_spmv` (2 launches per BiCGStab iteration) and, with the multigrid
        preconditioner, the finest-level smoother `k_amg_post<true>` (4 launches per iteration)."""
        c = self.ctx
        c.profile_enable(True)
        c.profile_read(reset=True)
        info = self.step()
        prof = c.profile_read(reset=True)
        c.profile_enable(False)
        nv, nnz, slices = self.nv_global, self.nnz_global, (self.nv_global + 63) // 64
        # algorithmic bytes per launch (fp64 values, int32 indices), SURVEY.md 8d
        b_spmv = 12 * nnz + 4 * (nv + 1) + 16 * nv                      # values, colidx, rowptr, x, y = 104 nv
        # multigrid smoother: float values + int32 indices; x (float), r (double), 1/diag (float), x' (float)
        b_post = 8 * nnz + 4 * (slices + 1) + 20 * nv                   # x' = x + w D^-1 (r - A x)
        b_asm = 12 * self.ne_global + 16 * nv + 88 * nv + 8 * nv + 8 * nnz  # 192 nv

        pmc = _pmc_traffic(getattr(self, "config_name", None), nv)

        def leg(phase, nbytes, name):
            n = max(prof[phase]["launches"], 1)
            ms = prof[phase]["ms"] / n
            ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "avg_launch_ms": ms, "launches": prof[phase]["launches"],
                    "bytes_per_launch": nbytes, "achieved": ach, "frac": ach / peak_gbs,
                    "traffic": pmc.get(phase)}

        legs = {"spmv": leg("spmv", b_spmv, "k_spmv<1|2> (SELL-64 SpMV + fused BiCGStab dots)"),
                "assemble": leg("assemble", b_asm, "k_assemble (fused residual + Jacobian)")}
        # what the SpMV actually has to move: fp64 values, 16-bit column offsets on the slices that allow them (the plan
        # reports the padded slot count), the float x of the multigrid cycle, y and the two dot operands in double
        slots = self.stats["sell_slots"]
        legs["spmv"]["needed_bytes"] = 8 * slots + 2 * slots + 16 * ((nv + 63) // 64) + 4 * nv + 8 * nv + 12 * nv
        legs["spmv"]["needed_note"] = ("bytes this implementation must move per launch (padded SELL slots, 16-bit columns, float "
                                       "x, double y and dot operands); `bytes_per_launch` is SURVEY.md 8d's algorithmic figure "
                                       "(int32 columns, double x) that `achieved` is priced with")
        legs["spmv"]["achieved_needed"] = legs["spmv"]["needed_bytes"] / (legs["spmv"]["avg_launch_ms"] * 1e-3) / 1e9 \
            if legs["spmv"]["avg_launch_ms"] > 0 else 0.0
        if prof["amg_fine"]["launches"]:
            legs["amg_fine"] = leg("amg_fine", b_post, "k_amg_post<true> (finest-level multigrid smoother, SELL-64 SpMV)")
        if prof.get("amg_first", {}).get("launches"):
            ap = c.plan_stats()["ap_nnz"]
            b_first = 8 * ap + 4 * (slices + 1) + 4 * nv + 16 * nv + 1 * nv    # A*P (float), agg, r / 1/diag / x', e
            legs["amg_first"] = leg("amg_first", b_first, "k_amg_first<true> (first sweep on the A*P operator)")
        dom = max((k for k in legs if k != "assemble"), key=lambda k: prof[k]["ms"])
        d = legs[dom]
        # The same SpMV launched 20 times back to back between ONE event pair: what a launch takes without the
        # dispatch latency that a per-launch event pair adds (this is the figure the rocprofv3 trace agrees with).
        c.assemble(self.dt)
        b2b_ms = c.time_kernel("spmv", 20)
        b2b = {"kernel": "k_spmv<0> (same product without the fused dots), 20 launches between one hipEvent pair",
                      "avg_launch_ms": b2b_ms, "bytes_per_launch": b_spmv,
                      "achieved": b_spmv / (b2b_ms * 1e-3) / 1e9, "frac": b_spmv / (b2b_ms * 1e-3) / 1e9 / peak_gbs}
        # north_star states its target on the "assembly + SpMV inner loop": both kernels' algorithmic bytes over
        # both kernels' time in the profiled step
        t_in = prof["assemble"]["ms"] + prof["spmv"]["ms"]
        by_in = b_asm * prof["assemble"]["launches"] + b_spmv * prof["spmv"]["launches"]
        inner = {"definition": "algorithmic bytes of all k_assemble + k_spmv launches of the profiled step / their summed "
                               "hipEvent durations", "achieved": by_in / (t_in * 1e-3) / 1e9 if t_in > 0 else 0.0,
                 "launches": {"assemble": prof["assemble"]["launches"], "spmv": prof["spmv"]["launches"]},
                 "note": "a mix figure: the assembly runs at ~13 % and the product at ~75-85 % of the roofline, so the aggregate "
                         "falls as a step needs fewer Krylov iterations per assembly (round 1: 92 iterations in the profiled "
                         "step, now ~26-34)"}
        inner["frac"] = inner["achieved"] / peak_gbs
        # the same mix of launches priced with back-to-back durations (no per-launch event overhead)
        asm_b2b = c.time_kernel("assemble", 3, self.dt)
        t_b2b = asm_b2b * prof["assemble"]["launches"] + b2b_ms * prof["spmv"]["launches"]
        inner["back_to_back"] = {"assemble_ms": asm_b2b, "spmv_ms": b2b_ms,
                                 "achieved": by_in / (t_b2b * 1e-3) / 1e9 if t_b2b > 0 else 0.0}
        inner["back_to_back"]["frac"] = inner["back_to_back"]["achieved"] / peak_gbs
        # SURVEY.md 8d also asks for a whole Krylov iteration and a whole Newton iteration against the roofline, priced with
        # ITS algorithmic bytes -- those of the textbook Jacobi-BiCGStab iteration (416 Nv) and of a Newton iteration with
        # k such iterations ((224 + 416 k) Nv).  The multigrid-preconditioned iteration moves several times more (two cycles
        # on top of the two products), so these fractions say how much time a Jacobi iteration's bytes would have been worth,
        # not how busy the memory system was.
        kits, nits = max(info.krylov_its, 1), max(info.newton_its, 1)
        t_kry = sum(prof[k]["ms"] for k in ("spmv", "vector", "amg_fine", "amg_coarse", "amg_first", "halo") if k in prof)
        t_all = sum(v["ms"] for v in prof.values())
        whole = {
            "krylov_iteration": {"ms": t_kry / kits, "algorithmic_bytes": 416 * nv,
                                 "achieved": 416 * nv / (t_kry / kits * 1e-3) / 1e9 if t_kry > 0 else 0.0},
            "newton_iteration": {"ms": t_all / nits, "krylov_its_per_newton": kits / nits,
                                 "algorithmic_bytes": (224 + 416 * kits / nits) * nv,
                                 "achieved": (224 + 416 * kits / nits) * nv / (t_all / nits * 1e-3) / 1e9 if t_all > 0 else 0.0},
            "note": "SURVEY.md 8d's algorithmic bytes of Jacobi-BiCGStab over the time of the multigrid-preconditioned "
                    "iteration of the profiled step (summed per-phase hipEvent durations)"}
        for v in (whole["krylov_iteration"], whole["newton_iteration"]):
            v["frac"] = v["achieved"] / peak_gbs
        return {
            "whole_iterations": whole,
            "bound": "hbm", "kernel": d["kernel"], "achieved": d["achieved"], "peak": peak_gbs, "unit": "GB/s",
            "frac": d["frac"], "traffic": d["traffic"], "bytes_per_launch": d["bytes_per_launch"],
            "traffic_source": pmc.get("source"),
            "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
            "kernels": legs,
            "spmv_back_to_back": b2b,
            "assembly_plus_spmv": inner,
            "phase_ms": {k: v["ms"] for k, v in prof.items()},
            "profiled_step": {"newton_its": info.newton_its, "krylov_its": info.krylov_its},
        }

    def close(self):
        self.ctx.close()


class PartitionedRunner(SingleRunner):
    """Same workload, vertices split over `world` subdomains (one process / GPU each): every rank builds
    the global synthetic mesh, keeps its subdomain and joins the communicator."""

    def __init__(self, rank, world, device, config="c4_10m", order="morton", dt=3600.0, storage=False,
                 moulins=0, krylov_rtol=1e-10, shape=None, transport="rccl", group=None, precond="amg", basin=0):
        from .distributed import make_context
        from .partition import partition

        nx, ny, Lx, Ly = CONFIGS[config] if shape is None else shape
        dom = basin_mesh(basin, order="random") if basin else rectangle_mesh(nx, ny, Lx, Ly, order=order)
        self.dt = dt
        sf = synthetic_fields(dom, storage_on=storage, moulins=moulins)
        bc_global = locate_boundary_dofs(dom, (lambda X: X[0] < 1e-9) if basin else outflow_predicate(dom))
        self.sub = sub = partition(dom, world, rank)
        self.nv_global, self.ne_global = dom.num_vertices, dom.num_cells
        g = sub.gid
        self.ctx = c = make_context(sub, device, transport, group)
        c.set_params(krylov_rtol=krylov_rtol, precond=_lib.PRECOND[precond])
        c.set_field("z_b", sf["z_b"][g]); c.set_field("z_s", sf["z_s"][g]); c.set_field("G", sf["G"][g])
        c.set_field("inputs", sf["inputs"][g]); c.set_field("storage", sf["lake_bdry"][g])
        c.set_field("b", np.abs(sf["b_init"][g]))
        c.set_field("N_n", sf["N_init"][g]); c.set_field("N", sf["N_init"][g])
        c.set_field("q", sf["q_init"][g]); c.set_field("melt_n", np.zeros(sub.n_loc))
        g2l = np.full(dom.num_vertices, -1, dtype=np.int64)
        g2l[g] = np.arange(g.size)
        loc = g2l[bc_global]
        self.bc = loc[loc >= 0].astype(np.int32)
        c.set_dirichlet(self.bc, N_BDRY)
        self.stats = c.plan_stats()
        # nnz of the global matrix = sum over subdomains of the owned rows' entries
        import torch
        import torch.distributed as dist
        t = torch.tensor([self.stats["nnz"]], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            t = t.cuda(device)
        dist.all_reduce(t, group=group)
        self.nnz_global = int(t.item())
        self.transport = transport
        mesh_txt = (f"Delaunay basin mesh of {dom.num_vertices} vertices (hole, curved outlet, random order)" if basin else
                    f"{Lx/1e3:.0f} km x {Ly/1e3:.0f} km rectangle, {nx}x{ny} jittered P1 mesh ({order} order)")
        self._desc = (f"{mesh_txt}, dt {dt:g} s (first step 0.1 dt), storage {'on' if storage else 'off'}, {moulins} moulins, "
                      f"Dirichlet N = {N_BDRY:g} Pa on x = 0; |b_init| for the reference's signed draw; {world} subdomains (RCB), "
                      f"{transport} halo")
        self.next_step = 0
        del dom, sf

    def roofline(self, peak_gbs: float) -> dict:
        # per-subdomain kernel: algorithmic bytes of THIS rank's SpMV launch
        nv, nnz = self.stats["nv"], self.stats["nnz"]
        save = (self.nv_global, self.nnz_global, self.ne_global)
        self.nv_global, self.nnz_global, self.ne_global = nv, nnz, self.stats["ne"]
        try:
            out = super().roofline(peak_gbs)
        finally:
            self.nv_global, self.nnz_global, self.ne_global = save
        out["kernel"] += f" on rank 0's subdomain ({nv} owned rows)"
        return out


def make_runner(args, rank: int, world: int, local_rank: int):
    """The runner bench.py drives.  With several ranks the transport is what was asked for (--transport): if RCCL is
    requested and cannot be used, this RAISES (on every rank: the failure is agreed on collectively) -- unless
    --allow-host-staged was given, in which case the same solver runs over the host-staged gloo transport and
    `run.transport` says so (a PCIe-staged number must never pass for an xGMI one)."""
    if world == 1:
        return SingleRunner(args.config, args.order, args.dt, bool(args.storage), args.moulins, local_rank,
                            args.krylov_rtol, precond=getattr(args, "precond", "amg"))
    transport = getattr(args, "transport", "rccl")
    kw = dict(precond=getattr(args, "precond", "amg"))
    failure = None
    try:
        return PartitionedRunner(rank, world, local_rank, args.config, args.order, args.dt, bool(args.storage),
                                 args.moulins, args.krylov_rtol, transport=transport, **kw)
    except _lib.ShaktiHipError as exc:
        if transport != "rccl" or "RCCL communicator" not in str(exc) or not getattr(args, "allow_host_staged", False):
            raise
        failure = str(exc)   # leave the except block first: the traceback keeps the failed context's frames alive
    import sys
    import torch.distributed as dist
    print(f"[bench] WARNING rank {rank}: {failure}; --allow-host-staged: continuing with the host-staged gloo transport",
          file=sys.stderr, flush=True)
    group = dist.new_group(backend="gloo")
    return PartitionedRunner(rank, world, local_rank, args.config, args.order, args.dt, bool(args.storage),
                             args.moulins, args.krylov_rtol, transport="gloo", group=group, **kw)
