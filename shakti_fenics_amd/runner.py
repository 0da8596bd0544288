"""Benchmark / smoke driver around one device context: builds a synthetic configuration
(SURVEY.md section 8d), keeps every field resident in HBM and advances the solve loop of
`/root/reference/source/solvers.py:168-229` one time step per call."""
from __future__ import annotations

import numpy as np

from . import _lib
from .bc import locate_boundary_dofs
from .mesh import basin_mesh, rectangle_mesh
from .synthetic import BASIN_CONFIGS, CONFIGS, N_BDRY, config_mesh, outflow_predicate, synthetic_fields


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
        out["assemble_valu_insts"] = k["assemble"].get("valu_wave_insts")
        out["assemble_wait_any_frac"] = k["assemble"].get("wait_any_frac")
        out["assemble_residual_valu_insts"] = k.get("assemble_residual_only", {}).get("valu_wave_insts")
    return out


class SingleRunner:
    def __init__(self, config="c4_10m", order="morton", dt=3600.0, storage=False, moulins=0, device=0,
                 krylov_rtol=1e-10, shape=None, precond="amg", basin=0):
        self.config_name = config if shape is None and not basin else None
        # basin=n: the unstructured Delaunay basin mesh of ~n vertices (mesh.basin_mesh) instead of the rectangle;
        # config "basin_*": the same kind of mesh on the bench footprint (synthetic.BASIN_CONFIGS)
        if shape is None and not basin and config in BASIN_CONFIGS:
            self.dom = config_mesh(config)
            basin, (nx, ny, Lx, Ly) = self.dom.num_vertices, (0, 0) + BASIN_CONFIGS[config][1:]
        else:
            nx, ny, Lx, Ly = CONFIGS[config] if shape is None else shape
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
        mesh_txt = (f"Delaunay basin mesh of {self.dom.num_vertices} vertices (graded, hole, curved outlet, valence-13 stars, random "
                    f"vertex order), rows up to {st['max_row_len']} entries" if basin else
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
        """Per-launch hipEvent timing of one more step on the library's stream (events ride on the dispatch packets).
        Headline = the dominant kernel, the SELL-64 SpMV `k_spmv<1|2>`, priced with the bytes it HAS to move (padded
        slots, 16-bit columns, float x: shk_profile.bytes, accounted by the library per launch); SURVEY.md 8d's
        algorithmic figure (fp64 values, int32 columns, double x) stands beside it.  `step` = every launch of the profiled
        step: needed bytes over kernel time = whole-step HBM utilisation.  The assembly is graded against the ceiling that
        binds it (fp64 VALU issue, from the committed counter file) as well as against HBM."""
        c = self.ctx
        c.profile_enable(True)
        c.profile_read(reset=True)
        import time
        c.sync()
        t0 = time.perf_counter()
        info = self.step()
        c.sync()
        wall_ms = 1e3 * (time.perf_counter() - t0)
        prof = c.profile_read(reset=True)
        c.profile_enable(False)
        nv, nnz, slices = self.nv_global, self.nnz_global, (self.nv_global + 63) // 64
        ph = {k: v for k, v in prof.items() if k in _lib.PHASES}   # (prof also carries the aggregate "amg_coarse")
        # SURVEY.md 8d's algorithmic bytes per launch (fp64 values, int32 indices)
        alg = {"spmv": 12 * nnz + 4 * (nv + 1) + 16 * nv,                          # 104 nv
               "assemble": 12 * self.ne_global + 16 * nv + 88 * nv + 8 * nv + 8 * nnz,  # 192 nv
               "amg_fine": 8 * nnz + 4 * (slices + 1) + 20 * nv,                   # float values + int32 indices + 5 vectors
               "amg_first": 8 * c.plan_stats()["ap_nnz"] + 4 * (slices + 1) + 21 * nv}
        names = {"spmv": "k_spmv<1|2> (SELL-64 SpMV + fused BiCGStab dots)", "assemble": "k_assemble (fused residual + Jacobian)",
                 "amg_fine": "k_amg_post<true> (finest-level multigrid smoother, SELL-64 SpMV)",
                 "amg_first": "k_amg_first<true> (first sweep on the A*P operator)"}
        pmc = _pmc_traffic(getattr(self, "config_name", None), nv)

        def leg(phase):
            n = max(ph[phase]["launches"], 1)
            ms = ph[phase]["ms"] / n
            need = ph[phase]["bytes"] / n
            ach = need / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            ach_alg = alg[phase] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": names[phase], "avg_launch_ms": ms, "launches": ph[phase]["launches"],
                    "bytes_per_launch": need, "achieved": ach, "frac": ach / peak_gbs,
                    "algorithmic": {"bytes_per_launch": alg[phase], "achieved": ach_alg, "frac": ach_alg / peak_gbs,
                                    "note": "SURVEY.md 8d's figure (fp64 values, int32 columns, double x): bytes this "
                                            "implementation does not all move"},
                    "traffic": pmc.get(phase)}

        legs = {k: leg(k) for k in ("spmv", "assemble", "amg_fine", "amg_first") if ph[k]["launches"]}
        if "assemble" in legs:
            a = legs["assemble"]
            # the two instances apart, each against ITS ceiling (the profiled step mixes them: two full passes and, when the
            # last Newton iteration was predicted, one residual-only pass): a few back-to-back launches of each
            c.assemble(self.dt)
            inst = {"full": (c.time_kernel("assemble", 3, self.dt), pmc.get("assemble_valu_insts")),
                    "residual_only": (c.time_assemble_residual(3, self.dt), pmc.get("assemble_residual_valu_insts"))}
            c.assemble(self.dt)
            per = {}
            for name, (ms, valu) in inst.items():
                floor_ms = valu * 4.0 / (1024 * 2.4e9) * 1e3 if valu else None   # 4 cycles per fp64 wave-instruction, 1024 SIMDs
                per[name] = {"avg_launch_ms": ms, "valu_wave_insts": valu, "valu_issue_floor_ms": floor_ms,
                             "frac": (floor_ms / ms) if floor_ms and ms > 0 else None}
            a.update(bound="fp64 valu", hbm_frac=a["frac"], frac=per["full"]["frac"], instances=per,
                     valu_wave_insts=per["full"]["valu_wave_insts"], valu_issue_floor_ms=per["full"]["valu_issue_floor_ms"],
                     wait_any_frac=pmc.get("assemble_wait_any_frac"),
                     bound_note="k_assemble is bound by fp64 vector issue, not HBM: frac = (VALU wave-instructions x 4 cycles / "
                                "(1024 SIMDs x 2.4 GHz)) / time of the FULL instance launched back to back; `instances` has the "
                                "residual-only instance too; hbm_frac = needed bytes / mean time of the profiled step's passes / 8 TB/s"
                                + ("" if per["full"]["frac"] else " (no counter file for these kernel sources: frac is null)"))
        dom = max((k for k in legs if k != "assemble"), key=lambda k: ph[k]["ms"])
        d = legs[dom]
        # whole step: every launch the profiled step made
        tot_ms = sum(v["ms"] for v in ph.values())
        tot_b = sum(v["bytes"] for v in ph.values())
        step = {"definition": "needed bytes of EVERY launch of the profiled step (shk_profile.bytes) / their summed hipEvent "
                              "durations / peak: whole-step HBM utilisation while a kernel runs; wall_ms adds launch gaps "
                              "and host synchronisations",
                "bytes": tot_b, "kernel_ms": tot_ms, "wall_ms": wall_ms,
                "achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0,
                "achieved_wall": tot_b / (wall_ms * 1e-3) / 1e9 if wall_ms > 0 else 0.0}
        step["frac"] = step["achieved"] / peak_gbs
        step["frac_wall"] = step["achieved_wall"] / peak_gbs
        phase_gbs = {k: (v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0) for k, v in ph.items() if v["launches"]}
        cycles = max(2 * info.krylov_its, 1)
        levels = {k: {"ms_per_cycle": ph[k]["ms"] / cycles, "launches_per_cycle": ph[k]["launches"] / cycles,
                      "gbs": phase_gbs.get(k, 0.0)}
                  for k in _lib.COARSE_PHASES if ph[k]["launches"]}
        # The same SpMV launched 20 times back to back between ONE event pair (cross-check of the per-launch events)
        c.assemble(self.dt)
        b2b_ms = c.time_kernel("spmv", 20)
        need_spmv = d["bytes_per_launch"] if dom == "spmv" else legs["spmv"]["bytes_per_launch"]
        st = c.storage_stats()   # what k_spmv<0> has to move: values + (16- or 32-bit) columns + slice descriptors + double x and y
        need_b2b = (8 * st["sell_slots"] + 2 * st["slots_with_16bit_columns"] + 4 * (st["sell_slots"] - st["slots_with_16bit_columns"])
                    + 16 * st["slices"] + 16 * nv)
        b2b = {"kernel": "k_spmv<0> (same product without the fused dots, double x), 20 launches between one hipEvent pair",
               "avg_launch_ms": b2b_ms, "bytes_per_launch": alg["spmv"],
               "achieved": alg["spmv"] / (b2b_ms * 1e-3) / 1e9, "frac": alg["spmv"] / (b2b_ms * 1e-3) / 1e9 / peak_gbs,
               "needed": {"bytes_per_launch": need_b2b, "achieved": need_b2b / (b2b_ms * 1e-3) / 1e9,
                          "frac": need_b2b / (b2b_ms * 1e-3) / 1e9 / peak_gbs},
               "note": "`frac` prices this leg with SURVEY.md 8d's algorithmic bytes (as rounds 1-2 did); `needed` with the bytes "
                       "the kernel has to move, the pricing of `roofline.frac`"}
        # north_star states its target on the "assembly + SpMV inner loop"
        t_in = ph["assemble"]["ms"] + ph["spmv"]["ms"]
        inner = {"definition": "all k_assemble + k_spmv launches of the profiled step: bytes / summed hipEvent durations",
                 "needed": {"achieved": (ph["assemble"]["bytes"] + ph["spmv"]["bytes"]) / (t_in * 1e-3) / 1e9 if t_in > 0 else 0.0},
                 "algorithmic": {"achieved": (alg["assemble"] * ph["assemble"]["launches"] + alg["spmv"] * ph["spmv"]["launches"])
                                 / (t_in * 1e-3) / 1e9 if t_in > 0 else 0.0},
                 "launches": {"assemble": ph["assemble"]["launches"], "spmv": ph["spmv"]["launches"]},
                 "note": "a mix figure: the assembly is bound by fp64 issue (~14 % of the HBM roofline), the product runs at "
                         "~65-80 % depending on the box, so the aggregate falls as a step needs fewer Krylov iterations per assembly"}
        for v in (inner["needed"], inner["algorithmic"]):
            v["frac"] = v["achieved"] / peak_gbs
        inner["achieved"], inner["frac"] = inner["needed"]["achieved"], inner["needed"]["frac"]
        # a whole Krylov / Newton iteration priced with the textbook Jacobi-BiCGStab bytes
        kits, nits = max(info.krylov_its, 1), max(info.newton_its, 1)
        t_kry = sum(ph[k]["ms"] for k in ph if k in ("spmv", "vector", "amg_fine", "amg_first", "halo") or k in _lib.COARSE_PHASES)
        whole = {
            "krylov_iteration": {"ms": t_kry / kits, "algorithmic_bytes": 416 * nv,
                                 "achieved": 416 * nv / (t_kry / kits * 1e-3) / 1e9 if t_kry > 0 else 0.0},
            "newton_iteration": {"ms": tot_ms / nits, "krylov_its_per_newton": kits / nits,
                                 "algorithmic_bytes": (224 + 416 * kits / nits) * nv,
                                 "achieved": (224 + 416 * kits / nits) * nv / (tot_ms / nits * 1e-3) / 1e9 if tot_ms > 0 else 0.0},
            "note": "SURVEY.md 8d's algorithmic bytes of Jacobi-BiCGStab over the time of the multigrid-preconditioned "
                    "iteration of the profiled step (the multigrid iteration moves several times those bytes: see `step`)"}
        for v in (whole["krylov_iteration"], whole["newton_iteration"]):
            v["frac"] = v["achieved"] / peak_gbs
        return {
            "bound": "hbm", "kernel": d["kernel"], "achieved": d["achieved"], "peak": peak_gbs, "unit": "GB/s",
            "frac": d["frac"], "bytes_per_launch": d["bytes_per_launch"],
            "bytes_note": "bytes the kernel HAS to move per launch (padded SELL slots, 16-bit columns where the slice allows, "
                          "float x of the multigrid cycle, double y and dot operands); `algorithmic` prices the same time with "
                          "SURVEY.md 8d's figure",
            "algorithmic": d["algorithmic"],
            "traffic": d["traffic"], "traffic_source": pmc.get("source"),
            "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
            "kernels": legs,
            "step": step,
            "coarse_levels": levels,
            "phase_ms": {k: v["ms"] for k, v in prof.items()},
            "phase_gbs": phase_gbs,
            "whole_iterations": whole,
            "spmv_back_to_back": b2b,
            "assembly_plus_spmv": inner,
            "profiled_step": {"newton_its": info.newton_its, "krylov_its": info.krylov_its},
        }

    def close(self):
        self.ctx.close()


class PartitionedRunner(SingleRunner):
    """Same workload, vertices split over `world` subdomains (one process / GPU each).  The global synthetic mesh is
    built, partitioned and cut ONCE, on rank 0, which leaves every rank's subdomain (mesh, halo plan, field slices,
    Dirichlet dofs) as one .npz in a scratch directory of the node (bench.py runs on ONE node): the other ranks never
    hold the 10M-vertex mesh (round 2: every rank generated and partitioned all of it, 17 s and several GB x world).
    Before the first solve the data path is exercised once (distributed.startup_check)."""

    def __init__(self, rank, world, device, config="c4_10m", order="morton", dt=3600.0, storage=False,
                 moulins=0, krylov_rtol=1e-10, shape=None, transport="rccl", group=None, precond="amg", basin=0):
        import torch
        import torch.distributed as dist
        from .distributed import make_context, scatter_subdomains, startup_check

        self.dt = dt

        def make_global():
            nonlocal basin
            if shape is None and not basin and config in BASIN_CONFIGS:
                dom = config_mesh(config)
                basin, (nx, ny, Lx, Ly) = dom.num_vertices, (0, 0) + BASIN_CONFIGS[config][1:]
            else:
                nx, ny, Lx, Ly = CONFIGS[config] if shape is None else shape
                dom = basin_mesh(basin, order="random") if basin else rectangle_mesh(nx, ny, Lx, Ly, order=order)
            sf = synthetic_fields(dom, storage_on=storage, moulins=moulins)
            sf["b_init"] = np.abs(sf["b_init"])
            bc_global = locate_boundary_dofs(dom, (lambda X: X[0] < 1e-9) if basin else outflow_predicate(dom))
            mesh_txt = (f"Delaunay basin mesh of {dom.num_vertices} vertices (graded, hole, curved outlet, random order)" if basin else
                        f"{Lx/1e3:.0f} km x {Ly/1e3:.0f} km rectangle, {nx}x{ny} jittered P1 mesh ({order} order)")
            return dom, sf, bc_global, dict(mesh=mesh_txt)

        self.sub, f, self.bc, info = scatter_subdomains(rank, world, make_global, group)
        sub = self.sub
        self.nv_global, self.ne_global = info["nv"], info["ne"]
        self.device = device
        self.ctx = c = make_context(sub, device, transport, group)
        self.wiring = startup_check(c, sub, group)   # one exchange + one all-reduce with value checks, before any solve
        self.wiring["device"] = int(device)
        c.set_params(krylov_rtol=krylov_rtol, precond=_lib.PRECOND[precond])
        c.set_field("z_b", f["z_b"]); c.set_field("z_s", f["z_s"]); c.set_field("G", f["G"])
        c.set_field("inputs", f["inputs"]); c.set_field("storage", f["lake_bdry"])
        c.set_field("b", f["b_init"])
        c.set_field("N_n", f["N_init"]); c.set_field("N", f["N_init"])
        c.set_field("q", f["q_init"]); c.set_field("melt_n", np.zeros(sub.n_loc))
        c.set_dirichlet(self.bc, N_BDRY)
        self.stats = c.plan_stats()
        # nnz of the global matrix = sum over subdomains of the owned rows' entries
        t = torch.tensor([self.stats["nnz"]], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            t = t.cuda(device)
        dist.all_reduce(t, group=group)
        self.nnz_global = int(t.item())
        self.transport = transport
        self._desc = (f"{info['mesh']}, dt {dt:g} s (first step 0.1 dt), storage {'on' if storage else 'off'}, {moulins} moulins, "
                      f"Dirichlet N = {N_BDRY:g} Pa on x = 0; |b_init| for the reference's signed draw; {world} subdomains (RCB), "
                      f"{transport} halo")
        self.next_step = 0

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
