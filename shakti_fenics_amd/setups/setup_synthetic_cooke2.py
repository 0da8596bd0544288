"""Synthetic stand-in for `/root/reference/setups/setup_cooke2.py`: the same parameter values and call
pattern (N_bdry, initial conditions, time axis, save cadence) on a seeded analytic geometry, because the
reference's mesh file and BedMachine / ATL14 / GHF data are not shipped."""
import os
from pathlib import Path

import numpy as np

from shakti_fenics_amd.mesh import rectangle_mesh
from shakti_fenics_amd.model_setup import model_setup
from shakti_fenics_amd.params import g, rho_i, rho_w
from shakti_fenics_amd.synthetic import bed, surface


def initialize(comm, nx=71, ny=71, L=100e3, days=10.0 / 24.0, results_root=None, ingest="device", mesh_file=None):
    lake_name = "Synthetic_E2"
    if mesh_file:                                                       # setup_cooke2.py:19 (gmshio.read_from_msh)
        from shakti_fenics_amd.gmsh_io import read_msh
        domain = read_msh(mesh_file)
        L = float(max(domain.xy[:, 0].max(), domain.xy[:, 1].max()))
    else:
        domain = rectangle_mesh(nx, ny, L, L, jitter=0.25, seed=1234)   # C1 of SURVEY.md 8d
    md = model_setup(comm, domain)
    md.ingest = ingest   # "device": interp_data / set_lake_bdry on the GPU; "host": the reference's scipy path
    md.setup_name = os.path.splitext(os.path.basename(__file__))[0]
    md.setup_file = os.path.abspath(__file__)
    md.lake_name = lake_name
    md.N_bdry = 3.7e5                                                   # setup_cooke2.py:29
    root = Path(results_root) if results_root else Path(__file__).resolve().parent.parent.parent / "results"
    md.results_name = f"{root}/{md.lake_name}_{int(md.N_bdry/1e3):d}kpa"

    # lake outline: a disc of radius 5 km at the domain centre, as a polygon
    th = np.linspace(0, 2 * np.pi, 65)[:-1]
    cx, cy = 0.5 * (domain.xy[:, 0].min() + domain.xy[:, 0].max()), 0.5 * (domain.xy[:, 1].min() + domain.xy[:, 1].max())
    r_lake = min(5e3, 0.2 * L)
    md.outline = np.column_stack((cx + r_lake * np.cos(th), cy + r_lake * np.sin(th)))
    md.set_lake_bdry(md.outline)

    # geometry on regular grids, pushed through the same interp_data path as the reference's NetCDF data
    xg = np.linspace(-5e3, L + 5e3, 441)
    yg = np.linspace(-5e3, L + 5e3, 441)
    X, Y = np.meshgrid(xg, yg)                                           # f[y, x]
    bed_interp = md.interp_data("z_b", xg, yg, bed(X, Y))
    h_interp = md.interp_data("z_s", xg, yg, surface(X, Y))
    _ = md.interp_data("G", xg, yg, np.full_like(X, 0.05))

    # initial conditions (setup_cooke2.py:66-69), seeded, entering the flux law through |b|^3
    rng = np.random.default_rng(0)
    md.b_init.x.array[:] = np.abs(0.001 + rng.normal(scale=0.005, size=np.size(md.b_init.x.array[:])))
    md.N_init.interpolate(lambda x: md.N_bdry + 0 * x[0])
    md.q_init.sub(0).interpolate(lambda x: 0 * x[0])
    md.q_init.sub(1).interpolate(lambda x: 0 * x[0])

    # outflow where the hydraulic potential is within half a standard deviation of its minimum
    # (setup_cooke2.py:72-80)
    potential_interp = lambda x, y: rho_i * g * h_interp((x, y)) + (rho_w - rho_i) * g * bed_interp((x, y))
    P = potential_interp(md.x, md.y)
    P_min, P_std = np.min(P), np.std(P)
    md.OutflowBoundary = lambda x: np.less(np.abs(potential_interp(x[0], x[1]) - P_min), 0.5 * P_std)

    md.outflow_on = True
    md.storage_on = True
    md.inputs.interpolate(lambda x: 0 * x[0])

    # time stepping: 24 steps per day as in the reference (setup_cooke2.py:92-95), `days` long
    nt_per_day = 24
    t_final = (days / 365) * 3.154e7
    md.timesteps = np.linspace(0, t_final, max(2, int(round(days * nt_per_day))))
    md.nt_save = 1
    md.nt_check = 50 * md.nt_save
    return md
