"""`model_setup`: the parameter / field container of `/root/reference/source/model_setup.py:18-119`,
attribute for attribute, over NumPy-backed functions; `md.solve()` runs the MI355X solve loop."""
from __future__ import annotations

import numpy as np
from scipy.interpolate import RegularGridInterpolator

from .fem import Function, functionspace


def _points_in_polygon(px, py, poly) -> np.ndarray:
    """Even-odd rule, vectorised over points; poly is (m,2) (closed or not)."""
    poly = np.asarray(poly, dtype=np.float64)
    x0, y0 = poly[:, 0], poly[:, 1]
    x1, y1 = np.roll(x0, -1), np.roll(y0, -1)
    inside = np.zeros(px.shape, dtype=bool)
    for a, b, c, d in zip(x0, y0, x1, y1):
        if b == d:
            continue
        hit = ((b > py) != (d > py)) & (px < (c - a) * (py - b) / (d - b) + a)
        inside ^= hit
    return inside


class model_setup:
    def __init__(self, comm, domain):
        # communicator (model_setup.py:21-23)
        self.comm = comm
        self.rank = comm.Get_rank()
        self.size = comm.Get_size()

        # domain, mesh, function spaces (model_setup.py:26-32)
        self.domain = domain
        self.x = domain.geometry.x[:, 0]
        self.y = domain.geometry.x[:, 1]
        self.V = functionspace(domain, ("CG", 1))
        self.V_flux = functionspace(domain, ("P", 1, (domain.geometry.dim,)))
        self.mask = self.ghost_mask(self.V)
        self.OutflowBoundary = None

        # bounding box for interpolating gridded data (model_setup.py:35-37)
        buffer = self.get_buffer()
        self.bounds = [self.x.min() - buffer, self.x.max() + buffer,
                       self.y.min() - buffer, self.y.max() + buffer]

        # boundary-condition switches (model_setup.py:40-41)
        self.outflow_on = True
        self.storage_on = True

        # physical input functions (model_setup.py:44-53)
        self.z_b = Function(self.V)
        self.z_s = Function(self.V)
        self.G = Function(self.V)
        self.inputs = Function(self.V)
        self.b_init = Function(self.V)
        self.N_init = Function(self.V)
        self.q_init = Function(self.V_flux)
        self.lake_bdry = Function(self.V)
        self.N_bdry = 0.0
        self.b_min = 1.0e-5

        self.outline = None
        self.lake_name = None
        self.results_name = None
        self.setup_name = None
        self.setup_file = None     # optional: path copied next to the results (solvers.py:125)

        self.timesteps = None
        self.nt_save = None
        self.nt_check = None

        # solver knobs of this build (reference: DOLFINx defaults + exact LU)
        self.device = 0
        self.krylov_rtol = 1e-10
        self.krylov_max_it = 20000
        self.preconditioner = "amg"      # "amg" (default) | "amg_local" | "jacobi" (north_star's solver)

    def set_lake_bdry(self, outline):
        """1 inside the lake outline, 0 outside (model_setup.py:68-72).  `outline` is an (m,2) polygon,
        or any object with the GeoDataFrame-style `.geometry.contains(point)` the reference passes."""
        if hasattr(outline, "geometry"):
            from shapely import Point  # only needed for GeoDataFrame outlines
            for j in range(self.lake_bdry.x.array.size):
                p = Point(self.x[j], self.y[j])
                self.lake_bdry.x.array[j] = outline.geometry.contains(p).iloc[0]
        else:
            self.lake_bdry.x.array[:] = _points_in_polygon(self.x, self.y, outline)
        self.lake_bdry.x.scatter_forward()

    def interp_data(self, var_name, x_d, y_d, f):
        """Bilinear interpolation of gridded data f[y, x] to the mesh nodes into `self.<var_name>`
        (model_setup.py:74-91); returns the interpolator, as the reference does."""
        x_d, y_d, f = np.asarray(x_d), np.asarray(y_d), np.asarray(f)
        mx = (x_d >= self.bounds[0]) & (x_d <= self.bounds[1])
        my = (y_d >= self.bounds[2]) & (y_d <= self.bounds[3])
        f_interp = RegularGridInterpolator((x_d[mx], y_d[my]), f[np.ix_(my, mx)].T, bounds_error=False,
                                           fill_value=None)
        target = self
        *path, last = f"{var_name}.x.array".split(".")
        for name in path:
            target = getattr(target, name)
        getattr(target, last)[:] = f_interp(np.column_stack((self.x, self.y)))
        getattr(self, var_name.split(".")[0]).x.scatter_forward()
        return f_interp

    def get_buffer(self):
        """10 x the largest gap between sorted node coordinates (model_setup.py:93-106)."""
        x__ = self.comm.gather(self.x[self.mask], root=0)
        y__ = self.comm.gather(self.y[self.mask], root=0)
        x_bfr = y_bfr = 0.0
        if self.rank == 0:
            x_bfr = 10 * np.max(np.diff(np.unique(np.concatenate(x__))))
            y_bfr = 10 * np.max(np.diff(np.unique(np.concatenate(y__))))
        self.comm.barrier()
        x_bfr, y_bfr = self.comm.bcast(x_bfr, root=0), self.comm.bcast(y_bfr, root=0)
        return max(x_bfr, y_bfr)

    def ghost_mask(self, V):
        """True for owned dofs (model_setup.py:108-116); host arrays are global here, so all True."""
        im = V.dofmap.index_map
        mask = np.ones(im.size_local + im.num_ghosts, dtype=bool)
        mask[im.global_to_local(im.ghosts)] = False
        return mask

    def solve(self):
        from .solvers import solve
        return solve(self)
