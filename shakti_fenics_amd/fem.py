"""The sliver of the DOLFINx object model that the reference's setup / solve code touches:
`functionspace`, `Function` with `.x.array`, `.x.scatter_forward()`, `.interpolate(...)`, `.sub(i)`
(`/root/reference/source/model_setup.py:29-51,71-72,89-90`, `setups/setup_cooke2.py:66-69,89`).
Data live in plain NumPy arrays; the device copies are managed by `solvers.py`."""
from __future__ import annotations

import numpy as np


class _IndexMap:
    def __init__(self, n):
        self.size_local = n
        self.size_global = n
        self.num_ghosts = 0
        self.ghosts = np.zeros(0, dtype=np.int64)

    def global_to_local(self, g):
        return np.asarray(g, dtype=np.int64)


class _DofMap:
    def __init__(self, n, bs):
        self.index_map = _IndexMap(n)
        self.index_map_bs = bs
        self.bs = bs


class _Element:
    def interpolation_points(self):
        return np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])  # P1: the reference-cell vertices


class FunctionSpace:
    """P1 Lagrange on a `mesh.Domain`; `shape=(2,)` gives the blocked vector space V_flux."""

    def __init__(self, domain, shape=()):
        self.mesh = domain
        self.shape = tuple(shape)
        self.bs = int(np.prod(shape)) if shape else 1
        self.dofmap = _DofMap(domain.num_vertices, self.bs)
        self.element = _Element()

    def tabulate_dof_coordinates(self):
        return self.mesh.geometry.x


def functionspace(domain, element):
    """`functionspace(domain, ("CG", 1))` or `functionspace(domain, ("P", 1, (2,)))`."""
    family, degree = element[0], element[1]
    if family not in ("CG", "P", "Lagrange") or degree != 1:
        raise NotImplementedError("only first-order Lagrange spaces exist in the SHAKTI hot path")
    return FunctionSpace(domain, element[2] if len(element) > 2 else ())


class _Vector:
    def __init__(self, n):
        self.array = np.zeros(n)

    def scatter_forward(self):
        """Ghost update: host arrays are global (replicated), so there is nothing to move."""
        return None


class _SubFunction:
    def __init__(self, parent, i):
        self._p, self._i = parent, i

    @property
    def values(self):
        return self._p.x.array[self._i::self._p.function_space.bs]

    def interpolate(self, f):
        self._p.x.array[self._i::self._p.function_space.bs] = _evaluate(f, self._p.function_space, scalar=True)


def _evaluate(f, V, scalar):
    if isinstance(f, _SubFunction):
        return f.values.copy()
    if isinstance(f, Function):
        return f.x.array.copy()
    X = V.mesh.geometry.x.T  # (3, npts), the layout DOLFINx hands to interpolation callables
    out = np.asarray(f(X), dtype=np.float64)
    n = V.mesh.num_vertices
    if scalar:
        return np.broadcast_to(out, (n,)).copy()
    return np.broadcast_to(out, (V.bs, n)).T.reshape(-1).copy()  # (bs, npts) -> blocked


class Function:
    def __init__(self, V, name=None):
        self.function_space = V
        self.name = name
        self.x = _Vector(V.mesh.num_vertices * V.bs)

    def interpolate(self, f):
        self.x.array[:] = _evaluate(f, self.function_space, scalar=self.function_space.bs == 1)

    def sub(self, i):
        if not 0 <= i < self.function_space.bs:
            raise IndexError(i)
        return _SubFunction(self, i)

    def copy(self):
        g = Function(self.function_space, self.name)
        g.x.array[:] = self.x.array
        return g


class Constant:
    """`dolfinx.fem.Constant(domain, value)`: a mutable scalar (`dt.value = ...`, solvers.py:82,176)."""

    def __init__(self, domain, value):
        self.value = float(value)

    def __float__(self):
        return float(self.value)
