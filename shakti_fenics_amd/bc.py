"""Dirichlet dof search -- the stand-in for `locate_entities_boundary` + `locate_dofs_topological`
that `get_bcs` uses at `/root/reference/source/solvers.py:17-26`."""
from __future__ import annotations

import numpy as np


def locate_boundary_dofs(domain, predicate) -> np.ndarray:
    """Vertex dofs of the boundary facets whose vertices ALL satisfy `predicate(x)`.

    `predicate` receives a (3, npts) coordinate array and returns a boolean array, exactly like
    the `md.OutflowBoundary` callables of the reference (`setups/setup_cooke2.py:80`).
    """
    facets = domain.boundary_facets()
    if facets.size == 0:
        return np.zeros(0, dtype=np.int32)
    x3 = np.zeros((3, domain.num_vertices))
    x3[0] = domain.xy[:, 0]
    x3[1] = domain.xy[:, 1]
    ok = np.asarray(predicate(x3), dtype=bool)
    if ok.shape != (domain.num_vertices,):
        raise ValueError("boundary predicate must return one boolean per point")
    keep = ok[facets[:, 0]] & ok[facets[:, 1]]
    return np.unique(facets[keep].ravel()).astype(np.int32)
