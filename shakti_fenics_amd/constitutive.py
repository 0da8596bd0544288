"""Point-wise (NumPy) forms of the closures the reference writes in UFL
(`/root/reference/source/constitutive.py:6-39`).  The HIP kernels evaluate the same expressions at
quadrature points / vertices; these host versions serve setup scripts and post-processing
(e.g. the potential-based outflow predicate of `setups/setup_cooke2.py:72-80`)."""
from __future__ import annotations

import numpy as np

from .params import A, Lh, g, n, nu, omega, rho_i, rho_w


def Head(N, z_b, z_s):
    """hydraulic head [m] (constitutive.py:6-9)"""
    return z_b + (rho_i / rho_w) * (z_s - z_b) - N / (rho_w * g)


def Reynolds(q):
    """local Reynolds number; q has shape (..., 2) (constitutive.py:18-20)"""
    q = np.asarray(q)
    return np.sqrt(np.sum(q * q, axis=-1)) / nu


def WaterFlux(b, grad_h, Re):
    """water discharge [m^2/s] from gap height, head GRADIENT (..., 2) and Reynolds number (constitutive.py:11-16)"""
    k = -(np.abs(b) ** 3) * g / (12 * nu * (1 + omega * Re))
    return np.asarray(k)[..., None] * np.asarray(grad_h)


def Melt(q, grad_h, G, b_n, melt_n, grad_b, grad_melt):
    """melt rate [kg m^-2 s^-1] with the Warburton et al. diffusion term expanded cell-wise (constitutive.py:22-27)"""
    m0 = (G - rho_w * g * np.sum(np.asarray(q) * np.asarray(grad_h), axis=-1)) / Lh
    gb2 = np.sum(np.asarray(grad_b) ** 2, axis=-1)
    m_diff = (melt_n * gb2 + b_n * np.sum(np.asarray(grad_melt) * np.asarray(grad_b), axis=-1)) / (1 + gb2)
    return m0 + m_diff


def Closure(b, N):
    """viscous creep closure [m/s] (constitutive.py:29-31)"""
    return A * b * N * np.abs(N) ** (n - 1)


def BackgroundPotential(z_b, z_s):
    """hydraulic potential at zero effective pressure (constitutive.py:38-41)"""
    return rho_w * g * Head(0 * z_b, z_b, z_s)
