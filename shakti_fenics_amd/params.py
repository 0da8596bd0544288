"""Physical constants of the SHAKTI model, importable by name (`from params import rho_i, rho_w, g`) exactly
like the reference's module of the same name; values per `/root/reference/source/params.py:4-11`.
The HIP library receives them through `shk_params` (include/shakti_hip.h)."""

_SI = {
    # name: (value, unit, meaning)
    "g": (9.81, "m s^-2", "gravitational acceleration"),
    "rho_i": (917, "kg m^-3", "density of ice"),
    "rho_w": (1000, "kg m^-3", "density of water"),
    "nu": (1.787e-6, "m^2 s^-1", "kinematic viscosity of water"),
    "Lh": (3.34e5, "J kg^-1", "latent heat of fusion"),
    "omega": (1e-3, "-", "laminar/turbulent transition parameter of the discharge law"),
    "n": (3, "-", "Glen flow-law exponent"),
    "A": (2.24e-24, "Pa^-n s^-1", "Glen flow-law coefficient"),
}

globals().update({name: entry[0] for name, entry in _SI.items()})
__all__ = sorted(_SI)

