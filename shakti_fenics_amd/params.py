"""Physical constants of the SHAKTI model -- same names and values as
`/root/reference/source/params.py:4-11`, so setup scripts can `from params import rho_i, rho_w, g`."""
g = 9.81          # gravitational acceleration [m/s^2]
rho_i = 917       # ice density [kg/m^3]
rho_w = 1000      # density of water [kg/m^3]
nu = 1.787e-6     # water viscosity [m^2/s]
Lh = 3.34e5       # latent heat [J/kg]
omega = 1e-3      # laminar-turbulent transition parameter of the discharge law
n = 3             # Glen's flow law exponent
A = 2.24e-24      # Glen's flow law coefficient [Pa^-n s^-1]
