#!/usr/bin/env python3
"""CPU analogue of BASELINE config 5 with the LU oracle (the reference's algorithmic configuration): storage + moulins on
an nx x ny mesh of the 100 km x 20 km geometry.  Per step: Newton residual history, and the sensitivity of F to ONE ulp
of N (random sign, Dirichlet dofs excluded) -- rounding N - dx to fp64 perturbs N by rms ulp / sqrt(12), so `est-rounding`
is the floor of ||F|| for ANY solver that stores N in fp64.  profiles/r02_oracle_lu_c5_analog_62k.log shows the LU path
ending every step on that floor (e.g. step 39: 3.5e-12 against 3.8e-12), growing with b_max like the GPU run's.
    python tools/oracle_c5_analog.py 560 112 12 40"""
import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, os.path.join(R,'oracle')]
import numpy as np, shakti_oracle as O
from shakti_fenics_amd.mesh import rectangle_mesh
from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields
nx,ny=int(sys.argv[1]),int(sys.argv[2]); moul=int(sys.argv[3]); steps=int(sys.argv[4])
dom=rectangle_mesh(nx,ny,100e3,20e3)
sf=synthetic_fields(dom,storage_on=True,moulins=moul)
nv=dom.num_vertices
f=O.Fields(N=sf["N_init"].copy(),N_n=sf["N_init"].copy(),b=np.abs(sf["b_init"]),q=sf["q_init"].copy(),melt_n=np.zeros(nv),
  z_b=sf["z_b"],z_s=sf["z_s"],G=sf["G"],storage=sf["lake_bdry"],inputs=sf["inputs"])
bc=O.boundary_dofs(dom.xy,dom.cells,outflow_predicate(dom))
prm=O.Params(); last,_=O.last_cell_of_vertex(nv,dom.cells)
rng=np.random.default_rng(0)
for i in range(steps):
    dt=360. if i==0 else 3600.
    t=time.time()
    n,conv,info=O.newton_solve(dom.xy,dom.cells,f,dt,prm,bc,N_BDRY)
    r=info["residuals"]
    # sensitivity of F to the fp64 representation of N: perturb by +-0.5 ulp
    F0,_=O.assemble(dom.xy,dom.cells,f,dt,prm,bc,N_BDRY,want_jacobian=False)
    g=f.copy(); sg=rng.choice([-1.0,1.0],nv); sg[bc]=0; g.N=f.N+np.spacing(f.N)*sg
    F1,_=O.assemble(dom.xy,dom.cells,g,dt,prm,bc,N_BDRY,want_jacobian=False)
    print(f"step {i} newton {n} conv {conv} r0 {r[0]:.3e} r {r[-1]:.3e} rel {r[-1]/r[0]:.2e} all {['%.1e'%v for v in r]} ulpN-sens(1ulp) {np.linalg.norm(F1-F0):.2e} est-rounding {np.linalg.norm(F1-F0)/12**0.5:.2e} bmax {f.b.max():.3f} {time.time()-t:.1f}s",flush=True)
    if not conv: break
    O.update_explicit(dom.xy,dom.cells,f,dt,prm,last)
