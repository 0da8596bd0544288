/*
 * shakti_hip.h -- C ABI of the MI355X (gfx950) SHAKTI hot-path library, libshakti_hip.so.
 *
 * The reference has no native/FFI boundary: its solve loop is Python driving DOLFINx/PETSc
 * (/root/reference/source/solvers.py).  This header is the boundary this build introduces
 * *underneath* the reference's Python API; each entry point names the reference lines whose
 * work it replaces.  The Python mirror of the reference API that calls it lives in
 * shakti_fenics_amd/solvers.py (ctypes binding: shakti_fenics_amd/_lib.py, and INTEGRATION.md).
 *
 * Conventions
 *  - every function returns int: 0 = ok, negative = error (text via shk_last_error(), thread-local);
 *  - shk_ctx is opaque; one context per GPU (per subdomain); a context is not thread-safe;
 *  - the caller owns all host arrays (borrowed for the duration of the call, C-contiguous,
 *    float64 / int32); the library owns all device memory;
 *  - vertex arrays are in the CALLER's numbering (the library renumbers internally for locality) and
 *    have length nv (= owned + ghost for a subdomain); SHK_Q is length 2*nv interleaved [x0,y0,x1,y1,...] (DOLFINx's
 *    blocked P1-vector layout, solvers.py:130,139-140);
 *  - all arithmetic is float64 (PETSc.ScalarType, solvers.py:24).
 */
#ifndef SHAKTI_HIP_H
#define SHAKTI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct shk_ctx shk_ctx;

/* Physical constants = /root/reference/source/params.py:4-11; b_min = model_setup.py:53;
 * Newton knobs = the DOLFINx NewtonSolver defaults the reference never overrides
 * (solvers.py:52); Krylov knobs belong to this build (the reference uses LU). */
typedef struct shk_params {
    double g, rho_i, rho_w, nu, Lh, omega, n, A;
    double b_min;
    double newton_rtol, newton_atol, newton_relax;
    double krylov_rtol, krylov_atol;  /* target of every linear solve, on the TRUE residual ||F - J dx|| */
    double krylov_fail_rtol;    /* a linear solve whose true relative residual ends above this (or is not finite) is a
                                   FAILURE (shk_solve_info.krylov_failed); between krylov_rtol and this it has merely
                                   stagnated at its fp64 floor eps * || |J| |dx| ||, which Newton absorbs.  Default 1e-6 */
    double krylov_newton_eta;   /* inside shk_newton_solve a linear solve also stops once its true residual is below
                                   eta * max(newton_atol, newton_rtol ||F_0||): Newton's own stopping threshold, beyond
                                   which more digits cannot change its decision.  Default 0.1 (= krylov_rtol ||F_0|| for
                                   the first iteration); 0 = always solve to krylov_rtol ||F_k|| */
    double krylov_forcing;      /* inexact Newton inside shk_newton_solve (round 3).  The previous solve recorded, per Newton
                                   iteration k, the ratio ||F_{k+1}|| / ||F_k||; e_k = ratio_k ||F_k|| is therefore what iteration
                                   k of THIS solve is expected to leave behind however well its linear system is solved (the
                                   nonlinear remainder).  While e_k > 10 x Newton's stopping threshold -- another iteration
                                   will follow anyway -- the linear solve stops at a true residual of krylov_forcing * e_k
                                   instead of ten digits below ||F_k||: digits the next iteration recovers for free.  The
                                   iteration expected to END the solve is solved as tightly as before, and a solve is only cut
                                   short when the history fits (same dt, ||F_0|| within a factor 2), the regime has settled
                                   (the ratio of the last two solves within a factor 2), the previous solve ended with the
                                   iteration after this one and well inside its threshold, and that iteration's remainder,
                                   predicted from the previous solve's quadratic constant, lies >= 100 x below the threshold:
                                   the converged state and (in every test) the Newton counts are those of exact solves.
                                   Default 0.1; 0 = off */
    int32_t newton_max_it;
    int32_t krylov_max_it;
    int32_t krylov_check_every; /* iterations enqueued between host stop-flag polls; 0 = automatic */
    int32_t precond;            /* shk_precond; shk_default_params gives SHK_PC_JACOBI (north_star's solver), the Python
                                   mirror and bench.py select SHK_PC_AMG */
    int32_t krylov_warm_start;  /* 0..4: inside shk_newton_solve the linear solve of Newton iteration k (k < 3) starts from
                                   the least-squares combination of the solutions of iteration k of that many previous
                                   steps (minimising ||F - J sum c_j g_j||: it can only lower the starting residual) instead
                                   of zero.  The stopping rule (true residual against ||F_k||) is untouched.  Default 4;
                                   0 = every solve starts from zero, as in round 1 */
} shk_params;

/* Right preconditioner of BiCGStab.  JACOBI is folded into the matrix (A D^-1).  AMG = one V(0,2) cycle (two
 * damped-Jacobi sweeps on the finest level on the way up, four Chebyshev-damped ones on every coarser level, no
 * smoothing on the way down) of a static-pattern aggregation multigrid stored in float. */
enum shk_precond {
    SHK_PC_JACOBI = 0,
    SHK_PC_AMG = 1,        /* on a communicator of > 1 subdomains: distributed hierarchy, built collectively by the
                              first shk_set_params that selects it */
    SHK_PC_AMG_LOCAL = 2   /* multigrid of each subdomain's own diagonal block (additive Schwarz, no communication) */
};

enum shk_field {
    SHK_N = 0,       /* effective pressure, the Newton unknown          (solvers.py:129) */
    SHK_N_N = 1,     /* N at the previous time step                     (solvers.py:134) */
    SHK_B = 2,       /* gap height                                      (solvers.py:131) */
    SHK_Q = 3,       /* water flux, 2*nv interleaved                    (solvers.py:130) */
    SHK_Z_B = 4,     /* bed elevation                                   (model_setup.py:44) */
    SHK_Z_S = 5,     /* surface elevation                               (model_setup.py:45) */
    SHK_G = 6,       /* geothermal heat flux                            (model_setup.py:46) */
    SHK_MELT_N = 7,  /* lagged melt rate                                (solvers.py:156) */
    SHK_STORAGE = 8, /* lake storage indicator                          (solvers.py:147-152) */
    SHK_INPUTS = 9,  /* moulin inputs                                   (model_setup.py:47) */
    SHK_QX = 10,     /* x component of q, length nv                     (solvers.py:144) */
    SHK_QY = 11,     /* y component of q, length nv                     (solvers.py:145) */
    SHK_DX = 12,     /* last Newton increment (read-only diagnostic) */
    SHK_FIELD_COUNT = 13
};

typedef struct shk_solve_info {
    int32_t newton_its;      /* NewtonSolver.solve's niter              (solvers.py:179) */
    int32_t converged;       /* ... and its converged flag */
    int32_t krylov_its;      /* BiCGStab iterations summed over the Newton iterations */
    int32_t krylov_failed;   /* 1 if a linear solve ended above krylov_fail_rtol (max_it, breakdown, divergence) */
    double residual0;        /* ||F|| before the first Newton iteration */
    double residual;         /* ||F|| at exit */
    double krylov_relres;    /* largest true relative residual ||F - J dx|| / ||F|| a linear solve of this call ended on */
} shk_solve_info;

/* Device-side timings accumulated with hipEvents on the library's stream while profiling is on. */
enum shk_phase {
    SHK_PH_ASSEMBLE = 0, SHK_PH_SPMV = 1, SHK_PH_VECTOR = 2, SHK_PH_UPDATE = 3, SHK_PH_OTHER = 4,
    SHK_PH_HALO = 5, SHK_PH_AMG_FINE = 6 /* finest-level smoothing SpMV, k_amg_post<true> */,
    SHK_PH_AMG_COARSE = 7 /* multigrid kernels not named below (prolongations, gathers, the one-workgroup tail) */,
    SHK_PH_AMG_FIRST = 8 /* finest-level first sweep on the A*P operator, k_amg_first<true> */,
    SHK_PH_AMG_REP = 9 /* every kernel of the REPLICATED coarse hierarchy of a decomposed mesh: the part of a cycle
                          that does not shrink with the number of GPUs */,
    SHK_PH_AMG_RESTRICT = 10 /* the restrictions of the way down */,
    SHK_PH_AMG_DENSE = 11 /* the dense coarsest solve (GEMV with the kept inverse) */,
    SHK_PH_AMG_L1 = 12 /* the smoothing sweeps of coarse level 1; level l is SHK_PH_AMG_L1 + l - 1, levels >= 8 share
                          the last slot */,
    SHK_PH_AMG_L8 = 19, SHK_PH_COUNT = 20
};
typedef struct shk_profile {
    double ms[SHK_PH_COUNT];      /* summed launch durations per phase */
    int64_t launches[SHK_PH_COUNT];
    double bytes[SHK_PH_COUNT];   /* bytes those launches HAD to move (what this implementation streams: padded SELL slots,
                                     16-bit columns, float preconditioner data, every vector once per kernel that reads or
                                     writes it; re-reads served by caches are not counted) -- with ms: HBM utilisation */
} shk_profile;

const char* shk_last_error(void);
int shk_version(void);

/* Upload the mesh, build the P1 CSR pattern and the static assembly / SpMV plans.
 * Replaces mesh + function-space + matrix preallocation: setup_cooke2.py:19, model_setup.py:29-30,
 * solvers.py:51-52.  xy is (nv,2), cells is (ne,3); cell order defines "last cell wins". */
int shk_create(int device_id, int64_t nv, int64_t ne, const double* xy, const int32_t* cells, shk_ctx** out);
/* One subdomain of a domain-decomposed mesh (the DOLFINx distributed mesh: owned + ghost dofs,
 * model_setup.py:108-116).  Local vertices [0, n_own) are owned, [n_own, n_own+n_ghost) are ghosts;
 * `cells` lists, in ascending GLOBAL cell order, every cell that touches an owned vertex, in local ids.
 * All vertex arrays of this context have length n_own + n_ghost.  shk_create == n_ghost 0. */
int shk_create_local(int device_id, int64_t n_own, int64_t n_ghost, int64_t ne, const double* xy,
                     const int32_t* cells, shk_ctx** out);
int shk_destroy(shk_ctx* ctx);

/* ---- domain decomposition (replaces the MPI layer under DOLFINx/PETSc) ----
 * Halo plan of this subdomain: neighbour ranks (ascending); for neighbour k the owned local vertices
 * send_idx[send_ptr[k] .. send_ptr[k+1]) are sent to it, and its values arrive in ghost vertices
 * n_own + recv_ptr[k] .. n_own + recv_ptr[k+1] (ghosts are numbered by owner, in the owner's send order). */
int shk_set_halo(shk_ctx* ctx, int32_t n_nbr, const int32_t* nbr_rank, const int64_t* send_ptr,
                 const int32_t* send_idx, const int64_t* recv_ptr);
/* RCCL transport over xGMI: rank 0 creates a 128-byte id, every rank joins with it (the id travels by
 * whatever bootstrap the host has, e.g. torch.distributed broadcast). */
int shk_comm_unique_id(void* id128);
int shk_comm_init_rccl(shk_ctx* ctx, int32_t rank, int32_t nranks, const void* id128);
/* Host-staged transport through caller callbacks (gloo / MPI / tests).  exchange(...): for neighbour k send
 * send[send_ptr[k] .. send_ptr[k+1]) to rank nbr[k] and receive recv[recv_ptr[k] .. recv_ptr[k+1]) from it (the
 * layout is passed on every call because multigrid levels exchange with their own, smaller plans);
 * allreduce(user, buf, n): in-place element-wise sum over ranks.  Both return 0 on success. */
typedef int (*shk_exchange_fn)(void* user, int32_t n_nbr, const int32_t* nbr, const double* send,
                               const int64_t* send_ptr, double* recv, const int64_t* recv_ptr);
typedef int (*shk_allreduce_fn)(void* user, double* buf, int64_t n);
int shk_comm_init_callbacks(shk_ctx* ctx, int32_t rank, int32_t nranks, shk_exchange_fn exchange,
                            shk_allreduce_fn allreduce, void* user);
/* Loop-back test of the RCCL call sequence of the data path on this context's communicator (grouped ncclSend /
 * ncclRecv to the rank itself on the context's stream and again on a second, event-ordered stream, ncclAllReduce,
 * async-error query); 0 = passed. */
int shk_comm_selftest(shk_ctx* ctx);
/* Timing-only transport (measurement aid, any transport, AFTER the collective setup): with on != 0 every ghost
 * exchange and reduction across subdomains returns at once -- pack / unpack kernels still run, the message itself is
 * skipped, ghosts keep stale values and reductions stay local.  RESULTS ARE WRONG BY CONSTRUCTION, kernel durations are
 * those of the subdomain: one rank of a P-way decomposition can be timed alone on one GPU (tools/scaling_model.py).
 * shk_step / shk_newton_solve report what they computed; nothing in the product path turns this on. */
int shk_comm_set_timing_only(shk_ctx* ctx, int32_t on);
/* Time `reps` back-to-back RCCL message rounds of one kind on the context's stream: kind 0 = grouped ncclSend + ncclRecv
 * of n doubles with the neighbouring rank (rank ^ 1; the rank itself on a one-rank communicator), 1 = ncclAllReduce of n
 * doubles, 2 = in-place ncclAllGather of n bytes per rank.  Microseconds per round.  On a one-GPU box (one-rank
 * communicator) this is the software floor of a round -- launch + protocol, no link -- i.e. the lower bound of the
 * message cost the strong-scaling model of profiles/r03_scaling_model.md leaves free. */
int shk_comm_time_round(shk_ctx* ctx, int32_t kind, int64_t n, int32_t reps, double* us_per_round);
/* Start-up check of the reduction path (either transport): value[0] is summed over the subdomains through the SAME
 * all-reduce the Krylov loop uses, on the context's stream, and the host wait goes through the RCCL deadline
 * (SHK_COMM_TIMEOUT_S): a mis-wired communicator fails here with an error instead of hanging the first solve.  The ghost
 * exchange has its own check in shk_halo_update on a field of known values (shakti_fenics_amd/distributed.py). */
int shk_comm_allreduce_check(shk_ctx* ctx, double* value);
/* What the RCCL deadline does when it fires, exposed for hosts that detect a dead peer themselves (and for tests): marks
 * the context POISONED.  Every later call that would wait for the device fails at once, and shk_destroy returns without
 * synchronising, destroying the communicator (ncclCommAbort if available) or freeing device memory -- all of which
 * would block behind the stalled collective.  The process is expected to exit non-zero right after. */
int shk_comm_mark_stalled(shk_ctx* ctx);
/* Message rounds this context has issued since creation: n[0] ghost exchanges, n[1] all-reduces, n[2] bytes sent in
 * exchanges, n[3] bytes all-reduced (per rank), n[4] in-place all-gathers (the replicated multigrid level's right-hand
 * side, once per cycle, and its operator values, once per Newton iteration), n[5] bytes received in them.  Differences
 * around a solve give rounds per Krylov iteration. */
int shk_comm_stats(shk_ctx* ctx, int64_t n[6]);
/* Interior / boundary split of the finest level's sweeps (several subdomains; off unless the environment sets
 * SHK_OVERLAP=1 -- DESIGN.md section 5 says why): the ghost
 * exchange of the two Krylov products and of the finest smoothing sweep travels on a second stream while the SELL
 * slices without ghost columns are swept; the others follow once it has arrived.  n[0] 1 = active, n[1] slices that
 * read ghost columns, n[2] all slices of the subdomain, n[3] exchanges issued this way since creation. */
int shk_comm_overlap(shk_ctx* ctx, int64_t n[4]);
/* Refresh the ghost entries of a field from their owners (scatter_forward, solvers.py:197,229). */
int shk_halo_update(shk_ctx* ctx, int32_t field);

/* Experiment switches: every SHK_* environment variable the library honours is read once per process
 * (csrc/shk_tunables.h).  Copies "NAME=value, ..." of the ones that are SET into buf (NUL-terminated, truncated to cap)
 * and returns how many there are; 0 = the run uses the defaults every committed measurement was taken with.  None of
 * them changes what is computed.  buf may be NULL. */
int64_t shk_env_overrides(char* buf, int64_t cap);
/* Set one of those switches from code instead of the environment (tests, probes): name as in the environment
 * ("SHK_AMG_ALPHA"), value as its text; value NULL = back to the default.  Process-wide; contexts read the switches when
 * they are created (a few tuning values at use).  Recorded in shk_env_overrides like an environment override. */
int shk_tunable_set(const char* name, const char* value);

int shk_default_params(shk_params* p);
int shk_set_params(shk_ctx* ctx, const shk_params* p);
int shk_get_params(shk_ctx* ctx, shk_params* p);

/* Quadrature on the reference triangle, xyw = (nq,3) rows x,y,w with sum(w) = 1/2; nq <= 32.
 * Default: the 15-point degree-7 rule of csrc/shk_quadrature.h (stands in for Basix's table). */
int shk_set_quadrature(shk_ctx* ctx, int32_t nq, const double* xyw);

int shk_set_field(shk_ctx* ctx, int32_t field, const double* host);
int shk_get_field(shk_ctx* ctx, int32_t field, double* host);

/* get_bcs (solvers.py:17-26): constant Dirichlet value on the listed dofs; n = 0 clears (outflow_on False). */
int shk_set_dirichlet(shk_ctx* ctx, int64_t n, const int32_t* dofs, double value);

/* Residual + Jacobian of the weak form solvers.py:35-45 with DOLFINx's Dirichlet algebra, one fused
 * element pass.  Exposed for parity tests. */
int shk_assemble(shk_ctx* ctx, double dt);
int shk_get_residual(shk_ctx* ctx, double* host);
int shk_csr_nnz(shk_ctx* ctx, int64_t* nnz);
int shk_get_csr(shk_ctx* ctx, int32_t* rowptr, int32_t* colidx, double* values);

/* Solve J dx = F for the system last assembled (BiCGStab with the right preconditioner shk_params.precond selects,
 * wrapped in true-residual refinement); dx via SHK_DX.  converged = the true residual met krylov_rtol. */
int shk_linear_solve(shk_ctx* ctx, int32_t* its, int32_t* converged, double* rel_residual);

/* y = J x with the assembled Jacobian (unscaled) -- parity / roofline probe of the SpMV kernel. */
int shk_spmv(shk_ctx* ctx, const double* x_host, double* y_host);

/* NewtonSolver.solve(N) (solvers.py:179): updates SHK_N in place.  Residual first, then per iteration J / linear solve /
 * x <- x - dx / residual, converged at ||F|| < max(newton_atol, newton_rtol ||F_0||) (DOLFINx defaults).  Each linear
 * solve runs to a TRUE residual of max(krylov_rtol ||F_k||, krylov_atol, krylov_newton_eta x that threshold) and starts
 * from the projected solutions of the previous steps (krylov_warm_start); info reports Newton and Krylov counts, the
 * worst true relative residual (krylov_relres) and whether a linear solve failed (krylov_failed). */
int shk_newton_solve(shk_ctx* ctx, double dt, shk_solve_info* info);

/* q, melt_n, b interpolations + clamp + N_n <- N (solvers.py:186-197,228-229). */
int shk_update_explicit(shk_ctx* ctx, double dt);

/* One time step: shk_newton_solve then shk_update_explicit (solvers.py:179-197,228-229). */
int shk_step(shk_ctx* ctx, double dt, shk_solve_info* info);

int shk_sync(shk_ctx* ctx);

/* Profiling: per-launch hipEvent pairs on the library's stream, summed per phase. */
int shk_profile_enable(shk_ctx* ctx, int32_t on);
int shk_profile_read(shk_ctx* ctx, shk_profile* out, int32_t reset);
/* Launch kernel `phase` (SHK_PH_ASSEMBLE or SHK_PH_SPMV) `reps` times between two events. */
int shk_time_kernel(shk_ctx* ctx, int32_t phase, int32_t reps, double dt, double* avg_ms);

/* The residual-only instance of the assembly kernel (what shk_newton_solve launches after the update it expects to be
 * the last of a solve: no element Jacobians, no slot phase), `reps` times between two events. */
int shk_time_assemble_residual(shk_ctx* ctx, int32_t reps, double dt, double* avg_ms);
/* Assembly passes since creation: n[0] full (residual + Jacobian), n[1] residual-only (predicted last iteration of a
 * Newton solve), n[2] full passes repeated because the prediction was wrong, n[3] Newton iterations of the last solve, n[4] linear solves
 * stopped by the forcing rule (shk_params.krylov_forcing). */
int shk_solver_stats(shk_ctx* ctx, int64_t n[5]);

/* Plan statistics for DESIGN.md / bench: n[0]=owned rows n[1]=ne n[2]=nnz n[3]=assembly blocks
 * n[4]=cells computed per assembly incl. cells shared between blocks n[5]=SELL slots (padded nnz)
 * n[6]=device bytes n[7]=max row length n[8]=entries of the finest A*P operator (0 if none)
 * n[9]=multigrid levels (sparse + dense) n[10]=rows of the dense coarsest level
 * n[11]=LDS bytes of one assembly workgroup | (most vertices one stages << 32) */
int shk_plan_stats(shk_ctx* ctx, int64_t n[12]);
/* Storage of the Jacobian and of the multigrid's smoother copies: n[0]=stored entries (nnz) n[1]=SELL-64 slots (nnz +
 * padding) n[2]=slots of slices whose columns are stored as 16-bit offsets n[3]=slices n[4]=sparse multigrid levels
 * that smooth on a packed bfloat16 copy (level 0 included) n[5]=rows of the largest level that does not. */
int shk_storage_stats(shk_ctx* ctx, int64_t n[6]);

/* ---- setup-time data ingestion (context-free: host arrays in and out, caller's node order) ----
 * Bilinear interpolation of gridded data to npts points: replaces the RegularGridInterpolator evaluation of
 * model_setup.interp_data (/root/reference/source/model_setup.py:84-86; linear, bounds_error=False,
 * fill_value=None -> points outside the grid extrapolate from the edge cell).  xg (nx) and yg (ny) strictly
 * ascending; f_xy is (nx, ny) row-major, f_xy[ix*ny + iy] -- the transposed array the reference passes to scipy.
 * Bit-identical to scipy's result: pairwise_weights = 0 rounds like scipy's 2-D float64 fast path,
 * (f*wx)*wy, which is what float64 data gets; 1 like its generic evaluator, f*(wx*wy), used for float32 or
 * read-only data. */
int shk_interp_regular_grid(int device_id, int64_t npts, const double* px, const double* py, int64_t nx, int64_t ny,
                            const double* xg, const double* yg, const double* f_xy, int32_t pairwise_weights,
                            double* out);
/* out[p] = 1.0 if point p lies inside the polygon poly_xy (m vertices, x0,y0,x1,y1,..., closed implicitly), else
 * 0.0, by the even-odd rule: replaces the per-node shapely loop of model_setup.set_lake_bdry
 * (model_setup.py:68-72). */
int shk_points_in_polygon(int device_id, int64_t npts, const double* px, const double* py, int64_t m,
                          const double* poly_xy, double* out);

#ifdef __cplusplus
}
#endif
#endif /* SHAKTI_HIP_H */
