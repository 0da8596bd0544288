// C ABI of libshakti_hip.so (declared in include/shakti_hip.h) and the host-side Newton / Krylov
// drivers.  Host code only decides "how many more iterations to enqueue"; every number the solve
// produces is computed on the device.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "shk_device.h"
#include "shk_quadrature.h"

using namespace shk;


static thread_local std::string g_err;

static int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                        std::to_string(__LINE__) + ")");                                          \
    } while (0)

#define CHECK_CTX(c) \
    if (!(c)) return fail("null context")

template <class T>
static hipError_t dev_alloc(Ctx* c, T** p, size_t n) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    c->allocs.push_back(q);
    c->device_bytes += (int64_t)bytes;
    *p = reinterpret_cast<T*>(q);
    return hipSuccess;
}

template <class T>
static hipError_t upload(Ctx* c, T** p, const std::vector<T>& h) {
    hipError_t e = dev_alloc(c, p, h.size());
    if (e != hipSuccess) return e;
    if (h.empty()) return hipSuccess;
    return hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

static void derive_params(Ctx* c) {
    const shk_params& p = c->params;
    DevParams& d = c->dp;
    d.g = p.g; d.rho_i = p.rho_i; d.rho_w = p.rho_w; d.nu = p.nu; d.Lh = p.Lh; d.omega = p.omega;
    d.n = p.n; d.A = p.A; d.b_min = p.b_min;
    d.rwg = p.rho_w * p.g;
    d.c_m = 1.0 / p.rho_i - 1.0 / p.rho_w;
    d.kcoef = p.g / (12.0 * p.nu);
    d.om_nu = p.omega / p.nu;
    d.ri_rw = p.rho_i / p.rho_w;
    d.n_is_3 = (p.n == 3.0) ? 1 : 0;
}

static int set_quadrature(Ctx* c, int nq, const double* xyw) {
    if (nq < 1 || nq > kMaxQuad) return fail("quadrature size must be in [1, 32]");
    double sw = 0.0;
    for (int k = 0; k < nq; ++k) sw += xyw[3 * k + 2];
    if (std::fabs(sw - 0.5) > 1e-12) return fail("quadrature weights must sum to 1/2 (reference triangle area)");
    c->quad.nq = nq;
    for (int k = 0; k < nq; ++k) {
        const double x = xyw[3 * k], y = xyw[3 * k + 1];
        c->quad.phi0[k] = 1.0 - x - y;
        c->quad.phi1[k] = x;
        c->quad.phi2[k] = y;
        c->quad.w2[k] = 2.0 * xyw[3 * k + 2];
    }
    c->assembled = false;
    return 0;
}

extern "C" {

const char* shk_last_error(void) { return g_err.c_str(); }
int shk_version(void) { return 100; }

int shk_default_params(shk_params* p) {
    if (!p) return fail("null params");
    std::memset(p, 0, sizeof(*p));
    // /root/reference/source/params.py:4-11
    p->g = 9.81; p->rho_i = 917.0; p->rho_w = 1000.0; p->nu = 1.787e-6; p->Lh = 3.34e5; p->omega = 1e-3;
    p->n = 3.0; p->A = 2.24e-24;
    p->b_min = 1.0e-5;  // model_setup.py:53
    // DOLFINx NewtonSolver defaults (never overridden at solvers.py:52)
    p->newton_rtol = 1e-9; p->newton_atol = 1e-10; p->newton_relax = 1.0; p->newton_max_it = 50;
    // the reference solves each Newton system exactly (LU); the Krylov loop is driven to 1e-10
    p->krylov_rtol = 1e-10; p->krylov_atol = 1e-50; p->krylov_max_it = 20000; p->krylov_check_every = 32;
    return 0;
}

int shk_create(int device_id, int64_t nv, int64_t ne, const double* xy, const int32_t* cells, shk_ctx** out) {
    if (!out) return fail("null output pointer");
    *out = nullptr;
    if (!xy || !cells) return fail("null mesh arrays");
    if (nv < 3 || ne < 1) return fail("mesh needs at least one triangle");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail("device_id out of range: no such GPU");
    HIPCHK(hipSetDevice(device_id));
    Ctx* c = new Ctx();
    c->device = device_id;
    c->nv = nv;
    c->ne = ne;
    shk_default_params(&c->params);
    derive_params(c);
    {
        const double* q = &SHK_QUAD_DEFAULT[0][0];
        if (set_quadrature(c, SHK_NQ_DEFAULT, q)) { delete c; return -1; }
    }
    PlanOptions opt;
    if (const char* s = getenv("SHK_ASM_ROWS")) opt.rows_max = std::max(16, atoi(s));
    if (const char* s = getenv("SHK_ASM_CELLS")) opt.cells_max = std::max(64, atoi(s));
    opt.spmv_nnz = kSpmvNnz;  // the LDS product buffer of k_spmv
    std::string err = build_plan(nv, ne, cells, opt, c->plan);
    if (!err.empty()) { delete c; return fail("plan: " + err); }
    c->nnz = c->plan.nnz;
    c->nblk = (int)c->plan.blk_row0.size() - 1;
    c->nsb = (int)c->plan.sp_row0.size() - 1;
    c->grid = (int)std::min<int64_t>(kMaxParts, std::max<int64_t>(1, (nv + kBlock - 1) / kBlock));
    {
        const size_t E = c->plan.cells_max, R = c->plan.rows_max;
        size_t lds = 12 * E * sizeof(double) + 3 * E * sizeof(int) + 2 * (R + 1) * sizeof(int) +
                     (size_t)c->plan.max_inc_per_block * sizeof(uint16_t);
        c->asm_lds = (lds + 15) & ~size_t(15);
        if (c->asm_lds > 160 * 1024) { delete c; return fail("assembly LDS budget exceeds 160 KiB"); }
    }
    auto bail = [&](hipError_t e, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        shk_destroy(reinterpret_cast<shk_ctx*>(c));
        return fail(m);
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "stream");
    // mesh
    if ((e = dev_alloc(c, &c->d_xy, (size_t)nv)) != hipSuccess) return bail(e, "alloc xy");
    if ((e = hipMemcpy(c->d_xy, xy, (size_t)nv * sizeof(double2), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "copy xy");
    if ((e = dev_alloc(c, &c->d_cells, (size_t)ne * 3)) != hipSuccess) return bail(e, "alloc cells");
    if ((e = hipMemcpy(c->d_cells, cells, (size_t)ne * 3 * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "copy cells");
    // plan
    const HostPlan& P = c->plan;
#define UP(dst, src) if ((e = upload(c, &c->dst, P.src)) != hipSuccess) return bail(e, "upload " #src)
    UP(d_rowptr, rowptr); UP(d_colidx, colidx); UP(d_diagpos, diagpos); UP(d_lastcell, lastcell);
    UP(d_blk_row0, blk_row0); UP(d_blk_cellptr, blk_cellptr); UP(d_blk_cells, blk_cells); UP(d_incptr, incptr);
    UP(d_inccode, inccode); UP(d_sp_row0, sp_row0);
#undef UP
    // fields
    for (int fidx = 0; fidx < SHK_FIELD_COUNT; ++fidx) {
        if (fidx == SHK_Q) continue;
        if ((e = dev_alloc(c, &c->f[fidx], (size_t)nv)) != hipSuccess) return bail(e, "alloc field");
        if ((e = hipMemset(c->f[fidx], 0, (size_t)nv * sizeof(double))) != hipSuccess) return bail(e, "memset");
    }
    double** vecs[] = {&c->d_melt_tmp, &c->d_b_tmp, &c->d_m0, &c->d_F, &c->d_dinv, &c->d_r, &c->d_rhat,
                       &c->d_p, &c->d_v, &c->d_s, &c->d_t, &c->d_y};
    for (double** v : vecs) {
        if ((e = dev_alloc(c, v, (size_t)nv)) != hipSuccess) return bail(e, "alloc vector");
        if ((e = hipMemset(*v, 0, (size_t)nv * sizeof(double))) != hipSuccess) return bail(e, "memset");
    }
    if ((e = dev_alloc(c, &c->d_vals, (size_t)c->nnz)) != hipSuccess) return bail(e, "alloc vals");
    if ((e = dev_alloc(c, &c->d_vals_s, (size_t)c->nnz)) != hipSuccess) return bail(e, "alloc vals_s");
    if ((e = dev_alloc(c, &c->d_bcflag, (size_t)nv)) != hipSuccess) return bail(e, "alloc bcflag");
    if ((e = hipMemset(c->d_bcflag, 0, (size_t)nv)) != hipSuccess) return bail(e, "memset");
    if ((e = dev_alloc(c, &c->d_part, (size_t)6 * kMaxParts)) != hipSuccess) return bail(e, "alloc partials");
    if ((e = hipMemset(c->d_part, 0, 6 * kMaxParts * sizeof(double))) != hipSuccess) return bail(e, "memset");
    if ((e = dev_alloc(c, &c->d_state, 1)) != hipSuccess) return bail(e, "alloc state");
    if ((e = hipMemset(c->d_state, 0, sizeof(KrylovState))) != hipSuccess) return bail(e, "memset");
    if ((e = hipHostMalloc((void**)&c->h_state, 2 * sizeof(KrylovState))) != hipSuccess) return bail(e, "pinned");
    if ((e = hipHostMalloc((void**)&c->h_part, kMaxParts * sizeof(double))) != hipSuccess) return bail(e, "pinned");
    if ((e = prepare_kernels(c)) != hipSuccess) return bail(e, "hipFuncSetAttribute(dynamic LDS)");
    *out = reinterpret_cast<shk_ctx*>(c);
    return 0;
}

int shk_destroy(shk_ctx* ctx) {
    if (!ctx) return 0;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& ev : c->ev_pool) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    for (void* p : c->allocs) hipFree(p);
    if (c->h_state) hipHostFree(c->h_state);
    if (c->h_part) hipHostFree(c->h_part);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int shk_set_params(shk_ctx* ctx, const shk_params* p) {
    CHECK_CTX(ctx);
    if (!p) return fail("null params");
    if (!(p->g > 0 && p->rho_i > 0 && p->rho_w > 0 && p->nu > 0 && p->Lh > 0)) return fail("non-positive constant");
    if (p->newton_max_it < 0 || p->krylov_max_it < 1 || p->krylov_check_every < 1) return fail("bad iteration limits");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    c->params = *p;
    derive_params(c);
    c->assembled = false;
    return 0;
}

int shk_get_params(shk_ctx* ctx, shk_params* p) {
    CHECK_CTX(ctx);
    if (!p) return fail("null params");
    *p = reinterpret_cast<Ctx*>(ctx)->params;
    return 0;
}

int shk_set_quadrature(shk_ctx* ctx, int32_t nq, const double* xyw) {
    CHECK_CTX(ctx);
    if (!xyw) return fail("null quadrature table");
    return set_quadrature(reinterpret_cast<Ctx*>(ctx), nq, xyw);
}

int shk_set_field(shk_ctx* ctx, int32_t field, const double* host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!host) return fail("null host array");
    if (field < 0 || field >= SHK_FIELD_COUNT || field == SHK_DX) return fail("field id not settable");
    HIPCHK(hipSetDevice(c->device));
    c->assembled = false;
    if (field == SHK_Q) {
        // interleaved (nv,2) -> qx, qy; d_r/d_p are free scratch outside a linear solve
        double* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, (size_t)c->nv * 2 * sizeof(double)));
        hipError_t e = hipMemcpyAsync(tmp, host, (size_t)c->nv * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) { launch_split_q(c, tmp); e = hipStreamSynchronize(c->stream); }
        hipFree(tmp);
        HIPCHK(e);
        return 0;
    }
    HIPCHK(hipMemcpyAsync(c->f[field], host, (size_t)c->nv * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int shk_get_field(shk_ctx* ctx, int32_t field, double* host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!host) return fail("null host array");
    if (field < 0 || field >= SHK_FIELD_COUNT) return fail("unknown field id");
    HIPCHK(hipSetDevice(c->device));
    if (field == SHK_Q) {
        double* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, (size_t)c->nv * 2 * sizeof(double)));
        launch_join_q(c, tmp);
        hipError_t e = hipMemcpyAsync(host, tmp, (size_t)c->nv * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(tmp);
        HIPCHK(e);
        return 0;
    }
    HIPCHK(hipMemcpyAsync(host, c->f[field], (size_t)c->nv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int shk_set_dirichlet(shk_ctx* ctx, int64_t n, const int32_t* dofs, double value) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (n < 0 || (n > 0 && !dofs)) return fail("bad dof list");
    HIPCHK(hipSetDevice(c->device));
    std::vector<uint8_t> flag((size_t)c->nv, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (dofs[i] < 0 || dofs[i] >= c->nv) return fail("Dirichlet dof outside [0, nv)");
        flag[dofs[i]] = 1;
    }
    HIPCHK(hipMemcpy(c->d_bcflag, flag.data(), (size_t)c->nv, hipMemcpyHostToDevice));
    c->has_bc = n > 0;
    c->bc_value = value;
    c->assembled = false;
    return 0;
}

int shk_assemble(shk_ctx* ctx, double dt) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!(dt > 0)) return fail("dt must be positive");
    HIPCHK(hipSetDevice(c->device));
    launch_assemble(c, dt);
    HIPCHK(hipGetLastError());
    c->assembled = true;
    c->assembled_dt = dt;
    return 0;
}

int shk_get_residual(shk_ctx* ctx, double* host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!host) return fail("null host array");
    if (!c->assembled) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(host, c->d_F, (size_t)c->nv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int shk_csr_nnz(shk_ctx* ctx, int64_t* nnz) {
    CHECK_CTX(ctx);
    if (!nnz) return fail("null output");
    *nnz = reinterpret_cast<Ctx*>(ctx)->nnz;
    return 0;
}

int shk_get_csr(shk_ctx* ctx, int32_t* rowptr, int32_t* colidx, double* values) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (rowptr) std::memcpy(rowptr, c->plan.rowptr.data(), (size_t)(c->nv + 1) * sizeof(int32_t));
    if (colidx) std::memcpy(colidx, c->plan.colidx.data(), (size_t)c->nnz * sizeof(int32_t));
    if (values) {
        if (!c->assembled) return fail("no assembled system: call shk_assemble first");
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpyAsync(values, c->d_vals, (size_t)c->nnz * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// Enqueue BiCGStab on the assembled system with rhs = F; returns after the device reports done.
static int krylov_solve(Ctx* c, int* its, int* converged, double* relres) {
    launch_scale(c);
    krylov_init(c);
    const int chunk = std::max(1, c->params.krylov_check_every);
    int it = 0;
    KrylovState* hs = c->h_state;
    for (;;) {
        for (int k = 0; k < chunk; ++k) krylov_iteration(c, it + k);
        it += chunk;
        HIPCHK(hipMemcpyAsync(hs, c->d_state, sizeof(KrylovState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (hs->done) break;
        if (it > c->params.krylov_max_it + chunk) return fail("Krylov driver ran past max_it without a stop flag");
    }
    HIPCHK(hipGetLastError());
    if (its) *its = hs->its;
    if (converged) *converged = hs->converged;
    if (relres) *relres = (hs->rhs2 > 0) ? std::sqrt(hs->rnorm2 / hs->rhs2) : 0.0;
    return 0;
}

int shk_linear_solve(shk_ctx* ctx, int32_t* its, int32_t* converged, double* rel_residual) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c->assembled) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    int k = 0, cv = 0;
    double rr = 0;
    if (krylov_solve(c, &k, &cv, &rr)) return -1;
    // dx = D^-1 y without touching N
    launch_newton_update(c, false);
    HIPCHK(hipStreamSynchronize(c->stream));
    if (its) *its = k;
    if (converged) *converged = cv;
    if (rel_residual) *rel_residual = rr;
    return 0;
}

int shk_spmv(shk_ctx* ctx, const double* x_host, double* y_host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!x_host || !y_host) return fail("null host array");
    if (!c->assembled) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_p, x_host, (size_t)c->nv * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_spmv_plain(c, c->d_vals, c->d_p, c->d_v);
    HIPCHK(hipMemcpyAsync(y_host, c->d_v, (size_t)c->nv * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// ||F||_2 with a fixed summation order (device partials, host sum of <= 1024 values).
static int residual_norm(Ctx* c, double* out) {
    launch_norm2(c, c->d_F, c->d_part + P_AUX * kMaxParts);
    HIPCHK(hipMemcpyAsync(c->h_part, c->d_part + P_AUX * kMaxParts, (size_t)c->grid * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double s = 0.0;
    for (int i = 0; i < c->grid; ++i) s += c->h_part[i];
    *out = std::sqrt(s);
    return 0;
}

int shk_newton_solve(shk_ctx* ctx, double dt, shk_solve_info* info) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!(dt > 0)) return fail("dt must be positive");
    HIPCHK(hipSetDevice(c->device));
    shk_solve_info I{};
    // DOLFINx NewtonSolver::solve: residual first, then J / solve / update / residual.
    launch_assemble(c, dt);  // F and J of the current iterate in one pass
    c->assembled = true;
    c->assembled_dt = dt;
    double r = 0.0;
    if (residual_norm(c, &r)) return -1;
    I.residual0 = r;
    I.residual = r;
    int it = 0;
    bool conv = r < c->params.newton_atol;  // relative residual is 1 at iteration 0
    while (!conv && it < c->params.newton_max_it) {
        int k = 0, kc = 0;
        if (krylov_solve(c, &k, &kc, nullptr)) return -1;
        I.krylov_its += k;
        if (!kc) I.krylov_failed = 1;
        launch_newton_update(c, true);
        launch_assemble(c, dt);
        ++it;
        if (residual_norm(c, &r)) return -1;
        I.residual = r;
        if (!std::isfinite(r)) break;
        conv = (r < c->params.newton_atol) || (I.residual0 > 0 && r / I.residual0 < c->params.newton_rtol);
    }
    HIPCHK(hipGetLastError());
    I.newton_its = it;
    I.converged = conv ? 1 : 0;
    if (info) *info = I;
    return 0;
}

int shk_update_explicit(shk_ctx* ctx, double dt) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!(dt > 0)) return fail("dt must be positive");
    HIPCHK(hipSetDevice(c->device));
    launch_update_explicit(c, dt);
    HIPCHK(hipGetLastError());
    c->assembled = false;
    return 0;
}

int shk_step(shk_ctx* ctx, double dt, shk_solve_info* info) {
    shk_solve_info I{};
    if (shk_newton_solve(ctx, dt, &I)) return -1;
    if (info) *info = I;
    if (!I.converged) return 0;  // caller decides (the reference raises, solvers.py:179-183)
    return shk_update_explicit(ctx, dt);
}

int shk_sync(shk_ctx* ctx) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int shk_profile_enable(shk_ctx* ctx, int32_t on) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIPCHK(hipStreamSynchronize(c->stream));
    c->profiling = on != 0;
    return 0;
}

int shk_profile_read(shk_ctx* ctx, shk_profile* out, int32_t reset) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        c->prof.ms[c->ev_pool[i].phase] += ms;
        c->prof.launches[c->ev_pool[i].phase] += 1;
    }
    c->ev_used = 0;
    if (out) *out = c->prof;
    if (reset) c->prof = shk_profile{};
    return 0;
}

int shk_time_kernel(shk_ctx* ctx, int32_t phase, int32_t reps, double dt, double* avg_ms) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (reps < 1 || !avg_ms) return fail("bad arguments");
    if (phase != SHK_PH_ASSEMBLE && phase != SHK_PH_SPMV) return fail("only ASSEMBLE and SPMV can be timed");
    if (phase == SHK_PH_SPMV && !c->assembled) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    const bool was = c->profiling;
    c->profiling = false;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    auto once = [&]() {
        if (phase == SHK_PH_ASSEMBLE) launch_assemble(c, dt);
        else launch_spmv_plain(c, c->d_vals, c->d_p, c->d_v);
    };
    once();  // warm
    HIPCHK(hipEventRecord(a, c->stream));
    for (int i = 0; i < reps; ++i) once();
    HIPCHK(hipEventRecord(b, c->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    c->profiling = was;
    *avg_ms = ms / reps;
    return 0;
}

int shk_plan_stats(shk_ctx* ctx, int64_t n[8]) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    n[0] = c->nv; n[1] = c->ne; n[2] = c->nnz; n[3] = c->nblk; n[4] = (int64_t)c->plan.blk_cells.size();
    n[5] = c->nsb; n[6] = c->device_bytes; n[7] = c->plan.max_row_len;
    return 0;
}

}  // extern "C"
