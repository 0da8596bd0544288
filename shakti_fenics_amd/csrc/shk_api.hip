// C ABI of libshakti_hip.so (declared in include/shakti_hip.h) and the host-side Newton / Krylov
// drivers.  Host code only decides "how many more iterations to enqueue"; every number the solve
// produces is computed on the device.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <thread>
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
namespace shk {
int set_error(const std::string& msg) { return fail(msg); }
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

// Waiting on the device while RCCL collectives are in flight must not be able to hang forever: a peer that died or a
// mismatched collective would otherwise block every rank inside hipEventSynchronize / hipStreamSynchronize with no
// message.  With an RCCL communicator of > 1 ranks the host polls with a deadline (SHK_COMM_TIMEOUT_S, default 300 s)
// and turns a stall into an error; the context is then POISONED: its stream will never drain, so shk_destroy skips
// every call that would wait for it.  Single-GPU contexts block as usual.
static const char* kStallMsg = "the device stream stalled: no progress for SHK_COMM_TIMEOUT_S seconds with RCCL collectives in "
                               "flight (a peer rank died, or the ranks diverged); the context is poisoned -- exit the process";
static hipError_t wait_event(Ctx* c, hipEvent_t ev) {
    if (c->poisoned) return hipErrorLaunchTimeOut;
    if (c->comm.kind != Comm::RCCL || c->comm.nranks <= 1) return hipEventSynchronize(ev);
    const double limit = tunables().comm_timeout_s;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            c->poisoned = true;
            return hipErrorLaunchTimeOut;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}
namespace shk {
hipError_t wait_stream(Ctx* c) {
    if (c->poisoned) return hipErrorLaunchTimeOut;
    if (c->comm.kind != Comm::RCCL || c->comm.nranks <= 1) return hipStreamSynchronize(c->stream);
    hipError_t e = hipEventRecord(c->poll_ev[2], c->stream);
    return e != hipSuccess ? e : wait_event(c, c->poll_ev[2]);
}
}
// every host wait of the API goes through the deadline
#define WAITCHK(c)                                                                       \
    do {                                                                                 \
        hipError_t w_ = wait_stream(c);                                                  \
        if (w_ == hipErrorLaunchTimeOut) return fail(kStallMsg);                         \
        if (w_ != hipSuccess) return fail(std::string("stream wait failed: ") + hipGetErrorString(w_)); \
    } while (0)

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
    return upload_sync(c, *p, h.data(), h.size() * sizeof(T));
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
    d.inv_rwg = 1.0 / d.rwg; d.inv_Lh = 1.0 / p.Lh; d.cm_Lh = d.c_m / p.Lh;
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

// Radon's 7-point degree-5 rule on the reference triangle (centroid + two S21 orbits), weights sum to 1/2.
static void set_poly_rule(Ctx* c) {
    const double r15 = std::sqrt(15.0);
    const double a1 = (6.0 - r15) / 21.0, a2 = (6.0 + r15) / 21.0;
    const double w0 = 9.0 / 80.0, w1 = (155.0 - r15) / 2400.0, w2 = (155.0 + r15) / 2400.0;
    const double pts[7][3] = {{1.0 / 3.0, 1.0 / 3.0, w0},
                              {a1, a1, w1}, {a1, 1.0 - 2.0 * a1, w1}, {1.0 - 2.0 * a1, a1, w1},
                              {a2, a2, w2}, {a2, 1.0 - 2.0 * a2, w2}, {1.0 - 2.0 * a2, a2, w2}};
    QuadArg& q = c->qpoly5;
    q.nq = 7;
    for (int k = 0; k < 7; ++k) {
        q.phi0[k] = 1.0 - pts[k][0] - pts[k][1];
        q.phi1[k] = pts[k][0];
        q.phi2[k] = pts[k][1];
        q.w2[k] = 2.0 * pts[k][2];
    }
}

// Fused multi-sweep smoother of one level: built on the host from the level's pattern, uploaded if the level allows it.
static hipError_t upload_sweep_plan(Ctx* c, const SellPattern& A, const SellPattern& AP, DevSweepPlan& D) {
    D = DevSweepPlan();
    if (!tunables().amg_fused_sweeps || AP.n_rows != A.n_rows) return hipSuccess;
    // the fused launch pays where a sweep is launch-bound; a level too large for that keeps its streaming sweeps
    // (measured at 10M rows, us per cycle in four launches / in one: level 5 of 9.8k rows 26 / 14, level 4 of 39k 26 / 16,
    //  level 3 of 156k 31 / 22, level 2 of 625k 51 / 63, level 1 of 2.5M 142 / 223; at 1M rows, level 1 of 250k 33 / 31:
    //  SHK_AMG_FUSED_ROWS = 320 000)
    if (A.n_rows > tunables().amg_fused_rows) return hipSuccess;
    SweepPlan P;
    if (!build_sweep_plan(A, AP, P).empty() || P.nblk == 0) return hipSuccess;   // no plan: one launch per sweep
    hipError_t e;
    if ((e = upload(c, &D.hdr, P.hdr)) != hipSuccess) return e;
    if ((e = upload(c, &D.ext_info, P.ext_info)) != hipSuccess) return e;
    if ((e = upload(c, &D.lcol_own, P.lcol_own)) != hipSuccess) return e;
    if ((e = upload(c, &D.ring_lcol, P.ring_lcol)) != hipSuccess) return e;
    D.nblk = P.nblk; D.width = P.width; D.max_local = P.max_local;
    D.plan_bytes = 4.0 * (double)(P.hdr.size() + P.ext_info.size()) + 2.0 * (double)(P.lcol_own.size() + P.ring_lcol.size());
    return hipSuccess;
}

namespace shk {
hipError_t amg_upload_rep_top(Ctx* c, AmgHierarchy& R, const SellPattern& G, const std::vector<int32_t>& diag_slot) {
    hipError_t e;

    if ((e = upload(c, &R.t_ptr, G.ptr)) != hipSuccess) return e;
    if ((e = upload(c, &R.t_col, G.col)) != hipSuccess) return e;
    if ((e = upload(c, &R.t_rowlen, G.rowlen)) != hipSuccess) return e;
    if ((e = upload(c, &R.t_cbase, G.cbase)) != hipSuccess) return e;
    if ((e = upload(c, &R.t_ptr16, G.ptr16)) != hipSuccess) return e;
    if ((e = upload(c, &R.t_col16, G.col16)) != hipSuccess) return e;
    if ((e = upload(c, &R.t_diag, diag_slot)) != hipSuccess) return e;
    R.t_slots = G.slots;
    if ((e = dev_alloc(c, &R.t_vals, (size_t)G.slots)) != hipSuccess) return e;
    if ((e = dev_alloc(c, &R.t_dinv, (size_t)G.nslice * kSlice)) != hipSuccess) return e;
    if ((e = zero_async(c, R.t_dinv, (size_t)G.nslice * kSlice * sizeof(float))) != hipSuccess) return e;
    R.topA = DevSell{G.n_rows, G.n_cols, G.nslice, sell_fits_cache(G.slots, kAmgSlotBytes), R.t_ptr, R.t_col, R.t_rowlen,
                     R.t_cbase, R.t_ptr16, R.t_col16};
    R.top_vals = R.t_vals;
    R.top_dinv = R.t_dinv;
    R.top_bytes = sell_bytes(G.slots, (int64_t)G.col16.size(), G.nslice, 4);
    return hipSuccess;
}

hipError_t amg_upload(Ctx* c, std::vector<AmgLevelPlan>& plans, AmgHierarchy& H, int64_t n_loc0,
                      const std::vector<int32_t>* krank0, const SellPattern* top) {
    const std::vector<int32_t>& kr0 = krank0 ? *krank0 : c->plan.krank;
    hipError_t e;
    const size_t nx = plans.size();
    H.xf.resize(nx);
    H.lv.resize(nx);  // [0] unused; sparse levels 1 .. nx-1
    if (H.sw.size() < nx) H.sw.resize(nx);
    SellPattern Aprev;            // pattern of the level the current transfer starts from (level l), kept for its sweep plan
    bool have_prev = false;
    std::vector<int32_t> prev_pos, prev_rank;
    for (size_t l = 0; l < nx; ++l) {
        AmgLevelPlan& LP = plans[l];
        AmgXfer& X = H.xf[l];
        X.n_fine = LP.n_fine; X.n_coarse = LP.n_coarse; X.n_coarse_cols = LP.n_coarse_cols; X.dense = LP.dense;
        X.onto_global = LP.onto_global;
        X.n_glist = (int64_t)LP.glist.size();
        if ((e = upload(c, &X.agg, LP.agg)) != hipSuccess) return e;
        if ((e = upload(c, &X.members, LP.members)) != hipSuccess) return e;
        if ((e = upload(c, &X.gptr, LP.gptr)) != hipSuccess) return e;
        if ((e = upload(c, &X.glist, LP.glist)) != hipSuccess) return e;
        {
            // tables of the fused restriction, indexed by k-d rank; they rely on rank k's members sitting in the
            // 256-row group k / 64 of the finer level (true for every hierarchy built here; checked anyway)
            const int32_t nc = LP.n_coarse;
            std::vector<int32_t> mk((size_t)4 * nc), pos(nc);
            bool ok = LP.kd_pos.empty() || (int32_t)LP.kd_pos.size() == nc;
            for (int32_t k = 0; ok && k < nc; ++k) {
                const int32_t I = LP.kd_pos.empty() ? k : LP.kd_pos[k];
                if (I < 0 || I >= nc) { ok = false; break; }
                pos[k] = I;
                for (int q = 0; q < 4; ++q) {
                    const int32_t m = LP.members[(size_t)4 * I + q];
                    mk[(size_t)4 * k + q] = m;
                    if (q == 0 && m < 0) ok = false;
                    if (m < 0) continue;
                    if (m / 256 != k / 64) ok = false;
                    // ... and the member's own k-d rank must be one of 4k .. 4k+3
                    const int32_t mr = l == 0 ? ((size_t)m < kr0.size() ? kr0[m] : -1)
                                              : (prev_pos.empty() ? m : prev_rank[m]);
                    if (mr / 4 != k) ok = false;
                }
            }
            // k-d rank of every row of the level just described (= the finer level of the next transfer)
            prev_pos = LP.kd_pos;
            prev_rank.assign(prev_pos.size(), -1);
            for (size_t k = 0; k < prev_pos.size(); ++k) prev_rank[prev_pos[k]] = (int32_t)k;
            if (ok) {
                if ((e = upload(c, &X.members_kd, mk)) != hipSuccess) return e;
                if ((e = upload(c, &X.kd_pos, pos)) != hipSuccess) return e;
            }
        }
        if (LP.with_ap) {
            // fused four-sweep smoother of level l (its first sweep runs on this transfer's A*P): coarse levels, and the
            // top level of a replicated hierarchy (`top`), which is a coarse level of the whole cycle
            if (l >= 1 && have_prev && (e = upload_sweep_plan(c, Aprev, LP.AP, H.sw[l])) != hipSuccess) return e;
            if (l == 0 && top && (e = upload_sweep_plan(c, *top, LP.AP, H.sw[0])) != hipSuccess) return e;
            X.with_ap = true; X.ap_nslice = LP.AP.nslice; X.ap_slots = LP.AP.slots; X.ap_slots16 = (int64_t)LP.AP.col16.size();
            X.ap_n_glist = (int64_t)LP.ap_glist.size();
            if (l == 0) H.ap_nnz0 = LP.AP.nnz;
            if ((e = upload(c, &X.ap_ptr, LP.AP.ptr)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_col, LP.AP.col)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_rowlen, LP.AP.rowlen)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_cbase, LP.AP.cbase)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_ptr16, LP.AP.ptr16)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_col16, LP.AP.col16)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_gptr, LP.ap_gptr)) != hipSuccess) return e;
            if ((e = upload(c, &X.ap_glist, LP.ap_glist)) != hipSuccess) return e;
            if ((e = dev_alloc(c, &X.ap_vals, (size_t)X.ap_slots)) != hipSuccess) return e;
            if (amg_packed_level(LP.AP.n_rows, X.ap_slots16)) {
                if ((e = dev_alloc(c, &X.ap_pk, (size_t)X.ap_slots16)) != hipSuccess) return e;
            }
        }
        if (!LP.ghost_col.empty()) {
            X.n_ghost = (int32_t)LP.ghost_col.size();
            if ((e = upload(c, &X.ghost_col, LP.ghost_col)) != hipSuccess) return e;
        }
        if (LP.onto_global) {
            if (l + 1 != nx) return hipErrorInvalidValue;  // the replicated level ends the distributed part
            // (the dummy rows that pad every subdomain's block of the gathered right-hand side stay zero for ever)
            const size_t ng = std::max<size_t>(64, (size_t)LP.n_coarse_cols);
            if ((e = dev_alloc(c, &H.rep_rglob, ng)) != hipSuccess) return e;
            if ((e = zero_async(c, H.rep_rglob, ng * sizeof(float))) != hipSuccess) return e;
            if ((e = dev_alloc(c, &H.rep_xglob, ng)) != hipSuccess) return e;
        } else if (!LP.dense) {
            if (l + 1 >= nx) return hipErrorInvalidValue;  // a hierarchy must end on a dense level
            AmgLevel& L = H.lv[l + 1];
            L.n = LP.Ac.n_rows; L.n_cols = LP.Ac.n_cols; L.nslice = LP.Ac.nslice; L.slots = LP.Ac.slots;
            L.slots16 = (int64_t)LP.Ac.col16.size();
            if ((e = upload(c, &L.ptr, LP.Ac.ptr)) != hipSuccess) return e;
            if ((e = upload(c, &L.col, LP.Ac.col)) != hipSuccess) return e;
            if ((e = upload(c, &L.rowlen, LP.Ac.rowlen)) != hipSuccess) return e;
            if ((e = upload(c, &L.cbase, LP.Ac.cbase)) != hipSuccess) return e;
            if ((e = upload(c, &L.ptr16, LP.Ac.ptr16)) != hipSuccess) return e;
            if ((e = upload(c, &L.col16, LP.Ac.col16)) != hipSuccess) return e;
            if ((e = upload(c, &L.diag_slot, LP.diag_slot)) != hipSuccess) return e;
            Aprev = std::move(LP.Ac);
            have_prev = true;
            const size_t nr = std::max<size_t>((size_t)L.nslice * kSlice, (size_t)L.n_cols);
            if ((e = dev_alloc(c, &L.vals, (size_t)L.slots)) != hipSuccess) return e;
            if (amg_packed_level(L.n, L.slots16)) {
                if ((e = dev_alloc(c, &L.pk, (size_t)L.slots16)) != hipSuccess) return e;
            }
            float** vs[] = {&L.dinv, &L.x, &L.x2, &L.x3, &L.r};
            for (float** v : vs) {
                if ((e = dev_alloc(c, v, nr)) != hipSuccess) return e;
                if ((e = zero_async(c, *v, nr * sizeof(float))) != hipSuccess) return e;
            }
        } else {
            const size_t rows = H.distributed ? (size_t)H.n_glob : (size_t)LP.n_coarse;
            const size_t cols = (size_t)LP.n_coarse_cols;
            if ((e = dev_alloc(c, &H.cdense, rows * cols)) != hipSuccess) return e;
            if ((e = dev_alloc(c, &H.cinv, rows * cols)) != hipSuccess) return e;
            if (!H.distributed && cols > 128 && (e = dev_alloc(c, &H.cinv32, rows * cols)) != hipSuccess) return e;
            if ((e = dev_alloc(c, &H.cr, std::max<size_t>(64, std::max(rows, cols)))) != hipSuccess) return e;
            if ((e = dev_alloc(c, &H.cx, std::max<size_t>(64, rows))) != hipSuccess) return e;
            if ((e = dev_alloc(c, &H.cglob, std::max<size_t>(64, std::max(rows, cols)))) != hipSuccess) return e;
            // scratch of the blocked Gauss-Jordan inverse: column panel, row panel, diagonal block (32 pivots)
            if ((e = dev_alloc(c, &H.gj, (2 * std::max<size_t>(1024, std::max(rows, cols)) + 32) * 32)) != hipSuccess) return e;
        }
        LP = AmgLevelPlan();  // host copy no longer needed
    }
    const Tunables& T = tunables();
    if (T.amg_alpha > 0.0) H.alpha = T.amg_alpha;
    if (T.amg_coarse4 >= 0) H.coarse4 = T.amg_coarse4 != 0;
    if (T.amg_coarse4_from > 0) H.coarse4_from = T.amg_coarse4_from;
    if (T.amg_dense_period > 0) H.dense_period = T.amg_dense_period;
    {   // cycle doubling at the first sparse level of at most SHK_AMG_W_ROWS rows (experiment; default 0 = V-cycle:
        // measured at 10M | 1M rows it saves 4 | 10 % of the iterations and costs 5 | 44 % more time per step)
        const int64_t wrows = T.amg_w_rows;
        H.w_level = 0;
        for (size_t l = 1; l < nx && wrows > 0; ++l)
            if (H.lv[l].n > 0 && H.lv[l].n <= wrows) { H.w_level = l; break; }
    }
    if (T.amg_damp_scale > 0.0) {   // robustness experiments: every damping of the cycle times f
        const double f = T.amg_damp_scale;
        H.c1 *= f; H.c2 *= f; for (double& v : H.c4) v *= f;
    }
    if ((e = dev_alloc(c, &H.x0, (size_t)n_loc0)) != hipSuccess) return e;
    if ((e = dev_alloc(c, &H.x1, (size_t)n_loc0)) != hipSuccess) return e;
    if ((e = dev_alloc(c, &H.x2, (size_t)n_loc0)) != hipSuccess) return e;
    if ((e = zero_async(c, H.x2, (size_t)n_loc0 * sizeof(float))) != hipSuccess) return e;
    if ((e = zero_async(c, H.x1, (size_t)n_loc0 * sizeof(float))) != hipSuccess) return e;
    return zero_async(c, H.x0, (size_t)n_loc0 * sizeof(float));
}
}  // namespace shk

extern "C" {

const char* shk_last_error(void) { return g_err.c_str(); }
int shk_version(void) { return 300; }   // round * 100: the ABI of include/shakti_hip.h as of round 3

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
    p->krylov_rtol = 1e-10; p->krylov_atol = 1e-50; p->krylov_max_it = 20000; p->krylov_check_every = 0;
    p->krylov_fail_rtol = 1e-6;
    p->krylov_newton_eta = 0.1;
    p->krylov_forcing = 0.1;
    p->krylov_warm_start = 4;
    p->precond = SHK_PC_JACOBI;
    return 0;
}

int shk_create_local(int device_id, int64_t n_own, int64_t n_ghost, int64_t ne, const double* xy,
                     const int32_t* cells, shk_ctx** out) {
    if (!out) return fail("null output pointer");
    *out = nullptr;
    if (!xy || !cells) return fail("null mesh arrays");
    if (n_own < 1 || n_ghost < 0 || n_own + n_ghost < 3 || ne < 1) return fail("mesh needs at least one triangle");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail("device_id out of range: no such GPU");
    HIPCHK(hipSetDevice(device_id));
    Ctx* c = new Ctx();
    c->device = device_id;
    c->n_own = n_own;
    c->n_loc = n_own + n_ghost;
    c->ne = ne;
    shk_default_params(&c->params);
    derive_params(c);
    {
        const double* q = &SHK_QUAD_DEFAULT[0][0];
        if (set_quadrature(c, SHK_NQ_DEFAULT, q)) { delete c; return -1; }
        set_poly_rule(c);
    }
    PlanOptions opt;
    const Tunables& T = tunables();
    // k_assemble finds a slot's slice with three compares: a block owns at most 4 slices
    if (T.asm_slices < 1 || T.asm_slices > 4) { delete c; return fail("SHK_ASM_SLICES must be in 1..4 (the assembly kernel's limit)"); }
    opt.slices_max = T.asm_slices;
    opt.cells_max = std::min(std::max(64, T.asm_cells), kAsmCellsMax);
    opt.slots_max = kAsmSlotsMax;
    opt.sort_window = std::max(64, T.sort_window);
    opt.rim_first = T.sort_rim;
    opt.reorder = T.reorder;
    opt.amg = T.amg;
    opt.amg_coarsest = T.amg_coarsest;
    std::string err = build_plan(c->n_own, c->n_loc, ne, xy, cells, opt, c->plan);
    if (!err.empty()) { delete c; return fail("plan: " + err); }
    const HostPlan& P = c->plan;
    c->nnz = P.A.nnz;
    c->slots = P.A.slots;
    c->nblk = (int)P.blk_slice0.size() - 1;
    c->cells_staged = (int64_t)P.blk_cells.size();
    c->grid = (int)std::min<int64_t>(kMaxParts, std::max<int64_t>(1, (c->n_own + kBlock - 1) / kBlock));
    c->slots16 = (int64_t)P.A.col16.size();
    // one assembly pass: block descriptors, halo lists, staged cells (8 B), incidence lists, one plan word per slot, the
    // 13 nodal doubles + Dirichlet flag of every own row (halo gathers are re-reads), and F, 1/diag, the values
    c->asm_bytes = 4.0 * (double)(P.blk_desc.size() + P.blk_halo.size() + P.incptr.size()) + 2.0 * (double)(P.blk_cellv.size() + P.inccode.size())
                   + 4.0 * (double)P.A.slots + 105.0 * (double)c->n_own + 16.0 * (double)c->n_own + 8.0 * (double)P.A.slots;
    c->np = c->grid;
    c->warm_its = std::max(1, std::min(T.warm_its, (int)Ctx::kWarmIts));   // experiments
    if (P.verts_max > kAsmVertsMax) { delete c; return fail("an assembly block touches more than 768 vertices (degenerate mesh?)"); }
    c->asm_lds = assemble_lds_bytes(P, &c->asm_region_a);
    c->asm_lds_res = assemble_lds_bytes(P, &c->asm_region_a_res, true);
    if (c->asm_lds > 160 * 1024) { delete c; return fail("assembly LDS budget exceeds 160 KiB"); }
    auto bail = [&](hipError_t e, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        shk_destroy(reinterpret_cast<shk_ctx*>(c));
        return fail(m);
    };
    hipError_t e;
    const size_t nl = (size_t)c->n_loc;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "stream");
    if ((e = hipEventCreateWithFlags(&c->poll_ev[0], hipEventDisableTiming)) != hipSuccess) return bail(e, "event");
    if ((e = hipEventCreateWithFlags(&c->poll_ev[1], hipEventDisableTiming)) != hipSuccess) return bail(e, "event");
    if ((e = hipEventCreateWithFlags(&c->poll_ev[2], hipEventDisableTiming)) != hipSuccess) return bail(e, "event");
    // mesh (internal numbering)
    if ((e = dev_alloc(c, &c->d_xy, nl)) != hipSuccess) return bail(e, "alloc xy");
    if ((e = upload_sync(c, c->d_xy, P.xy.data(), nl * sizeof(double2))) != hipSuccess)
        return bail(e, "copy xy");
#define UP(dst, src) if ((e = upload(c, &c->dst, P.src)) != hipSuccess) return bail(e, "upload " #src)
    UP(d_cells, cells); UP(d_perm, perm); UP(d_sell_ptr, A.ptr); UP(d_sell_col, A.col); UP(d_rowlen, A.rowlen);
    UP(d_cbase, A.cbase); UP(d_ptr16, A.ptr16); UP(d_col16, A.col16);
    UP(d_lastcell, lastcell); UP(d_blk_desc, blk_desc); UP(d_blk_halo, blk_halo); UP(d_blk_cellv, blk_cellv);
    UP(d_incptr, incptr); UP(d_inccode, inccode); UP(d_slotsrc, slotsrc);
#undef UP
    // the host copies of the big plan arrays are no longer needed (the SELL pattern stays for get_csr)
    c->plan.xy = std::vector<double>();
    c->plan.cells = std::vector<int32_t>();
    c->plan.blk_cells = std::vector<int32_t>();
    c->plan.blk_halo = std::vector<int32_t>();
    c->plan.blk_cellv = std::vector<uint16_t>();
    c->plan.inccode = std::vector<uint16_t>();
    c->plan.slotsrc = std::vector<uint32_t>();
    c->plan.incptr = std::vector<int32_t>();
    c->plan.lastcell = std::vector<int32_t>();
    if ((e = dev_alloc(c, &c->d_io, 2 * nl)) != hipSuccess) return bail(e, "alloc io");
    for (int fidx = 0; fidx < SHK_FIELD_COUNT; ++fidx) {
        if (fidx == SHK_Q) continue;
        if ((e = dev_alloc(c, &c->f[fidx], nl)) != hipSuccess) return bail(e, "alloc field");
        if ((e = zero_async(c, c->f[fidx], nl * sizeof(double))) != hipSuccess) return bail(e, "memset");
    }
    double** vecs[] = {&c->d_melt_tmp, &c->d_b_tmp, &c->d_m0, &c->d_F, &c->d_dinv, &c->d_r, &c->d_rhat,
                       &c->d_p, &c->d_v, &c->d_s, &c->d_t, &c->d_y, &c->d_ytot, &c->d_rhs};
    for (double** v : vecs) {
        if ((e = dev_alloc(c, v, nl)) != hipSuccess) return bail(e, "alloc vector");
        if ((e = zero_async(c, *v, nl * sizeof(double))) != hipSuccess) return bail(e, "memset");
    }
    if ((e = dev_alloc(c, &c->d_vals, (size_t)c->slots)) != hipSuccess) return bail(e, "alloc vals");
    if ((e = dev_alloc(c, &c->d_vals_s, (size_t)c->slots)) != hipSuccess) return bail(e, "alloc vals_s");
    if ((e = dev_alloc(c, &c->d_vals32, (size_t)c->slots)) != hipSuccess) return bail(e, "alloc vals32");
    if (amg_packed_level(c->n_own, c->slots16)) {
        if ((e = dev_alloc(c, &c->d_pk, (size_t)c->slots16)) != hipSuccess) return bail(e, "alloc packed values");
    }
    if ((e = dev_alloc(c, &c->d_dinv32, nl)) != hipSuccess) return bail(e, "alloc dinv32");
    if ((e = zero_async(c, c->d_dinv32, nl * sizeof(float))) != hipSuccess) return bail(e, "memset");
    if ((e = dev_alloc(c, &c->d_bcflag, nl)) != hipSuccess) return bail(e, "alloc bcflag");
    if ((e = zero_async(c, c->d_bcflag, nl)) != hipSuccess) return bail(e, "memset");
    if ((e = dev_alloc(c, &c->d_part, (size_t)P_COUNT * kMaxParts)) != hipSuccess) return bail(e, "alloc partials");
    if ((e = zero_async(c, c->d_part, P_COUNT * kMaxParts * sizeof(double))) != hipSuccess) return bail(e, "memset");
    c->d_red = c->d_part;
    if ((e = dev_alloc(c, &c->d_state, 1)) != hipSuccess) return bail(e, "alloc state");
    if ((e = zero_async(c, c->d_state, sizeof(KrylovState))) != hipSuccess) return bail(e, "memset");
    if ((e = hipHostMalloc((void**)&c->h_state, 2 * sizeof(KrylovState))) != hipSuccess) return bail(e, "pinned");
    if ((e = hipHostMalloc((void**)&c->h_part, kMaxParts * sizeof(double))) != hipSuccess) return bail(e, "pinned");
    if ((e = prepare_kernels(c)) != hipSuccess) return bail(e, "hipFuncSetAttribute(dynamic LDS)");
    // multigrid hierarchy of the owned diagonal block
    if (!c->plan.amg.empty()) {
        if ((e = amg_upload(c, c->plan.amg, c->amg_local, c->n_loc)) != hipSuccess) return bail(e, "amg upload");
        c->plan.amg = std::vector<AmgLevelPlan>();
    }
    {
        float** vs[] = {&c->d_phat, &c->d_shat, &c->d_p32, &c->d_s32};
        for (float** v : vs) {
            if ((e = dev_alloc(c, v, nl)) != hipSuccess) return bail(e, "amg alloc");
            if ((e = zero_async(c, *v, nl * sizeof(float))) != hipSuccess) return bail(e, "memset");
        }
    }
    // (every upload and zero fill above travelled on the context's own stream)
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "synchronize");
    *out = reinterpret_cast<shk_ctx*>(c);
    return 0;
}

int shk_create(int device_id, int64_t nv, int64_t ne, const double* xy, const int32_t* cells, shk_ctx** out) {
    return shk_create_local(device_id, nv, 0, ne, xy, cells, out);
}

int shk_destroy(shk_ctx* ctx) {
    if (!ctx) return 0;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    (void)hipSetDevice(c->device);
    if (c->poisoned) {
        // The stream holds a collective that will never complete: hipStreamSynchronize, ncclCommDestroy, hipFree and
        // hipStreamDestroy would all wait for it.  Abort the communicator if the library can, leave the device memory
        // to the process exit, and return -- so that the rank can exit non-zero and the launcher tears the job down.
        comm_abort(c);
        delete c;
        return 0;
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    comm_destroy(c);
    if (c->comm.h_send) (void)hipHostFree(c->comm.h_send);
    if (c->comm.h_recv) (void)hipHostFree(c->comm.h_recv);
    if (c->comm.h_red) (void)hipHostFree(c->comm.h_red);
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (hipEvent_t ev : c->poll_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_part) (void)hipHostFree(c->h_part);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int shk_set_params(shk_ctx* ctx, const shk_params* p) {
    CHECK_CTX(ctx);
    if (!p) return fail("null params");
    if (!(p->g > 0 && p->rho_i > 0 && p->rho_w > 0 && p->nu > 0 && p->Lh > 0)) return fail("non-positive constant");
    if (p->newton_max_it < 0 || p->krylov_max_it < 1 || p->krylov_check_every < 0) return fail("bad iteration limits");
    if (!(p->krylov_fail_rtol >= 0)) return fail("krylov_fail_rtol must be >= 0");
    if (!(p->krylov_newton_eta >= 0 && p->krylov_newton_eta <= 1)) return fail("krylov_newton_eta must be in [0, 1]");
    if (!(p->krylov_forcing >= 0 && p->krylov_forcing <= 1)) return fail("krylov_forcing must be in [0, 1]");
    if (p->krylov_warm_start < 0 || p->krylov_warm_start > Ctx::kWarmDepth) return fail("krylov_warm_start must be in 0..4");
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (p->precond != SHK_PC_JACOBI && p->precond != SHK_PC_AMG && p->precond != SHK_PC_AMG_LOCAL)
        return fail("unknown preconditioner id");
    AmgHierarchy* H = nullptr;
    if (p->precond == SHK_PC_AMG_LOCAL || (p->precond == SHK_PC_AMG && c->comm.nranks <= 1)) {
        H = &c->amg_local;
        if (!H->ready())
            return fail("multigrid hierarchy unavailable for this context (mesh of <= 64 vertices, or SHK_AMG=0)");
    } else if (p->precond == SHK_PC_AMG) {
        // distributed hierarchy: built on first use, COLLECTIVELY (every subdomain must make this call)
        H = &c->amg_dist;
        if (!H->ready()) {
            HIPCHK(hipSetDevice(c->device));
            std::string err;
            if (amg_setup_distributed(c, err)) return fail("distributed multigrid setup: " + err);
        }
    }
    c->params = *p;
    c->use_amg = H != nullptr;
    c->amg = H;
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
    const size_t n = (size_t)c->n_loc * (field == SHK_Q ? 2 : 1);
    HIPCHK(hipMemcpyAsync(c->d_io, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (field == SHK_Q) launch_split_q(c, c->d_io);
    else launch_permute_in(c, c->d_io, c->f[field]);
    WAITCHK(c);
    HIPCHK(hipGetLastError());
    return 0;
}

static int get_vector(Ctx* c, const double* dev, double* host) {
    launch_permute_out(c, dev, c->d_io);
    HIPCHK(hipMemcpyAsync(host, c->d_io, (size_t)c->n_loc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    WAITCHK(c);
    HIPCHK(hipGetLastError());
    return 0;
}

int shk_get_field(shk_ctx* ctx, int32_t field, double* host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!host) return fail("null host array");
    if (field < 0 || field >= SHK_FIELD_COUNT) return fail("unknown field id");
    HIPCHK(hipSetDevice(c->device));
    if (field == SHK_Q) {
        launch_join_q(c, c->d_io);
        HIPCHK(hipMemcpyAsync(host, c->d_io, (size_t)c->n_loc * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        WAITCHK(c);
        HIPCHK(hipGetLastError());
        return 0;
    }
    return get_vector(c, c->f[field], host);
}

int shk_set_dirichlet(shk_ctx* ctx, int64_t n, const int32_t* dofs, double value) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (n < 0 || (n > 0 && !dofs)) return fail("bad dof list");
    HIPCHK(hipSetDevice(c->device));
    std::vector<uint8_t> flag((size_t)c->n_loc, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (dofs[i] < 0 || dofs[i] >= c->n_loc) return fail("Dirichlet dof outside [0, nv)");
        flag[c->plan.iperm[dofs[i]]] = 1;
    }
    // on the context's stream (behind whatever still reads the old flags; the stream is non-blocking, so a null-stream
    // copy would be ordered with nothing); `flag` lives until the synchronisation below
    HIPCHK(hipMemcpyAsync(c->d_bcflag, flag.data(), (size_t)c->n_loc, hipMemcpyHostToDevice, c->stream));
    launch_slot_bc(c);   // per-slot Dirichlet codes into the plan words (all zero when n == 0)
    WAITCHK(c);
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
    c->jac_valid = true;
    c->assembled_dt = dt;
    return 0;
}

int shk_get_residual(shk_ctx* ctx, double* host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!host) return fail("null host array");
    if (!c->assembled) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    return get_vector(c, c->d_F, host);
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
    std::vector<double> sv;
    if (values) {
        if (!c->assembled || !c->jac_valid) return fail("no assembled Jacobian: call shk_assemble first");
        HIPCHK(hipSetDevice(c->device));
        sv.resize((size_t)c->slots);
        HIPCHK(hipMemcpyAsync(sv.data(), c->d_vals, sv.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        WAITCHK(c);
    }
    std::vector<int32_t> rp, ci;
    std::vector<double> va;
    sell_to_csr(c->plan, values ? sv.data() : nullptr, rp, ci, values ? &va : nullptr);
    if (rowptr) std::memcpy(rowptr, rp.data(), rp.size() * sizeof(int32_t));
    if (colidx) std::memcpy(colidx, ci.data(), ci.size() * sizeof(int32_t));
    if (values) std::memcpy(values, va.data(), va.size() * sizeof(double));
    return 0;
}

// One BiCGStab run on A' y = rhs (x0 = 0).  The host only polls a stop flag: chunk k+1 is already queued
// when chunk k's flag is read, so the GPU never idles; kernels after the stop return immediately.
static int krylov_inner(Ctx* c, const double* rhs, int max_it, KrylovState* out) {
    krylov_init(c, rhs);
    // iterations enqueued per stop-flag poll: 0 = auto (a multigrid iteration is ~90 launches: poll often)
    const int env_chunk = tunables().krylov_chunk;   // experiments
    const int chunk = c->params.krylov_check_every > 0 ? c->params.krylov_check_every
                      : env_chunk > 0 ? env_chunk : c->use_amg ? 2 : 16;
    int it = 0, slot = 0;
    const int saved_max = c->params.krylov_max_it;
    c->params.krylov_max_it = max_it;
    // while profiling: the event-pool position at which every enqueued iteration begins, so that the launches queued
    // behind the stop flag can be taken out of the profile by POSITION once the run knows where it stopped.  (Their
    // duration alone does not identify them: a launch that returns at once measures 2-5 us, the small multigrid levels'
    // real kernels 6-10 us -- and counted as launches of 3-5 us they diluted the average k_spmv launch of the profiled
    // step from ~190 to ~140 us until this was found in a kernel trace of that step.)
    std::vector<size_t> iter_ev;
    auto enqueue = [&](int sl, int n) -> hipError_t {
        hipError_t e = hipSuccess;
        for (int k = 0; k < n; ++k) {
            if (c->profiling) iter_ev.push_back(c->ev_used);
            if ((e = krylov_iteration(c, it + k)) != hipSuccess) return e;
        }
        it += n;
        e = hipMemcpyAsync(&c->h_state[sl], c->d_state, sizeof(KrylovState), hipMemcpyDeviceToHost,
                                      c->stream);
        if (e != hipSuccess) return e;
        return hipEventRecord(c->poll_ev[sl], c->stream);
    };
    hipError_t e = enqueue(slot, chunk);
    int rc = 0;
    // (profiling uses the same pipelined polling: with the queue kept full the per-launch event durations agree
    // with a rocprofv3 trace -- a GPU left idle between iterations runs every kernel ~10 % slower -- and the launches
    // that return at once behind the stop flag are dropped when the events are read)
    // The next chunk is queued before the previous one's flag is read (the GPU never waits for the host); everything
    // queued behind the stop costs ~90 empty launches per iteration.  SHK_KRYLOV_NEAR=f (experiment, off by default) stops
    // queueing ahead once ||r|| is within a factor f of the target and then waits for every iteration: measured at 10M DOF
    // with f = 10, 58.4 -> 57.5 ms per step, but the GPU idles during those round trips and every kernel launched after an
    // idle gap runs slower (k_spmv 165 -> 188 us on average over the step, the smoothers alike) -- the launches saved are
    // paid back in kernel time, and the per-kernel figures of the bench line stop describing the kernels.
    const double near2 = std::pow(tunables().krylov_near, 2);
    bool careful = false;
    while (e == hipSuccess) {
        if (!careful) {
            const int prev = slot;
            slot ^= 1;
            if ((e = enqueue(slot, chunk)) != hipSuccess) break;
            if ((e = wait_event(c, c->poll_ev[prev])) != hipSuccess) break;
            const KrylovState& st = c->h_state[prev];
            if (st.done) { *out = st; break; }
            careful = near2 > 0.0 && st.target2 > 0.0 && st.rr_last <= near2 * st.target2;
        } else {
            if ((e = wait_event(c, c->poll_ev[slot])) != hipSuccess) break;
            if (c->h_state[slot].done) { *out = c->h_state[slot]; break; }
            if ((e = enqueue(slot, 1)) != hipSuccess) break;
        }
        if (it > max_it + 4 * chunk) { rc = fail("Krylov driver ran past max_it without a stop flag"); break; }
    }
    c->params.krylov_max_it = saved_max;
    if (e == hipErrorLaunchTimeOut) return fail(kStallMsg);
    if (e != hipSuccess) return fail(std::string("krylov enqueue: ") + hipGetErrorString(e));
    if (c->profiling && rc == 0 && out->done && out->its >= 0 && (size_t)out->its < iter_ev.size()) {
        // iterations 0 .. its-1 ran; everything queued for iterations > its returned at once.  Iteration `its` itself: with
        // the early stop test only its first launch (k_krylov_check) did any work; without it the cycle and the product that
        // open the iteration ran and the rest returned at once (those keep the duration filter of shk_profile_read).
        const size_t its = (size_t)out->its;
        const size_t later = its + 1 < iter_ev.size() ? iter_ev[its + 1] : c->ev_used;
        for (size_t i = later; i < c->ev_used; ++i) c->ev_pool[i].phase = -1;
        const bool early = c->use_amg && (c->comm.kind == Comm::NONE || c->comm.nranks <= 1) && tunables().krylov_early_check;
        if (early) for (size_t i = iter_ev[its] + 1; i < later; ++i) c->ev_pool[i].phase = -1;
    }
    return rc;
}

static int read_aux_norm(Ctx* c, double* out) {  // fixed-order host sum of the (reduced) P_AUX partials
    HIPCHK(hipMemcpyAsync(c->h_part, c->d_red + (size_t)P_AUX * c->red_stride, (size_t)c->np * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    WAITCHK(c);
    double s = 0.0;
    for (int i = 0; i < c->np; ++i) s += c->h_part[i];
    *out = std::sqrt(s);
    return 0;
}

// Solve A' y = F to  ||F - A' y|| <= max(rtol ||F||, atol)  measured on the TRUE residual: BiCGStab's
// recursive residual is only trusted to stop an inner run; each run is followed by one explicit
// residual, and the correction equation is solved again if the target was missed.
// Storage of the warm start (launch_warm_start), allocated when a solve first asks for it.
static int warm_alloc(Ctx* c, int newton_it) {
    if (!c->d_part_w) {
        HIPCHK(dev_alloc(c, &c->d_part_w, (size_t)Ctx::kWarmDots * kMaxParts));
        // (on the context's stream: it is a non-blocking stream, which a null-stream hipMemset would NOT be ordered with --
        //  the copy that follows could be overtaken by the zeroing)
        HIPCHK(hipMemsetAsync(c->d_part_w, 0, (size_t)Ctx::kWarmDots * kMaxParts * sizeof(double), c->stream));
        HIPCHK(dev_alloc(c, &c->d_red_w, (size_t)Ctx::kWarmDots));
    }
    for (int j = 0; j < Ctx::kWarmDepth; ++j)
        if (!c->d_guess[newton_it][j]) {
            HIPCHK(dev_alloc(c, &c->d_guess[newton_it][j], (size_t)c->n_loc));
            HIPCHK(hipMemsetAsync(c->d_guess[newton_it][j], 0, (size_t)c->n_loc * sizeof(double), c->stream));
        }
    return 0;
}

static int krylov_solve(Ctx* c, int* its, int* converged, double* relres, int newton_it = 0,
                        double newton_floor = 0.0, double fnorm = -1.0, bool iterate_moved_little = false) {
    const bool first_of_step = newton_it == 0;
    if (c->use_amg) {
        // Galerkin coarse operators (and the float copy) of the Jacobian just assembled.  (Keeping the first Newton system's
        // hierarchy for the later iterations of a solve was measured: 96.6 ms per step either way at 10M DOF -- not kept.)
        HIPCHK(amg_numeric_setup(c, *c->amg, first_of_step, false, !first_of_step && iterate_moved_little));
    } else {
        HIPCHK(halo_exchange(c, c->d_dinv));  // ghost columns of A' = A D^-1 need their owners' diagonal
        launch_scale(c);
    }
    // newton_floor: absolute residual below which more digits cannot change Newton's own stopping decision
    // (shk_newton_solve); the solve's target is max(rtol ||F||, atol, newton_floor)
    const double rtol = c->params.krylov_rtol, atol = std::max(c->params.krylov_atol, newton_floor);
    int total = 0, conv = 0;
    double target = 0.0, rhs_norm = 0.0, rt = 0.0, rt_prev = 0.0, r_start = 0.0;
    const int max_outer = 12;
    // warm start (first solve of a time step inside shk_newton_solve, which knows ||F||): the loop then begins with
    // the correction equation of the projected guess, as if a first pass had already run
    const int depth = c->params.krylov_warm_start;
    const bool warmable = depth > 0 && fnorm > 0.0 && newton_it >= 0 && newton_it < c->warm_its;
    const bool warm = warmable && std::min(c->n_guess[newton_it], depth) > 0;
    if (warm) {
        HIPCHK(launch_warm_start(c, newton_it));
        rhs_norm = fnorm;
        target = std::max(rtol * rhs_norm, atol);
        rt_prev = HUGE_VAL;
    }
    for (int outer = warm ? 1 : 0; outer < max_outer; ++outer) {
        if (outer == 0) { c->cur_rtol2 = rtol * rtol; c->cur_atol2 = atol * atol; }
        else { c->cur_rtol2 = 0.0; c->cur_atol2 = target * target; }
        KrylovState st{};
        const int budget = c->params.krylov_max_it - total;
        if (budget <= 0) break;
        if (krylov_inner(c, outer == 0 ? c->d_F : c->d_rhs, budget, &st)) return -1;
        total += st.its;
        if (warm && outer == 1) r_start = std::sqrt(st.rhs2);
        if (outer == 0) {
            rhs_norm = std::sqrt(st.rhs2);
            target = std::max(rtol * rhs_norm, atol);
        }
        launch_accumulate(c, outer == 0);
        HIPCHK(launch_true_residual(c));
        if (read_aux_norm(c, &rt)) return -1;
        if (!(rt > target)) { conv = std::isfinite(rt) ? 1 : 0; break; }
        if (st.breakdown && st.its == 0) break;  // no progress possible
        // round-off floor of the true residual (eps * cond(J)): another pass that gained less than 2x will not
        // get there either -- stop instead of burning the iteration budget (reported as not converged)
        if (outer > 0 && !(rt < 0.5 * rt_prev)) break;
        rt_prev = rt;
    }
    HIPCHK(hipGetLastError());
    if (warmable && conv) {   // keep this solution for the same Newton iteration of the next steps: newest first
        if (warm_alloc(c, newton_it)) return -1;
        double** g = c->d_guess[newton_it];
        std::rotate(g, g + Ctx::kWarmDepth - 1, g + Ctx::kWarmDepth);
        // the copy rides on the Newton update that follows (k_newton_update streams the solution anyway): the runtime's
        // own device-to-device copy took 205 us for these 80 MB at 10M DOF, 0.4 TB/s
        c->pending_keep = c->d_guess[newton_it][0];
        c->n_guess[newton_it] = std::min(c->n_guess[newton_it] + 1, (int)Ctx::kWarmDepth);
    }
    if (c->use_amg && first_of_step) {  // feedback for the coarsest-inverse refresh policy (like with like:
        int cost = total;               // only the first Newton system of each solve is compared)
        // a warm start covers fewer decades: scale to the full span -- only when the start is clearly above the target
        // (a projected guess that already sits at the target would blow the ratio up), by at most 4x, within max_it
        if (warm && r_start > 10.0 * target && rhs_norm > target) {
            const double f = std::min(4.0, std::log(rhs_norm / target) / std::log(r_start / target));
            cost = (int)std::min<double>(std::lround(total * f), (double)c->params.krylov_max_it);
        }
        c->amg->its_last = cost;
        if (c->amg->its_fresh == 0) c->amg->its_fresh = cost;
    }
    if (its) *its = total;
    if (converged) *converged = conv;
    if (relres) *relres = rhs_norm > 0 ? rt / rhs_norm : 0.0;
    return 0;
}

int shk_linear_solve(shk_ctx* ctx, int32_t* its, int32_t* converged, double* rel_residual) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c->assembled || !c->jac_valid) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    int k = 0, cv = 0;
    double rr = 0;
    if (krylov_solve(c, &k, &cv, &rr)) return -1;
    launch_newton_update(c, false);  // dx = D^-1 y without touching N
    WAITCHK(c);
    if (its) *its = k;
    if (converged) *converged = cv;
    if (rel_residual) *rel_residual = rr;
    return 0;
}

int shk_spmv(shk_ctx* ctx, const double* x_host, double* y_host) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!x_host || !y_host) return fail("null host array");
    if (!c->assembled || !c->jac_valid) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_io, x_host, (size_t)c->n_loc * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_permute_in(c, c->d_io, c->d_p);
    HIPCHK(hipMemsetAsync(c->d_v, 0, (size_t)c->n_loc * sizeof(double), c->stream));
    launch_spmv_plain(c, c->d_vals, c->d_p, c->d_v);
    return get_vector(c, c->d_v, y_host);
}

// ||F||_2 with a fixed summation order (device partials, host sum of <= 1024 values).
static int residual_norm(Ctx* c, double* out) {
    launch_norm2(c, c->d_F, c->d_part + P_AUX * kMaxParts);
    HIPCHK(allreduce_parts(c, P_AUX, 1));
    return read_aux_norm(c, out);
}

// Absolute residual at which the linear solve of Newton iteration `it` may stop (shk_newton_solve): Newton's own floor
// krylov_newton_eta x T, or -- inexact Newton, shk_params.krylov_forcing -- more, where more cannot change the outcome.
//   r = ||F_it||, T = Newton's stopping threshold, hist1[k] = ||F_{k+1}|| / ||F_k|| of the previous solve.
// e = hist1[it] r is what this iteration is expected to leave behind however well its system is solved (the nonlinear
// remainder).  While e is an order of magnitude above T another iteration follows anyway and inherits a right-hand side of
// ~e: digits below forcing x e are digits the next iteration recovers for free.  The prediction is only believed where it
// has earned it; every condition answers a test that took one Newton iteration more than the LU oracle without it:
//   similar   the history comes from a solve like this one: same dt, ||F_0|| within a factor 2 (a forcing that jumps between
//             two steps -- moulin input x 20, then off -- moves ||F_0|| by more);
//   settled   the ratio this iteration left in the last TWO solves within a factor 2 of each other (where the Newton count
//             is about to drop the ratios change by an order of magnitude per step -- 3.6e-3, 4.8e-4, 4e-5 -- and the rule
//             cut a solve short at 1.04 T that exact solves finish at 0.9 T; in the settled transient of the bench they
//             change by 0.5 % per step);
//   was_last  the previous solve ENDED with the iteration after this one, well inside its threshold (<= 0.3 T: mostly the
//             linear floor 0.1 T, so that twice the nonlinear part still fits);
//   pred2     that iteration's remainder, predicted with the previous solve's quadratic constant kappa = ||F_{k+1}|| /
//             ||F_k||^2 as kappa e^2, lies 100 x below T (an inexact solve enlarges the next right-hand side and that
//             remainder with its square).
// The iteration expected to end the solve, and every iteration without such a history, is solved as before.
static double forcing_floor(Ctx* c, int it, double r, double T, bool similar, const double* hist1) {
    double floor = c->params.krylov_newton_eta * T;
    if (!(c->params.krylov_forcing > 0.0) || !similar || it >= Ctx::kNewtonHist) return floor;
    const double r1 = hist1[it], r2 = c->newton_ratio2[it], fk = c->newton_fk[it];
    const bool settled = r1 > 0.0 && r2 > 0.0 && r1 < 2.0 * r2 && r2 < 2.0 * r1;
    if (!settled || !(r1 * r > 10.0 * T)) return floor;
    const double e = r1 * r;
    const double pred2 = fk > 0.0 ? (r1 / fk) * e * e : HUGE_VAL;
    const bool was_last = c->newton_prev == it + 2 && c->newton_hist_margin > 0.0 && c->newton_hist_margin <= 0.3;
    const double f = c->params.krylov_forcing * std::min(r1, 1e-2) * r;
    const bool forced = was_last && pred2 < 0.01 * T && f > floor && f > c->params.krylov_rtol * r;
    if (forced) { floor = f; c->n_forced += 1; }
    if (tunables().debug)
        fprintf(stderr, "[shk] newton it %d: ||F|| %.3e (%.3g T), previous solve: %d its, ratio %.3e (before it %.3e), ended at "
                        "%.3g T; expected after this iteration %.3g T, after the next %.3g T -> linear floor %.3g T%s\n", it, r,
                r / T, c->newton_prev, r1, r2, c->newton_hist_margin, e / T, pred2 / T, floor / T, forced ? " (inexact)" : "");
    return floor;
}

int shk_newton_solve(shk_ctx* ctx, double dt, shk_solve_info* info) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!(dt > 0)) return fail("dt must be positive");
    HIPCHK(hipSetDevice(c->device));
    shk_solve_info I{};
    // DOLFINx NewtonSolver::solve: residual first, then J / solve / update / residual.
    HIPCHK(halo_exchange(c, c->f[SHK_N]));  // form(x): ghost update of the iterate
    launch_assemble(c, dt);  // F and J of the current iterate in one pass
    c->n_asm_full += 1;
    c->assembled = true;
    c->jac_valid = true;
    c->assembled_dt = dt;
    // The pass after the LAST update of a solve needs no Jacobian (only ||F|| is looked at): when this solve is expected to
    // end with iteration `expect` -- the previous solve's count: consecutive steps of a transient take the same number --
    // that pass runs the residual-only kernel instance (no element Jacobians, no slot phase, 0.7 GB less to write at 10M
    // DOF).  If Newton then does NOT stop, the Jacobian of that iterate is assembled after all, by the full pass (whose F
    // replaces the residual-only one's): a misprediction costs one extra pass, never a different result.
    const int expect = (tunables().predict_last && c->newton_prev > 0) ? c->newton_prev : 0;
    double hist1[Ctx::kNewtonHist];   // the previous solve's ratios: c->newton_ratio is rewritten as this solve proceeds
    for (int k = 0; k < Ctx::kNewtonHist; ++k) hist1[k] = c->newton_ratio[k];
    double r = 0.0;
    if (residual_norm(c, &r)) return -1;
    I.residual0 = r;
    I.residual = r;
    // (history of the inexact-Newton rule below: it only speaks for a solve that resembles the one it came from)
    const bool similar = c->newton_hist_dt == dt && I.residual0 < 2.0 * c->newton_hist_f0 && c->newton_hist_f0 < 2.0 * I.residual0;
    int it = 0;
    bool conv = r < c->params.newton_atol;  // relative residual is 1 at iteration 0
    while (!conv && it < c->params.newton_max_it) {
        int k = 0, kc = 0;
        double rr = 0.0;
        // Newton stops at ||F|| < max(atol, rtol ||F_0||) (DOLFINx defaults, solvers.py:52).  A linear residual a factor
        // krylov_newton_eta (0.1) below that threshold cannot change that decision any more, so later Newton iterations
        // -- whose right-hand side is already ~1e-5 ||F_0|| -- are not driven ten digits below their own ||F|| (the first
        // iteration is unaffected: 0.1 x 1e-9 ||F_0|| = krylov_rtol ||F_0||).  Newton counts stay those of the LU oracle
        // in every parity test; 0 restores the pure relative rule.
        const double newton_target = std::max(c->params.newton_atol, c->params.newton_rtol * I.residual0);
        const double floor = forcing_floor(c, it, r, newton_target, similar, hist1);
        const double r_before = r;
        // (an iterate whose residual is already 1e-3 of the step's first one has barely moved: the multigrid keeps its coarse
        //  operators, amg_numeric_setup's top_only)
        if (krylov_solve(c, &k, &kc, &rr, it, floor, r, it > 0 && r < 1e-3 * I.residual0)) return -1;
        I.krylov_its += k;
        // a solve that stagnated between krylov_rtol and krylov_fail_rtol sits on its fp64 floor
        // eps || |J| |dx| || (Newton absorbs it); beyond that -- max_it, breakdown, divergence -- it has failed
        if (!kc && !(rr <= std::max(c->params.krylov_rtol, c->params.krylov_fail_rtol))) I.krylov_failed = 1;
        if (!(rr <= I.krylov_relres)) I.krylov_relres = rr;   // also keeps a NaN
        launch_newton_update(c, true);
        HIPCHK(halo_exchange(c, c->f[SHK_N]));
        ++it;
        const bool guess_last = expect > 0 && it >= expect;
        launch_assemble(c, dt, guess_last);
        (guess_last ? c->n_asm_res : c->n_asm_full) += 1;
        c->jac_valid = !guess_last;
        if (residual_norm(c, &r)) return -1;
        I.residual = r;
        if (!std::isfinite(r)) break;
        if (it - 1 < Ctx::kNewtonHist) { c->newton_ratio[it - 1] = r_before > 0.0 ? r / r_before : 0.0; c->newton_fk[it - 1] = r_before; }
        conv = (r < c->params.newton_atol) || (I.residual0 > 0 && r / I.residual0 < c->params.newton_rtol);
        if (guess_last && !conv && it < c->params.newton_max_it) {   // mispredicted: this iterate's Jacobian is needed
            launch_assemble(c, dt);
            c->n_asm_redo += 1;
            c->jac_valid = true;
        }
    }
    c->newton_prev = it;
    for (int k = it; k < Ctx::kNewtonHist; ++k) c->newton_ratio[k] = 0.0;   // iterations this solve did not run: no history
    for (int k = 0; k < Ctx::kNewtonHist; ++k) c->newton_ratio2[k] = similar ? hist1[k] : 0.0;
    c->newton_hist_f0 = I.residual0;
    c->newton_hist_dt = dt;
    {
        const double T = std::max(c->params.newton_atol, c->params.newton_rtol * I.residual0);
        c->newton_hist_margin = (conv && T > 0.0) ? I.residual / T : 0.0;
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
    HIPCHK(launch_update_explicit(c, dt));
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

int shk_set_halo(shk_ctx* ctx, int32_t n_nbr, const int32_t* nbr_rank, const int64_t* send_ptr,
                 const int32_t* send_idx, const int64_t* recv_ptr) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (n_nbr < 0 || (n_nbr > 0 && (!nbr_rank || !send_ptr || !recv_ptr))) return fail("bad halo plan");
    if (!c->comm.plans.empty()) return fail("halo plan already installed");
    HIPCHK(hipSetDevice(c->device));
    Comm& m = c->comm;
    HaloPlan P;
    P.n_own = c->n_own;
    P.nbr.assign(nbr_rank, nbr_rank + n_nbr);
    P.send_ptr.assign(1, 0);
    P.recv_ptr.assign(1, 0);
    if (n_nbr > 0) {
        P.send_ptr.assign(send_ptr, send_ptr + n_nbr + 1);
        P.recv_ptr.assign(recv_ptr, recv_ptr + n_nbr + 1);
    }
    for (int k = 0; k < n_nbr; ++k)
        if (P.send_ptr[k + 1] < P.send_ptr[k] || P.recv_ptr[k + 1] < P.recv_ptr[k] || (k > 0 && P.nbr[k] <= P.nbr[k - 1]))
            return fail("halo plan: offsets must be non-decreasing and neighbour ranks ascending");
    const int64_t nsend = P.send_ptr.back(), nrecv = P.recv_ptr.back();
    if (P.send_ptr[0] != 0 || P.recv_ptr[0] != 0) return fail("halo plan: offsets must start at 0");
    if (nrecv != c->n_loc - c->n_own) return fail("halo plan: receive counts do not cover the ghost vertices");
    if (nsend > 0 && !send_idx) return fail("halo plan: null send list");
    P.h_send_idx.resize((size_t)nsend);
    for (int64_t i = 0; i < nsend; ++i) {
        if (send_idx[i] < 0 || send_idx[i] >= c->n_own) return fail("halo plan: send index is not an owned vertex");
        P.h_send_idx[i] = c->plan.iperm[send_idx[i]];
    }
    if (dev_alloc(c, &P.d_send_idx, (size_t)nsend) != hipSuccess) return fail("halo alloc");
    if (dev_alloc(c, &m.d_sendbuf, (size_t)nsend) != hipSuccess) return fail("halo alloc");
    if (dev_alloc(c, &m.d_recvbuf, (size_t)nrecv) != hipSuccess) return fail("halo alloc");
    if (nsend > 0)
        HIPCHK(upload_sync(c, P.d_send_idx, P.h_send_idx.data(), (size_t)nsend * sizeof(int32_t)));
    HIPCHK(hipHostMalloc((void**)&m.h_send, std::max<size_t>(1, (size_t)nsend) * sizeof(double)));
    HIPCHK(hipHostMalloc((void**)&m.h_recv, std::max<size_t>(1, (size_t)nrecv) * sizeof(double)));
    m.h_red_cap = (size_t)P_COUNT * kMaxParts;
    HIPCHK(hipHostMalloc((void**)&m.h_red, m.h_red_cap * sizeof(double)));
    m.plans.push_back(std::move(P));
    return 0;
}

// Interior / boundary split of the level-0 sweeps (SplitSell): flag the SELL slices that store a ghost column --
// padding entries included, whatever column they carry -- and create the stream the exchanges travel on.
static hipError_t overlap_setup(Ctx* c) {
    const SellPattern& A = c->plan.A;
    std::vector<uint8_t> flag((size_t)std::max(A.nslice, 1), 0);
    std::vector<int32_t> list;
    for (int32_t s = 0; s < A.nslice; ++s) {
        bool ghost = false;
        for (int32_t k = A.ptr[s]; k < A.ptr[s + 1] && !ghost; ++k) ghost = A.col[k] >= c->n_own;
        if (A.cbase[s] >= 0)
            for (int32_t k = A.ptr16[s]; k < A.ptr16[s + 1] && !ghost; ++k) ghost = A.cbase[s] + (int32_t)A.col16[k] >= c->n_own;
        if (ghost) { flag[s] = 1; list.push_back(s); }
    }
    hipError_t e;
    if ((e = upload(c, &c->d_slice_ghost, flag)) != hipSuccess) return e;
    if ((e = upload(c, &c->d_bslices, list)) != hipSuccess) return e;
    c->n_bslices = (int)list.size();
    if ((e = dev_alloc(c, &c->d_part_b, (size_t)P_COUNT * kMaxParts)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(c->d_part_b, 0, (size_t)P_COUNT * kMaxParts * sizeof(double), c->stream)) != hipSuccess) return e;
    if ((e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming)) != hipSuccess) return e;
    c->overlap = true;
    return hipStreamSynchronize(c->stream);   // (uploads and fills above are on the context's stream)
}

static int comm_common(Ctx* c, int rank, int nranks, bool overlap_default) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("bad rank / nranks");
    if (c->comm.plans.empty()) return fail("call shk_set_halo before initialising the communicator");
    for (int r : c->comm.plans[0].nbr)
        if (r < 0 || r >= nranks || r == rank) return fail("halo plan names a neighbour outside the communicator");
    if (nranks > 1 && c->d_red == c->d_part) {
        double* red = nullptr;
        if (dev_alloc(c, &red, (size_t)P_COUNT) != hipSuccess) return fail("alloc reduced scalars");
        HIPCHK(hipMemsetAsync(red, 0, (size_t)P_COUNT * sizeof(double), c->stream));
        c->d_red = red;
        c->np = 1;           // every subdomain reads the same all-reduced scalars
        c->red_stride = 1;
    }
    // Interior / boundary overlap of the finest level's exchanges: OFF unless SHK_OVERLAP=1 asks for it.  Handing work to
    // a second stream and back costs ~15 us on this machine (tools/probe_stream_handoff.py), as much as a small grouped
    // send / recv is expected to take, and RCCL has never run with several ranks in this build's environment: the first
    // multi-GPU runs should measure the plain path first.  (Host-staged transport, 2 subdomains sharing one GPU at 1M
    // rows: 133 -> 183 ms per step with it -- the callback blocks the host anyway.)
    const bool want_overlap = tunables().overlap >= 0 ? tunables().overlap != 0 : overlap_default;
    if (nranks > 1 && want_overlap && !c->overlap) HIPCHK(overlap_setup(c));
    c->comm.rank = rank;
    c->comm.nranks = nranks;
    return 0;
}

int shk_comm_unique_id(void* id128) {
    if (!id128) return fail("null id buffer");
    if (const char* e = rccl_load()) return fail(e);
    if (rccl_unique_id(id128)) return fail("ncclGetUniqueId failed");
    return 0;
}

int shk_comm_init_rccl(shk_ctx* ctx, int32_t rank, int32_t nranks, const void* id128) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!id128) return fail("null id buffer");
    if (c->comm.kind != Comm::NONE) return fail("communicator already initialised");
    HIPCHK(hipSetDevice(c->device));
    if (const char* e = rccl_load()) return fail(e);
    if (comm_common(c, rank, nranks, false)) return -1;
    if (const char* e = rccl_init(c, rank, nranks, id128)) return fail(std::string("ncclCommInitRank: ") + e);
    return 0;
}

int shk_comm_init_callbacks(shk_ctx* ctx, int32_t rank, int32_t nranks, shk_exchange_fn exchange,
                            shk_allreduce_fn allreduce, void* user) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!exchange || !allreduce) return fail("null callback");
    if (c->comm.kind != Comm::NONE) return fail("communicator already initialised");
    if (comm_common(c, rank, nranks, false)) return -1;
    c->comm.cb_exchange = exchange;
    c->comm.cb_allreduce = allreduce;
    c->comm.cb_user = user;
    c->comm.kind = Comm::CALLBACK;
    return 0;
}

int shk_comm_selftest(shk_ctx* ctx) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIPCHK(hipSetDevice(c->device));
    if (const char* e = rccl_selftest(c)) return fail(std::string("RCCL self-test: ") + e);
    return 0;
}

int shk_comm_set_timing_only(shk_ctx* ctx, int32_t on) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (c->comm.kind == Comm::NONE || c->comm.nranks <= 1) return fail("timing-only transport needs a communicator of several subdomains");
    HIPCHK(hipSetDevice(c->device));
    WAITCHK(c);
    c->comm.timing_only = on != 0;
    return 0;
}

int shk_comm_time_round(shk_ctx* ctx, int32_t kind, int64_t n, int32_t reps, double* us_per_round) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!us_per_round || kind < 0 || kind > 2) return fail("bad arguments");
    HIPCHK(hipSetDevice(c->device));
    const double us = rccl_time_round(c, kind, n, reps);
    if (us < 0.0) return fail("no RCCL communicator on this context (or the timed rounds failed)");
    *us_per_round = us;
    return 0;
}

int shk_comm_allreduce_check(shk_ctx* ctx, double* value) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!value) return fail("null value");
    HIPCHK(hipSetDevice(c->device));
    double* d = c->d_io;   // staging vector, free between calls
    HIPCHK(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(allreduce_buffer(c, d, d, 1));
    HIPCHK(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    WAITCHK(c);
    return 0;
}

int shk_comm_mark_stalled(shk_ctx* ctx) {
    CHECK_CTX(ctx);
    reinterpret_cast<Ctx*>(ctx)->poisoned = true;
    return 0;
}

int64_t shk_env_overrides(char* buf, int64_t cap) {
    const Tunables& T = tunables();
    const std::string o = T.overrides();
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>((size_t)cap - 1, o.size());
        std::memcpy(buf, o.data(), n);
        buf[n] = '\0';
    }
    return (int64_t)T.set.size();
}

int shk_tunable_set(const char* name, const char* value) {
    if (!name) return fail("null switch name");
    if (!tunable_set(name, value)) return fail(std::string("unknown experiment switch: ") + name);
    return 0;
}

int shk_comm_stats(shk_ctx* ctx, int64_t n[6]) {
    CHECK_CTX(ctx);
    if (!n) return fail("null output");
    const Comm& m = reinterpret_cast<Ctx*>(ctx)->comm;
    n[0] = m.n_exchange; n[1] = m.n_allreduce; n[2] = m.bytes_exchange; n[3] = m.bytes_allreduce;
    n[4] = m.n_allgather; n[5] = m.bytes_allgather;
    return 0;
}

int shk_comm_overlap(shk_ctx* ctx, int64_t n[4]) {
    CHECK_CTX(ctx);
    if (!n) return fail("null output");
    const Ctx* c = reinterpret_cast<Ctx*>(ctx);
    n[0] = c->overlap ? 1 : 0; n[1] = c->n_bslices; n[2] = c->plan.A.nslice; n[3] = c->comm.n_overlapped;
    return 0;
}

int shk_halo_update(shk_ctx* ctx, int32_t field) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (field < 0 || field >= SHK_FIELD_COUNT || field == SHK_Q) return fail("field id has no ghost segment");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(halo_exchange(c, c->f[field]));
    WAITCHK(c);
    return 0;
}

int shk_sync(shk_ctx* ctx) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIPCHK(hipSetDevice(c->device));
    WAITCHK(c);
    return 0;
}

int shk_profile_enable(shk_ctx* ctx, int32_t on) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    WAITCHK(c);
    c->profiling = on != 0;
    c->pending_bytes = 0.0;
    return 0;
}

int shk_profile_read(shk_ctx* ctx, shk_profile* out, int32_t reset) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIPCHK(hipSetDevice(c->device));
    WAITCHK(c);
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        if (c->ev_pool[i].phase < 0) continue;   // queued behind a Krylov stop flag (krylov_inner marks them by position)
        if (ms < 0.003f) continue;   // a launch that returned at once behind a solver's stop flag (~1 us)
        c->prof.ms[c->ev_pool[i].phase] += ms;
        c->prof.launches[c->ev_pool[i].phase] += 1;
        c->prof.bytes[c->ev_pool[i].phase] += c->ev_pool[i].bytes;
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
    if (phase != SHK_PH_ASSEMBLE && phase != SHK_PH_SPMV && phase != SHK_PH_OTHER)
        return fail("only ASSEMBLE, SPMV and OTHER (streaming-read calibration) can be timed");
    if (phase == SHK_PH_SPMV && !c->assembled) return fail("no assembled system: call shk_assemble first");
    HIPCHK(hipSetDevice(c->device));
    const bool was = c->profiling;
    c->profiling = false;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    auto once = [&]() {
        if (phase == SHK_PH_ASSEMBLE) launch_assemble(c, dt);
        else if (phase == SHK_PH_SPMV) launch_spmv_plain(c, c->d_vals, c->d_p, c->d_v);
        else launch_stream_read(c);  // k_norm2 over the value array: exactly 8 * slots bytes, 8 B per lane
    };
    once();  // warm
    HIPCHK(hipEventRecord(a, c->stream));
    for (int i = 0; i < reps; ++i) once();
    HIPCHK(hipEventRecord(b, c->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    c->profiling = was;
    *avg_ms = ms / reps;
    return 0;
}

int shk_time_assemble_residual(shk_ctx* ctx, int32_t reps, double dt, double* avg_ms) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (reps < 1 || !avg_ms || !(dt > 0)) return fail("bad arguments");
    HIPCHK(hipSetDevice(c->device));
    const bool was = c->profiling;
    c->profiling = false;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    launch_assemble(c, dt, true);  // warm
    HIPCHK(hipEventRecord(a, c->stream));
    for (int i = 0; i < reps; ++i) launch_assemble(c, dt, true);
    HIPCHK(hipEventRecord(b, c->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    c->profiling = was;
    c->assembled = true;      // d_F is the residual of the current state ...
    c->jac_valid = false;     // ... but d_vals was not touched
    c->assembled_dt = dt;
    *avg_ms = ms / reps;
    return 0;
}

int shk_solver_stats(shk_ctx* ctx, int64_t n[5]) {
    CHECK_CTX(ctx);
    if (!n) return fail("null output");
    const Ctx* c = reinterpret_cast<Ctx*>(ctx);
    n[0] = c->n_asm_full; n[1] = c->n_asm_res; n[2] = c->n_asm_redo; n[3] = c->newton_prev; n[4] = c->n_forced;
    return 0;
}

int shk_plan_stats(shk_ctx* ctx, int64_t n[12]) {
    CHECK_CTX(ctx);
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    n[0] = c->n_own; n[1] = c->ne; n[2] = c->nnz; n[3] = c->nblk; n[4] = c->cells_staged;
    n[5] = c->slots; n[6] = c->device_bytes; n[7] = c->plan.A.max_row_len;
    const AmgHierarchy& H = c->amg ? *c->amg : c->amg_local;
    n[8] = (H.ready() && H.xf[0].with_ap) ? H.ap_nnz0 : 0;
    n[9] = H.ready() ? (int64_t)H.xf.size() + 1 : 0;
    n[10] = H.ready() ? (H.distributed ? H.n_glob : H.xf.back().n_coarse) : 0;
    n[11] = (int64_t)c->asm_lds | ((int64_t)c->plan.verts_max << 32);   // LDS bytes of an assembly workgroup | its vertex stride
    return 0;
}

int shk_storage_stats(shk_ctx* ctx, int64_t n[6]) {
    CHECK_CTX(ctx);
    if (!n) return fail("null output");
    const Ctx* c = reinterpret_cast<Ctx*>(ctx);
    n[0] = c->nnz; n[1] = c->slots; n[2] = c->slots16; n[3] = c->plan.A.nslice;
    const AmgHierarchy& H = c->amg ? *c->amg : c->amg_local;
    int64_t packed = c->d_pk ? 1 : 0, largest_float = c->d_pk ? 0 : c->n_own;
    if (H.ready())
        for (size_t l = 1; l < H.lv.size(); ++l) {
            if (H.lv[l].n <= 0) continue;
            if (H.lv[l].pk) ++packed;
            else largest_float = std::max<int64_t>(largest_float, H.lv[l].n);
        }
    n[4] = packed; n[5] = largest_float;
    return 0;
}

}  // extern "C"
