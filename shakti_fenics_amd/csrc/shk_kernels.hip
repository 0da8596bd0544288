// gfx950 kernels of the SHAKTI hot path.  All arithmetic is fp64 vector (no MFMA: the path is
// sparse and bandwidth-bound).  Wave = 64 lanes, workgroup = 256 threads throughout.
//
//   k_assemble      R1+R2+R3  residual + CSR Jacobian of the weak form at
//                             /root/reference/source/solvers.py:35-45 (closures constitutive.py:6-31),
//                             atomic-free: each workgroup owns a row range, stages the element tensors
//                             of every cell touching those rows in LDS, then writes each CSR value and
//                             residual entry exactly once (deterministic summation order).
//   k_spmv          SELL-64 SpMV (one row per lane, coalesced column-major slices, row sums in
//                   registers) with the BiCGStab dot products fused into the row epilogue.
//   k_bicg_*        fused vector updates of right-Jacobi-preconditioned BiCGStab; scalars are
//                   re-derived in every workgroup from per-workgroup partial sums, so there is no
//                   host sync, no atomics, and results are bitwise reproducible.
//   k_update_a/b    R6+R7 and R8 (+clamp, + N_n <- N): solvers.py:186-197,228.
#include "shk_device.h"

namespace shk {

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// Sum over the 256 threads of the workgroup; every thread returns the same value.
__device__ __forceinline__ double block_sum(double v, double* sh4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect sh4 against a previous use
    if (lane == 0) sh4[w] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

// Fixed-order sum of a partial array (identical in every workgroup and every run).
__device__ __forceinline__ double reduce_partials(const double* __restrict__ P, int n, double* sh4) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) a += P[i];
    return block_sum(a, sh4);
}

// Difference of the head between two vertices from the differences of its coefficients (constitutive.py:6-9)
__device__ __forceinline__ double head_diff(double dzb, double dzs, double dN, const DevParams& p) {
    return dzb + p.ri_rw * (dzs - dzb) - dN / p.rwg;
}

__device__ __forceinline__ double glen_pow(double N, const DevParams& p) {
    // |N|^(n-1), constitutive.py:31; n = 3 (params.py:10) is the square
    return p.n_is_3 ? N * N : pow(fabs(N), p.n - 1.0);
}

// A' = A D^-1 (right Jacobi preconditioning folded into the matrix once per Newton iteration)
__global__ __launch_bounds__(kBlock) void k_scale(int64_t slots, const int32_t* __restrict__ col,
                                                  const double* __restrict__ vals,
                                                  const double* __restrict__ dinv, double* __restrict__ out) {
    for (int64_t s = blockIdx.x * (int64_t)kBlock + threadIdx.x; s < slots; s += (int64_t)gridDim.x * kBlock)
        out[s] = vals[s] * dinv[col[s]];
}

void launch_scale(Ctx* c) {
    PhaseTimer t(c, SHK_PH_OTHER);
    note_bytes(c, 20.0 * (double)c->slots + 8.0 * (double)c->n_own);
    int g = (int)std::min<int64_t>((c->slots + kBlock - 1) / kBlock, 8192);
    hipLaunchKernelGGL(k_scale, dim3(g), dim3(kBlock), 0, c->stream, c->slots, c->d_sell_col, c->d_vals, c->d_dinv,
                       c->d_vals_s);
}

// ------------------------------------------------------------------ SpMV (SELL-64)
// One wavefront per slice, one row per lane; the slice is stored column-major so every load of
// values / column indices is a contiguous 512 B / 256 B wave access and the row sum never leaves
// its register.
//   MODE 0: y = A x
//   MODE 1: BiCGStab first product  v = A p, partial (rhat . v)
//   MODE 2: BiCGStab second product t = A s, partials (t . s), (t . t), (rhat . t)
struct SpmvArgs {
    DevSell A;
    const double* vals;
    const void* x;           // double (Jacobi: p, s themselves) or float (the multigrid cycle's M^-1 p, M^-1 s): template TX
    double* y;
    const double* rhat;
    const double* sdot;      // MODE 2: the vector s of (t . s) (== x unless a preconditioner sits in between)
    double* part;            // partial arrays (MODE 1, 2)
    KrylovState* st;
    SplitSell split;         // PASS 1, 2 only
};

template <int MODE, class TX, int PASS = 0>   // PASS: SplitSell (shk_device.h)
__global__ __launch_bounds__(kBlock) void k_spmv(const SpmvArgs a) {
    __shared__ double sh4[4];
    const int tid = threadIdx.x;
    if (MODE != 0) {
        if (a.st->done) return;
    }
    const int lane = tid & 63;
    double d0 = 0.0, d1 = 0.0, d2 = 0.0;
    auto slice = [&](int s, const SellMeta& m) {
        const int row = min(s * kSlice + lane, a.A.n_rows - 1);   // tail rows of the last slice: clamped, not used
        // the dot-product operands are requested ahead of the slice stream
        const double rh = MODE != 0 ? a.rhat[row] : 0.0;
        const double sd = MODE == 2 ? a.sdot[row] : 0.0;
        const bool skip = PASS == 1 ? a.split.ghost[s] != 0 : false;   // scalar load, needed only after the stream
        const double sum = sell_row_sum<12>(a.A, m, a.vals, reinterpret_cast<const TX*>(a.x), lane);
        if (s * kSlice + lane < a.A.n_rows && !skip) {
            a.y[row] = sum;
            if (MODE == 1) d0 += rh * sum;
            if (MODE == 2) { d0 += sum * sd; d1 += sum * sum; d2 += rh * sum; }
        }
    };
    if (PASS == 2) {
        const int w = wave_index();
        for (int k = 4 * blockIdx.x + w; k < a.split.n_list; k += 4 * gridDim.x) {
            const int s = __builtin_amdgcn_readfirstlane(a.split.list[k]);
            slice(s, sell_meta(a.A, s));
        }
    } else {
        for (SliceLoop it(a.A, wave_index()); it.valid(); it.next()) slice(it.s, it.m);
    }
    if (MODE == 1) {
        d0 = block_sum(d0, sh4);
        if (tid == 0) a.part[P_RHV * kMaxParts + blockIdx.x] = d0;
    }
    if (MODE == 2) {
        d0 = block_sum(d0, sh4);
        d1 = block_sum(d1, sh4);
        d2 = block_sum(d2, sh4);
        if (tid == 0) {
            a.part[P_TS * kMaxParts + blockIdx.x] = d0;
            a.part[P_TT * kMaxParts + blockIdx.x] = d1;
            a.part[P_RHT * kMaxParts + blockIdx.x] = d2;
        }
    }
}

static SpmvArgs spmv_args(Ctx* c, const double* vals, const void* x, double* y) {
    SpmvArgs a;
    a.A = c->sell();
    a.vals = vals; a.x = x; a.y = y; a.rhat = c->d_rhat; a.sdot = nullptr; a.part = c->d_part; a.st = c->d_state;
    a.split = SplitSell{c->d_slice_ghost, c->d_bslices, c->n_bslices};
    return a;
}

// bytes one Krylov product moves: the SELL stream, x (TX), y, and the dot-product operands of its mode
static double spmv_bytes(const Ctx* c, int mode, int x_bytes) {
    return sell_bytes(c->slots, c->slots16, c->plan.A.nslice, 8) + (double)c->n_own * (x_bytes + 8 + (mode == 1 ? 8 : mode == 2 ? 16 : 0));
}
void launch_spmv_plain(Ctx* c, const double* vals, const double* x, double* y) {
    note_bytes(c, spmv_bytes(c, 0, 8));
    launch_phase(c, SHK_PH_SPMV, k_spmv<0, double>, dim3(c->grid), dim3(kBlock), 0, spmv_args(c, vals, x, y));
}

// y_j = A x_j for M vectors in one sweep of the matrix (the warm start's images of the kept solutions: the 0.73 GB of
// values and columns are read once instead of M times).  Same slice bodies as k_spmv: all value / column loads of a
// slice, then the M x width gathers, FMAs in slot order per vector (the operation order of M separate products).
template <int M>
struct SpmmArgs {
    DevSell A;
    const double* vals;
    const double* x[M];
    double* y[M];
};
template <bool NT, int W, int M, class TC>
__device__ __forceinline__ void sell_fixed_m(const double* __restrict__ vp, const TC* __restrict__ cp,
                                             const double* const (&xb)[M], double (&sum)[M]) {
    double v[W];
    TC c[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { v[k] = sell_ld<NT>(vp + k * kSlice); c[k] = sell_ld<NT>(cp + k * kSlice); }
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int k = 0; k < W; ++k) sum[j] += v[k] * xb[j][c[k]];
}
template <bool NT, int M, class TC>
__device__ __forceinline__ void sell_width_m(const double* __restrict__ vp, const TC* __restrict__ cp,
                                             const double* const (&xb)[M], int width, double (&sum)[M]) {
    while (width > 12) {
        sell_fixed_m<NT, 12, M>(vp, cp, xb, sum);
        vp += 12 * kSlice; cp += 12 * kSlice; width -= 12;
    }
    switch (width) {
        case 1: sell_fixed_m<NT, 1, M>(vp, cp, xb, sum); break;
        case 2: sell_fixed_m<NT, 2, M>(vp, cp, xb, sum); break;
        case 3: sell_fixed_m<NT, 3, M>(vp, cp, xb, sum); break;
        case 4: sell_fixed_m<NT, 4, M>(vp, cp, xb, sum); break;
        case 5: sell_fixed_m<NT, 5, M>(vp, cp, xb, sum); break;
        case 6: sell_fixed_m<NT, 6, M>(vp, cp, xb, sum); break;
        case 7: sell_fixed_m<NT, 7, M>(vp, cp, xb, sum); break;
        case 8: sell_fixed_m<NT, 8, M>(vp, cp, xb, sum); break;
        case 9: sell_fixed_m<NT, 9, M>(vp, cp, xb, sum); break;
        case 10: sell_fixed_m<NT, 10, M>(vp, cp, xb, sum); break;
        case 11: sell_fixed_m<NT, 11, M>(vp, cp, xb, sum); break;
        case 12: sell_fixed_m<NT, 12, M>(vp, cp, xb, sum); break;
        default: break;
    }
}
template <int M>
__global__ __launch_bounds__(kBlock) void k_spmm(const SpmmArgs<M> a) {
    const int lane = threadIdx.x & 63;
    for (SliceLoop it(a.A, wave_index()); it.valid(); it.next()) {
        double sum[M];
        const double* xb[M];
#pragma unroll
        for (int j = 0; j < M; ++j) { sum[j] = 0.0; xb[j] = a.x[j] + (it.m.cb >= 0 ? it.m.cb : 0); }
        const double* __restrict__ vp = a.vals + it.m.base + lane;
        if (it.m.cb >= 0) {
            const uint16_t* cp = a.A.col16 + it.m.p16 + lane;
            if (a.A.xcd_local) sell_width_m<false, M>(vp, cp, xb, it.m.width, sum);
            else sell_width_m<true, M>(vp, cp, xb, it.m.width, sum);
        } else {
            const int32_t* cp = a.A.col + it.m.base + lane;
            if (a.A.xcd_local) sell_width_m<false, M>(vp, cp, xb, it.m.width, sum);
            else sell_width_m<true, M>(vp, cp, xb, it.m.width, sum);
        }
        const int row = it.s * kSlice + lane;
        if (row < a.A.n_rows) {
#pragma unroll
            for (int j = 0; j < M; ++j) a.y[j][row] = sum[j];
        }
    }
}
// y[j] = A x[j], j < m <= 4 (m = 3 runs the four-vector kernel with its last vector doubled)
static void launch_spmm(Ctx* c, const double* vals, int m, double* const* x, double* const* y) {
    if (m == 1) { launch_spmv_plain(c, vals, x[0], y[0]); return; }
    note_bytes(c, sell_bytes(c->slots, c->slots16, c->plan.A.nslice, 8) + 16.0 * (double)c->n_own * (m == 2 ? 2 : 4));
    if (m == 2) {
        SpmmArgs<2> a{c->sell(), vals, {x[0], x[1]}, {y[0], y[1]}};
        launch_phase(c, SHK_PH_VECTOR, k_spmm<2>, dim3(c->grid), dim3(kBlock), 0, a);   // (not the Krylov product's leg)
        return;
    }
    SpmmArgs<4> a{c->sell(), vals, {x[0], x[1], x[2], x[m > 3 ? 3 : 2]}, {y[0], y[1], y[2], y[m > 3 ? 3 : 2]}};
    launch_phase(c, SHK_PH_VECTOR, k_spmm<4>, dim3(c->grid), dim3(kBlock), 0, a);
}


// ------------------------------------------------------------------ vector kernels
__global__ __launch_bounds__(kBlock) void k_norm2(int64_t n, const double* __restrict__ x, double* __restrict__ part) {
    __shared__ double sh4[4];
    double a = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        a += x[i] * x[i];
    a = block_sum(a, sh4);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// Calibration of the HBM counters: a pure streaming read of known size (8 B per lane, like the SpMV's
// value stream).
void launch_stream_read(Ctx* c) {
    hipLaunchKernelGGL(k_norm2, dim3(c->grid), dim3(kBlock), 0, c->stream, c->slots, c->d_vals,
                       c->d_part + P_AUX * kMaxParts);
}

void launch_norm2(Ctx* c, const double* x, double* partials) {
    PhaseTimer t(c, SHK_PH_OTHER);
    note_bytes(c, 8.0 * (double)c->n_own);
    hipLaunchKernelGGL(k_norm2, dim3(c->grid), dim3(kBlock), 0, c->stream, c->n_own, x, partials);
}

// Merged-reduction BiCGStab (right preconditioning folded into A' = A D^-1), x0 = 0:
//   init      r = rhat = p = rhs, y = 0, partial ||rhs||^2
//   spmv<1>   v = A' p, (rhat.v)                                         -- reduction point 1: ||r||^2, (rhat.v)
//   k_bicg_s  [stop test on ||r||^2]  alpha = rho / (rhat.v);  s = r - alpha v, (rhat.s)
//   spmv<2>   t = A' s, (t.s), (t.t), (rhat.t)                           -- reduction point 2
//   k_bicg_u  omega = (t.s)/(t.t); rho' = (rhat.r') = (rhat.s) - omega (rhat.t)
//             beta = (rho'/rho)(alpha/omega)
//             y += alpha p + omega s;  r = s - omega t;  p = r + beta (p - omega v);  partial ||r||^2
// Two reduction points per iteration instead of three, four kernels instead of five.
__global__ __launch_bounds__(kBlock) void k_bicg_init(int64_t n, const double* __restrict__ rhs,
                                                      double* __restrict__ r, double* __restrict__ rhat,
                                                      double* __restrict__ p, double* __restrict__ y,
                                                      double* __restrict__ part, KrylovState* __restrict__ st,
                                                      float* __restrict__ p32) {   // p32: the multigrid cycle's input (or null)
    __shared__ double sh4[4];
    double a = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double f = rhs[i];
        r[i] = f; rhat[i] = f; p[i] = f; y[i] = 0.0;
        if (p32) p32[i] = (float)f;
        a += f * f;
    }
    a = block_sum(a, sh4);
    if (threadIdx.x == 0) {
        part[P_RR * kMaxParts + blockIdx.x] = a;
        if (blockIdx.x == 0) {
            st->rho[0] = 0.0; st->rho[1] = 0.0; st->alpha = 0.0; st->omega = 0.0;
            st->target2 = 0.0; st->rnorm2 = 0.0; st->rhs2 = 0.0; st->rr_last = 0.0;
            st->done = 0; st->converged = 0; st->breakdown = 0; st->its = 0;
        }
    }
}

// The stop test of k_bicg_s, one kernel EARLIER (round 3; one context only -- across subdomains ||r||^2 is reduced together
// with (rhat.v), i.e. after the product, and an extra all-reduce per iteration would cost more than it saves).  k_bicg_s
// learns that iterate `it` has converged only after the multigrid cycle and the product that open iteration `it` have
// run: every solve paid one cycle + one product (0.55 ms at 10M rows, two solves per step) for a vector nobody reads.
// One workgroup reduces the same partials in the same order first; behind its flag the cycle's launches return at once.
// It takes the same decision from the same numbers as k_bicg_s would, and writes the same state.
__global__ __launch_bounds__(kBlock) void k_krylov_check(int it, int max_it, double rtol2, double atol2, int np, int rs,
                                                         const double* red, KrylovState* __restrict__ st) {
    __shared__ double sh4[4];
    if (st->done) return;
    const double rr = reduce_partials(red + P_RR * rs, np, sh4);
    const double target2 = (it == 0) ? fmax(rtol2 * rr, atol2) : st->target2;
    int stop = 0, conv = 0;
    if (!(rr > target2)) { stop = 1; conv = (rr <= target2); }
    else if (it >= max_it) stop = 1;
    if (stop && threadIdx.x == 0) {
        if (it == 0) { st->target2 = target2; st->rhs2 = rr; st->rho[0] = rr; }
        st->rr_last = rr;
        st->converged = conv; st->its = it; st->rnorm2 = rr; st->done = 1;
    }
}

__global__ __launch_bounds__(kBlock) void k_bicg_s(int64_t n, int it, int max_it, double rtol2, double atol2,
                                                   int np, int rs, const double* red, double* part,
                                                   const double* __restrict__ r, const double* __restrict__ v,
                                                   const double* __restrict__ rhat, double* __restrict__ s,
                                                   KrylovState* __restrict__ st, float* __restrict__ s32) {
    __shared__ double sh4[4];
    if (st->done) return;
    // ||r||^2 of the iterate that opened this iteration decides whether to go on (the test sits here,
    // one kernel after the first product, so that it shares that product's reduction point)
    // red: slot X starts at red[X * rs]; np entries (one context: its own partial arrays, rs = kMaxParts, np = grid;
    // several subdomains: the all-reduced scalars, rs = 1, np = 1)
    const double rr = reduce_partials(red + P_RR * rs, np, sh4);
    const double rhv = reduce_partials(red + P_RHV * rs, np, sh4);
    const double target2 = (it == 0) ? fmax(rtol2 * rr, atol2) : st->target2;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (lead && it == 0) { st->target2 = target2; st->rhs2 = rr; st->rho[0] = rr; }
    if (lead) st->rr_last = rr;
    int stop = 0, conv = 0;
    if (!(rr > target2)) { stop = 1; conv = (rr <= target2); }  // also stops on NaN
    else if (it >= max_it) stop = 1;
    if (stop) {
        if (lead) { st->converged = conv; st->its = it; st->rnorm2 = rr; st->done = 1; }
        return;
    }
    const double rho = (it == 0) ? rr : st->rho[it & 1];  // rho_0 = (rhat, r_0) = ||rhs||^2
    const double alpha = rho / rhv;
    if (!isfinite(alpha)) {
        if (lead) { st->breakdown = 1; st->converged = 0; st->its = it; st->rnorm2 = rr; st->done = 1; }
        return;
    }
    if (lead) st->alpha = alpha;
    double a = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double si = r[i] - alpha * v[i];
        s[i] = si;
        if (s32) s32[i] = (float)si;
        a += rhat[i] * si;
    }
    a = block_sum(a, sh4);
    if (threadIdx.x == 0) part[P_RHS * kMaxParts + blockIdx.x] = a;
}

// Closes iteration `it` (it >= 0) and prepares p for iteration it+1.
template <class TP>   // double: phat, shat alias p, s (Jacobi); float: the multigrid cycle's outputs
__global__ __launch_bounds__(kBlock) void k_bicg_u(int64_t n, int it, int np, int rs, const double* red, double* part,
                                                   const double* __restrict__ s, const double* __restrict__ t,
                                                   const double* __restrict__ v, double* p, const TP* phat,
                                                   const TP* shat, double* __restrict__ y,
                                                   double* __restrict__ r, KrylovState* __restrict__ st, float* __restrict__ p32) {
    __shared__ double sh4[4];
    if (st->done) return;
    double ts = 0.0, tt = 0.0, rht = 0.0, rhs = 0.0;
    for (int i = threadIdx.x; i < np; i += kBlock) {
        ts += red[P_TS * rs + i];
        tt += red[P_TT * rs + i];
        rht += red[P_RHT * rs + i];
        rhs += red[P_RHS * rs + i];
    }
    ts = block_sum(ts, sh4);
    tt = block_sum(tt, sh4);
    rht = block_sum(rht, sh4);
    rhs = block_sum(rhs, sh4);
    const double alpha = st->alpha, rho = st->rho[it & 1];
    const double omega = (tt > 0.0) ? ts / tt : 0.0;
    // (rhat, r_new) = (rhat, s) - omega (rhat, t); (rhat, s) vanishes only in exact arithmetic and is
    // kept so that rho stays the inner product of the vectors actually stored
    const double rho_new = rhs - omega * rht;
    const double beta = (omega != 0.0) ? (rho_new / rho) * (alpha / omega) : 0.0;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (!isfinite(beta) || !isfinite(omega)) {
        if (lead) { st->breakdown = 1; st->converged = 0; st->its = it; st->done = 1; }
        return;
    }
    if (lead) { st->omega = omega; st->rho[(it + 1) & 1] = rho_new; }
    double a = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double si = s[i], pi = p[i];
        y[i] += alpha * (double)phat[i] + omega * (double)shat[i];  // phat = M^-1 p, shat = M^-1 s (aliases of p, s for Jacobi)
        const double ri = si - omega * t[i];
        r[i] = ri;
        const double pn = ri + beta * (pi - omega * v[i]);
        p[i] = pn;
        if (p32) p32[i] = (float)pn;
        a += ri * ri;
    }
    a = block_sum(a, sh4);
    if (threadIdx.x == 0) part[P_RR * kMaxParts + blockIdx.x] = a;
}

void krylov_init(Ctx* c, const double* rhs) {
    PhaseTimer t(c, SHK_PH_VECTOR);
    note_bytes(c, (c->use_amg ? 44.0 : 40.0) * (double)c->n_own);
    hipLaunchKernelGGL(k_bicg_init, dim3(c->grid), dim3(kBlock), 0, c->stream, c->n_own, rhs, c->d_r, c->d_rhat,
                       c->d_p, c->d_y, c->d_part, c->d_state, c->use_amg ? c->d_p32 : nullptr);
}

// Iterative refinement around BiCGStab (its recursive residual drifts from b - A x over thousands of
// iterations): ytot (+)= y, then rt = F - A' ytot with ||rt||^2 partials.
__global__ __launch_bounds__(kBlock) void k_accumulate(int64_t n, int first, const double* __restrict__ y,
                                                       double* __restrict__ ytot) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        ytot[i] = first ? y[i] : ytot[i] + y[i];
}
__global__ __launch_bounds__(kBlock) void k_true_residual(int64_t n, const double* __restrict__ F,
                                                          const double* __restrict__ Ay, double* __restrict__ rt,
                                                          double* __restrict__ part) {
    __shared__ double sh4[4];
    double a = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double r = F[i] - Ay[i];
        rt[i] = r;
        a += r * r;
    }
    a = block_sum(a, sh4);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
void launch_accumulate(Ctx* c, bool first) {
    PhaseTimer t(c, SHK_PH_VECTOR);
    note_bytes(c, (first ? 16.0 : 24.0) * (double)c->n_own);
    hipLaunchKernelGGL(k_accumulate, dim3(c->grid), dim3(kBlock), 0, c->stream, c->n_own, first ? 1 : 0, c->d_y,
                       c->d_ytot);
}
// Warm start of the linear solve of Newton iteration k of a time step from the solutions g_1 .. g_m of iteration k
// of the previous m <= kWarmDepth steps (Ctx::d_guess, newest first): the combination sum_j c_j g_j that minimises
// ||F - A' sum_j c_j g_j|| becomes the initial iterate, so the guesses can only lower the starting residual.  The normal
// equations (Gram matrix of the images t_j = A' g_j, m + m (m + 1) / 2 dot products in one pass) are solved in every
// workgroup by a Cholesky factorisation that drops a guess whose image is numerically inside the span of the newer
// ones.  Consecutive steps of a slow transient ask for nearly the same Newton updates, and m of them extrapolate.
struct WarmArgs {
    int64_t n;
    int m, np, rs;
    const double* g[Ctx::kWarmDepth];
    const double* t[Ctx::kWarmDepth];
    const double* F;
    double* part;          // kWarmDots partial arrays
    const double* red;     // what k_warm_apply sums: the partial arrays, or the all-reduced scalars
    double *ytot, *rhs;
};
__global__ __launch_bounds__(kBlock) void k_warm_dots(const WarmArgs a) {
    __shared__ double sh4[4];
    constexpr int M = Ctx::kWarmDepth;
    double b[M], G[M * (M + 1) / 2];
#pragma unroll
    for (int j = 0; j < M; ++j) b[j] = 0.0;
#pragma unroll
    for (int j = 0; j < M * (M + 1) / 2; ++j) G[j] = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const double f = a.F[i];
        double t[M];
#pragma unroll
        for (int j = 0; j < M; ++j) t[j] = j < a.m ? a.t[j][i] : 0.0;
#pragma unroll
        for (int j = 0, q = 0; j < M; ++j) {
            b[j] += t[j] * f;
#pragma unroll
            for (int l = 0; l <= j; ++l, ++q) G[q] += t[j] * t[l];
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const double v = block_sum(b[j], sh4);
        if (threadIdx.x == 0) a.part[j * kMaxParts + blockIdx.x] = v;
    }
#pragma unroll
    for (int q = 0; q < M * (M + 1) / 2; ++q) {
        const double v = block_sum(G[q], sh4);
        if (threadIdx.x == 0) a.part[(M + q) * kMaxParts + blockIdx.x] = v;
    }
}
__global__ __launch_bounds__(kBlock) void k_warm_apply(const WarmArgs a) {
    __shared__ double sh4[4];
    constexpr int M = Ctx::kWarmDepth;
    double b[M], G[M][M], c[M];
#pragma unroll
    for (int j = 0; j < M; ++j) b[j] = reduce_partials(a.red + j * a.rs, a.np, sh4);
#pragma unroll
    for (int j = 0, q = 0; j < M; ++j)
#pragma unroll
        for (int l = 0; l <= j; ++l, ++q) G[j][l] = reduce_partials(a.red + (M + q) * a.rs, a.np, sh4);
    // Cholesky G = L L^T in place (lower triangle), forward / backward substitution; a pivot below 1e-8 of its
    // diagonal entry (the image is, to 4 digits, a combination of the newer ones) removes that guess
    bool keep[M];
    bool finite = true;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        double d = G[j][j];
        const double d0 = d;
#pragma unroll
        for (int l = 0; l < j; ++l) if (keep[l]) d -= G[j][l] * G[j][l];
        keep[j] = j < a.m && d0 > 0.0 && d > 1e-8 * d0;
        finite = finite && isfinite(d0);
        if (!keep[j]) continue;
        G[j][j] = sqrt(d);
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            double v = G[i][j];
#pragma unroll
            for (int l = 0; l < j; ++l) if (keep[l]) v -= G[i][l] * G[j][l];
            G[i][j] = v / G[j][j];
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {   // L z = b
        double v = b[j];
#pragma unroll
        for (int l = 0; l < j; ++l) if (keep[l]) v -= G[j][l] * c[l];
        c[j] = keep[j] ? v / G[j][j] : 0.0;
    }
#pragma unroll
    for (int j = M - 1; j >= 0; --j) {   // L^T c = z
        double v = c[j];
#pragma unroll
        for (int l = j + 1; l < M; ++l) if (keep[l]) v -= G[l][j] * c[l];
        c[j] = keep[j] ? v / G[j][j] : 0.0;
        finite = finite && isfinite(c[j]);
    }
    // guesses holding NaN / Inf: start from zero
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        double y = 0.0, r = a.F[i];
        if (finite) {
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (c[j] != 0.0) { y += c[j] * a.g[j][i]; r -= c[j] * a.t[j][i]; }
        }
        a.ytot[i] = y;
        a.rhs[i] = r;
    }
}
hipError_t launch_warm_start(Ctx* c, int k) {   // d_ytot = the projected guess, d_rhs = F - A' d_ytot
    const double* A = c->use_amg ? c->d_vals : c->d_vals_s;
    double* images[Ctx::kWarmDepth] = {c->d_t, c->d_v, c->d_s, c->d_p};   // Krylov vectors, free between solves
    WarmArgs a{};
    a.n = c->n_own; a.m = std::min(c->n_guess[k], (int)c->params.krylov_warm_start); a.F = c->d_F; a.part = c->d_part_w; a.ytot = c->d_ytot; a.rhs = c->d_rhs;
    hipError_t e;
    for (int j = 0; j < a.m; ++j) {
        if ((e = halo_exchange(c, c->d_guess[k][j])) != hipSuccess) return e;
        a.g[j] = c->d_guess[k][j];
        a.t[j] = images[j];
    }
    launch_spmm(c, A, a.m, c->d_guess[k], images);
    {
        PhaseTimer t(c, SHK_PH_VECTOR);
        note_bytes(c, 8.0 * (a.m + 1) * (double)c->n_own);
        hipLaunchKernelGGL(k_warm_dots, dim3(c->grid), dim3(kBlock), 0, c->stream, a);
    }
    // one subdomain: the consumers sum the partial arrays themselves; several: fixed-order local sums, one all-reduce
    a.red = c->d_part_w; a.np = c->grid; a.rs = kMaxParts;
    if (c->comm.kind != Comm::NONE && c->comm.nranks > 1) {
        if ((e = allreduce_part_arrays(c, c->d_part_w, c->d_red_w, Ctx::kWarmDots)) != hipSuccess) return e;
        a.red = c->d_red_w; a.np = 1; a.rs = 1;
    }
    PhaseTimer t(c, SHK_PH_VECTOR);
    note_bytes(c, 8.0 * (2 * a.m + 3) * (double)c->n_own);
    hipLaunchKernelGGL(k_warm_apply, dim3(c->grid), dim3(kBlock), 0, c->stream, a);
    return hipSuccess;
}

// Boundary pass of a split product: the flagged slices, after the ghosts have arrived.  Accounted to the halo
// phase (it is the part of the product that had to wait for the exchange).
template <int MODE, class TX>
static void launch_spmv_boundary(Ctx* c, SpmvArgs a) {
    if (c->n_bslices <= 0) return;
    a.part = c->d_part_b;
    PhaseTimer t(c, SHK_PH_HALO);
    note_bytes(c, 12.0 * 64.0 * 8.0 * (double)c->n_bslices);   // ~8 entries per row of the flagged slices
    hipLaunchKernelGGL((k_spmv<MODE, TX, 2>), dim3(std::min((c->n_bslices + 3) / 4, kMaxParts)), dim3(kBlock), 0, c->stream, a);
}

hipError_t launch_true_residual(Ctx* c) {  // d_rhs = F - A' ytot, partials in P_AUX
    const double* A = c->use_amg ? c->d_vals : c->d_vals_s;
    hipError_t e;
    if (c->overlap) {
        if ((e = hipEventRecord(c->ev_ready, c->stream)) != hipSuccess) return e;
        const SpmvArgs a = spmv_args(c, A, c->d_ytot, c->d_t);
        note_bytes(c, spmv_bytes(c, 0, 8));
        launch_phase(c, SHK_PH_SPMV, k_spmv<0, double, 1>, dim3(c->grid), dim3(kBlock), 0, a);
        if ((e = halo_begin(c, c->d_ytot)) != hipSuccess) return e;
        if ((e = halo_end(c)) != hipSuccess) return e;
        launch_spmv_boundary<0, double>(c, a);
    } else {
        if ((e = halo_exchange(c, c->d_ytot)) != hipSuccess) return e;
        launch_spmv_plain(c, A, c->d_ytot, c->d_t);
    }
    PhaseTimer t(c, SHK_PH_VECTOR);
    note_bytes(c, 24.0 * (double)c->n_own);
    hipLaunchKernelGGL(k_true_residual, dim3(c->grid), dim3(kBlock), 0, c->stream, c->n_own, c->d_F, c->d_t,
                       c->d_rhs, c->d_part + P_AUX * kMaxParts);
    return allreduce_parts(c, P_AUX, 1);
}

// One product of the Krylov loop, v = A' x (MODE 1) or t = A' x (MODE 2), with the ghost update of x it needs.
// Several subdomains: the exchange travels on comm_stream while the slices without ghost columns are swept.
template <int MODE>
static hipError_t krylov_product(Ctx* c, const double* A, const void* x, double* y, const double* sdot) {
    const dim3 g(c->grid), b(kBlock);
    const bool amg = c->use_amg;
    SpmvArgs a = spmv_args(c, A, x, y);
    a.sdot = sdot;
    hipError_t e;
    if (c->overlap) {
        if ((e = hipEventRecord(c->ev_ready, c->stream)) != hipSuccess) return e;
        note_bytes(c, spmv_bytes(c, MODE, amg ? 4 : 8));
        if (amg) launch_phase(c, SHK_PH_SPMV, k_spmv<MODE, float, 1>, g, b, 0, a);
        else launch_phase(c, SHK_PH_SPMV, k_spmv<MODE, double, 1>, g, b, 0, a);
        if ((e = amg ? halo_begin_f32(c, (float*)const_cast<void*>(x)) : halo_begin(c, (double*)const_cast<void*>(x))) != hipSuccess) return e;
        if ((e = halo_end(c)) != hipSuccess) return e;
        if (amg) launch_spmv_boundary<MODE, float>(c, a);
        else launch_spmv_boundary<MODE, double>(c, a);
        return hipSuccess;
    }
    if (amg) {
        if (!c->comm.plans.empty() && (e = halo_exchange_plan_f32(c, c->comm.plans[0], (float*)const_cast<void*>(x))) != hipSuccess) return e;
        note_bytes(c, spmv_bytes(c, MODE, 4));
        launch_phase(c, SHK_PH_SPMV, k_spmv<MODE, float>, g, b, 0, a);
    } else {
        if ((e = halo_exchange(c, (double*)const_cast<void*>(x))) != hipSuccess) return e;
        note_bytes(c, spmv_bytes(c, MODE, 8));
        launch_phase(c, SHK_PH_SPMV, k_spmv<MODE, double>, g, b, 0, a);
    }
    return hipSuccess;
}

hipError_t krylov_iteration(Ctx* c, int it) {
    const dim3 g(c->grid), b(kBlock);
    double* part = c->d_part;
    hipError_t e;
    // right preconditioner: Jacobi is folded into the matrix (A' = A D^-1, phat = p); multigrid is applied and
    // leaves M^-1 p, M^-1 s in float (the product and the solution update read them as such)
    const double* A = c->use_amg ? c->d_vals : c->d_vals_s;
    const bool amg = c->use_amg;
    if (amg && (c->comm.kind == Comm::NONE || c->comm.nranks <= 1) && tunables().krylov_early_check) {
        PhaseTimer t(c, SHK_PH_VECTOR);
        hipLaunchKernelGGL(k_krylov_check, dim3(1), b, 0, c->stream, it, c->params.krylov_max_it, c->cur_rtol2, c->cur_atol2,
                           c->np, c->red_stride, c->d_red, c->d_state);
    }
    if (amg && (e = amg_vcycle(c, *c->amg, (const float*)c->d_p32, c->d_phat)) != hipSuccess) return e;
    if ((e = krylov_product<1>(c, A, amg ? (const void*)c->d_phat : (const void*)c->d_p, c->d_v, nullptr)) != hipSuccess) return e;
    if ((e = allreduce_parts(c, P_RR, 2)) != hipSuccess) return e;
    {
        PhaseTimer t(c, SHK_PH_VECTOR);
        note_bytes(c, (amg ? 36.0 : 32.0) * (double)c->n_own);
        hipLaunchKernelGGL(k_bicg_s, g, b, 0, c->stream, c->n_own, it, c->params.krylov_max_it, c->cur_rtol2,
                           c->cur_atol2, c->np, c->red_stride, c->d_red, part, c->d_r, c->d_v, c->d_rhat, c->d_s, c->d_state,
                           amg ? c->d_s32 : nullptr);
    }
    if (amg && (e = amg_vcycle(c, *c->amg, (const float*)c->d_s32, c->d_shat, tunables().amg_warm_s ? c->d_s : nullptr)) != hipSuccess) return e;
    if ((e = krylov_product<2>(c, A, amg ? (const void*)c->d_shat : (const void*)c->d_s, c->d_t, c->d_s)) != hipSuccess) return e;
    if ((e = allreduce_parts(c, P_TS, 4)) != hipSuccess) return e;
    {
        PhaseTimer t(c, SHK_PH_VECTOR);
        note_bytes(c, (amg ? 76.0 : 64.0) * (double)c->n_own);
        if (amg)
            hipLaunchKernelGGL(k_bicg_u<float>, g, b, 0, c->stream, c->n_own, it, c->np, c->red_stride, c->d_red, part, c->d_s, c->d_t, c->d_v,
                               c->d_p, (const float*)c->d_phat, (const float*)c->d_shat, c->d_y, c->d_r, c->d_state, c->d_p32);
        else
            hipLaunchKernelGGL(k_bicg_u<double>, g, b, 0, c->stream, c->n_own, it, c->np, c->red_stride, c->d_red, part, c->d_s, c->d_t, c->d_v,
                               c->d_p, (const double*)c->d_p, (const double*)c->d_s, c->d_y, c->d_r, c->d_state, (float*)nullptr);
    }
    return hipSuccess;
}

// dx = D^-1 y (undo the right preconditioning); N <- N - relax dx   (NewtonSolver update)
__global__ __launch_bounds__(kBlock) void k_newton_update(int64_t n, double relax, int apply, int unscale,
                                                          const double* __restrict__ y,
                                                          const double* __restrict__ dinv, double* __restrict__ dx,
                                                          double* __restrict__ N, double* __restrict__ keep) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double yi = y[i];
        const double d = unscale ? yi * dinv[i] : yi;
        dx[i] = d;
        if (apply) N[i] -= relax * d;
        if (keep) keep[i] = yi;   // the solution in the solver's own scaling: a starting guess of the next steps' solves
    }
}

void launch_newton_update(Ctx* c, bool apply) {
    PhaseTimer t(c, SHK_PH_OTHER);
    double* keep = c->pending_keep;
    c->pending_keep = nullptr;
    note_bytes(c, (apply ? 32.0 : 16.0) * (double)c->n_own + (c->use_amg ? 0.0 : 8.0 * (double)c->n_own) + (keep ? 8.0 * (double)c->n_own : 0.0));
    hipLaunchKernelGGL(k_newton_update, dim3(c->grid), dim3(kBlock), 0, c->stream, c->n_own, c->params.newton_relax,
                       apply ? 1 : 0, c->use_amg ? 0 : 1, c->d_ytot, c->d_dinv, c->f[SHK_DX], c->f[SHK_N], keep);
}

// ------------------------------------------------------------------ explicit updates (R6-R8)
struct UpdArgs {
    Mesh m;
    const int32_t* lastcell;
    const double *N, *z_b, *z_s, *G;
    double *b, *qx, *qy, *melt_n, *N_n;
    double *melt_tmp, *b_tmp, *m0;
    double dt;
    int64_t nv;
    DevParams p;
};

struct CellGeom { int v0, v1, v2; double g0x, g0y, g1x, g1y, g2x, g2y; };

__device__ __forceinline__ CellGeom cell_geom(const Mesh& m, int c) {
    CellGeom g;
    g.v0 = m.cells[3 * (size_t)c]; g.v1 = m.cells[3 * (size_t)c + 1]; g.v2 = m.cells[3 * (size_t)c + 2];
    const double2 p0 = m.xy[g.v0], p1 = m.xy[g.v1], p2 = m.xy[g.v2];
    const double d1x = p1.x - p0.x, d1y = p1.y - p0.y, d2x = p2.x - p0.x, d2y = p2.y - p0.y;
    const double inv = 1.0 / (d1x * d2y - d1y * d2x);
    g.g1x = d2y * inv; g.g1y = -d2x * inv; g.g2x = -d1y * inv; g.g2y = d1x * inv;
    g.g0x = -(g.g1x + g.g2x); g.g0y = -(g.g1y + g.g2y);
    return g;
}

// q <- WaterFlux(b, Head(N), Reynolds(q_old)) and melt_n <- Melt(q_new, ., ., b, melt_old) at each vertex,
// gradients taken on T*(v) = highest cell containing v (solvers.py:186,189).
__global__ __launch_bounds__(kBlock) void k_update_a(const UpdArgs a) {
    const DevParams& p = a.p;
    for (int64_t v = blockIdx.x * (int64_t)kBlock + threadIdx.x; v < a.nv; v += (int64_t)gridDim.x * kBlock) {
        const CellGeom g = cell_geom(a.m, a.lastcell[v]);
        const double zb0 = a.z_b[g.v0], zs0 = a.z_s[g.v0], N0 = a.N[g.v0];
        const double dh1 = head_diff(a.z_b[g.v1] - zb0, a.z_s[g.v1] - zs0, a.N[g.v1] - N0, p);
        const double dh2 = head_diff(a.z_b[g.v2] - zb0, a.z_s[g.v2] - zs0, a.N[g.v2] - N0, p);
        const double ghx = dh1 * g.g1x + dh2 * g.g2x, ghy = dh1 * g.g1y + dh2 * g.g2y;
        const double b0 = a.b[g.v0], b1 = a.b[g.v1], b2 = a.b[g.v2];
        const double gbx = (b1 - b0) * g.g1x + (b2 - b0) * g.g2x, gby = (b1 - b0) * g.g1y + (b2 - b0) * g.g2y;
        const double m0_ = a.melt_n[g.v0], m1_ = a.melt_n[g.v1], m2_ = a.melt_n[g.v2];
        const double gmx = (m1_ - m0_) * g.g1x + (m2_ - m0_) * g.g2x, gmy = (m1_ - m0_) * g.g1y + (m2_ - m0_) * g.g2y;
        const double bv = a.b[v], mv = a.melt_n[v];
        const double qxo = a.qx[v], qyo = a.qy[v];
        const double qn = sqrt(qxo * qxo + qyo * qyo);
        const double ab = fabs(bv);
        const double K = ab * ab * ab * p.kcoef / (1.0 + p.om_nu * qn);
        const double qxn = -K * ghx, qyn = -K * ghy;
        const double melt0 = (a.G[v] - p.rwg * (qxn * ghx + qyn * ghy)) / p.Lh;
        const double gb2 = gbx * gbx + gby * gby;
        const double meltn = melt0 + (mv * gb2 + bv * (gmx * gbx + gmy * gby)) / (1.0 + gb2);
        a.qx[v] = qxn; a.qy[v] = qyn;   // depends on q_old of this vertex only: in place is safe
        a.melt_tmp[v] = meltn;          // neighbours still read the old melt_n
        a.m0[v] = melt0;
    }
}

// b <- max(b + dt (Melt(q_new, h, G, b, melt_new)/rho_i - Closure(b, N)), b_min); N_n <- N
// (solvers.py:162,192,196,228)
__global__ __launch_bounds__(kBlock) void k_update_b(const UpdArgs a) {
    const DevParams& p = a.p;
    for (int64_t v = blockIdx.x * (int64_t)kBlock + threadIdx.x; v < a.nv; v += (int64_t)gridDim.x * kBlock) {
        const CellGeom g = cell_geom(a.m, a.lastcell[v]);
        const double b0 = a.b[g.v0], b1 = a.b[g.v1], b2 = a.b[g.v2];
        const double gbx = (b1 - b0) * g.g1x + (b2 - b0) * g.g2x, gby = (b1 - b0) * g.g1y + (b2 - b0) * g.g2y;
        const double m0_ = a.melt_tmp[g.v0], m1_ = a.melt_tmp[g.v1], m2_ = a.melt_tmp[g.v2];
        const double gmx = (m1_ - m0_) * g.g1x + (m2_ - m0_) * g.g2x, gmy = (m1_ - m0_) * g.g1y + (m2_ - m0_) * g.g2y;
        const double bv = a.b[v], mv = a.melt_tmp[v], Nv = a.N[v];
        const double gb2 = gbx * gbx + gby * gby;
        const double melt = a.m0[v] + (mv * gb2 + bv * (gmx * gbx + gmy * gby)) / (1.0 + gb2);
        const double closure = p.A * bv * Nv * glen_pow(Nv, p);
        double bn = bv + a.dt * (melt / p.rho_i - closure);
        bn = (bn < p.b_min) ? p.b_min : bn;
        a.b_tmp[v] = bn;
        a.N_n[v] = Nv;
    }
}

hipError_t launch_update_explicit(Ctx* c, double dt) {
    UpdArgs a;
    a.m.xy = c->d_xy; a.m.cells = c->d_cells;
    a.lastcell = c->d_lastcell;
    a.N = c->f[SHK_N]; a.z_b = c->f[SHK_Z_B]; a.z_s = c->f[SHK_Z_S]; a.G = c->f[SHK_G];
    a.b = c->f[SHK_B]; a.qx = c->f[SHK_QX]; a.qy = c->f[SHK_QY]; a.melt_n = c->f[SHK_MELT_N];
    a.N_n = c->f[SHK_N_N];
    a.melt_tmp = c->d_melt_tmp; a.b_tmp = c->d_b_tmp; a.m0 = c->d_m0;
    a.dt = dt; a.nv = c->n_own; a.p = c->dp;
    int g = (int)std::min<int64_t>((c->n_own + kBlock - 1) / kBlock, 4096);
    {
        PhaseTimer t(c, SHK_PH_UPDATE);
        // per vertex: last cell + its three ids, coordinates, z_b z_s N b melt_n G qx qy in; qx qy melt m0 out
        note_bytes(c, (4.0 + 12.0 + 16.0 + 64.0 + 32.0) * (double)c->n_own);
        hipLaunchKernelGGL(k_update_a, dim3(g), dim3(kBlock), 0, c->stream, a);
    }
    hipError_t e = halo_exchange(c, c->d_melt_tmp);  // neighbours' new melt rate enters grad(melt_n) below
    if (e != hipSuccess) return e;
    {
        PhaseTimer t(c, SHK_PH_UPDATE);
        note_bytes(c, (4.0 + 12.0 + 16.0 + 32.0 + 16.0) * (double)c->n_own);
        hipLaunchKernelGGL(k_update_b, dim3(g), dim3(kBlock), 0, c->stream, a);
    }
    std::swap(c->f[SHK_MELT_N], c->d_melt_tmp);
    std::swap(c->f[SHK_B], c->d_b_tmp);
    if (c->n_loc > c->n_own) {  // ghost copies of the updated state (scatter_forward, solvers.py:197,229)
        if ((e = halo_exchange(c, c->f[SHK_B])) != hipSuccess) return e;
        if ((e = halo_exchange(c, c->f[SHK_QX])) != hipSuccess) return e;
        if ((e = halo_exchange(c, c->f[SHK_QY])) != hipSuccess) return e;
        e = hipMemcpyAsync(c->f[SHK_N_N] + c->n_own, c->f[SHK_N] + c->n_own,
                           (size_t)(c->n_loc - c->n_own) * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    }
    return e;
}

// ------------------------------------------------------------------ field I/O in the caller's numbering
__global__ void k_permute_in(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ io,
                             double* __restrict__ dst) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        dst[i] = io[perm[i]];
}
__global__ void k_permute_out(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src,
                              double* __restrict__ io) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        io[perm[i]] = src[i];
}
__global__ void k_split_q(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ q,
                          double* __restrict__ qx, double* __restrict__ qy) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t e = perm[i];
        qx[i] = q[2 * e]; qy[i] = q[2 * e + 1];
    }
}
__global__ void k_join_q(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ qx,
                         const double* __restrict__ qy, double* __restrict__ q) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t e = perm[i];
        q[2 * e] = qx[i]; q[2 * e + 1] = qy[i];
    }
}
static int io_grid(Ctx* c) { return (int)std::min<int64_t>((c->n_loc + kBlock - 1) / kBlock, 4096); }
void launch_permute_in(Ctx* c, const double* io, double* dst) {
    hipLaunchKernelGGL(k_permute_in, dim3(io_grid(c)), dim3(kBlock), 0, c->stream, c->n_loc, c->d_perm, io, dst);
}
void launch_permute_out(Ctx* c, const double* src, double* io) {
    hipLaunchKernelGGL(k_permute_out, dim3(io_grid(c)), dim3(kBlock), 0, c->stream, c->n_loc, c->d_perm, src, io);
}
void launch_split_q(Ctx* c, const double* io) {
    hipLaunchKernelGGL(k_split_q, dim3(io_grid(c)), dim3(kBlock), 0, c->stream, c->n_loc, c->d_perm, io,
                       c->f[SHK_QX], c->f[SHK_QY]);
}
void launch_join_q(Ctx* c, double* io) {
    hipLaunchKernelGGL(k_join_q, dim3(io_grid(c)), dim3(kBlock), 0, c->stream, c->n_loc, c->d_perm, c->f[SHK_QX],
                       c->f[SHK_QY], io);
}

// ------------------------------------------------------------------ profiling
int profile_slot(Ctx* c, int phase) {
    if (c->ev_used == c->ev_pool.size()) {
        Ctx::Ev e;
        // no system-scope fence at the events: a default event flushes / invalidates the caches around every
        // timed launch, which showed as ~15 us per launch against the rocprofv3 trace of the same kernels
        (void)hipEventCreateWithFlags(&e.a, hipEventDisableSystemFence);
        (void)hipEventCreateWithFlags(&e.b, hipEventDisableSystemFence);
        c->ev_pool.push_back(e);
    }
    const int idx = (int)c->ev_used++;
    c->ev_pool[idx].phase = phase;
    c->ev_pool[idx].bytes = c->pending_bytes;   // noted by the launch site (launch_phase); a PhaseTimer adds at its end
    c->pending_bytes = 0.0;
    return idx;
}
PhaseTimer::PhaseTimer(Ctx* c_, int phase) : c(c_) {
    if (!c->profiling) return;
    idx = profile_slot(c, phase);
    (void)hipEventRecord(c->ev_pool[idx].a, c->stream);
}
PhaseTimer::~PhaseTimer() {
    if (idx < 0) return;
    (void)hipEventRecord(c->ev_pool[idx].b, c->stream);
    c->ev_pool[idx].bytes += c->pending_bytes;   // what the launches inside the scope noted
    c->pending_bytes = 0.0;
}

}  // namespace shk
