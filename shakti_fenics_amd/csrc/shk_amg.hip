// Static-pattern aggregation multigrid used as a right preconditioner of BiCGStab (SURVEY.md 8f rank 1:
// "stronger preconditioning behind the same Newton / C ABI").
//
// Hierarchy (built once on the host, shk_plan.cpp): aggregates are runs of 4 consecutive vertices of the
// k-d order, prolongation is piecewise constant, coarse operators are Galerkin products whose sparsity is
// fixed, so refreshing them after each assembly is one gather-sum kernel per level.  The coarsest level
// (<= 64 rows) is inverted densely in LDS.  One V(0,2) cycle with damped Jacobi (omega = 0.7) -- on this
// operator it needs as many BiCGStab iterations as V(1,1) and its smoothing kernels are plain SpMVs:
//   down  k_amg_restrict   r_c = P^T r                       (no smoothing on the way down)
//   up    k_amg_prolong    x = P e_c ;  k_amg_post x' = x + w D^-1 (r - A x), twice
// All levels use SELL-64; an aligned group of 4 slices = 256 rows contains, by construction, all 4 members
// of each of its 64 aggregates.
#include "shk_device.h"

namespace shk {

constexpr double kAmgOmega = 0.7;

__global__ __launch_bounds__(kBlock) void k_galerkin(int64_t nslots, const int32_t* __restrict__ gptr,
                                                     const int32_t* __restrict__ glist,
                                                     const double* __restrict__ fine, double* __restrict__ coarse) {
    for (int64_t s = blockIdx.x * (int64_t)kBlock + threadIdx.x; s < nslots; s += (int64_t)gridDim.x * kBlock) {
        double a = 0.0;
        for (int32_t k = gptr[s]; k < gptr[s + 1]; ++k) a += fine[glist[k]];  // ascending fine slot: fixed order
        coarse[s] = a;
    }
}

__global__ __launch_bounds__(kBlock) void k_diag_inv(int32_t n, const int32_t* __restrict__ diag_slot,
                                                     const double* __restrict__ vals, double* __restrict__ dinv) {
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double d = vals[diag_slot[i]];
        dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
    }
}

// In-place Gauss-Jordan inverse of the dense coarsest operator (n <= 64) in LDS, one workgroup.  No
// pivoting: the operator is a Galerkin projection of a (negated) M-matrix-like Jacobian.
__global__ __launch_bounds__(kBlock) void k_dense_invert(int n, const double* __restrict__ A, double* __restrict__ inv) {
    __shared__ double M[64 * 64];
    __shared__ double mult[64];
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += kBlock) M[(e / n) * 64 + (e % n)] = A[e];
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        const double piv = M[p * 64 + p];
        const double d = (piv != 0.0) ? 1.0 / piv : 0.0;
        __syncthreads();
        if (tid == 0) M[p * 64 + p] = 1.0;
        __syncthreads();
        if (tid < n) M[p * 64 + tid] *= d;
        if (tid >= 64 && tid < 64 + n) mult[tid - 64] = (tid - 64 != p) ? M[(tid - 64) * 64 + p] : 0.0;
        __syncthreads();
        if (tid < n && tid != p) M[tid * 64 + p] = 0.0;
        __syncthreads();
        for (int e = tid; e < n * n; e += kBlock) {
            const int i = e / n, j = e % n;
            if (i != p) M[i * 64 + j] -= mult[i] * M[p * 64 + j];
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += kBlock) inv[e] = M[(e / n) * 64 + (e % n)];
}

__global__ __launch_bounds__(64) void k_dense_apply(int n, const double* __restrict__ inv,
                                                    const double* __restrict__ r, double* __restrict__ x,
                                                    const int* __restrict__ done) {
    if (*done) return;
    const int i = threadIdx.x;
    if (i < n) {
        double a = 0.0;
        for (int j = 0; j < n; ++j) a += inv[i * n + j] * r[j];
        x[i] = a;
    }
}

struct AmgSmoothArgs {
    DevSell A;
    const double* vals;
    const double* dinv;
    const double* r;        // right-hand side of this level
    const double* x;        // k_amg_post: current iterate
    double* xo;             // down: w D^-1 r ; post: smoothed iterate
    double* rc;             // down: restricted residual (coarse rhs)
    const int32_t* members; // down: 4 fine rows per aggregate
    int32_t n_coarse;
    double omega;
    const int* done;        // Krylov stop flag: once set, every later kernel of the queue returns at once
};

__global__ __launch_bounds__(kBlock) void k_amg_restrict(int32_t n_coarse, const int32_t* __restrict__ members,
                                                         const double* __restrict__ r, double* __restrict__ rc,
                                                         const int* __restrict__ done) {
    if (*done) return;
    for (int32_t I = blockIdx.x * kBlock + threadIdx.x; I < n_coarse; I += gridDim.x * kBlock) {
        const int4 m = reinterpret_cast<const int4*>(members)[I];
        double acc = r[m.x];                      // every aggregate has at least one member
        if (m.y >= 0) acc += r[m.y];
        if (m.z >= 0) acc += r[m.z];
        if (m.w >= 0) acc += r[m.w];
        rc[I] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void k_amg_prolong(int32_t n, const int32_t* __restrict__ agg,
                                                        const double* __restrict__ ec, double* __restrict__ x,
                                                        const int* __restrict__ done) {
    if (*done) return;
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] = ec[agg[i]];
}

// FINE only gives the finest level its own symbol, so that profilers report its launches separately
template <bool FINE>
__global__ __launch_bounds__(kBlock) void k_amg_post(const AmgSmoothArgs a) {
    if (*a.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GroupSweep sw = xcd_sweep((a.A.nslice + 3) >> 2, a.A.xcd_local);
    for (int g = sw.begin; g < sw.end; g += sw.step) {
        const int s = 4 * g + wave;
        if (s >= a.A.nslice) break;
        const int base = __builtin_amdgcn_readfirstlane(a.A.ptr[s]);
        const int width = (__builtin_amdgcn_readfirstlane(a.A.ptr[s + 1]) - base) >> 6;
        const double* __restrict__ vp = a.vals + base + lane;
        const int32_t* __restrict__ cp = a.A.col + base + lane;
        double sum = 0.0;
        if (a.A.xcd_local) {
#pragma unroll 4
            for (int k = 0; k < width; ++k) sum += vp[k * kSlice] * a.x[cp[k * kSlice]];
        } else {  // matrix larger than the Infinity Cache: stream it non-temporally, keep x cached
#pragma unroll 4
            for (int k = 0; k < width; ++k)
                sum += __builtin_nontemporal_load(vp + k * kSlice) * a.x[__builtin_nontemporal_load(cp + k * kSlice)];
        }
        const int row = s * kSlice + lane;
        if (row < a.A.n_rows) a.xo[row] = a.x[row] + a.omega * a.dinv[row] * (a.r[row] - sum);
    }
}

static int small_grid(int64_t n) { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }

// Refresh the coarse operators from the Jacobian just assembled (d_vals, d_dinv).
void amg_numeric_setup(Ctx* c) {
    PhaseTimer t(c, SHK_PH_OTHER);
    const double* fine = c->d_vals;
    for (size_t l = 0; l < c->amg_xf.size(); ++l) {
        const AmgXfer& X = c->amg_xf[l];
        if (X.dense) {
            const int64_t ns = (int64_t)X.n_coarse * X.n_coarse;
            hipLaunchKernelGGL(k_galerkin, dim3(small_grid(ns)), dim3(kBlock), 0, c->stream, ns, X.gptr, X.glist, fine,
                               c->d_cdense);
            hipLaunchKernelGGL(k_dense_invert, dim3(1), dim3(kBlock), 0, c->stream, X.n_coarse, c->d_cdense, c->d_cinv);
        } else {
            AmgLevel& L = c->amg_lv[l + 1];
            hipLaunchKernelGGL(k_galerkin, dim3(small_grid(L.slots)), dim3(kBlock), 0, c->stream, L.slots, X.gptr,
                               X.glist, fine, L.vals);
            hipLaunchKernelGGL(k_diag_inv, dim3(small_grid(L.n)), dim3(kBlock), 0, c->stream, L.n, L.diag_slot, L.vals,
                               L.dinv);
            fine = L.vals;
        }
    }
}

static DevSell level_sell(const Ctx* c, size_t l) {
    if (l == 0) return c->sell();
    const AmgLevel& L = c->amg_lv[l];
    return DevSell{L.n, L.n, L.nslice, sell_fits_cache(L.slots), L.ptr, L.col, L.rowlen};
}

// z = M^-1 r : one V(0,2) cycle.  r and z have the fine level's length; r is not modified.
void amg_vcycle(Ctx* c, const double* rin, double* zout) {
    const size_t nx = c->amg_xf.size();  // levels 0..nx-1 are sparse, level nx is the dense coarsest
    const int* done = &c->d_state->done;
    auto vals = [&](size_t l) { return l == 0 ? c->d_vals : c->amg_lv[l].vals; };
    auto dinv = [&](size_t l) { return l == 0 ? c->d_dinv : c->amg_lv[l].dinv; };
    auto rhs = [&](size_t l) -> const double* { return l == 0 ? rin : c->amg_lv[l].r; };
    auto bufA = [&](size_t l) { return l == 0 ? zout : c->amg_lv[l].x2; };       // where the level's result lands
    auto bufB = [&](size_t l) { return l == 0 ? c->d_amg_x0 : c->amg_lv[l].x; };
    {
        PhaseTimer t(c, SHK_PH_AMG_COARSE);
        for (size_t l = 0; l < nx; ++l) {
            const AmgXfer& X = c->amg_xf[l];
            double* rc = X.dense ? c->d_cr : c->amg_lv[l + 1].r;
            hipLaunchKernelGGL(k_amg_restrict, dim3(small_grid(X.n_coarse)), dim3(kBlock), 0, c->stream, X.n_coarse,
                               X.members, rhs(l), rc, done);
        }
        hipLaunchKernelGGL(k_dense_apply, dim3(1), dim3(64), 0, c->stream, c->amg_xf[nx - 1].n_coarse, c->d_cinv,
                           c->d_cr, c->d_cx, done);
    }
    for (size_t l = nx; l-- > 0;) {
        const AmgXfer& X = c->amg_xf[l];
        const double* ec = X.dense ? c->d_cx : c->amg_lv[l + 1].x2;
        {
            PhaseTimer t(c, SHK_PH_AMG_COARSE);
            hipLaunchKernelGGL(k_amg_prolong, dim3(small_grid(X.n_fine)), dim3(kBlock), 0, c->stream, X.n_fine, X.agg,
                               ec, bufA(l), done);
        }
        AmgSmoothArgs a;
        a.A = level_sell(c, l);
        a.vals = vals(l); a.dinv = dinv(l); a.r = rhs(l);
        a.rc = nullptr; a.members = nullptr; a.n_coarse = 0; a.omega = kAmgOmega; a.done = done;
        const dim3 g(std::min((a.A.nslice + 3) / 4, 2048));
        for (int sweep = 0; sweep < 2; ++sweep) {
            a.x = sweep == 0 ? bufA(l) : bufB(l);
            a.xo = sweep == 0 ? bufB(l) : bufA(l);
            PhaseTimer t(c, l == 0 ? SHK_PH_AMG_FINE : SHK_PH_AMG_COARSE);
            if (l == 0) hipLaunchKernelGGL(k_amg_post<true>, g, dim3(kBlock), 0, c->stream, a);
            else hipLaunchKernelGGL(k_amg_post<false>, g, dim3(kBlock), 0, c->stream, a);
        }
    }
}

}  // namespace shk
