// Static-pattern aggregation multigrid used as a right preconditioner of BiCGStab (SURVEY.md 8f rank 1:
// "stronger preconditioning behind the same Newton / C ABI").
//
// Hierarchy (built once on the host, shk_plan.cpp): aggregates are runs of 4 consecutive vertices of the
// k-d order, prolongation is piecewise constant, coarse operators are Galerkin products whose sparsity is
// fixed, so refreshing them after each assembly is one gather-sum kernel per level.  The hierarchy stops early
// on a dense coarsest level (<= 4096 rows; n^3 <= 250 nnz) whose inverse is kept and applied by a GEMV.
// One V(0, .) cycle with damped Jacobi, dampings scaled by a power-iteration estimate of lambda_max(D^-1 A):
//   down  k_amg_restrict(4)  r_c = P^T r                     (no smoothing on the way down)
//   up    k_amg_first        x1 = a P e + w D^-1 (r - a (A P) e)   on the precomputed A*P operator
//         k_amg_post         x' = x + w D^-1 (r - A x):  once more on the finest level, three more times on
//                            every coarser one (four Chebyshev-damped sweeps there: they are cheap)
//   levels <= 4096 rows run inside one workgroup (k_amg_tail)
// All levels use SELL-64; an aligned group of 4 slices = 256 rows contains, by construction, all 4 members
// of each of its 64 aggregates.
//
// Precision: the cycle is a preconditioner, so its operators (a float copy of the Jacobian, the Galerkin
// products, A*P) and its vectors are stored in float -- the smoothing sweeps are HBM-bound SpMVs and move
// half the bytes.  BiCGStab's own vectors, its operator, the true-residual refinement and the dense coarsest
// solve stay in double; the recurrence uses p^ = M^-1 p exactly as computed, so rounding inside M^-1 changes
// (marginally) the iteration count, never the solution the stopping test certifies.
#include <climits>
#include <cmath>
#include <cstdio>

#include "shk_device.h"

namespace shk {


// coarse[s] = sum of the finer values listed for slot s (ascending fine slot: fixed order), accumulated in double.
// ILP slots per thread at a time, kBlock apart (their list bounds first, then the gather chains side by side): the
// kernel is a chain of three dependent loads per entry.  Which ILP / grid is fastest differs between the operators
// (measured at 10M rows, profiles/r02_galerkin_variants.md): see launch_galerkin.
template <class TC, int ILP>
__global__ __launch_bounds__(kBlock) void k_galerkin(int64_t nslots, const int32_t* __restrict__ gptr,
                                                     const int32_t* __restrict__ glist,
                                                     const float* __restrict__ fine, TC* __restrict__ coarse, int contiguous) {
    constexpr int64_t kChunk = (int64_t)ILP * kBlock;
    // contiguous: a workgroup walks ONE run of consecutive slots (the k-columns of a few coarse slices, whose lists share
    // fine cache lines) instead of striding through the whole array
    const int64_t nchunk = (nslots + kChunk - 1) / kChunk, per = (nchunk + gridDim.x - 1) / gridDim.x;
    const int64_t first = contiguous ? blockIdx.x * per * kChunk : blockIdx.x * kChunk;
    const int64_t last = contiguous ? min(nslots, (blockIdx.x + 1) * per * kChunk) : nslots;
    const int64_t stride = contiguous ? kChunk : (int64_t)gridDim.x * kChunk;
    for (int64_t c0 = first; c0 < last; c0 += stride) {
        int32_t b[ILP], e[ILP];
        double a[ILP];
        int len = 0;
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            const int64_t s = c0 + threadIdx.x + (int64_t)j * kBlock;
            b[j] = s < nslots ? gptr[s] : 0;
            e[j] = s < nslots ? gptr[s + 1] : 0;
            a[j] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < ILP; ++j) len = max(len, e[j] - b[j]);
        if (ILP == 1) {
            // one slot per thread: four list entries at a time (indices first, then the four gathers; added in list order)
            for (int k = b[0]; k < e[0]; k += 4) {
                int32_t idx[4];
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) idx[u] = k + u < e[0] ? glist[k + u] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = idx[u] >= 0 ? fine[idx[u]] : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) if (idx[u] >= 0) a[0] += (double)v[u];
            }
        } else {
            for (int k = 0; k < len; ++k) {
                int32_t idx[ILP];
#pragma unroll
                for (int j = 0; j < ILP; ++j) idx[j] = b[j] + k < e[j] ? glist[b[j] + k] : -1;
#pragma unroll
                for (int j = 0; j < ILP; ++j) if (idx[j] >= 0) a[j] += (double)fine[idx[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            const int64_t s = c0 + threadIdx.x + (int64_t)j * kBlock;
            if (s < nslots) coarse[s] = (TC)a[j];
        }
    }
}
// kind 0: a coarse operator P^T A P (long, uneven lists), kind 1: A*P (1-2 entries per slot)
template <class TC>
static void launch_galerkin(Ctx* c, int kind, int64_t nslots, const int32_t* gptr, const int32_t* glist, const float* fine, TC* coarse) {
    const Tunables& T = tunables();
    note_bytes(c, 4.0 * (double)(nslots + 1) + (double)sizeof(TC) * (double)nslots);   // + the gather lists, noted by the caller
    const int contig = T.gal_contig[kind];
    const int ilp = T.gal_ilp[kind];
    const int g = (int)std::min<int64_t>(T.gal_grid[kind], std::max<int64_t>(1, (nslots + (int64_t)ilp * kBlock - 1) / ((int64_t)ilp * kBlock)));
    if (ilp == 4) hipLaunchKernelGGL((k_galerkin<TC, 4>), dim3(g), dim3(kBlock), 0, c->stream, nslots, gptr, glist, fine, coarse, contig);
    else if (ilp == 2) hipLaunchKernelGGL((k_galerkin<TC, 2>), dim3(g), dim3(kBlock), 0, c->stream, nslots, gptr, glist, fine, coarse, contig);
    else hipLaunchKernelGGL((k_galerkin<TC, 1>), dim3(g), dim3(kBlock), 0, c->stream, nslots, gptr, glist, fine, coarse, contig);
}

__global__ __launch_bounds__(kBlock) void k_diag_inv(int32_t n, const int32_t* __restrict__ diag_slot,
                                                     const float* __restrict__ vals, float* __restrict__ dinv) {
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float d = vals[diag_slot[i]];
        dinv[i] = (d != 0.0f) ? 1.0f / d : 1.0f;
    }
}

// float copies of the Jacobian and of its inverse diagonal (level 0 of the preconditioner)
// Smoother copy of a level's values (DevSell::pk): bfloat16, round to nearest even, beside the slot's 16-bit column.
// The smoothing sweeps stream two thirds of the bytes; what they lose is 2^-9 relative per entry of a preconditioner
// whose cycle reduces the error by ~0.7: the Krylov iteration counts do not move.
__device__ __forceinline__ uint32_t bf16_bits(float v) {   // round to nearest even, in the upper half word
    uint32_t u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u & 0xffff0000u;
}
// own / own_off: column of row i's own slot = own_off + own[i] (A*P: its aggregate), or i itself (own == nullptr)
__global__ __launch_bounds__(kBlock) void k_pack_bf16(const DevSell A, const float* __restrict__ vals, uint32_t* __restrict__ pk,
                                                      const int32_t* __restrict__ own, int32_t own_off) {
    const int lane = threadIdx.x & 63;
    for (int s = 4 * blockIdx.x + wave_index(); s < A.nslice; s += 4 * gridDim.x) {
        const SellMeta m = sell_meta(A, s);
        if (m.cb < 0) continue;
        const int row = min(s * kSlice + lane, A.n_rows - 1);
        const uint32_t mine = (uint32_t)((own ? own_off + own[row] : row) - m.cb);
        float sum = 0.0f;
        int kd = -1;
        for (int k = 0; k < m.width; ++k) {
            const float v = vals[m.base + k * kSlice + lane];
            const uint32_t col = A.col16[m.p16 + k * kSlice + lane];
            sum += v;
            if (col == mine && v != 0.0f) kd = k;
            pk[m.p16 + k * kSlice + lane] = bf16_bits(v) | col;
        }
        if (kd >= 0) pk[m.p16 + kd * kSlice + lane] = bf16_bits(sum) | mine;
    }
}
static void launch_pack(Ctx* c, const DevSell& A, const float* vals, int64_t slots16, const int32_t* own = nullptr, int32_t own_off = 0) {
    if (!A.pk) return;
    note_bytes(c, 10.0 * (double)slots16);
    hipLaunchKernelGGL(k_pack_bf16, dim3(std::min((A.nslice + 3) / 4, 4096)), dim3(kBlock), 0, c->stream, A, vals,
                       const_cast<uint32_t*>(A.pk), own, own_off);
}
__global__ __launch_bounds__(kBlock) void k_narrow(int64_t n, const double* __restrict__ a, float* __restrict__ b) {
    const int64_t n2 = n >> 1;
    typedef double dvec2 __attribute__((ext_vector_type(2)));
    const dvec2* __restrict__ a2 = reinterpret_cast<const dvec2*>(a);
    float2* __restrict__ b2 = reinterpret_cast<float2*>(b);
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
        const dvec2 v = __builtin_nontemporal_load(a2 + i);
        b2[i] = make_float2((float)v.x, (float)v.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) b[n - 1] = (float)a[n - 1];
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

// Shared coarsest level: every subdomain contributes its slice of the right-hand side (zeros elsewhere); the
// element-wise sum over subdomains is then the gathered vector.
__global__ __launch_bounds__(kBlock) void k_coarse_scatter(int n, int row0, int ncols, const float* __restrict__ rc,
                                                           double* __restrict__ rglob, const int* __restrict__ done) {
    if (*done) return;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < ncols; j += gridDim.x * kBlock)
        rglob[j] = (j >= row0 && j < row0 + n) ? (double)rc[j - row0] : 0.0;
}

// Gauss-Jordan inverse of a dense coarsest operator with 64 < n <= 1024 rows, in global memory (L2-resident):
// two launches per pivot -- a single workgroup scales the pivot row and lifts column p out, then the whole
// chip applies the rank-1 update.  ~6 us per pivot; a one-workgroup version took 140 ms at n = 977.
__global__ __launch_bounds__(1024) void k_gj_prep(int n, int p, double* __restrict__ M, double* __restrict__ mult,
                                                  double* __restrict__ prow) {
    const int tid = threadIdx.x;
    const double piv = M[(size_t)p * n + p];
    const double d = (piv != 0.0) ? 1.0 / piv : 0.0;
    for (int t = tid; t < n; t += 1024) mult[t] = (t != p) ? M[(size_t)t * n + p] : 0.0;
    __syncthreads();
    for (int t = tid; t < n; t += 1024) {
        const double v = ((t == p) ? 1.0 : M[(size_t)p * n + t]) * d;
        prow[t] = v;
        M[(size_t)p * n + t] = v;
        if (t != p) M[(size_t)t * n + p] = 0.0;
    }
}
__global__ __launch_bounds__(kBlock) void k_gj_update(int n, int p, double* __restrict__ M,
                                                      const double* __restrict__ mult, const double* __restrict__ prow) {
    const int total = n * n;
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
        const int i = e / n, j = e - i * n;
        if (i != p) M[e] -= mult[i] * prow[j];
    }
}
static void dense_invert_pivotwise(Ctx* c, int n, const double* A, double* inv, double* scratch /* >= 2n */) {
    (void)hipMemcpyAsync(inv, A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    const int g = std::min(2048, (n * n + kBlock - 1) / kBlock);
    for (int p = 0; p < n; ++p) {
        hipLaunchKernelGGL(k_gj_prep, dim3(1), dim3(1024), 0, c->stream, n, p, inv, scratch, scratch + n);
        hipLaunchKernelGGL(k_gj_update, dim3(g), dim3(kBlock), 0, c->stream, n, p, inv, scratch, scratch + n);
    }
}

// ---- blocked Gauss-Jordan inverse (no pivoting, like the pivot-wise version): per block of kGjB pivots K
//        [ D  R ]        [  D^-1      D^-1 R       ]
//        [ C  E ]  --->  [ -C D^-1    E - C D^-1 R ]
//   k_bgj_prepare  copies the column panel C (all rows x K) aside and inverts the diagonal block D in LDS
//   k_bgj_rows     Rp = D^-1 [row panel], with D^-1 itself in the columns of K (so that one formula serves below)
//   k_bgj_update   M[i][j] = Rp[i][j] for rows in K, else (j in K ? 0 : M[i][j]) - sum_s C[i][s] Rp[s][j]
// 3 launches per 32 pivots and a rank-32 update with LDS tiles instead of 2 launches and a rank-1 sweep of the
// whole matrix per pivot: the 2441-row inverse of the 10M-DOF hierarchy takes ~8 ms instead of 63.
constexpr int kGjB = 32;
__global__ __launch_bounds__(kBlock) void k_bgj_prepare(int n, int k0, int b, const double* __restrict__ M,
                                                        double* __restrict__ Cp, double* __restrict__ Dinv) {
    if (blockIdx.x + 1 < gridDim.x) {   // column panel: one row per thread
        const int i = blockIdx.x * kBlock + threadIdx.x;
        if (i < n)
            for (int s = 0; s < kGjB; ++s) Cp[(size_t)i * kGjB + s] = s < b ? M[(size_t)i * n + k0 + s] : 0.0;
        return;
    }
    __shared__ double D[kGjB][kGjB + 1], mult[kGjB];
    const int tid = threadIdx.x;
    for (int e = tid; e < kGjB * kGjB; e += kBlock) {
        const int r = e / kGjB, q = e % kGjB;
        D[r][q] = (r < b && q < b) ? M[(size_t)(k0 + r) * n + k0 + q] : (r == q ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int p = 0; p < b; ++p) {
        const double piv = D[p][p];
        const double d = (piv != 0.0) ? 1.0 / piv : 0.0;
        __syncthreads();
        if (tid < kGjB) mult[tid] = (tid != p) ? D[tid][p] : 0.0;
        __syncthreads();
        if (tid < kGjB) { D[p][tid] = (tid == p ? 1.0 : D[p][tid]) * d; if (tid != p) D[tid][p] = 0.0; }
        __syncthreads();
        for (int e = tid; e < kGjB * kGjB; e += kBlock) {
            const int r = e / kGjB, q = e % kGjB;
            if (r != p) D[r][q] -= mult[r] * D[p][q];
        }
        __syncthreads();
    }
    for (int e = tid; e < kGjB * kGjB; e += kBlock) Dinv[e] = D[e / kGjB][e % kGjB];
}
__global__ __launch_bounds__(kBlock) void k_bgj_rows(int n, int k0, int b, const double* __restrict__ M,
                                                     const double* __restrict__ Dinv, double* __restrict__ Rp) {
    __shared__ double Di[kGjB * kGjB];
    for (int e = threadIdx.x; e < kGjB * kGjB; e += kBlock) Di[e] = Dinv[e];
    __syncthreads();
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    double col[kGjB];
#pragma unroll
    for (int s = 0; s < kGjB; ++s) col[s] = s < b ? M[(size_t)(k0 + s) * n + j] : 0.0;
    const bool inK = j >= k0 && j < k0 + b;
    for (int r = 0; r < kGjB; ++r) {
        double a = 0.0;
        if (inK) a = Di[r * kGjB + (j - k0)];
        else
#pragma unroll
            for (int s = 0; s < kGjB; ++s) a += Di[r * kGjB + s] * col[s];
        Rp[(size_t)r * n + j] = a;
    }
}
__global__ __launch_bounds__(kBlock) void k_bgj_update(int n, int k0, int b, double* __restrict__ M,
                                                       const double* __restrict__ Cp, const double* __restrict__ Rp) {
    __shared__ double Cs[64][kGjB + 1];      // rows of the tile x pivots
    __shared__ double Rs[kGjB][64 + 1];      // pivots x columns of the tile
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * kGjB; e += kBlock) {
        const int r = e / kGjB, s = e % kGjB;
        Cs[r][s] = (i0 + r < n) ? Cp[(size_t)(i0 + r) * kGjB + s] : 0.0;
        const int s2 = e / 64, q = e % 64;
        Rs[s2][q] = (j0 + q < n) ? Rp[(size_t)s2 * n + j0 + q] : 0.0;
    }
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 8
    for (int s = 0; s < kGjB; ++s) {
        double cr[4], rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { cr[u] = Cs[ty * 4 + u][s]; rr[u] = Rs[s][tx * 4 + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] += cr[u] * rr[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + ty * 4 + u;
        if (i >= n) continue;
        const bool rowK = i >= k0 && i < k0 + b;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = j0 + tx * 4 + v;
            if (j >= n) continue;
            const bool colK = j >= k0 && j < k0 + b;
            double* m = M + (size_t)i * n + j;
            *m = rowK ? Rs[i - k0][tx * 4 + v] : (colK ? 0.0 : *m) - acc[u][v];
        }
    }
}
// scratch: >= (2 n + kGjB) * kGjB doubles
static void dense_invert_big(Ctx* c, int n, const double* A, double* inv, double* scratch) {
    if (tunables().gj_pivotwise) { dense_invert_pivotwise(c, n, A, inv, scratch); return; }   // cross-check switch
    (void)hipMemcpyAsync(inv, A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    double *Cp = scratch, *Rp = scratch + (size_t)n * kGjB, *Dinv = Rp + (size_t)n * kGjB;
    const int gv = (n + kBlock - 1) / kBlock, gt = (n + 63) / 64;
    for (int k0 = 0; k0 < n; k0 += kGjB) {
        const int b = std::min(kGjB, n - k0);
        hipLaunchKernelGGL(k_bgj_prepare, dim3(gv + 1), dim3(kBlock), 0, c->stream, n, k0, b, (const double*)inv, Cp, Dinv);
        hipLaunchKernelGGL(k_bgj_rows, dim3(gv), dim3(kBlock), 0, c->stream, n, k0, b, (const double*)inv,
                           (const double*)Dinv, Rp);
        hipLaunchKernelGGL(k_bgj_update, dim3(gt, gt), dim3(kBlock), 0, c->stream, n, k0, b, inv, (const double*)Cp,
                           (const double*)Rp);
    }
}

// r_c = P^T r
template <class TI>
__global__ __launch_bounds__(kBlock) void k_amg_restrict(int32_t n_coarse, const int32_t* __restrict__ members,
                                                         const TI* __restrict__ r, float* __restrict__ rc,
                                                         const int* __restrict__ done) {
    if (*done) return;
    for (int32_t I = blockIdx.x * kBlock + threadIdx.x; I < n_coarse; I += gridDim.x * kBlock) {
        const int4 m = reinterpret_cast<const int4*>(members)[I];
        TI acc = r[m.x];                          // every aggregate has at least one member
        if (m.y >= 0) acc += r[m.y];
        if (m.z >= 0) acc += r[m.z];
        if (m.w >= 0) acc += r[m.w];
        rc[I] = (float)acc;
    }
}

template <class TO>
__global__ __launch_bounds__(kBlock) void k_amg_prolong(int32_t n, float alpha, const int32_t* __restrict__ agg,
                                                        const float* __restrict__ ec, TO* __restrict__ x,
                                                        const int* __restrict__ done) {
    if (*done) return;
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] = (TO)(alpha * ec[agg[i]]);
}

// x' = x + w D^-1 (r - A x).  The finest level reads the Krylov vector r (double) and writes either the scratch
// iterate (float) or the preconditioned Krylov vector (double); coarser levels are float throughout.
template <class TX, class TR, class TO>
struct AmgSmoothArgs {
    DevSell A;
    const float* vals;
    const float* dinv;
    const TR* r;            // right-hand side of this level
    const TX* x;            // current iterate
    TO* xo;                 // smoothed iterate
    float omega;
    const int* done;        // Krylov stop flag: once set, every later kernel of the queue returns at once
    SplitSell split;        // PASS 1, 2 only (finest level of a decomposed mesh)
};
// FINE only gives the finest level its own symbol, so that profilers report its launches separately
template <bool FINE, class TX, class TR, class TO, int PASS = 0, bool PK = false>   // PASS: SplitSell; PK: DevSell::pk (shk_device.h)
__global__ __launch_bounds__(kBlock) void k_amg_post(const AmgSmoothArgs<TX, TR, TO> a) {
    if (*a.done) return;
    const int lane = threadIdx.x & 63;
    auto slice = [&](int s, const SellMeta& m) {
        const int row = min(s * kSlice + lane, a.A.n_rows - 1);   // tail rows of the last slice: clamped, not stored
        const TX xr = a.x[row];            // the row's own entries are requested ahead of the slice stream
        const TR rr = a.r[row];
        const float di = a.dinv[row];
        const bool skip = PASS == 1 ? a.split.ghost[s] != 0 : false;
        const auto sum = sell_row_sum_pk<PK>(a.A, m, a.vals, a.x, lane, row, (float)xr);   // (own column = the row)
        if (s * kSlice + lane < a.A.n_rows && !skip) a.xo[row] = (TO)(xr + a.omega * di * (rr - sum));
    };
    if (PASS == 2) {
        for (int k = 4 * blockIdx.x + wave_index(); k < a.split.n_list; k += 4 * gridDim.x) {
            const int s = __builtin_amdgcn_readfirstlane(a.split.list[k]);
            slice(s, sell_meta(a.A, s));
        }
    } else {
        for (SliceLoop it(a.A, wave_index()); it.valid(); it.next()) slice(it.s, it.m);
    }
}

// Second visit of the doubled cycle: r <- r - A x (the level's residual after its first solution), x3 <- x.
__global__ __launch_bounds__(kBlock) void k_amg_residual_save(const AmgSmoothArgs<float, float, float> a, float* __restrict__ rout) {
    if (*a.done) return;
    const int lane = threadIdx.x & 63;
    for (SliceLoop it(a.A, wave_index()); it.valid(); it.next()) {
        const int row = min(it.s * kSlice + lane, a.A.n_rows - 1);
        const float xr = a.x[row];
        const float rr = a.r[row];
        const float sum = sell_row_sum(a.A, it.m, a.vals, a.x, lane);
        if (it.s * kSlice + lane < a.A.n_rows) { rout[row] = rr - sum; a.xo[row] = xr; }
    }
}
__global__ __launch_bounds__(kBlock) void k_amg_add(int32_t n, const float* __restrict__ a, float* __restrict__ x,
                                                    const int* __restrict__ done) {
    if (*done) return;
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] += a[i];
}

// First smoothing sweep after the prolongation, without ever forming x0 = alpha P e:
//   x1 = alpha e[agg] + w D^-1 (r - alpha (A P) e)       (A P has ~4 entries per row, e is 4x shorter than x)
template <class TR>
struct AmgFirstArgs {
    DevSell AP;              // fine rows x coarse columns
    const float* vals;       // A*P values
    const float* dinv;
    const TR* r;
    const float* e;          // coarse correction, indexed by the A*P columns
    const int32_t* agg;
    int32_t agg_off;         // e[agg_off + agg[row]] is the row's own aggregate (agg holds a subdomain's local ids; the
                             // columns of a transfer onto the replicated global level are global)
    float* xo;
    float omega, alpha;
    const int* done;
    // decomposed level with frozen ghosts: ghost column g of the iterate gets alpha * e[ghost_col[g]] (in xo and, for the
    // levels whose later sweeps alternate between two vectors, in xo2 as well); nullptr / 0 otherwise
    const int32_t* ghost_col;
    int32_t n_ghost;
    float* xo2;
};
// GHOSTS: the instance that also fills the ghost columns (decomposed levels with frozen ghosts).  A separate instance
// because the mere presence of that (zero-trip) loop cost the one-subdomain kernel 25 % (97 -> 121 us at 10M rows: 16 more
// registers and a differently scheduled slice loop); without it the kernel is the one rounds 1-2 measured.
template <bool FINE, class TR, bool GHOSTS = false, bool PK = false>
__global__ __launch_bounds__(kBlock) void k_amg_first(const AmgFirstArgs<TR> a) {
    if (*a.done) return;
    const int lane = threadIdx.x & 63;
    if constexpr (GHOSTS) {
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_ghost; i += gridDim.x * kBlock) {
            const float v = a.alpha * a.e[a.ghost_col[i]];
            a.xo[a.AP.n_rows + i] = v;
            if (a.xo2) a.xo2[a.AP.n_rows + i] = v;
        }
    }
    for (SliceLoop it(a.AP, wave_index()); it.valid(); it.next()) {
        const int row = min(it.s * kSlice + lane, a.AP.n_rows - 1);
        const int ag = a.agg[row];
        const float rr = (float)a.r[row];
        const float di = a.dinv[row];
        const float eo = a.e[a.agg_off + ag];
        const float sum = sell_row_sum_pk<PK>(a.AP, it.m, a.vals, a.e, lane, a.agg_off + ag, eo);
        if (it.s * kSlice + lane < a.AP.n_rows) a.xo[row] = a.alpha * eo + a.omega * di * (rr - a.alpha * sum);
    }
}

// ---- fused multi-sweep smoother: the four sweeps of a level in ONE launch (SweepPlan, shk_plan.h).
//   x1 = alpha e[agg] + w0 D^-1 (r - alpha (A P) e)   on S3 = rows within distance 3 of the workgroup's 256-row block
//   x2 = x1 + w1 D^-1 (r - A x1)                      on S2
//   x3 = x2 + w2 D^-1 (r - A x2)                      on S1
//   x4 = x3 + w3 D^-1 (r - A x3)                      on S0 = the block  -> xo
// A thread owns up to three rows of the extended block -- row tid of the block itself, local rows tid + 256 and
// tid + 512 of the rings -- and keeps what it needs of them in REGISTERS: the matrix rows it recomputes in sweeps 2-4
// (its own, and the first ring row if that lies in S2), right-hand side and 1/diag.  The kernel is a latency chain
// on the small levels it exists for, so every global load is issued as early as its address is known: block header
// (scalar) -> one 16-byte record per ring row -> all row data of the three rows at once -> the gathers of the coarse
// correction.  Only the iterates pass through LDS (two buffers, ping-pong).  Rows of the block read the level's SELL
// arrays coalesced; ring rows gather their entries (served by L2: they are the neighbouring blocks' own rows).
// Ghost columns of a decomposed level (frozen-ghost smoothing) are constants alpha e[ghost_col].
// Same formulas and slot order as k_amg_first / k_amg_post: the result equals the four separate launches'.
template <class TR>
struct AmgSweepArgs {
    DevSell A;               // the level's operator (ptr used; columns come from the plan)
    const float* vals;
    const float* dinv;
    const TR* r;
    DevSell AP;              // first sweep: fine rows x coarse columns (ptr, col used)
    const float* ap_vals;
    const float* e;          // coarse correction, indexed by the A*P columns
    const int32_t* agg;
    int32_t agg_off;
    const int32_t* ghost_col;   // coarse column of ghost column n_rows + g (nullptr: the level has no ghost columns)
    int32_t n_ghost;
    float* xo;               // result: own rows (and, with ghost_col, the ghost segment = the frozen values)
    float w[4];
    float alpha;
    const int* done;
    int32_t nblk, width;     // plan
    const int32_t *hdr, *ext_info;
    const uint16_t *lcol_own, *ring_lcol;
};
template <class TR, int W>
__global__ __launch_bounds__(kSweepRows) void k_amg_sweeps(const AmgSweepArgs<TR> a) {
    constexpr int WP = kSweepMaxWidthAP;
    __shared__ float xa[kSweepMaxLocal], xb[kSweepMaxLocal];
    if (*a.done) return;
    const int tid = threadIdx.x;
    const int n = a.A.n_rows;
    for (int i = blockIdx.x * kSweepRows + tid; i < a.n_ghost; i += gridDim.x * kSweepRows)
        a.xo[n + i] = a.alpha * a.e[a.ghost_col[i]];
    for (int b = blockIdx.x; b < a.nblk; b += gridDim.x) {
        const int r0 = b * kSweepRows, n0 = min(n - r0, kSweepRows);
        const int32_t* __restrict__ hd = a.hdr + (size_t)8 * b;     // wave-uniform: scalar loads
        // local row ids: 0 .. n0-1 the block, kSweepRows .. the rings (SweepPlan), then the fixed entries
        const int ext0 = hd[0], rl0 = hd[1], nS1 = kSweepRows + hd[2], nS2 = nS1 + hd[3], nS3 = nS2 + hd[4], nfix = hd[5];
        const int4* __restrict__ info = reinterpret_cast<const int4*>(a.ext_info) + ext0;
        // ---- this thread's rows: local ids t0 = tid (own), t1 = tid + 256, t2 = tid + 512 (rings); row records first
        const int t1 = tid + kSweepRows, t2 = tid + 2 * kSweepRows;
        const bool has0 = tid < n0, has1 = t1 < nS3, has2 = t2 < nS3, a1 = t1 < nS2;   // a1: ring row 1 is recomputed later
        const int4 i1 = has1 ? info[t1 - kSweepRows] : make_int4(0, 0, 0, 0);
        const int4 i2 = has2 ? info[t2 - kSweepRows] : make_int4(0, 0, 0, 0);
        const int4 ifx = tid < nfix ? info[nS3 - kSweepRows + tid] : make_int4(n, 0, 0, 0);
        const int g0 = min(r0 + tid, n - 1), s0 = g0 >> 6;
        const int base0 = a.A.ptr[s0] + (g0 & 63), wid0 = has0 ? (a.A.ptr[s0 + 1] - a.A.ptr[s0]) >> 6 : 0;
        const int pbase0 = a.AP.ptr[s0] + (g0 & 63), pwid0 = has0 ? (a.AP.ptr[s0 + 1] - a.AP.ptr[s0]) >> 6 : 0;
        const int g1 = i1.x, g2 = i2.x;
        const int len1 = a1 ? (i1.w & 255) : 0, plen1 = has1 ? (i1.w >> 8) : 0, plen2 = has2 ? (i2.w >> 8) : 0;
        // ---- all row data at once
        float v0[W], v1[W];
        uint16_t c0[W], c1[W];
        const uint16_t* __restrict__ lc1 = a.ring_lcol + (size_t)(rl0 + tid) * a.width;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const bool m0 = k < wid0, m1 = k < len1;
            v0[k] = m0 ? a.vals[base0 + k * kSlice] : 0.0f;
            c0[k] = m0 ? a.lcol_own[base0 + k * kSlice] : (uint16_t)0;
            v1[k] = m1 ? a.vals[i1.y + k * kSlice] : 0.0f;
            c1[k] = m1 ? lc1[k] : (uint16_t)0;
        }
        float pv0[WP], pv1[WP], pv2[WP];
        int32_t pc0[WP], pc1[WP], pc2[WP];
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            const bool m0 = k < pwid0, m1 = k < plen1, m2 = k < plen2;
            pv0[k] = m0 ? a.ap_vals[pbase0 + k * kSlice] : 0.0f;
            pc0[k] = m0 ? a.AP.col[pbase0 + k * kSlice] : 0;
            pv1[k] = m1 ? a.ap_vals[i1.z + k * kSlice] : 0.0f;
            pc1[k] = m1 ? a.AP.col[i1.z + k * kSlice] : 0;
            pv2[k] = m2 ? a.ap_vals[i2.z + k * kSlice] : 0.0f;
            pc2[k] = m2 ? a.AP.col[i2.z + k * kSlice] : 0;
        }
        const float rr0 = has0 ? (float)a.r[g0] : 0.0f, rr1 = has1 ? (float)a.r[g1] : 0.0f, rr2 = has2 ? (float)a.r[g2] : 0.0f;
        const float d0 = has0 ? a.dinv[g0] : 0.0f, d1 = has1 ? a.dinv[g1] : 0.0f, d2 = has2 ? a.dinv[g2] : 0.0f;
        const int ag0 = has0 ? a.agg[g0] : 0, ag1 = has1 ? a.agg[g1] : 0, ag2 = has2 ? a.agg[g2] : 0;
        const int gfx = tid < nfix ? a.ghost_col[ifx.x - n] : 0;
        // ---- gathers of the coarse correction
        float s0_ = 0.0f, s1_ = 0.0f, s2_ = 0.0f;
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            s0_ += pv0[k] * a.e[pc0[k]];
            s1_ += pv1[k] * a.e[pc1[k]];
            s2_ += pv2[k] * a.e[pc2[k]];
        }
        const float e0 = a.e[a.agg_off + ag0], e1 = a.e[a.agg_off + ag1], e2 = a.e[a.agg_off + ag2];
        const float efx = tid < nfix ? a.alpha * a.e[gfx] : 0.0f;
        __syncthreads();   // the previous block's readers are done with the LDS buffers
        // ---- sweep 1 on S3 (through A*P), constants of the ghost columns
        if (has0) xa[tid] = a.alpha * e0 + a.w[0] * d0 * (rr0 - a.alpha * s0_);
        if (has1) xa[t1] = a.alpha * e1 + a.w[0] * d1 * (rr1 - a.alpha * s1_);
        if (has2) xa[t2] = a.alpha * e2 + a.w[0] * d2 * (rr2 - a.alpha * s2_);
        if (tid < nfix) { xa[nS3 + tid] = efx; xb[nS3 + tid] = efx; }
        __syncthreads();
        // ---- sweeps 2 .. 4 on S2, S1, S0: from registers and LDS
        auto row = [&](const float* xin, const float (&v)[W], const uint16_t (&cc)[W]) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < W; ++k) sum += v[k] * xin[cc[k]];
            return sum;
        };
        {
            if (has0) xb[tid] = xa[tid] + a.w[1] * d0 * (rr0 - row(xa, v0, c0));
            if (a1) xb[t1] = xa[t1] + a.w[1] * d1 * (rr1 - row(xa, v1, c1));
        }
        __syncthreads();
        {
            if (has0) xa[tid] = xb[tid] + a.w[2] * d0 * (rr0 - row(xb, v0, c0));
            if (t1 < nS1) xa[t1] = xb[t1] + a.w[2] * d1 * (rr1 - row(xb, v1, c1));
        }
        __syncthreads();
        if (has0) a.xo[r0 + tid] = xa[tid] + a.w[3] * d0 * (rr0 - row(xa, v0, c0));
    }
}

// ---- tail: every level with <= kTailRows rows runs inside ONE workgroup (restrictions, dense coarsest solve,
// prolongations and smoothing sweeps separated by workgroup barriers) instead of ~4 tiny launches per level.
constexpr int kTailThreads = 1024;
constexpr int kTailMaxLevels = 8;

struct TailLevel {
    int32_t n, nslice, n_coarse;         // rows, slices, rows of the next level
    const int32_t *ptr, *col, *agg, *members;
    const float *vals, *dinv;
    float *x, *x2, *r;
};
struct TailArgs {
    int nlev;
    TailLevel lv[kTailMaxLevels];
    int n_c, row0, ncols;                // dense coarsest: my rows, first row, columns
    const double* inv;
    float *cr, *cx;
    const double* cglob;                 // distributed: gathered coarsest rhs (phase 2); nullptr: use cr
    float omega, omega2, alpha;
    const int* done;
    int dense_in_tail;                   // 0: the coarsest solve was done by k_dense_gemv before phase 2
};

// x[i] = sum_j inv[(row0 + i) * ncols + j] * r[j]: one wave per row, over the whole chip (the one-workgroup
// tail would read a 1024 x 1024 inverse through a single CU: 150 us instead of 5).
// TI: the inverse as it is applied -- double for a shared dense level of a decomposed hierarchy, a float copy for a
// hierarchy's own dense level (the cycle is a float preconditioner; 2441^2 entries: 48 -> 24 MB per application)
template <class TR, class TI = double>
__global__ __launch_bounds__(kBlock) void k_dense_gemv(int n, int row0, int ncols, const TI* __restrict__ inv,
                                                       const TR* __restrict__ r, float* __restrict__ x,
                                                       const int* __restrict__ done) {
    if (*done) return;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const TI* row = inv + (size_t)(row0 + i) * ncols;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int j = lane;
        for (; j + 192 < ncols; j += 256) {  // four independent streams per lane
            a0 += (double)row[j] * (double)r[j];
            a1 += (double)row[j + 64] * (double)r[j + 64];
            a2 += (double)row[j + 128] * (double)r[j + 128];
            a3 += (double)row[j + 192] * (double)r[j + 192];
        }
        for (; j < ncols; j += 64) a0 += (double)row[j] * (double)r[j];
        double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) x[i] = (float)acc;
    }
}

__device__ __forceinline__ void tail_post(const TailLevel& L, const float* x, float* xo, float omega) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < L.nslice; s += kTailThreads / 64) {
        const int base = L.ptr[s];
        const int width = (L.ptr[s + 1] - base) >> 6;
        float sum = 0.0f;
        for (int k = 0; k < width; ++k) sum += L.vals[base + k * kSlice + lane] * x[L.col[base + k * kSlice + lane]];
        const int row = s * kSlice + lane;
        if (row < L.n) xo[row] = x[row] + omega * L.dinv[row] * (L.r[row] - sum);
    }
}

// PHASE 0: whole tail (one context).  PHASE 1: restrictions only.  PHASE 2: coarsest solve + way up (the
// gathered coarsest right-hand side was summed over subdomains between the two).
template <int PHASE>
__global__ __launch_bounds__(kTailThreads) void k_amg_tail(const TailArgs a) {
    if (*a.done) return;
    const int tid = threadIdx.x;
    if (PHASE != 2) {
        for (int k = 0; k < a.nlev; ++k) {
            const TailLevel& L = a.lv[k];
            float* rc = (k + 1 < a.nlev) ? a.lv[k + 1].r : a.cr;
            for (int I = tid; I < L.n_coarse; I += kTailThreads) {
                const int4 m = reinterpret_cast<const int4*>(L.members)[I];
                float acc = L.r[m.x];
                if (m.y >= 0) acc += L.r[m.y];
                if (m.z >= 0) acc += L.r[m.z];
                if (m.w >= 0) acc += L.r[m.w];
                rc[I] = acc;
            }
            __syncthreads();
        }
    }
    if (PHASE == 1) return;
    if (a.dense_in_tail) {   // dense coarsest solve: one wave per row, lanes over columns
        const int lane = tid & 63, wave = tid >> 6;
        for (int i = wave; i < a.n_c; i += kTailThreads / 64) {
            const double* row = a.inv + (size_t)(a.row0 + i) * a.ncols;
            double acc = 0.0;
            if (a.cglob) for (int j = lane; j < a.ncols; j += 64) acc += row[j] * a.cglob[j];
            else for (int j = lane; j < a.ncols; j += 64) acc += row[j] * (double)a.cr[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            if (lane == 0) a.cx[i] = (float)acc;
        }
        __syncthreads();
    }
    for (int k = a.nlev - 1; k >= 0; --k) {
        const TailLevel& L = a.lv[k];
        const float* ec = (k + 1 < a.nlev) ? a.lv[k + 1].x2 : a.cx;
        for (int i = tid; i < L.n; i += kTailThreads) L.x2[i] = a.alpha * ec[L.agg[i]];
        __syncthreads();
        tail_post(L, L.x2, L.x, a.omega);
        __syncthreads();
        tail_post(L, L.x, L.x2, a.omega2);
        __syncthreads();
    }
}

// One power-iteration step xo = D^-1 A x; with NORMS the workgroups also leave their shares of |xo|^2 and |x|^2
// in part[0 .. grid) and part[kMaxParts .. kMaxParts + grid) (summed on the host in a fixed order).
template <bool NORMS>
__global__ __launch_bounds__(kBlock) void k_power_step(const DevSell A, const float* __restrict__ vals,
                                                       const float* __restrict__ dinv, const float* __restrict__ x,
                                                       float* __restrict__ xo, double* __restrict__ part) {
    __shared__ double sh[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double so = 0.0, sx = 0.0;
    for (SliceLoop it(A, wave_index()); it.valid(); it.next()) {
        const float sum = sell_row_sum(A, it.m, vals, x, lane);
        const int row = it.s * kSlice + lane;
        if (row < A.n_rows) {
            const float y = dinv[row] * sum;
            xo[row] = y;
            if (NORMS) { so += (double)y * y; sx += (double)x[row] * x[row]; }
        }
    }
    if (NORMS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { so += __shfl_down(so, o, 64); sx += __shfl_down(sx, o, 64); }
        if (lane == 0) { sh[wave] = so; sh[4 + wave] = sx; }
        __syncthreads();
        if (tid == 0) {
            part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
            part[kMaxParts + blockIdx.x] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_power_init(int32_t n, int32_t n_cols, float* __restrict__ x) {
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_cols; i += gridDim.x * kBlock) {
        uint32_t h = (uint32_t)i * 2654435761u;   // fixed pseudo-random start, ghosts columns stay zero
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = i < n ? 0.5f + (float)(h & 0xFFFF) * (1.0f / 65536.0f) * ((h & 0x10000) ? 1.0f : -1.0f) : 0.0f;
    }
}

// ---- Lanczos estimate of lambda_max(D^-1 A) in the |D| inner product (D^-1 A is self-adjoint in it when A is
// symmetric; the Jacobian's advection-like part is a small perturbation).  Three kernels per step:
//   k_lanczos_w   w = D^-1 A v - beta v_prev,  partial (w, v)_D
//   k_lanczos_n   w -= alpha v,                partial (w, w)_D
//   k_lanczos_s   v_next = w / beta   (written over v_prev)
__global__ __launch_bounds__(kBlock) void k_lanczos_w(const DevSell A, const float* __restrict__ vals,
                                                      const float* __restrict__ dinv, const float* __restrict__ v,
                                                      const float* __restrict__ vprev, float beta, float* __restrict__ w,
                                                      double* __restrict__ part) {
    __shared__ double sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc = 0.0;
    for (SliceLoop it(A, wave_index()); it.valid(); it.next()) {
        const float sum = sell_row_sum(A, it.m, vals, v, lane);
        const int row = it.s * kSlice + lane;
        if (row < A.n_rows) {
            const float di = dinv[row];
            const float wi = di * sum - beta * vprev[row];
            w[row] = wi;
            acc += (double)wi * v[row] / fabs((double)di);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (tid == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(kBlock) void k_lanczos_n(int32_t n, float alpha, const float* __restrict__ v,
                                                      const float* __restrict__ dinv, float* __restrict__ w,
                                                      double* __restrict__ part) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float wi = w[i] - alpha * v[i];
        w[i] = wi;
        acc += (double)wi * wi / fabs((double)dinv[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(kBlock) void k_lanczos_s(int32_t n, float inv_beta, const float* __restrict__ w, float* __restrict__ vnext) {
    for (int32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) vnext[i] = w[i] * inv_beta;
}

// Largest eigenvalue of the symmetric tridiagonal (alpha, beta) by bisection on the Sturm count.
static double tridiag_lambda_max(const std::vector<double>& al, const std::vector<double>& be) {
    const int m = (int)al.size();
    double lo = al[0], hi = al[0];
    for (int i = 0; i < m; ++i) {
        const double r = (i > 0 ? std::fabs(be[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(be[i]) : 0.0);
        lo = std::min(lo, al[i] - r);
        hi = std::max(hi, al[i] + r);
    }
    auto count_below = [&](double x) {   // eigenvalues < x
        int cnt = 0;
        double d = 1.0;
        for (int i = 0; i < m; ++i) {
            const double b2 = i > 0 ? be[i - 1] * be[i - 1] : 0.0;
            d = (al[i] - x) - (i > 0 ? b2 / (d == 0.0 ? 1e-300 : d) : 0.0);
            if (d < 0.0) ++cnt;
        }
        return cnt;
    };
    for (int it = 0; it < 100; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_below(mid) >= m) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

// Gershgorin bound of the spectrum of D^-1 A: max_i sum_j |a_ij| |1 / a_ii| (one partial maximum per workgroup)
__global__ __launch_bounds__(kBlock) void k_gershgorin(const DevSell A, const float* __restrict__ vals,
                                                       const float* __restrict__ dinv, double* __restrict__ part) {
    __shared__ double sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double m = 0.0;
    for (int s = blockIdx.x * 4 + wave; s < A.nslice; s += gridDim.x * 4) {
        const int base = A.ptr[s], width = (A.ptr[s + 1] - base) >> 6, row = s * kSlice + lane;
        float sum = 0.0f;
        for (int k = 0; k < width; ++k) sum += fabsf(vals[base + k * kSlice + lane]);   // padding slots hold zeros
        if (row < A.n_rows) m = fmax(m, (double)(sum * fabsf(dinv[row])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

static int small_grid(int64_t n) { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + kBlock - 1) / kBlock)); }

static DevSell level_sell(const Ctx* c, const AmgHierarchy& H, size_t l);
static DevSell ap_sell(const AmgXfer& X);

// Largest common factor <= 1 of a sweep sequence's dampings w_k = f c_k / lambda such that the sequence's error
// polynomial  prod_k (1 - w_k t)  stays within [-1, 1] at t = G, an upper bound of the spectrum of D^-1 A (the
// Lanczos estimate + 5 %, capped by Gershgorin's bound -- which alone is 45 % too pessimistic on these operators:
// 3.76 against 2.6): beyond the last root 1 / w_k the polynomial grows monotonically, so |p(G)| <= 1 means no
// eigenmode in (1 / w_max, G] is amplified (between the roots |p| < 1 for the damping ratios used here).  This is what keeps the cycle off the
// cliff measured at 10M rows: dampings 20 % above the tuned ones amplify the top of the spectrum by 1.6 per cycle
// and BiCGStab needs 585 iterations instead of 47; 30 % above, it diverges.
static double cap_factor(const double* cs, int n, double lambda, double G) {
    if (!(G > 0.0) || !(lambda > 0.0)) return 1.0;
    auto amp = [&](double f) {
        double p = 1.0;
        for (int k = 0; k < n; ++k) p *= (1.0 - f * cs[k] / lambda * G);
        return std::fabs(p);
    };
    // amp(0) = 1 and amp < 1 just above 0: the first factor (scanning upwards) at which it exceeds 1 again bounds
    // the safe range, which is therefore contiguous from 0
    constexpr int steps = 4000;
    for (int i = 1; i <= steps; ++i)
        if (amp((double)i / steps) > 1.0) return (double)(i - 1) / steps;
    return 1.0;
}
static void damping_caps(AmgHierarchy& H) {
    const double c2s[2] = {H.c1, H.c2};
    H.cap2 = cap_factor(c2s, 2, H.lambda, H.lam_max);
    H.cap4 = cap_factor(H.c4, 4, H.lambda, H.lam_max);
}

// `steps` Lanczos steps on one level (owned block; ghost columns read as zero); v0, v1, w: scratch vectors of the
// level.  Returns the largest Ritz value (a lower bound that is within a few per cent after ~30 steps, where 64 power
// steps are still 5-10 % low on these operators), or 0 on breakdown.
static hipError_t lanczos_lambda(Ctx* c, const DevSell& A, const float* vals, const float* dinv, float* v0, float* v1,
                                 float* w, int steps, double* out) {
    *out = 0.0;
    const int n = A.n_rows, grid = std::min((A.nslice + 3) / 4, 2048), vgrid = small_grid(n);
    double* part = c->d_part + (size_t)P_AUX * kMaxParts;
    std::vector<double> h((size_t)std::max(grid, vgrid));
    hipError_t e;
    auto host_sum = [&](int cnt, double* sum) -> hipError_t {
        hipError_t e2 = hipMemcpyAsync(h.data(), part, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e2 != hipSuccess) return e2;
        if ((e2 = wait_stream(c)) != hipSuccess) return e2;
        double a = 0.0;
        for (int i = 0; i < cnt; ++i) a += h[i];
        *sum = a;
        return hipSuccess;
    };
    const dim3 g1(std::min(1024, (A.n_cols + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(k_power_init, g1, dim3(kBlock), 0, c->stream, n, A.n_cols, v1);          // start vector
    if ((e = hipMemsetAsync(v0, 0, (size_t)A.n_cols * sizeof(float), c->stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w, 0, (size_t)A.n_cols * sizeof(float), c->stream)) != hipSuccess) return e;
    // normalise v1 in the D norm: reuse k_lanczos_n with alpha = 0 on a copy in w
    if ((e = hipMemcpyAsync(w, v1, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, c->stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_lanczos_n, dim3(vgrid), dim3(kBlock), 0, c->stream, n, 0.0f, (const float*)v1, dinv, w, part);
    double nn = 0.0;
    if ((e = host_sum(vgrid, &nn)) != hipSuccess) return e;
    if (!(nn > 0.0) || !std::isfinite(nn)) return hipSuccess;
    hipLaunchKernelGGL(k_lanczos_s, dim3(vgrid), dim3(kBlock), 0, c->stream, n, (float)(1.0 / std::sqrt(nn)), (const float*)w, v1);
    std::vector<double> al, be;
    float* vprev = v0;
    float* v = v1;
    double beta = 0.0;
    for (int j = 0; j < steps; ++j) {
        hipLaunchKernelGGL(k_lanczos_w, dim3(grid), dim3(kBlock), 0, c->stream, A, vals, dinv, (const float*)v,
                           (const float*)vprev, (float)beta, w, part);
        double a = 0.0, b2 = 0.0;
        if ((e = host_sum(grid, &a)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_lanczos_n, dim3(vgrid), dim3(kBlock), 0, c->stream, n, (float)a, (const float*)v, dinv, w, part);
        if ((e = host_sum(vgrid, &b2)) != hipSuccess) return e;
        if (!std::isfinite(a) || !std::isfinite(b2)) break;
        al.push_back(a);
        beta = std::sqrt(std::max(b2, 0.0));
        if (!(beta > 1e-6 * std::fabs(a)) || j + 1 == steps) break;   // invariant subspace found / done
        be.push_back(beta);
        hipLaunchKernelGGL(k_lanczos_s, dim3(vgrid), dim3(kBlock), 0, c->stream, n, (float)(1.0 / beta), (const float*)w, vprev);
        std::swap(v, vprev);   // v_next was written over v_prev
    }
    if (al.empty()) return hipSuccess;
    be.resize(al.size() - 1);
    *out = tridiag_lambda_max(al, be);
    return hipMemsetAsync(part, 0, (size_t)std::max(grid, vgrid) * sizeof(double), c->stream);
}

// Largest eigenvalue of D^-1 A over the levels that run as separate launches (the one-workgroup tail levels
// are coarser Galerkin products of the same operator), by `steps` power iterations each; one host sync.
static hipError_t estimate_lambda(Ctx* c, AmgHierarchy& H) {
    // 16 un-normalised steps (no float overflow: growth < 3^16).  The result is not lambda_max -- 32 / 64 steps give
    // 2.37 / 2.54 at 10M rows where 16 give 2.07 -- but a mesh-independent measure (2.04 .. 2.07 from 12k to 10M rows)
    // that the dampings H.c1, H.c2, H.c4 were tuned against.
    constexpr int steps = 16;
    const int kLanczosSteps = tunables().amg_lanczos;
    double lam = 0.0, gersh = 0.0, lanczos = 0.0;
    std::vector<double> h(2 * (size_t)kMaxParts);
    // Gershgorin bound G >= lambda_max(D^-1 A) over every sparse level: the dampings are capped so that no sweep
    // sequence amplifies anything in (0, G] (damping_caps below) -- a guarantee that does not depend on how well the
    // power iteration has converged
    for (size_t l = 0; l < H.xf.size(); ++l) {
        const DevSell A = level_sell(c, H, l);
        const int grid = std::min((A.nslice + 3) / 4, 1024);
        hipLaunchKernelGGL(k_gershgorin, dim3(grid), dim3(kBlock), 0, c->stream, A, l == 0 ? H.top_vals : H.lv[l].vals,
                           l == 0 ? H.top_dinv : H.lv[l].dinv, c->d_part + (size_t)P_AUX * kMaxParts);
        hipError_t e = hipMemcpyAsync(h.data(), c->d_part + (size_t)P_AUX * kMaxParts, (size_t)grid * sizeof(double),
                                      hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return e;
        if ((e = wait_stream(c)) != hipSuccess) return e;
        for (int b = 0; b < grid; ++b) if (std::isfinite(h[b])) gersh = std::max(gersh, h[b]);
        if ((e = hipMemsetAsync(c->d_part + (size_t)P_AUX * kMaxParts, 0, (size_t)grid * sizeof(double), c->stream)) != hipSuccess)
            return e;
    }
    for (size_t l = 0; l < H.xf.size(); ++l) {
        if (l > 0 && H.lv[l].n <= kTailRows) break;
        const DevSell A = level_sell(c, H, l);
        const float* vals = l == 0 ? H.top_vals : H.lv[l].vals;
        const float* dinv = l == 0 ? H.top_dinv : H.lv[l].dinv;
        float* xa = l == 0 ? H.x0 : H.lv[l].x;
        float* xb = l == 0 ? H.x1 : H.lv[l].x2;
        const int grid = std::min((A.nslice + 3) / 4, 2048);
        hipLaunchKernelGGL(k_power_init, dim3(std::min(1024, (A.n_cols + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           A.n_rows, A.n_cols, xa);
        // the second buffer's ghost columns must read as zero (the kernels only write owned rows)
        hipError_t e = hipSuccess;
        if (A.n_cols > A.n_rows &&
            (e = hipMemsetAsync(xb + A.n_rows, 0, (size_t)(A.n_cols - A.n_rows) * sizeof(float), c->stream)) != hipSuccess)
            return e;
        for (int k = 0; k < steps; ++k) {
            if (k + 1 < steps)
                hipLaunchKernelGGL(k_power_step<false>, dim3(grid), dim3(kBlock), 0, c->stream, A, vals, dinv,
                                   (const float*)xa, xb, c->d_part + (size_t)P_AUX * kMaxParts);
            else
                hipLaunchKernelGGL(k_power_step<true>, dim3(grid), dim3(kBlock), 0, c->stream, A, vals, dinv,
                                   (const float*)xa, xb, c->d_part + (size_t)P_AUX * kMaxParts);
            std::swap(xa, xb);
        }
        if ((e = hipMemcpyAsync(h.data(), c->d_part + (size_t)P_AUX * kMaxParts, 2 * (size_t)kMaxParts * sizeof(double),
                                hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
        if ((e = wait_stream(c)) != hipSuccess) return e;
        double so = 0.0, sx = 0.0;
        for (int b = 0; b < grid; ++b) { so += h[b]; sx += h[kMaxParts + b]; }
        if (sx > 0.0 && std::isfinite(so)) lam = std::max(lam, std::sqrt(so / sx));
        {   // converged estimate for the damping caps (the power value above only scales the tuned constants)
            double ll = 0.0;
            float* third = l == 0 ? H.x2 : H.lv[l].r;
            if (third && (e = lanczos_lambda(c, A, vals, dinv, l == 0 ? H.x0 : H.lv[l].x, l == 0 ? H.x1 : H.lv[l].x2, third,
                                             kLanczosSteps, &ll)) != hipSuccess)
                return e;
            lanczos = std::max(lanczos, ll);
        }
        // (the scratch vectors' ghost columns are still zero, and every V-cycle overwrites their owned rows first)
        // The two partial arrays go back to zero: a replicated global level can have more row groups than this
        // subdomain's own grid, and entries beyond that grid must stay zero for the reductions across subdomains.
        if ((e = hipMemsetAsync(c->d_part + (size_t)P_AUX * kMaxParts, 0, 2 * (size_t)kMaxParts * sizeof(double), c->stream)) != hipSuccess)
            return e;
    }
    // subdomains must agree on the damping: take the largest estimate
    if (c->comm.kind != Comm::NONE && c->comm.nranks > 1) {
        const int R = c->comm.nranks;
        std::vector<double> buf((size_t)3 * R, 0.0);
        buf[c->comm.rank] = lam;
        buf[R + c->comm.rank] = gersh;
        buf[2 * R + c->comm.rank] = lanczos;
        double* d = c->d_part + (size_t)P_AUX * kMaxParts;
        hipError_t e = hipMemcpyAsync(d, buf.data(), 3 * R * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return e;
        if ((e = allreduce_buffer(c, d, d, (size_t)3 * R)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(buf.data(), d, 3 * R * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
        if ((e = wait_stream(c)) != hipSuccess) return e;
        for (int r = 0; r < R; ++r) {
            lam = std::max(lam, buf[r]);
            gersh = std::max(gersh, buf[R + r]);
            lanczos = std::max(lanczos, buf[2 * R + r]);
        }
        if ((e = hipMemsetAsync(d, 0, 3 * R * sizeof(double), c->stream)) != hipSuccess) return e;
    }
    H.lambda = lam > 0.0 ? 1.1 * lam : 2.0 / 0.7;   // no estimate: fall back to w = 0.7 ... 
    H.gersh = gersh;
    // spectral bound the caps use: the Lanczos value + 5 % (it converges from below), never above Gershgorin's
    H.lam_max = lanczos > 0.0 ? std::min(gersh > 0.0 ? gersh : 1e300, 1.05 * lanczos) : gersh;
    damping_caps(H);
    if (tunables().debug)
        fprintf(stderr, "[shk] multigrid smoother: 16-step power estimate %.4f, Lanczos(%d) %.4f, Gershgorin bound %.4f, "
                        "damping caps %.3f (2 sweeps) %.3f (4 sweeps)\n", lam, kLanczosSteps, lanczos, gersh, H.cap2, H.cap4);
    return hipSuccess;
}

// Refresh the coarse operators from the Jacobian just assembled (d_vals, d_dinv).
// A context's own hierarchies sit on the float copy of its Jacobian.
static void bind_top_to_jacobian(Ctx* c, AmgHierarchy& H) {
    H.topA = c->sell32();
    H.top_vals = c->d_vals32;
    H.top_dinv = c->d_dinv32;
    H.top_bytes = amg_sell_bytes(c->slots, c->slots16, c->plan.A.nslice, c->d_pk != nullptr);
}

hipError_t amg_numeric_setup(Ctx* c, AmgHierarchy& H, bool refresh_dense, bool decided, bool top_only) {
    const bool primary = &H == &c->amg_local || &H == &c->amg_dist;
    if (primary) bind_top_to_jacobian(c, H);
    // top_only: a later Newton iteration whose iterate has barely moved (shk_newton_solve decides) keeps the coarse
    // operators and A*P of the step's first system and renews only the finest level's float copy: the cycle stays a
    // fixed linear operator, marginally staler (10M rows: +4..9 Krylov iterations in 550, -1.4 ms per step).
    const bool reuse = tunables().amg_reuse;
    if (top_only && reuse && primary && !refresh_dense && !decided && H.lambda != 0.0) {
        PhaseTimer t(c, SHK_PH_OTHER);
        note_bytes(c, 12.0 * (double)c->slots + 12.0 * (double)c->n_own);
        hipLaunchKernelGGL(k_narrow, dim3(c->grid), dim3(kBlock), 0, c->stream, c->slots, c->d_vals, c->d_vals32);
        hipLaunchKernelGGL(k_narrow, dim3(small_grid(c->n_own)), dim3(kBlock), 0, c->stream, c->n_own, c->d_dinv, c->d_dinv32);
        launch_pack(c, H.topA, c->d_vals32, c->slots16);
        return hipSuccess;
    }
    // a large dense coarsest inverse (2 launches per pivot) is only rebuilt when asked to: between the Newton
    // iterations of one time step the coarsest operator barely moves, and a slightly stale inverse only makes
    // the (fixed, linear) preconditioner marginally weaker
    // ... and across time steps: a big inverse (> 512 rows) is rebuilt every `dense_period`-th request, or at
    // once when the last solve needed 25 % more iterations than the first one after the previous rebuild.
    // (With a replicated coarse part the dense level lives in `rep`; the iteration feedback stays with H.)
    if (!decided) {
        AmgHierarchy& D = H.rep ? *H.rep : H;
        const AmgXfer& XD = D.xf.back();
        const int nd = D.distributed ? D.n_glob : XD.n_coarse;
        if (refresh_dense && D.dense_valid && nd > 512) {
            const bool degraded = H.its_fresh > 0 && H.its_last > 1.25 * H.its_fresh;
            if (++D.dense_age < D.dense_period && !degraded) refresh_dense = false;
            if (degraded) H.lambda_age = INT_MAX / 2;   // ... and the spectral estimates with it
        }
        if (!D.dense_valid) refresh_dense = true;
        if (refresh_dense) { D.dense_age = 0; H.its_fresh = 0; }
    }
    PhaseTimer t(c, SHK_PH_OTHER);
    if (primary) {
        note_bytes(c, 12.0 * (double)c->slots + 12.0 * (double)c->n_own);
        hipLaunchKernelGGL(k_narrow, dim3(c->grid), dim3(kBlock), 0, c->stream, c->slots, c->d_vals, c->d_vals32);
        hipLaunchKernelGGL(k_narrow, dim3(small_grid(c->n_own)), dim3(kBlock), 0, c->stream, c->n_own, c->d_dinv, c->d_dinv32);
        launch_pack(c, H.topA, c->d_vals32, c->slots16);
    }
    const float* fine = H.top_vals;
    for (size_t l = 0; l < H.xf.size(); ++l) {
        const AmgXfer& X = H.xf[l];
        // (every finer value is read once by each of the two gather plans: 8 B per list entry)
        note_bytes(c, 8.0 * (double)(X.n_glist + (X.with_ap ? X.ap_n_glist : 0)));
        if (X.with_ap)
        {
            launch_galerkin<float>(c, 1, X.ap_slots, X.ap_gptr, X.ap_glist, fine, X.ap_vals);
            launch_pack(c, ap_sell(X), X.ap_vals, X.ap_slots16, X.agg, X.onto_global ? H.rep_row0 : 0);
        }
        if (X.onto_global) {
            // my rows of the replicated global level -- whole SELL slices, since every subdomain's block is a multiple of
            // 1024 rows: the slots [rep_val_off[me], rep_val_off[me + 1]) of its value array -- written in place (float,
            // summed in double like every Galerkin product); ONE in-place all-gather brings in the other subdomains'
            // slices, then the replicated hierarchy refreshes itself from it, identically on every subdomain.
            // (Rounds 1-2 all-reduced the whole array in double, zero outside each subdomain's slices: 2 P times the bytes.)
            AmgHierarchy& R = *H.rep;
            const int me = c->comm.rank;
            const int64_t s0 = H.rep_val_off[me] / (int64_t)sizeof(float), s1 = H.rep_val_off[me + 1] / (int64_t)sizeof(float);
            launch_galerkin<float>(c, 0, s1 - s0, X.gptr + s0, X.glist, fine, R.t_vals + s0);
            hipError_t e = allgather_blocks(c, R.t_vals, H.rep_val_off);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_diag_inv, dim3(small_grid(H.rep_n)), dim3(kBlock), 0, c->stream, H.rep_n, R.t_diag,
                               (const float*)R.t_vals, R.t_dinv);
            if ((e = amg_numeric_setup(c, R, refresh_dense, true)) != hipSuccess) return e;
        } else if (X.dense) {
            const int64_t ns = (int64_t)X.n_coarse * X.n_coarse_cols;  // my rows of the coarsest operator
            if (H.distributed) {
                const size_t all = (size_t)H.n_glob * H.n_glob;
                hipError_t e = hipMemsetAsync(H.cdense, 0, all * sizeof(double), c->stream);
                if (e != hipSuccess) return e;
                launch_galerkin<double>(c, 0, ns, X.gptr,
                                   X.glist, fine, H.cdense + (size_t)H.offset * H.n_glob);
                if ((e = allreduce_buffer(c, H.cdense, H.cdense, all)) != hipSuccess) return e;
                if (refresh_dense || H.n_glob <= 64) dense_invert_big(c, H.n_glob, H.cdense, H.cinv, H.gj);
                H.dense_valid = true;
            } else {
                launch_galerkin<double>(c, 0, ns, X.gptr,
                                   X.glist, fine, H.cdense);
                if (X.n_coarse <= 64)
                    hipLaunchKernelGGL(k_dense_invert, dim3(1), dim3(kBlock), 0, c->stream, X.n_coarse, H.cdense, H.cinv);
                else if (refresh_dense) {
                    dense_invert_big(c, X.n_coarse, H.cdense, H.cinv, H.gj);
                    if (H.cinv32)   // the copy the cycle applies
                        hipLaunchKernelGGL(k_narrow, dim3(small_grid(ns)), dim3(kBlock), 0, c->stream, ns, (const double*)H.cinv, H.cinv32);
                }
                H.dense_valid = true;
            }
        } else {
            AmgLevel& L = H.lv[l + 1];
            launch_galerkin<float>(c, 0, L.slots, X.gptr, X.glist, fine, L.vals);
            launch_pack(c, level_sell(c, H, l + 1), L.vals, L.slots16);
            hipLaunchKernelGGL(k_diag_inv, dim3(small_grid(L.n)), dim3(kBlock), 0, c->stream, L.n, L.diag_slot, L.vals,
                               L.dinv);
            fine = L.vals;
        }
    }
    // Spectral estimates (power steps, Lanczos, Gershgorin on every sparse level: ~22 ms at 10M rows, most of it the
    // host round trips of the Lanczos recurrences) belong to the slowest-moving part of the setup: lambda_max(D^-1 A)
    // depends on the mesh and on the anisotropy of the coefficients, not on their size.  They are renewed with every
    // lambda_period-th refresh of the dense inverse (4: every 32nd step), at once when the iteration feedback reports
    // a degraded preconditioner, and with every refresh on hierarchies too small to be worth the bookkeeping.
    const int lambda_period = tunables().amg_lambda_period;
    const bool small = H.topA.n_rows < 200000;
    if (H.lambda == 0.0 || (refresh_dense && (small || ++H.lambda_age >= lambda_period))) {
        H.lambda_age = 0;
        return estimate_lambda(c, H);
    }
    return hipSuccess;
}

static DevSell level_sell(const Ctx* c, const AmgHierarchy& H, size_t l) {
    if (l == 0) return H.topA;
    const AmgLevel& L = H.lv[l];
    return DevSell{L.n, L.n_cols, L.nslice, sell_fits_cache(L.slots, L.pk ? kAmgPackedSlotBytes : kAmgSlotBytes), L.ptr, L.col,
                   L.rowlen, L.cbase, L.ptr16, L.col16, L.pk};
}
// A*P of a transfer: fine rows x coarse columns
static DevSell ap_sell(const AmgXfer& X) {
    return DevSell{X.n_fine, X.n_coarse_cols, X.ap_nslice, sell_fits_cache(X.ap_slots, X.ap_pk ? kAmgPackedSlotBytes : kAmgSlotBytes),
                   X.ap_ptr, X.ap_col, X.ap_rowlen, X.ap_cbase, X.ap_ptr16, X.ap_col16, X.ap_pk};
}

// first level handled by the single-workgroup tail (levels that exchange ghosts never are)
static size_t tail_start(const AmgHierarchy& H) {
    const size_t nx = H.xf.size();
    size_t lt = nx;
    while (lt > 1 && H.lv[lt - 1].n <= kTailRows && !(H.distributed && (int)(lt - 1) < H.halo_levels) &&
           nx - (lt - 1) <= (size_t)kTailMaxLevels)
        --lt;
    return lt;
}

template <class TR>
__global__ __launch_bounds__(kBlock) void k_amg_restrict4(int32_t n, const TR* __restrict__ r, const RestrictArgs ra,
                                                          const int* __restrict__ done) {
    __shared__ double rbuf0[kBlock];
    __shared__ float rbuf[kFusedRestrict - 1][256];
    if (*done) return;
    const int ngroups = (n + kBlock - 1) / kBlock;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int i = g * kBlock + threadIdx.x;
        fused_restrict(ra, g, i < n ? (double)r[i] : 0.0, rbuf0, rbuf);
    }
}

// Tables of the four-level restriction for the active hierarchy (nlev = 0: unavailable).  It starts at the first level
// `*first` of at most 2^21 rows (a workgroup then has few 256-row groups to walk through, each a memory round trip
// and four barriers: at 10M rows the finest level's own restriction is faster as a plain launch, and level 2 of 625k
// rows heads the cascade; at 1M rows the finest level does) and covers up to four transfers, down to `lt`.
static RestrictArgs amg_restrict_args(const AmgHierarchy& H, size_t lt, size_t* first) {
    RestrictArgs ra{};
    *first = 0;
    if (!tunables().fused_restrict) return ra;
    size_t ls = 0;
    while (ls < lt && (ls == 0 ? H.topA.n_rows : H.lv[ls].n) > (1 << 21)) ++ls;
    for (size_t j = 0; j < (size_t)kFusedRestrict && ls + j < lt; ++j) {
        const AmgXfer& X = H.xf[ls + j];
        if (!X.members_kd || !X.kd_pos) break;
        ra.members[j] = reinterpret_cast<const int4*>(X.members_kd);
        ra.pos[j] = X.kd_pos;
        ra.rc[j] = X.dense ? H.cr : X.onto_global ? H.rep_rglob + H.rep_row0 : H.lv[ls + j + 1].r;
        ra.nc[j] = X.n_coarse;
        ra.nlev = (int)j + 1;
    }
    *first = ls;
    return ra;
}

template <bool FINE, class TX, class TR, class TO>
static void launch_post(Ctx* c, const DevSell& A, const float* vals, const float* dinv, const TR* r, const TX* x, TO* xo,
                        float w, const int* done, int phase = SHK_PH_AMG_FINE, double post_bytes = 0.0) {   // post_bytes: the operator's stream
    AmgSmoothArgs<TX, TR, TO> a{A, vals, dinv, r, x, xo, w, done};
    const dim3 g(std::min((A.nslice + 3) / 4, 2048));
    note_bytes(c, post_bytes + (double)A.n_rows * (sizeof(TX) + sizeof(TR) + 4 + sizeof(TO)));
    if (FINE) {
        if (A.pk) launch_phase(c, phase, k_amg_post<FINE, TX, TR, TO, 0, true>, g, dim3(kBlock), 0, a);
        else launch_phase(c, phase, k_amg_post<FINE, TX, TR, TO>, g, dim3(kBlock), 0, a);
    } else {
        if (A.pk) hipLaunchKernelGGL((k_amg_post<FINE, TX, TR, TO, 0, true>), g, dim3(kBlock), 0, c->stream, a);
        else hipLaunchKernelGGL((k_amg_post<FINE, TX, TR, TO>), g, dim3(kBlock), 0, c->stream, a);
    }
}

// The finest level's sweep on a decomposed mesh, overlapped with the ghost update of its input (Ctx::overlap): the
// slices without ghost columns are swept while the exchange travels on comm_stream, the others after it has arrived.
template <class TR>
static hipError_t launch_post_split(Ctx* c, const DevSell& A, const float* vals, const float* dinv, const TR* r,
                                    float* x, float* xo, float w, const int* done) {
    AmgSmoothArgs<float, TR, float> a{A, vals, dinv, r, x, xo, w, done,
                                          SplitSell{c->d_slice_ghost, c->d_bslices, c->n_bslices}};
    hipError_t e;
    if ((e = hipEventRecord(c->ev_ready, c->stream)) != hipSuccess) return e;
    if (A.pk) launch_phase(c, SHK_PH_AMG_FINE, k_amg_post<true, float, TR, float, 1, true>, dim3(std::min((A.nslice + 3) / 4, 2048)),
                           dim3(kBlock), 0, a);
    else launch_phase(c, SHK_PH_AMG_FINE, k_amg_post<true, float, TR, float, 1>, dim3(std::min((A.nslice + 3) / 4, 2048)),
                      dim3(kBlock), 0, a);
    if ((e = halo_begin_f32(c, x)) != hipSuccess) return e;
    if ((e = halo_end(c)) != hipSuccess) return e;
    if (c->n_bslices > 0) {
        PhaseTimer t(c, SHK_PH_HALO);
        if (A.pk) hipLaunchKernelGGL((k_amg_post<true, float, TR, float, 2, true>), dim3(std::min((c->n_bslices + 3) / 4, 2048)),
                                     dim3(kBlock), 0, c->stream, a);
        else hipLaunchKernelGGL((k_amg_post<true, float, TR, float, 2>), dim3(std::min((c->n_bslices + 3) / 4, 2048)),
                                dim3(kBlock), 0, c->stream, a);
    }
    return hipSuccess;
}

// The four sweeps of one level in one launch (k_amg_sweeps), if the level has a plan.
template <class TR>
static bool launch_sweeps(Ctx* c, const DevSweepPlan& S, const DevSell& A, const float* vals, const float* dinv, const TR* r,
                          const AmgXfer& X, const float* e_cols, int32_t agg_off, bool frozen, float* xo, const float (&w)[4],
                          float alpha, const int* done, double a_bytes) {
    if (!S.ready()) return false;
    AmgSweepArgs<TR> a{};
    a.A = A; a.vals = vals; a.dinv = dinv; a.r = r;
    a.AP = DevSell{X.n_fine, X.n_coarse_cols, X.ap_nslice, 1, X.ap_ptr, X.ap_col, X.ap_rowlen, X.ap_cbase, X.ap_ptr16, X.ap_col16};
    a.ap_vals = X.ap_vals; a.e = e_cols; a.agg = X.agg; a.agg_off = agg_off;
    a.ghost_col = frozen ? X.ghost_col : nullptr; a.n_ghost = frozen ? X.n_ghost : 0;
    a.xo = xo;
    for (int k = 0; k < 4; ++k) a.w[k] = w[k];
    a.alpha = alpha; a.done = done;
    a.nblk = S.nblk; a.width = S.width; a.hdr = S.hdr; a.ext_info = S.ext_info; a.lcol_own = S.lcol_own; a.ring_lcol = S.ring_lcol;
    // the operator and A*P streams (the latter with 32-bit columns), the plan arrays, r / 1/diag / agg / result per row
    note_bytes(c, a_bytes + 8.0 * (double)X.ap_slots + S.plan_bytes + 16.0 * (double)A.n_rows + 4.0 * (double)X.n_coarse_cols);
    const dim3 g(std::min(S.nblk, 8192));
    if (S.width <= 12) hipLaunchKernelGGL((k_amg_sweeps<TR, 12>), g, dim3(kSweepRows), 0, c->stream, a);
    else hipLaunchKernelGGL((k_amg_sweeps<TR, kSweepMaxWidth>), g, dim3(kSweepRows), 0, c->stream, a);
    return true;
}

// z = M^-1 r : one V(0,2) cycle.  r and z have the fine level's length; r is not modified.  TR: double for a context's
// own hierarchies (the Krylov vector), float for the replicated hierarchy (the gathered right-hand side).
template <class TR>
static hipError_t amg_vcycle_t(Ctx* c, AmgHierarchy& H, const TR* rin, float* zout, const double* rin_last = nullptr) {
    const int32_t n_top = H.topA.n_rows, ncol_top = H.topA.n_cols;
    const size_t nx = H.xf.size();  // levels 0..nx-1 are sparse, level nx is the dense coarsest
    const int* done = &c->d_state->done;
    const double fw1 = tunables().amg_w1, fw2 = tunables().amg_w2;   // experiment overrides
    const float w1 = (float)(fw1 > 0.0 ? fw1 : H.cap2 * H.c1 / H.lambda), w2 = (float)(fw2 > 0.0 ? fw2 : H.cap2 * H.c2 / H.lambda);
    const double l4 = H.lambda / H.cap4;   // the four-sweep sequences divide their c4[k] by this
    const float alpha = (float)H.alpha;
    hipError_t e;
    const size_t lt = H.rep ? nx : tail_start(H);   // a replicated coarse part takes over after the last launch level
    // accounting: every launch of a replicated hierarchy is SHK_PH_AMG_REP (the part of a cycle that does not shrink
    // with the number of GPUs); a context's own hierarchy splits into finest level, restrictions, dense solve and one
    // slot per coarse level
    const bool is_rep = H.top_four;
    auto ph = [&](int p) { return is_rep ? (int)SHK_PH_AMG_REP : p; };
    auto ph_level = [&](size_t l) { return is_rep ? (int)SHK_PH_AMG_REP : (int)SHK_PH_AMG_L1 + (int)std::min<size_t>(l, 8) - 1; };
    if (ncol_top > n_top && !(H.distributed && H.halo_levels > 0)) {
        // block-local smoothing on the finest level: the output vector's ghost entries (left over from the Krylov
        // loop's own exchange) must read as zero, or the preconditioner would change from call to call
        if ((e = hipMemsetAsync(zout + n_top, 0, (size_t)(ncol_top - n_top) * sizeof(float), c->stream)) != hipSuccess)
            return e;
    }
    {
        PhaseTimer t(c, ph(SHK_PH_AMG_RESTRICT));
        size_t ls = 0;
        const RestrictArgs ra = amg_restrict_args(H, lt, &ls);
        // one launch for up to four levels from level `ls` on (measured at 1M rows, from the finest level: 46.9 ->
        // 45.4 ms/step; at 10M rows from level 2: four launches of ~4.4 us become one), plain launches around it
        for (size_t l = 0; l < lt; ++l) {
            if (ra.nlev > 1 && l == ls) {
                const int32_t nl = l == 0 ? n_top : H.lv[l].n;
                double rb = (double)nl * (l == 0 ? sizeof(TR) : 4);
                for (int j = 0; j < ra.nlev; ++j) rb += 24.0 * (double)ra.nc[j];   // members, position, output
                note_bytes(c, rb);
                const dim3 gf(std::min((nl + kBlock - 1) / kBlock, 2048));
                if (l == 0) hipLaunchKernelGGL(k_amg_restrict4<TR>, gf, dim3(kBlock), 0, c->stream, nl, rin, ra, done);
                else hipLaunchKernelGGL(k_amg_restrict4<float>, gf, dim3(kBlock), 0, c->stream, nl, (const float*)H.lv[l].r, ra, done);
                l += (size_t)ra.nlev - 1;
                continue;
            }
            const AmgXfer& X = H.xf[l];
            float* rc = X.dense ? H.cr : X.onto_global ? H.rep_rglob + H.rep_row0 : H.lv[l + 1].r;
            const dim3 g(small_grid(X.n_coarse));
            note_bytes(c, (double)X.n_fine * (l == 0 ? sizeof(TR) : 4) + 20.0 * (double)X.n_coarse);
            if (l == 0)
                hipLaunchKernelGGL(k_amg_restrict<TR>, g, dim3(kBlock), 0, c->stream, X.n_coarse, X.members, rin, rc, done);
            else
                hipLaunchKernelGGL(k_amg_restrict<float>, g, dim3(kBlock), 0, c->stream, X.n_coarse, X.members,
                                   (const float*)H.lv[l].r, rc, done);
        }
    }
    const AmgXfer& XL = H.xf[nx - 1];
    if (H.rep) {
        // the restriction has written my block of the replicated level's right-hand side in place; one in-place
        // all-gather of floats brings in the others', and the rest of the cycle runs redundantly on every subdomain
        if ((e = allgather_blocks(c, H.rep_rglob, H.rep_rhs_off)) != hipSuccess) return e;
        if ((e = amg_vcycle_t<float>(c, *H.rep, H.rep_rglob, H.rep_xglob)) != hipSuccess) return e;
    }
    TailArgs ta;
    ta.nlev = (int)(nx - lt);
    for (size_t l = lt; l < nx; ++l) {
        const AmgLevel& L = H.lv[l];
        const AmgXfer& X = H.xf[l];
        ta.lv[l - lt] = TailLevel{L.n, L.nslice, X.n_coarse, L.ptr, L.col, X.agg, X.members, L.vals, L.dinv, L.x, L.x2, L.r};
    }
    ta.n_c = XL.n_coarse; ta.row0 = H.distributed ? H.offset : 0; ta.ncols = H.distributed ? H.n_glob : XL.n_coarse;
    ta.inv = H.cinv; ta.cr = H.cr; ta.cx = H.cx; ta.cglob = H.distributed ? H.cglob : nullptr;
    ta.omega = w1; ta.omega2 = w2; ta.alpha = alpha; ta.done = done;
    ta.dense_in_tail = ta.ncols <= 128 ? 1 : 0;
    const int gemv_grid = std::min(2048, (ta.n_c + 3) / 4);
    auto bottom = [&]() -> hipError_t {
        if (H.rep) {
            // (done above)
        } else if (H.distributed) {
            {
                PhaseTimer t(c, SHK_PH_AMG_COARSE);
                if (ta.nlev > 0) hipLaunchKernelGGL(k_amg_tail<1>, dim3(1), dim3(kTailThreads), 0, c->stream, ta);
                hipLaunchKernelGGL(k_coarse_scatter, dim3(1), dim3(kBlock), 0, c->stream, XL.n_coarse, H.offset, H.n_glob,
                                   H.cr, H.cglob, done);
            }
            if ((e = allreduce_buffer(c, H.cglob, H.cglob, (size_t)H.n_glob)) != hipSuccess) return e;
            PhaseTimer t(c, SHK_PH_AMG_DENSE);
            if (!ta.dense_in_tail)
                hipLaunchKernelGGL((k_dense_gemv<double, double>), dim3(gemv_grid), dim3(kBlock), 0, c->stream, ta.n_c, ta.row0,
                                   ta.ncols, ta.inv, (const double*)H.cglob, ta.cx, done);
            if (ta.nlev > 0 || ta.dense_in_tail) hipLaunchKernelGGL(k_amg_tail<2>, dim3(1), dim3(kTailThreads), 0, c->stream, ta);
        } else if (ta.dense_in_tail) {
            PhaseTimer t(c, ph(SHK_PH_AMG_COARSE));
            hipLaunchKernelGGL(k_amg_tail<0>, dim3(1), dim3(kTailThreads), 0, c->stream, ta);
        } else {
            if (ta.nlev > 0) {
                PhaseTimer t(c, ph(SHK_PH_AMG_COARSE));
                hipLaunchKernelGGL(k_amg_tail<1>, dim3(1), dim3(kTailThreads), 0, c->stream, ta);
            }
            {
                PhaseTimer t(c, ph(SHK_PH_AMG_DENSE));
                note_bytes(c, (H.cinv32 ? 4.0 : 8.0) * (double)ta.n_c * ta.ncols + 4.0 * (double)(ta.n_c + ta.ncols));
                if (H.cinv32)
                    hipLaunchKernelGGL((k_dense_gemv<float, float>), dim3(gemv_grid), dim3(kBlock), 0, c->stream, ta.n_c, ta.row0,
                                       ta.ncols, (const float*)H.cinv32, (const float*)H.cr, ta.cx, done);
                else
                    hipLaunchKernelGGL((k_dense_gemv<float, double>), dim3(gemv_grid), dim3(kBlock), 0, c->stream, ta.n_c, ta.row0,
                                       ta.ncols, ta.inv, (const float*)H.cr, ta.cx, done);
            }
            if (ta.nlev > 0) {
                PhaseTimer t(c, ph(SHK_PH_AMG_COARSE));
                hipLaunchKernelGGL(k_amg_tail<2>, dim3(1), dim3(kTailThreads), 0, c->stream, ta);
            }
        }
        return hipSuccess;
    };
    if ((e = bottom()) != hipSuccess) return e;
    auto up_level = [&](size_t l) -> hipError_t {
        const AmgXfer& X = H.xf[l];
        const float* ec = X.dense ? H.cx : X.onto_global ? H.rep_xglob + H.rep_row0 : H.lv[l + 1].x2;
        const bool halo = H.distributed && (int)l < H.halo_levels;
        const HaloPlan* HP = halo ? &c->comm.plans[H.plan_of[l]] : nullptr;
        // first sweep on A*P: its columns are the coarser level's columns, so a decomposed coarser level must
        // exchange its result first (a smaller message than exchanging the prolongated vector); the replicated
        // level's result is complete on every subdomain
        // (towards a SHARED dense level of a decomposed hierarchy the solve leaves only this subdomain's rows behind;
        //  a hierarchy's own dense level is an ordinary coarse vector)
        const bool fused = X.with_ap && (H.distributed ? !X.dense && (X.onto_global || (int)(l + 1) < H.halo_levels) : true);
        // frozen-ghost smoothing (AmgHierarchy::frozen_ghosts): the first sweep also writes this level's ghost columns,
        // alpha * e[coarse column of the ghost], and no sweep of the level exchanges
        const bool frozen = halo && fused && ((H.frozen_mask >> std::min<size_t>(l, 30)) & 1u) && X.ghost_col != nullptr;
        const float* e_cols = X.onto_global ? H.rep_xglob : ec;
        const int32_t agg_off = X.onto_global ? H.rep_row0 : 0;
        // (a frozen coarser level whose bit of e_exchange_mask is clear keeps the ghosts its first sweep wrote -- the
        //  prolongated correction of the level below it, without that level's smoothing -- and sends nothing)
        const bool coarser_frozen = l + 1 < nx && ((H.frozen_mask >> std::min<size_t>(l + 1, 30)) & 1u) && H.xf[l + 1].ghost_col != nullptr;
        const bool send_e = !coarser_frozen || ((H.e_exchange_mask >> std::min<size_t>(l + 1, 30)) & 1u);
        if (fused && H.distributed && !X.onto_global && send_e &&
            (e = halo_exchange_plan_f32(c, c->comm.plans[H.plan_of[l + 1]], H.lv[l + 1].x2)) != hipSuccess)
            return e;
        const DevSell A = level_sell(c, H, l);
        const dim3 g(std::min((A.nslice + 3) / 4, 2048));
        // byte accounting (shk_profile.bytes): the streams of this level's operator and of its A*P
        const double a_bytes = l == 0 ? H.top_bytes : amg_sell_bytes(H.lv[l].slots, H.lv[l].slots16, H.lv[l].nslice, H.lv[l].pk != nullptr);
        const double ap_bytes = X.with_ap ? amg_sell_bytes(X.ap_slots, X.ap_slots16, X.ap_nslice, X.ap_pk != nullptr) : 0.0;
        const double first_bytes = ap_bytes + (double)X.n_fine * (4 + (l == 0 ? sizeof(TR) : 4) + 4 + 4) + 4.0 * (double)X.n_coarse_cols;
        if (l == 0) {
            // level 0: right-hand side = the Krylov vector (double); the iterate and the result are float.
            // A replicated hierarchy's top level is a coarse level of the whole cycle: four sweeps, like its peers.
            bool split_done = false;
            const float w2_fine = w2;
            const bool four = H.top_four && H.coarse4;
            const float omega = four ? (float)(H.c4[0] / l4) : w1;
            const float w2 = four ? (float)(H.c4[1] / l4) : w2_fine;
            const float w4[4] = {(float)(H.c4[0] / l4), (float)(H.c4[1] / l4), (float)(H.c4[2] / l4), (float)(H.c4[3] / l4)};
            if (four && fused && !halo && l < H.sw.size()) {
                PhaseTimer t(c, ph(SHK_PH_AMG_FINE));
                if (launch_sweeps<TR>(c, H.sw[l], A, H.top_vals, H.top_dinv, rin, X, e_cols, agg_off, false, zout, w4, alpha, done, a_bytes))
                    return hipSuccess;
            }
            if (fused) {
                AmgFirstArgs<TR> f{ap_sell(X), X.ap_vals, H.top_dinv, rin, e_cols, X.agg, agg_off, H.x0, omega, alpha, done,
                                   frozen ? X.ghost_col : nullptr, frozen ? X.n_ghost : 0, nullptr};
                note_bytes(c, first_bytes);
                if (frozen) {
                    if (f.AP.pk) launch_phase(c, ph(SHK_PH_AMG_FIRST), k_amg_first<true, TR, true, true>, g, dim3(kBlock), 0, f);
                    else launch_phase(c, ph(SHK_PH_AMG_FIRST), k_amg_first<true, TR, true>, g, dim3(kBlock), 0, f);
                } else {
                    if (f.AP.pk) launch_phase(c, ph(SHK_PH_AMG_FIRST), k_amg_first<true, TR, false, true>, g, dim3(kBlock), 0, f);
                    else launch_phase(c, ph(SHK_PH_AMG_FIRST), k_amg_first<true, TR>, g, dim3(kBlock), 0, f);
                }
            } else {
                {
                    PhaseTimer t(c, ph(SHK_PH_AMG_COARSE));
                    hipLaunchKernelGGL(k_amg_prolong<float>, dim3(small_grid(X.n_fine)), dim3(kBlock), 0, c->stream,
                                       X.n_fine, alpha, X.agg, ec, zout, done);
                }
                if (halo && (e = halo_exchange_plan_f32(c, *HP, zout)) != hipSuccess) return e;
                launch_post<true>(c, A, H.top_vals, H.top_dinv, rin, (const float*)zout, H.x0, omega, done, ph(SHK_PH_AMG_FINE), a_bytes);
            }
            if (halo && !frozen && c->overlap && H.plan_of[0] == 0 && A.ptr == c->d_sell_ptr) {   // (a context's own level 0)
                if ((e = launch_post_split<TR>(c, A, H.top_vals, H.top_dinv, rin, H.x0, zout, w2, done)) != hipSuccess) return e;
                split_done = true;
            }
            if (!split_done) {
                if (halo && !frozen && (e = halo_exchange_plan_f32(c, *HP, H.x0)) != hipSuccess) return e;
                // (rin_last: the cycle's LAST sweep reads the double right-hand side instead of its float copy, so that the
                //  consumer that follows finds it warm in the Infinity Cache: amg_vcycle below)
                if (rin_last && !four) launch_post<true>(c, A, H.top_vals, H.top_dinv, rin_last, (const float*)H.x0, zout, w2, done, ph(SHK_PH_AMG_FINE), a_bytes);
                else launch_post<true>(c, A, H.top_vals, H.top_dinv, rin, (const float*)H.x0, zout, w2, done, ph(SHK_PH_AMG_FINE), a_bytes);
            }
            if (four) {
                launch_post<true>(c, A, H.top_vals, H.top_dinv, rin, (const float*)zout, H.x0, (float)(H.c4[2] / l4), done, ph(SHK_PH_AMG_FINE), a_bytes);
                launch_post<true>(c, A, H.top_vals, H.top_dinv, rin, (const float*)H.x0, zout, (float)(H.c4[3] / l4), done, ph(SHK_PH_AMG_FINE), a_bytes);
            }
        } else {
            const AmgLevel& L = H.lv[l];
            // coarse levels are cheap next to the finest one, and a better coarse solve pays: four sweeps with the
            // Chebyshev dampings H.c4 instead of two.  On a level that exchanges ghosts only the first two sweeps
            // do: the last two read the ghost values those exchanges left behind (lagged, still a fixed linear
            // operator) -- rehearsed with 4 subdomains at 1M rows: 57 -> 54 iterations per Newton iteration with
            // two exchanging levels, 51 -> 38 with all (one subdomain: 40), at no extra message.
            const bool more = H.coarse4 && (int)l >= H.coarse4_from;
            const float lw1 = more ? (float)(H.c4[0] / l4) : w1;
            const float lw2 = more ? (float)(H.c4[1] / l4) : w2;
            if (fused && more && (frozen || A.n_cols == A.n_rows) && l < H.sw.size()) {
                const float w4[4] = {lw1, lw2, (float)(H.c4[2] / l4), (float)(H.c4[3] / l4)};
                PhaseTimer t(c, ph_level(l));
                if (launch_sweeps<float>(c, H.sw[l], A, L.vals, L.dinv, (const float*)L.r, X, e_cols, agg_off, frozen, L.x2, w4,
                                         alpha, done, a_bytes))
                    return hipSuccess;
            }
            if (fused) {
                AmgFirstArgs<float> f{ap_sell(X), X.ap_vals, L.dinv, L.r, e_cols, X.agg, agg_off, L.x, lw1, alpha, done,
                                      frozen ? X.ghost_col : nullptr, frozen ? X.n_ghost : 0, L.x2};
                PhaseTimer t(c, ph_level(l));
                note_bytes(c, first_bytes);
                if (frozen) {
                    if (f.AP.pk) hipLaunchKernelGGL((k_amg_first<false, float, true, true>), g, dim3(kBlock), 0, c->stream, f);
                    else hipLaunchKernelGGL((k_amg_first<false, float, true>), g, dim3(kBlock), 0, c->stream, f);
                } else {
                    if (f.AP.pk) hipLaunchKernelGGL((k_amg_first<false, float, false, true>), g, dim3(kBlock), 0, c->stream, f);
                    else hipLaunchKernelGGL((k_amg_first<false, float>), g, dim3(kBlock), 0, c->stream, f);
                }
            } else {
                {
                    PhaseTimer t(c, ph_level(l));
                    hipLaunchKernelGGL(k_amg_prolong<float>, dim3(small_grid(X.n_fine)), dim3(kBlock), 0, c->stream,
                                       X.n_fine, alpha, X.agg, ec, L.x2, done);
                }
                // distributed smoothing: the sweep needs the neighbours' current iterate on the ghost columns
                // (without the exchange the ghost entries stay zero = block-local smoothing on that level)
                if (halo && (e = halo_exchange_plan_f32(c, *HP, L.x2)) != hipSuccess) return e;
                PhaseTimer t(c, ph_level(l));
                launch_post<false>(c, A, L.vals, L.dinv, (const float*)L.r, (const float*)L.x2, L.x, lw1, done, 0, a_bytes);
            }
            if (halo && !frozen && (e = halo_exchange_plan_f32(c, *HP, L.x)) != hipSuccess) return e;
            if (halo && !frozen && fused && more) {
                // The third sweep reads x2, whose ghost segment no exchange of THIS cycle has filled when the first
                // sweep ran on A*P (it would hold the previous cycle's values: the preconditioner would stop being a
                // function of its input, and BiCGStab breaks down).  Give it the ghosts just received for x.
                const int64_t ng = HP->recv_ptr.back();
                if (ng > 0 && (e = hipMemcpyAsync(L.x2 + HP->n_own, L.x + HP->n_own, (size_t)ng * sizeof(float),
                                                  hipMemcpyDeviceToDevice, c->stream)) != hipSuccess)
                    return e;
            }
            PhaseTimer t(c, ph_level(l));
            launch_post<false>(c, A, L.vals, L.dinv, (const float*)L.r, (const float*)L.x, L.x2, lw2, done, 0, a_bytes);
            if (more) {
                launch_post<false>(c, A, L.vals, L.dinv, (const float*)L.r, (const float*)L.x2, L.x, (float)(H.c4[2] / l4), done, 0, a_bytes);
                launch_post<false>(c, A, L.vals, L.dinv, (const float*)L.r, (const float*)L.x, L.x2, (float)(H.c4[3] / l4), done, 0, a_bytes);
            }
        }
        return hipSuccess;
    };
    // Cycle doubling at ONE level (a W-cycle's second visit, at the level where visits cost launch latency, not
    // bandwidth): after the level's first solution x1, the cycle below it runs again on r - A x1 and the two add up.
    // Plain aggregation with piecewise-constant prolongation needs more than a V-cycle as the hierarchy deepens; a full
    // W-cycle would visit the tiny levels 2^l times.  A CPU prototype at 1M rows (same aggregates, sweeps, dampings, but a
    // sparse coarsest level) promised 44 -> 31 iterations; on the real hierarchy, whose dense coarsest level of 977 /
    // 2441 rows already solves the deep part exactly, it is 39.5 -> 35.4 at 1M rows and 45.8 -> 44.1 at 10M for 44 % / 5 %
    // more time per step.  OFF by default (SHK_AMG_W_ROWS), single-context hierarchies only.
    const size_t lw = (!H.distributed && !H.rep) ? H.w_level : 0;
    for (size_t l = lt; l-- > 0;) {
        if ((e = up_level(l)) != hipSuccess) return e;
        if (l == lw && lw >= 1 && lw < lt) {
            AmgLevel& L = H.lv[l];
            const DevSell A = level_sell(c, H, l);
            {
                PhaseTimer t(c, SHK_PH_AMG_COARSE);
                AmgSmoothArgs<float, float, float> ra{A, L.vals, L.dinv, L.r, L.x2, L.x3, 0.0f, done};
                hipLaunchKernelGGL(k_amg_residual_save, dim3(std::min((A.nslice + 3) / 4, 2048)), dim3(kBlock), 0, c->stream, ra, L.r);
                for (size_t ll = l; ll < lt; ++ll) {
                    const AmgXfer& X = H.xf[ll];
                    float* rc = X.dense ? H.cr : H.lv[ll + 1].r;
                    hipLaunchKernelGGL(k_amg_restrict<float>, dim3(small_grid(X.n_coarse)), dim3(kBlock), 0, c->stream,
                                       X.n_coarse, X.members, (const float*)H.lv[ll].r, rc, done);
                }
            }
            if ((e = bottom()) != hipSuccess) return e;
            for (size_t ll = lt; ll-- > l;)
                if ((e = up_level(ll)) != hipSuccess) return e;
            PhaseTimer t(c, SHK_PH_AMG_COARSE);
            hipLaunchKernelGGL(k_amg_add, dim3(small_grid(L.n)), dim3(kBlock), 0, c->stream, L.n, (const float*)L.x3, L.x2, done);
        }
    }
    return hipSuccess;
}

hipError_t amg_vcycle(Ctx* c, AmgHierarchy& H, const double* rin, float* zout) { return amg_vcycle_t<double>(c, H, rin, zout); }
// The Krylov loop hands the cycle the float copies its vector kernels write beside p and s (Ctx::d_p32, d_s32): the cycle
// reads its right-hand side three times on the finest level (restriction, first sweep, second sweep) and rounds it to
// float on the way in either way.
// rin_last (optional): the same vector in double.  The cycle's last sweep then reads IT: 40 MB more for that sweep, but the
// Krylov product that follows the cycle on s needs s itself (the dot products t.s, rhat.t are taken in double) and found it
// cold once the cycle stopped touching it: k_spmv<2> 150 -> 198 us at 10M rows beside k_spmv<1>'s 176.
hipError_t amg_vcycle(Ctx* c, AmgHierarchy& H, const float* rin, float* zout, const double* rin_last) {
    return amg_vcycle_t<float>(c, H, rin, zout, rin_last);
}

// ------------------------------------------------------------------ distributed setup (collective)
// One integer per row of a level travels as a double through that level's halo plan.
static hipError_t exchange_ids(Ctx* c, const HaloPlan& P, int64_t n_ghost, const std::vector<int32_t>& owned,
                               std::vector<int32_t>& ghost) {
    std::vector<double> buf((size_t)(P.n_own + n_ghost), -1.0);
    for (int64_t i = 0; i < P.n_own; ++i) buf[i] = (double)owned[i];
    hipError_t e = hipMemcpyAsync(c->d_io, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return e;
    if ((e = halo_exchange_plan(c, P, c->d_io)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(buf.data(), c->d_io, buf.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return e;
    if ((e = wait_stream(c)) != hipSuccess) return e;
    ghost.resize((size_t)n_ghost);
    for (int64_t g = 0; g < n_ghost; ++g) ghost[g] = (int32_t)buf[P.n_own + g];
    return hipSuccess;
}

static hipError_t allgather_int(Ctx* c, int32_t mine, std::vector<int32_t>& all) {
    const int R = c->comm.nranks;
    std::vector<double> buf((size_t)R, 0.0);
    buf[c->comm.rank] = (double)mine;
    double* d = c->d_io;
    hipError_t e = hipMemcpyAsync(d, buf.data(), R * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return e;
    if ((e = allreduce_buffer(c, d, d, (size_t)R)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(buf.data(), d, R * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
    if ((e = wait_stream(c)) != hipSuccess) return e;
    all.resize(R);
    for (int r = 0; r < R; ++r) all[r] = (int32_t)buf[r];
    return hipSuccess;
}

// Concatenation over subdomains of integer arrays (mine at [my_off, my_off + mine.size()) of `total`), through one
// all-reduce of a zero-padded double array on a temporary device buffer.  Setup-time only.
static hipError_t allgather_i32(Ctx* c, const std::vector<int32_t>& mine, int64_t my_off, int64_t total,
                                std::vector<int32_t>& all) {
    std::vector<double> buf((size_t)std::max<int64_t>(total, 1), 0.0);
    for (size_t i = 0; i < mine.size(); ++i) buf[(size_t)my_off + i] = (double)mine[i];
    double* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, buf.size() * sizeof(double));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(d, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = allreduce_buffer(c, d, d, buf.size());
    if (e == hipSuccess) e = hipMemcpyAsync(buf.data(), d, buf.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = wait_stream(c);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    all.resize((size_t)total);
    for (int64_t i = 0; i < total; ++i) all[(size_t)i] = (int32_t)buf[(size_t)i];
    return hipSuccess;
}

// Build the distributed hierarchy: aggregates are local (runs of 4 owned rows in k-d order), but every level
// keeps its ghost columns (the neighbours' aggregates), so the Galerkin operators are exactly those of the
// undecomposed matrix and a V-cycle with per-level ghost exchanges is the same operator as on one GPU.
// Every subdomain must call this at the same time (it exchanges aggregate ids level by level).
int amg_setup_distributed(Ctx* c, std::string& err) {
    Comm& m = c->comm;
    if (m.kind == Comm::NONE || m.plans.empty()) { err = "no communicator"; return -1; }
    if (c->plan.krank.size() != (size_t)c->n_own) { err = "k-d ranks unavailable"; return -1; }
    const int R = m.nranks, me = m.rank;
    if ((size_t)2 * c->n_loc < (size_t)R) { err = "too many ranks for the staging buffer"; return -1; }
    AmgHierarchy& H = c->amg_dist;
    H.distributed = true;
    // SHK_AMG_GHOST_EXCHANGE = bit mask of the decomposed levels that KEEP the exchange after their first sweep
    H.frozen_mask = tunables().amg_ghost_exchange >= 0 ? ~(uint32_t)tunables().amg_ghost_exchange : H.frozen_mask;
    if (tunables().amg_e_exchange >= 0) H.e_exchange_mask = (uint32_t)tunables().amg_e_exchange;
    if (tunables().amg_halo_levels >= 0) H.halo_levels = tunables().amg_halo_levels;
    // first level gathered and replicated on every subdomain: the first whose global size is at most this (0: never)
    // (default 200 000: level 3 of a 10M-row mesh, level 2 at 1M rows; the gathered right-hand side is then at most
    // 1.6 MB per cycle.  Rehearsed with 4 | 5 subdomains at 10M rows: 48 | 56 iterations per Newton iteration, the same
    // as with every level exchanging ghosts, against 76 | 86 with two exchanging levels and block-local ones below.)
    int64_t rep_rows = 200000;
    if (tunables().amg_rep_rows >= 0) rep_rows = tunables().amg_rep_rows;
    SellPattern G;                       // the replicated global level, if any
    std::vector<int32_t> G_diag;
    std::vector<AmgLevelPlan> plans;
    plans.reserve(40);
    const SellPattern* Af = &c->plan.A;
    int64_t n_own = c->n_own, n_ghost = c->n_loc - c->n_own;
    std::vector<int32_t> agg((size_t)n_own);
    for (int64_t i = 0; i < n_own; ++i) agg[i] = c->plan.krank[i] / 4;
    size_t cur_plan = 0;  // index into m.plans of the current level's halo plan
    // shared dense coarsest level: same cost rule as the local hierarchy (n^3 <= 250 nnz of the whole matrix,
    // estimated as R times this subdomain's), split evenly over the subdomains
    const int glob_dense = std::min(4096, std::max(64, (int)std::cbrt(250.0 * (double)c->nnz * R)));
    const int per_rank_dense = std::max(16, glob_dense / R);
    H.plan_of.clear();
    hipError_t e;
    for (int level = 0; level < 40; ++level) {
        std::vector<int32_t> all_n;
        if ((e = allgather_int(c, (int32_t)n_own, all_n)) != hipSuccess) { err = hipGetErrorString(e); return -1; }
        int32_t maxn = 0;
        for (int32_t v : all_n) maxn = std::max(maxn, v);
        if (maxn <= per_rank_dense) { err = "subdomains too small to coarsen"; return -1; }
        std::vector<int32_t> all_nc(R), offs(R + 1, 0);
        int32_t maxnc = 0;
        for (int r = 0; r < R; ++r) {
            all_nc[r] = (all_n[r] + 3) / 4;
            maxnc = std::max(maxnc, all_nc[r]);
            offs[r + 1] = offs[r] + all_nc[r];
        }
        const bool next_dense = maxnc <= per_rank_dense;
        if (next_dense && offs[R] > 4608) { err = "shared coarsest level too large (too many subdomains)"; return -1; }
        const bool next_rep = !next_dense && rep_rows > 0 && (int64_t)offs[R] <= rep_rows && offs[R] > 4096;
        const bool glob_cols = next_dense || next_rep;   // the next level's columns are global row ids
        if (next_rep) {
            // Every subdomain's block of the replicated level starts at a multiple of 4^5 (dummy identity rows fill
            // the gap): the replicated hierarchy aggregates runs of 4 GLOBAL rows level after level, and only an
            // aligned block keeps those runs inside the subdomain's own k-d cells (unaligned, every aggregate
            // straddles two cells and the cycle needs twice the iterations).
            // ... and all blocks have the SAME size (that of the largest subdomain's share; recursive bisection balances
            // the subdomains to +-1 vertex anyway): the right-hand side is then gathered by one ncclAllGather in place.
            const int32_t blk = ((maxnc + 1023) / 1024) * 1024;
            for (int r = 0; r < R; ++r) offs[r + 1] = offs[r] + blk;
        }
        const int32_t nc_own = all_nc[me];
        const HaloPlan& P = m.plans[cur_plan];
        std::vector<int32_t> gagg;
        if ((e = exchange_ids(c, P, n_ghost, agg, gagg)) != hipSuccess) { err = hipGetErrorString(e); return -1; }
        // coarse ghosts per neighbour = sorted unique aggregate ids of that neighbour's ghosts; coarse sends =
        // sorted unique aggregates of what I send it (the two sides derive the same ordered lists)
        HaloPlan CP;
        CP.n_own = nc_own;
        CP.nbr = P.nbr;
        CP.send_ptr.assign(1, 0);
        CP.recv_ptr.assign(1, 0);
        std::vector<int32_t> colmap((size_t)(n_own + n_ghost), -1);
        for (int64_t i = 0; i < n_own; ++i) colmap[i] = glob_cols ? offs[me] + agg[i] : agg[i];
        int64_t cghost = 0;
        for (size_t k = 0; k < P.nbr.size(); ++k) {
            std::vector<int32_t> ids(gagg.begin() + P.recv_ptr[k], gagg.begin() + P.recv_ptr[k + 1]);
            std::sort(ids.begin(), ids.end());
            ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
            for (int64_t g = P.recv_ptr[k]; g < P.recv_ptr[k + 1]; ++g) {
                if (gagg[g] < 0 || gagg[g] >= all_nc[P.nbr[k]]) { err = "inconsistent aggregate id from a neighbour"; return -1; }
                const int32_t pos = (int32_t)(std::lower_bound(ids.begin(), ids.end(), gagg[g]) - ids.begin());
                colmap[n_own + g] = glob_cols ? offs[P.nbr[k]] + gagg[g] : (int32_t)(nc_own + cghost + pos);
            }
            cghost += (int64_t)ids.size();
            CP.recv_ptr.push_back(cghost);
            std::vector<int32_t> sids;
            for (int64_t q = P.send_ptr[k]; q < P.send_ptr[k + 1]; ++q) sids.push_back(agg[P.h_send_idx[q]]);
            std::sort(sids.begin(), sids.end());
            sids.erase(std::unique(sids.begin(), sids.end()), sids.end());
            CP.h_send_idx.insert(CP.h_send_idx.end(), sids.begin(), sids.end());
            CP.send_ptr.push_back((int64_t)CP.h_send_idx.size());
        }
        plans.emplace_back();
        // first sweep on the A*P operator (like the single-GPU hierarchy), except towards a shared dense level, whose
        // solve leaves only this subdomain's rows of the correction behind
        plans.back().with_ap = !next_dense && Af->n_rows > kTailRows;
        const int32_t ncols = glob_cols ? offs[R] : (int32_t)(nc_own + cghost);
        std::string perr;
        if (next_rep) {
            // every subdomain contributes the rows it owns of the global level; all of them assemble its pattern
            const int32_t nblk = offs[me + 1] - offs[me];   // my block: nc_own real rows, then dummy identity rows
            std::vector<int32_t> rp, ci, len(nblk, 1), all_len, all_ci, all_nnz;
            perr = coarse_rows(*Af, agg, colmap, nc_own, offs[me], rp, ci);
            if (!perr.empty()) { err = perr; return -1; }
            for (int32_t I = 0; I < nc_own; ++I) len[I] = rp[I + 1] - rp[I];
            for (int32_t I = nc_own; I < nblk; ++I) ci.push_back(offs[me] + I);
            if ((e = allgather_int(c, (int32_t)ci.size(), all_nnz)) != hipSuccess) { err = hipGetErrorString(e); return -1; }
            int64_t nnz_off = 0, nnz_tot = 0;
            for (int r = 0; r < R; ++r) { if (r < me) nnz_off += all_nnz[r]; nnz_tot += all_nnz[r]; }
            if (nnz_tot > INT32_MAX) { err = "replicated level too large"; return -1; }
            if ((e = allgather_i32(c, len, offs[me], offs[R], all_len)) != hipSuccess ||
                (e = allgather_i32(c, ci, nnz_off, nnz_tot, all_ci)) != hipSuccess) { err = hipGetErrorString(e); return -1; }
            std::vector<int32_t> grp((size_t)offs[R] + 1, 0);
            for (int32_t I = 0; I < offs[R]; ++I) grp[I + 1] = grp[I] + all_len[I];
            if (grp[offs[R]] != (int32_t)nnz_tot) { err = "replicated level: inconsistent row lengths"; return -1; }
            // rows arrive with their diagonal first, in global ids; sell_from_csr wants the diagonal to be the row id
            perr = sell_from_csr(offs[R], offs[R], grp, all_ci, G, G_diag);
            if (perr.empty()) perr = coarsen_onto_global(*Af, agg, colmap, nc_own, offs[me], G, plans.back());
        } else {
            perr = coarsen(*Af, agg, colmap, nc_own, ncols, next_dense, plans.back());
        }
        if (!perr.empty()) { err = perr; return -1; }
        // coarse column of every ghost column of this level (frozen-ghost smoothing)
        plans.back().ghost_col.assign(colmap.begin() + n_own, colmap.end());
        H.plan_of.push_back((int)cur_plan);
        if (next_rep) {
            H.rep_row0 = offs[me];
            H.rep_n = offs[R];
            H.rep_rhs_off.resize(R + 1);
            H.rep_val_off.resize(R + 1);
            for (int r = 0; r <= R; ++r) {
                H.rep_rhs_off[r] = (int64_t)offs[r] * (int64_t)sizeof(float);
                H.rep_val_off[r] = (int64_t)G.ptr[offs[r] / kSlice] * (int64_t)sizeof(float);   // blocks are whole slices
            }
            break;
        }
        if (next_dense) {
            H.n_glob = offs[R];
            H.offset = offs[me];
            break;
        }
        // install the coarse level's halo plan on the device
        if (!CP.h_send_idx.empty()) {
            void* q = nullptr;
            if ((e = hipMalloc(&q, CP.h_send_idx.size() * sizeof(int32_t))) != hipSuccess) { err = hipGetErrorString(e); return -1; }
            c->allocs.push_back(q);
            CP.d_send_idx = reinterpret_cast<int32_t*>(q);
            if ((e = upload_sync(c, q, CP.h_send_idx.data(), CP.h_send_idx.size() * sizeof(int32_t))) != hipSuccess) {
                err = hipGetErrorString(e); return -1;
            }
        }
        m.plans.push_back(std::move(CP));
        cur_plan = m.plans.size() - 1;
        Af = &plans.back().Ac;
        n_own = nc_own;
        n_ghost = cghost;
        agg.resize((size_t)n_own);
        for (int64_t I = 0; I < n_own; ++I) agg[I] = (int32_t)(I / 4);
    }
    if ((e = amg_upload(c, plans, H, c->n_loc)) != hipSuccess) { err = hipGetErrorString(e); return -1; }
    if (H.rep_n > 0) {
        // the rest of the hierarchy: a single-GPU one on the global level, built identically by every subdomain
        // (aggregates = runs of 4 global rows: rows of one subdomain are consecutive and in its k-d order)
        std::vector<int32_t> ident((size_t)G.n_rows);
        for (int32_t i = 0; i < G.n_rows; ++i) ident[i] = i;
        PlanOptions opt;
        opt.amg_coarsest = tunables().amg_coarsest;
        opt.amg_cost_nnz = (double)c->nnz * R;   // the dense level is sized against the whole fine operator
        std::vector<AmgLevelPlan> rplans;
        std::string perr = build_amg_levels(G, ident, opt, rplans);
        if (!perr.empty() || rplans.empty()) { err = perr.empty() ? "replicated level too small for a hierarchy" : perr; return -1; }
        H.rep = new AmgHierarchy();
        H.rep->top_four = true;
        if ((e = amg_upload_rep_top(c, *H.rep, G, G_diag)) != hipSuccess ||
            (e = amg_upload(c, rplans, *H.rep, G.n_rows, &ident, &G)) != hipSuccess) { err = hipGetErrorString(e); return -1; }
    }
    if (wait_stream(c) != hipSuccess) { err = "synchronize"; return -1; }
    return 0;
}

}  // namespace shk
