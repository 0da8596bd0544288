// Device-side argument blocks and the context layout shared by the kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/shakti_hip.h"
#include "shk_plan.h"
#include "shk_tunables.h"

namespace shk {

constexpr int kBlock = 256;        // threads per workgroup: 4 waves of 64
constexpr int kMaxParts = 2048;    // upper bound of the reduction-partial arrays (= max grid: 8 workgroups per CU)
constexpr int kMaxQuad = 32;

// Constants of /root/reference/source/params.py:4-11 and derived products, passed by value
// (kernarg segment -> scalar loads).
struct DevParams {
    double g, rho_i, rho_w, nu, Lh, omega, n, A, b_min;
    double rwg;      // rho_w * g
    double c_m;      // 1/rho_i - 1/rho_w
    double kcoef;    // g / (12 nu)
    double om_nu;    // omega / nu
    double ri_rw;    // rho_i / rho_w
    double inv_rwg, inv_Lh, cm_Lh;   // 1 / (rho_w g), 1 / Lh, c_m / Lh: the assembly multiplies where the forms divide
    int n_is_3;
};

struct QuadArg {
    int nq;
    double phi0[kMaxQuad], phi1[kMaxQuad], phi2[kMaxQuad], w2[kMaxQuad];  // w2 = 2 * w
};

// Scalars of the BiCGStab recurrence.  A slot is written by block 0 of exactly one kernel and read
// only by LATER kernels (visibility = kernel boundary on one stream).
struct KrylovState {
    double rho[2];      // (rhat, r) by iteration parity
    double alpha;
    double omega;
    double target2;     // squared stopping threshold
    double rnorm2;      // ||r||^2 at exit
    double rhs2;        // ||rhs||^2
    double rr_last;     // ||r||^2 seen by the latest stop test (the host reads it to stop queueing ahead near the target)
    int done;           // 1 once converged / stopped; later kernels return immediately
    int converged;
    int breakdown;
    int its;            // iterations completed when done was set
};

struct Mesh {  // device pointers, internal numbering
    const double2* xy;
    const int32_t* cells;  // 3*ne
};

// SELL-64 matrix of one level: slot(s, k, lane) = ptr[s] + 64 k + lane, row = 64 s + lane.
struct DevSell {
    int32_t n_rows, n_cols, nslice;
    int32_t xcd_local;   // 1: matrix fits the 256 MiB Infinity Cache -> XCD-contiguous sweep (measured: +11 % at
                         // 1M rows, -9 % at 10M rows where the eight far-apart HBM streams cost more than the
                         // x-vector re-fetches the shared cache already absorbs)
    const int32_t* ptr;
    const int32_t* col;
    const uint8_t* rowlen;
    const int32_t* cbase;    // per slice: base column of the 16-bit offsets, or -1
    const int32_t* ptr16;
    const uint16_t* col16;
    // Packed smoother copy (multigrid levels of >= SHK_AMG_BF16_ROWS rows; nullptr otherwise): one 32-bit word per slot of
    // the slices with 16-bit columns, (bfloat16 value << 16) | column offset, indexed like col16.  ONE load per slot instead
    // of a 4-byte and a 2-byte one, 4 B instead of 6.  Only smoothing sweeps read it: the Galerkin products, 1/diag and the
    // spectral estimates keep the float values (slices with 32-bit columns too).
    // The rounded entries are applied to DIFFERENCES:  (A x)_i = rs_i x_i + sum_{j != i} a~_ij (x_j - x_i),  rs_i = sum_j a_ij
    // summed in float.  Rounding then costs 2^-9 of the LOCAL VARIATION of x and nothing on its smooth part -- which is
    // what the operator annihilates (rs_i << a_ii) and what plain rounding was measured to spoil: applied to x_j itself,
    // the row sums move by ~2^-9 a_ii and the Krylov iterations of the 10M-row mesh rise from 555 to 664 (with the
    // differences: 554).  The row's own slot -- the diagonal; for A*P the column of the row's own aggregate -- multiplies
    // x_i - x_i = 0 in that form, so it is the slot that carries rs_i (as bfloat16: 2^-9 of a small number).
    const uint32_t* pk = nullptr;
};

// One SELL-64 slice times x: the lane's row sum.  Four streams: 16- or 32-bit columns, and non-temporal
// loads when the matrix cannot stay in the Infinity Cache (so that it does not evict x; measured at 10M
// rows: -10 % time, while a cache-resident matrix is +25 % slower when read non-temporally).
template <bool NT, class T>
__device__ __forceinline__ T sell_ld(const T* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
// TV / TX: value and vector element types (double for the Krylov operator; the multigrid preconditioner stores
// its operators and vectors in float, see shk_amg.hip).  The sum is accumulated in their common type.
// A slice is 4 .. 11 entries wide.  A loop (even unrolled by 4, with its one-at-a-time remainder) would chain
// a load -> gather round trip per trip; instead the width, which is wave-uniform, selects a fully unrolled
// body that issues all value/column loads of the slice, then all gathers, then the FMAs in slot order
// (measured on MI355X, 10M rows, hipEvent legs: k_spmv 224 -> 205 us, k_amg_first<true> 137 -> 119 us).
template <bool NT, int W, class TV, class TC, class TX>
__device__ __forceinline__ auto sell_fixed(const TV* __restrict__ vp, const TC* __restrict__ cp,
                                           const TX* __restrict__ xb) -> decltype(TV() * TX()) {
    TV v[W];
    TC c[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { v[k] = sell_ld<NT>(vp + k * kSlice); c[k] = sell_ld<NT>(cp + k * kSlice); }
    decltype(TV() * TX()) sum = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) sum += v[k] * xb[c[k]];
    return sum;
}
template <bool NT, int WMAX, class TV, class TC, class TX>
__device__ __forceinline__ auto sell_width(const TV* __restrict__ vp, const TC* __restrict__ cp,
                                           const TX* __restrict__ xb, int width) -> decltype(TV() * TX()) {
    decltype(TV() * TX()) sum = 0;
    while (width > WMAX) {
        sum += sell_fixed<NT, WMAX>(vp, cp, xb);
        vp += WMAX * kSlice; cp += WMAX * kSlice; width -= WMAX;
    }
    switch (width) {
        case 1: return sum + sell_fixed<NT, 1>(vp, cp, xb);
        case 2: return sum + sell_fixed<NT, 2>(vp, cp, xb);
        case 3: return sum + sell_fixed<NT, 3>(vp, cp, xb);
        case 4: return sum + sell_fixed<NT, 4>(vp, cp, xb);
        case 5: return sum + sell_fixed<NT, 5>(vp, cp, xb);
        case 6: return sum + sell_fixed<NT, 6>(vp, cp, xb);
        case 7: return sum + sell_fixed<NT, 7>(vp, cp, xb);
        case 8: return sum + sell_fixed<NT, 8>(vp, cp, xb);
        case 9: if (WMAX >= 9) return sum + sell_fixed<NT, 9>(vp, cp, xb);
        case 10: if (WMAX >= 10) return sum + sell_fixed<NT, 10>(vp, cp, xb);
        case 11: if (WMAX >= 11) return sum + sell_fixed<NT, 11>(vp, cp, xb);
        case 12: if (WMAX >= 12) return sum + sell_fixed<NT, 12>(vp, cp, xb);
        default: return sum;
    }
}
// The same for the packed smoother copy (DevSell::pk).
// own: the row's own column as a 16-bit offset; xi = x[own column]
template <bool NT, int W, class TX>
__device__ __forceinline__ float sell_fixed_pk(const uint32_t* __restrict__ pp, const TX* __restrict__ xb, uint32_t own, float xi) {
    uint32_t w[W];
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = sell_ld<NT>(pp + k * kSlice);
    float sum = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const uint32_t c = w[k] & 0xffffu;
        const float d = (float)xb[c] - xi;
        sum += __uint_as_float(w[k] & 0xffff0000u) * (c == own ? xi : d);
    }
    return sum;
}
template <bool NT, int WMAX, class TX>
__device__ __forceinline__ float sell_width_pk(const uint32_t* __restrict__ pp, const TX* __restrict__ xb, int width, uint32_t own, float xi) {
    float sum = 0;
    while (width > WMAX) {
        sum += sell_fixed_pk<NT, WMAX>(pp, xb, own, xi);
        pp += WMAX * kSlice; width -= WMAX;
    }
    switch (width) {
        case 1: return sum + sell_fixed_pk<NT, 1>(pp, xb, own, xi);
        case 2: return sum + sell_fixed_pk<NT, 2>(pp, xb, own, xi);
        case 3: return sum + sell_fixed_pk<NT, 3>(pp, xb, own, xi);
        case 4: return sum + sell_fixed_pk<NT, 4>(pp, xb, own, xi);
        case 5: return sum + sell_fixed_pk<NT, 5>(pp, xb, own, xi);
        case 6: return sum + sell_fixed_pk<NT, 6>(pp, xb, own, xi);
        case 7: return sum + sell_fixed_pk<NT, 7>(pp, xb, own, xi);
        case 8: return sum + sell_fixed_pk<NT, 8>(pp, xb, own, xi);
        default: return sum;
    }
}
// XCD-aware work distribution for streaming kernels (MI355X: 8 XCDs with private 4 MiB L2s, workgroups are
// dealt round-robin, so blockIdx % 8 labels the XCD).  Each XCD gets one contiguous eighth of the slice groups
// and its workgroups sweep it side by side, so the x-vector lines shared by neighbouring rows are fetched into
// ONE L2 instead of up to eight.  Placement only affects speed, never results.
struct GroupSweep {
    int begin, end, step;
};
// (Dealing chunks of 4 .. 1024 consecutive groups to the XCDs in turn -- close streams, still XCD-local runs --
// measured the same as plain round robin at 10M rows: the x vector is served by the Infinity Cache either way.)
__device__ __forceinline__ GroupSweep xcd_sweep(int ngroups, int xcd_local) {
    const int nb = gridDim.x, b = blockIdx.x;
    if (!xcd_local || (nb & 7) != 0) return GroupSweep{b, ngroups, nb};
    const int per = (ngroups + 7) >> 3, xcd = b & 7;
    const int g0 = xcd * per;
    return GroupSweep{g0 + (b >> 3), min(ngroups, g0 + per), nb >> 3};
}

// Slice descriptor, wave-uniform: kept in SGPRs and fetched through the scalar cache (the slice index must be
// provably uniform for that: take the wave index through wave_index()).
struct SellMeta {
    int base, width, cb, p16;
};
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ SellMeta sell_meta(const DevSell& A, int s) {
    const int base = A.ptr[s];
    return SellMeta{base, (A.ptr[s + 1] - base) >> 6, A.cbase[s], A.ptr16[s]};
}
template <bool NT, int WMAX, class TV, class TX>
__device__ __forceinline__ auto sell_row_sum_t(const DevSell& A, const SellMeta& m, const TV* __restrict__ vals,
                                               const TX* __restrict__ x, int lane) -> decltype(TV() * TX()) {
    const TV* __restrict__ vp = vals + m.base + lane;
    if (m.cb >= 0) return sell_width<NT, WMAX>(vp, A.col16 + m.p16 + lane, x + m.cb, m.width);
    return sell_width<NT, WMAX>(vp, A.col + m.base + lane, x, m.width);
}
// WMAX = 8 or 12: widest fully unrolled body (registers against round trips; 12 pays for the double-precision
// Krylov product, whose slices are up to 11 wide, 8 for the float smoothers)
template <int WMAX = 8, class TV, class TX>
__device__ __forceinline__ auto sell_row_sum(const DevSell& A, const SellMeta& m, const TV* __restrict__ vals,
                                             const TX* __restrict__ x, int lane) -> decltype(TV() * TX()) {
    return A.xcd_local ? sell_row_sum_t<false, WMAX>(A, m, vals, x, lane) : sell_row_sum_t<true, WMAX>(A, m, vals, x, lane);
}
// PK: slices with 16-bit columns come from the packed copy (A.pk must be set), the others from vals / col as before
// (own_col: the row's own column -- the row itself for a square operator, its own aggregate for A*P; xi = x[own_col])
template <bool PK, class TX>
__device__ __forceinline__ float sell_row_sum_pk(const DevSell& A, const SellMeta& m, const float* __restrict__ vals,
                                                 const TX* __restrict__ x, int lane, int own_col, float xi) {
    if constexpr (!PK) return sell_row_sum(A, m, vals, x, lane);
    else {
        if (m.cb >= 0) {
            const uint32_t* __restrict__ pp = A.pk + m.p16 + lane;
            const uint32_t own = (uint32_t)(own_col - m.cb);
            return A.xcd_local ? sell_width_pk<false, 8>(pp, x + m.cb, m.width, own, xi)
                               : sell_width_pk<true, 8>(pp, x + m.cb, m.width, own, xi);
        }
        return sell_row_sum(A, m, vals, x, lane);
    }
}
template <int WMAX = 8, class TV, class TX>
__device__ __forceinline__ auto sell_row_sum(const DevSell& A, const TV* __restrict__ vals,
                                             const TX* __restrict__ x, int s, int lane) -> decltype(TV() * TX()) {
    return sell_row_sum<WMAX>(A, sell_meta(A, s), vals, x, lane);
}

// The loop every SELL kernel runs: this wave's slices s = 4 g + wave over the workgroup's groups g, with the
// NEXT slice's descriptor requested (scalar loads) before the current slice's row sum, so that its latency
// hides behind the value / column stream instead of preceding it.
//   for (SliceLoop it(A, wave); it.valid(); it.next()) { ... it.s, it.m ... }
struct SliceLoop {
    const DevSell& A;
    GroupSweep sw;
    int wave, g, s;
    SellMeta m, mn;
    __device__ __forceinline__ SliceLoop(const DevSell& A_, int wave_) : A(A_), wave(wave_) {
        sw = xcd_sweep((A.nslice + 3) >> 2, A.xcd_local);
        g = sw.begin;
        s = 4 * g + wave;
        if (valid()) { m = sell_meta(A, s); prefetch(); }
    }
    __device__ __forceinline__ bool valid() const { return g < sw.end && s < A.nslice; }
    __device__ __forceinline__ void prefetch() {
        const int sn = 4 * (g + sw.step) + wave;
        if (g + sw.step < sw.end && sn < A.nslice) mn = sell_meta(A, sn);
    }
    __device__ __forceinline__ void next() {
        g += sw.step;
        s = 4 * g + wave;
        m = mn;
        if (valid()) prefetch();
    }
};

// Interior / boundary passes of a level-0 sweep over a decomposed mesh.  PASS 0: every slice (one subdomain, or no
// overlap).  PASS 1 ("interior"): the ordinary sweep, but slices flagged in `ghost` keep their hands off the output
// (their row sums, computed from ghost values that may be arriving at that moment, are dropped: ~3 % redundant
// reads instead of a second slice order).  PASS 2 ("boundary"): only the flagged slices, from the list, after the
// receive has completed.
struct SplitSell {
    const uint8_t* ghost;    // nslice flags (PASS 1)
    const int32_t* list;     // flagged slices (PASS 2)
    int32_t n_list;
};

// Multigrid down-sweep over four levels in one launch (k_amg_restrict4).  An aligned group of 256 rows holds
// complete aggregate trees four levels deep (256 = 4^4): one workgroup pass leaves the right-hand sides of
// levels 1 .. nlev behind, through LDS, instead of one launch per level.  Tables are indexed by the coarse row's
// k-d rank, which is what a workgroup can enumerate: group g owns ranks [g 4^(4-l), (g+1) 4^(4-l)) of level l.
// (Fusing this into k_bicg_s / k_bicg_u, which produce the vectors, was measured slower: the barriers stall
// their load streams -- vector phase 25 -> 51 ms per step at 10M rows for 8 ms saved here.)
constexpr int kFusedRestrict = 4;
struct RestrictArgs {
    int nlev;                              // 0: off
    const int4* members[kFusedRestrict];   // [l][k]: the (<= 4, -1 padded) rows of level l under rank k of level l+1
    const int32_t* pos[kFusedRestrict];    // [l][k]: storage position of rank k of level l+1
    float* rc[kFusedRestrict];             // right-hand side of level l+1
    int32_t nc[kFusedRestrict];            // rows of level l+1
};
// All 256 threads call this once per group g with their entry of the vector (0 beyond the end).  Threads 0..63
// produce the group's level-1 rows, 64..79 its level-2 rows, 80..83 level 3, 84 level 4; every role fetches its
// table entries up front (one memory round trip per group), the sums then cascade through LDS.  Sums run in the
// member order of k_amg_restrict, so both paths give the same bits.
__device__ __forceinline__ void fused_restrict(const RestrictArgs& ra, int g, double val, double* buf0,
                                               float (*buf)[256]) {
    const int tid = threadIdx.x;
    const int lvl = tid < 64 ? 0 : tid < 80 ? 1 : tid < 84 ? 2 : tid == 84 ? 3 : -1;
    const int j = tid < 64 ? tid : tid < 80 ? tid - 64 : tid < 84 ? tid - 80 : 0;
    bool act = lvl >= 0 && lvl < ra.nlev;
    int4 m = make_int4(-1, -1, -1, -1);
    int p = 0;
    float* rc = nullptr;
    if (act) {
        const int k = g * (64 >> (2 * lvl)) + j;
        const int4* mt = lvl == 0 ? ra.members[0] : lvl == 1 ? ra.members[1] : lvl == 2 ? ra.members[2] : ra.members[3];
        const int32_t* pt = lvl == 0 ? ra.pos[0] : lvl == 1 ? ra.pos[1] : lvl == 2 ? ra.pos[2] : ra.pos[3];
        const int32_t nc = lvl == 0 ? ra.nc[0] : lvl == 1 ? ra.nc[1] : lvl == 2 ? ra.nc[2] : ra.nc[3];
        rc = lvl == 0 ? ra.rc[0] : lvl == 1 ? ra.rc[1] : lvl == 2 ? ra.rc[2] : ra.rc[3];
        act = k < nc;
        if (act) { m = mt[k]; p = pt[k]; }
    }
    __syncthreads();   // the previous group's readers are done
    buf0[tid] = val;
    __syncthreads();
    if (act && lvl == 0) {
        const int b0 = g << 8;
        double acc = buf0[m.x - b0];
        if (m.y >= 0) acc += buf0[m.y - b0];
        if (m.z >= 0) acc += buf0[m.z - b0];
        if (m.w >= 0) acc += buf0[m.w - b0];
        const float rv = (float)acc;
        rc[p] = rv;
        buf[0][p & 255] = rv;
    }
#pragma unroll
    for (int l = 1; l < kFusedRestrict; ++l) {
        if (l >= ra.nlev) break;
        __syncthreads();
        if (act && lvl == l) {
            float acc = buf[l - 1][m.x & 255];
            if (m.y >= 0) acc += buf[l - 1][m.y & 255];
            if (m.z >= 0) acc += buf[l - 1][m.z & 255];
            if (m.w >= 0) acc += buf[l - 1][m.w & 255];
            rc[p] = acc;
            if (l + 1 < kFusedRestrict) buf[l][p & 255] = acc;
        }
    }
}

constexpr int kAsmCellsMax = 640;   // cells one assembly workgroup stages (PlanOptions::cells_max is capped to it)
constexpr int kAsmVertsMax = 768;   // vertices (own rows + halo) one assembly workgroup stages
constexpr int kAsmSlotsMax = 3072;  // SELL slots of one assembly workgroup (4 slices x 64 rows x mean width <= 12)
struct AsmArgs {
    Mesh m;
    const double* fld[11];   // N, N_n, b, qx, qy, z_b, z_s, G, melt_n, storage, inputs
    const uint8_t* bcflag;   // nullptr if no Dirichlet dofs
    const uint32_t* slotsrc; // per SELL slot: its (at most two) staged cells, 14 bits each ((block-local cell slot << 4)
                             // | (3 li + lj), cell slot kSrcNone = none) and the slot's Dirichlet code in bits 28-29
    double bc_value;
    double inv_rwg_dt;       // 1 / (rho_w g dt)
    DevSell A;
    const int32_t *blk_desc, *blk_halo, *incptr;   // blk_desc: kBlkDesc ints per block (shk_plan.h)
    const uint16_t *blk_cellv, *inccode;
    int cells_max;           // LDS stride E of the element tensors
    int verts_max;           // LDS stride V of the staged fields
    int slices_max, inc_max;
    int lds_region_a;        // bytes of the fields / element-tensor region
#ifdef SHK_EXPERIMENTS
    int ablate;              // probe builds only: 1 skip the element computation, 2 the slot phase, 4 the field loads
#endif
    // outputs
    double* F;
    double* vals;
    double* dinv;            // 1 / diag(J)
    DevParams p;
    QuadArg quad;            // degree-7 rule: the transmissivity integral
    QuadArg qpoly;           // degree-5 rule (7 points): every polynomial term (quad again if n != 3)
};

inline int32_t sell_fits_cache(int64_t slots, int bytes_per_slot = 12) {   // 8 B value + 4 B column
    const int force = tunables().xcd;   // experiment switch: 0 / 1
    if (force >= 0) return force;
    return slots * bytes_per_slot < (int64_t)192 << 20 ? 1 : 0;
}
constexpr int kAmgSlotBytes = 8;   // float value + (mostly 16-bit) column
constexpr int kAmgPackedSlotBytes = 4;   // DevSell::pk
// rows from which a multigrid level's sweeps stream the packed bfloat16 copy (SHK_AMG_BF16_ROWS; 0: never)
inline bool amg_packed_level(int64_t n_rows, int64_t slots16) {
    const int64_t t = tunables().amg_bf16_rows;
    return t > 0 && n_rows >= t && slots16 > 0;
}

// Multigrid hierarchy on the device (shk_amg.hip).  Level 0 is the Jacobian itself (Ctx::d_vals, d_dinv).
struct AmgLevel {
    int32_t n = 0, n_cols = 0, nslice = 0;
    int64_t slots = 0, slots16 = 0;   // stored slots; of them with 16-bit columns
    int32_t *ptr = nullptr, *col = nullptr, *diag_slot = nullptr, *cbase = nullptr, *ptr16 = nullptr;
    uint16_t* col16 = nullptr;
    uint8_t* rowlen = nullptr;
    float *vals = nullptr, *dinv = nullptr, *x = nullptr, *x2 = nullptr, *x3 = nullptr, *r = nullptr;   // preconditioner precision
    uint32_t* pk = nullptr;           // packed smoother copy (DevSell::pk), large levels only
};
struct AmgXfer {  // level l -> l+1
    int32_t n_fine = 0, n_coarse = 0, n_coarse_cols = 0;
    int32_t *agg = nullptr, *members = nullptr, *gptr = nullptr, *glist = nullptr;
    int32_t *members_kd = nullptr, *kd_pos = nullptr;   // fused restriction tables (RestrictArgs); null: unavailable
    bool dense = false;
    bool onto_global = false;   // lands on my rows of the replicated global level (AmgHierarchy::rep)
    // A*P of the fine level (optional): thinner operator for the first smoothing sweep after the prolongation
    bool with_ap = false;
    int32_t ap_nslice = 0;
    int64_t ap_slots = 0, ap_slots16 = 0;
    int64_t n_glist = 0, ap_n_glist = 0;   // entries of the Galerkin gather lists (byte accounting of the refresh)
    int32_t *ap_ptr = nullptr, *ap_col = nullptr, *ap_cbase = nullptr, *ap_ptr16 = nullptr, *ap_gptr = nullptr,
            *ap_glist = nullptr;
    uint16_t* ap_col16 = nullptr;
    uint8_t* ap_rowlen = nullptr;
    float* ap_vals = nullptr;
    uint32_t* ap_pk = nullptr;        // packed smoother copy of A*P (DevSell::pk)
    // decomposed levels: coarse column of each ghost column of the fine level (frozen-ghost smoothing, AmgHierarchy)
    int32_t* ghost_col = nullptr;
    int32_t n_ghost = 0;
};
// Device copy of a SweepPlan (shk_plan.h): the fused multi-sweep smoother of one level.
struct DevSweepPlan {
    int32_t nblk = 0, width = 0, max_local = 0;
    int32_t *hdr = nullptr, *ext_info = nullptr;
    uint16_t *lcol_own = nullptr, *ring_lcol = nullptr;
    double plan_bytes = 0;   // bytes of the plan arrays one launch streams (byte accounting)
    bool ready() const { return nblk > 0; }
};
// A hierarchy is either block-local (the owned diagonal block of a subdomain, or the whole matrix of a single
// context: no communication) or distributed (ghost columns kept on every level, per-level halo plans, one
// dense coarsest operator shared by all subdomains).
struct AmgHierarchy {
    // The operator the hierarchy sits on ("level 0"): the Jacobian's float copy for a context's own hierarchies
    // (bound by amg_bind_top before use), or the gathered global level of a replicated coarse hierarchy.
    DevSell topA{};
    const float *top_vals = nullptr, *top_dinv = nullptr;
    double top_bytes = 0.0;      // bytes one sweep streams of the top operator (byte accounting)
    std::vector<AmgLevel> lv;    // [0] unused, [l] = sparse coarse level l
    std::vector<AmgXfer> xf;     // [l] : level l -> l+1; the last one lands on the dense coarsest level
    std::vector<int> plan_of;    // distributed: index into Comm::plans of level l's halo plan (size = xf.size())
    std::vector<DevSweepPlan> sw;   // [l]: fused four-sweep smoother of sparse level l >= 1 ([0]: a replicated hierarchy's top)
    bool distributed = false;
    double alpha = 1.5;          // over-correction x = alpha * P e_c: piecewise-constant prolongation under-estimates
                                 // the coarse correction.  Measured ms/step at 10M | 1M rows: alpha 1.0: 870 | 108,
                                 // 1.3: 641 | 81, 1.5: 646 | 81, 1.7: 703 | 80, 2.0: 899 | 92; with the float cycle and the
                                 // (1.0, 0.6) sweeps: 1.4: 282 | 52, 1.5: 274 | 49, 1.6: 268 | 49   (SHK_AMG_ALPHA)
    bool dense_valid = false;    // the dense coarsest inverse has been built at least once
    int dense_age = 0, dense_period = 8;   // Newton solves since / between rebuilds of a big inverse (SHK_AMG_DENSE_PERIOD)
    int its_fresh = 0, its_last = 0;       // Krylov iterations of the first solve after a rebuild / of the last solve
    int halo_levels = 99;        // decomposed levels [0, halo_levels) exchange ghosts inside the smoother
                                 // (SHK_AMG_HALO_LEVELS): all of them by default -- below the decomposed levels
                                 // sits the replicated coarse part (`rep`), which needs no exchange.  Block-local
                                 // smoothing instead is expensive: 4 subdomains at 10M rows, 1 / 2 / 3 / 4 / all
                                 // levels exchanging = 118 / 76 / 67 / 61 / 48 iterations (one subdomain: 45).
    // Two damped-Jacobi sweeps x <- x + (c_k / lambda) D^-1 (r - A x), lambda ~ the largest eigenvalue of D^-1 A
    // over the levels (power iteration at the rebuilds of the dense inverse, +10 % because it converges from
    // below).  c = (2.35, 1.41) is the pair measured best on the 10M-row system (lambda there = 2.35: w = 1.0,
    // 0.6; 61 iterations per Newton step against 69 for 0.7, 0.7 -- and 0.9, 0.9 diverges).  (SHK_AMG_W1/W2)
    double lambda = 0.0;             // 0: not estimated yet
    int lambda_age = 0;              // dense-inverse refreshes since the spectral estimates were renewed
    double gersh = 0.0;              // Gershgorin bound of lambda_max(D^-1 A) over the levels (rigorous, unlike lambda)
    double lam_max = 0.0;            // spectral bound the caps use: min(gersh, 1.05 x Lanczos estimate)
    double cap2 = 1.0, cap4 = 1.0;   // factors <= 1 on the two- / four-sweep dampings: no amplification on (0, gersh]
    double c1 = 2.35, c2 = 1.25;     // w = (1.03, 0.55) on the synthetic meshes.  Iterations per Newton iteration at 10M |
                                     // 1M rows with the final cycle: (1.03, 0.62) 45.1 | 40.9, (1.03, 0.55) 44.8 | 39.8,
                                     // (0.95, 0.55) 46.0 | 41.2, (0.85, 0.50) 48.0 | 40.9 -- and (1.03, 0.70) 97.6 | 41.9: the
                                     // second damping sits well below that edge
    // Levels >= 1 without ghost exchanges run FOUR sweeps, dampings c4[k] / lambda = the Chebyshev roots for
    // [0.25, 0.9] lambda, small and large steps interleaved.  Measured ms/step at 10M | 1M rows: two sweeps
    // 253 | 43.0; four with (c1, c2) twice 238 | 43.3; Chebyshev on [0.37, 0.77] 242 | 44, [0.25, 0.8] 230 | 39.4,
    // [0.2, 0.8] 223 | 41.5, [0.25, 0.9] 225 | 41.7; [0.15, 0.8] and [0.25, 0.7] diverge at 10M rows (SHK_AMG_COARSE4=0)
    size_t w_level = 0;          // level whose cycle runs twice per visit (0: plain V-cycle); set at upload
    bool coarse4 = true;
    int coarse4_from = 1;        // first level that runs the four-sweep sequence (levels above it: the finest level's two)
    bool top_four = false;       // the top level is itself a coarse level of a larger cycle (replicated hierarchy)
    double c4[4] = {1.143, 3.640, 1.430, 2.219};
    int64_t ap_nnz0 = 0;             // stored entries of the finest level's A*P operator
    int32_t n_glob = 0, offset = 0;  // dense coarsest operator: n_glob x n_glob, my rows start at `offset`
    float *x0 = nullptr, *x1 = nullptr, *x2 = nullptr, *cr = nullptr, *cx = nullptr;   // x1, x2: eigenvalue-estimate scratch (finest level)
    double *cdense = nullptr, *cinv = nullptr, *cglob = nullptr;   // the coarsest operator and its inverse are built in double
    float* cinv32 = nullptr;     // ... and applied from a float copy by a hierarchy's own dense level (> 128 rows)
    double* gj = nullptr;        // 2 * 1024 doubles of Gauss-Jordan scratch
    // Replicated coarse part of a decomposed hierarchy: from the first level whose GLOBAL size is small enough, the
    // level is gathered (one all-reduce of the right-hand side per cycle) and the rest of the cycle runs on every
    // GPU redundantly, as the single-GPU hierarchy `rep` whose top operator is that global level -- no ghost
    // exchange below it, and no loss of couplings.  The arrays below belong to the distributed hierarchy ...
    AmgHierarchy* rep = nullptr;
    int32_t rep_row0 = 0, rep_n = 0;          // my rows [rep_row0, rep_row0 + n) of the rep_n global rows
    float *rep_rglob = nullptr, *rep_xglob = nullptr;   // gathered right-hand side (my block is written in place by the
                                              // restriction, the others arrive by ONE in-place all-gather of floats per
                                              // cycle: 1 / (2 P) of the bytes of rounds 1-2's zero-padded double all-reduce)
    std::vector<int64_t> rep_rhs_off, rep_val_off;   // byte offsets of every subdomain's block of rep_rglob / of the
                                              // global level's SELL values (whole slices: blocks are multiples of 1024 rows)
    // Frozen-ghost smoothing of the decomposed COARSE levels: inside a level's sweeps a ghost column holds the prolongated
    // coarse correction alpha * e[coarse column of the ghost] -- known locally once the coarser level's result has been
    // exchanged for the A*P sweep -- instead of the neighbour's smoothed value.  Still a fixed linear operator; one
    // exchange per decomposed coarse level and cycle instead of two, and all sweeps of such a level can run inside one
    // launch.  Measured on the 8-way split of the 10M-DOF mesh (Krylov iterations of the first four steps | ghost
    // exchanges per Krylov iteration; tools/sweep_ghost_mask.sh, profiles/r03_ghost_policy_sweep.md): every level
    // exchanging after its first sweep (rounds 1-2) 397 | 13.7; every level frozen 520 | 6.9; only the finest level
    // exchanging 394 | 9.2 -- the default; additionally dropping the exchange of a frozen level's result 422-439 | 6.9
    // and 797 | 4.6 (no gain at 10-30 us per round).  SHK_AMG_GHOST_EXCHANGE = bit mask of the levels that exchange
    // after their first sweep (default 1); SHK_AMG_E_EXCHANGE = mask of the frozen levels whose result is exchanged.
    uint32_t frozen_mask = ~1u;   // bit l: level l runs with frozen ghosts
    uint32_t e_exchange_mask = ~0u;   // bit l: level l's final result is exchanged for the finer level's A*P sweep
    // ... and these to `rep` itself: its own top operator (global level), values refreshed by the owner hierarchy
    int32_t *t_ptr = nullptr, *t_col = nullptr, *t_cbase = nullptr, *t_ptr16 = nullptr, *t_diag = nullptr;
    uint16_t* t_col16 = nullptr;
    uint8_t* t_rowlen = nullptr;
    float *t_vals = nullptr, *t_dinv = nullptr;
    int64_t t_slots = 0;
    bool ready() const { return !xf.empty(); }
    AmgHierarchy() = default;
    AmgHierarchy(const AmgHierarchy&) = delete;
    AmgHierarchy& operator=(const AmgHierarchy&) = delete;
    ~AmgHierarchy() { delete rep; }
};

// Ghost-exchange plan of one level of a subdomain (level 0 = the mesh, l >= 1 = multigrid levels).
struct HaloPlan {
    int64_t n_own = 0;                        // the ghost segment of this level's vectors starts here
    std::vector<int> nbr;                     // neighbour ranks, ascending
    std::vector<int64_t> send_ptr, recv_ptr;  // per-neighbour offsets into the packed buffers (size nbr+1)
    std::vector<int32_t> h_send_idx;          // owned rows to send (internal numbering), host copy
    int32_t* d_send_idx = nullptr;
};

// Communication state of a subdomain context (shk_comm.hip).
struct Comm {
    enum Kind { NONE = 0, RCCL = 1, CALLBACK = 2 } kind = NONE;
    int rank = 0, nranks = 1;
    void* nccl = nullptr;                    // ncclComm_t
    shk_exchange_fn cb_exchange = nullptr;   // CALLBACK transport
    shk_allreduce_fn cb_allreduce = nullptr;
    void* cb_user = nullptr;
    std::vector<HaloPlan> plans;             // [0] fine level, [l] multigrid level l (distributed hierarchy)
    int64_t n_exchange = 0, n_allreduce = 0, bytes_exchange = 0, bytes_allreduce = 0;   // message rounds since creation
    int64_t n_overlapped = 0;                // of n_exchange: issued on comm_stream behind an interior pass
    int64_t n_allgather = 0, bytes_allgather = 0;   // in-place all-gathers (replicated level: right-hand side, operator values)
    bool timing_only = false;                // shk_comm_set_timing_only: messages are skipped (results wrong, durations right)
    double* d_sendbuf = nullptr;             // sized for plans[0], the largest
    double* d_recvbuf = nullptr;             // staging of a float vector's ghosts (they travel as doubles)
    double *h_send = nullptr, *h_recv = nullptr, *h_red = nullptr;  // pinned staging (CALLBACK)
    size_t h_red_cap = 0;
};

struct Ctx {
    int device = 0;
    Comm comm;
    int np = 0;     // entries of a reduction slot the consumers sum: grid (one context) or 1 (all-reduced scalars)
    int red_stride = kMaxParts;   // distance between reduction slots in d_red: kMaxParts, or 1 across subdomains
    hipStream_t stream = nullptr;
    int64_t n_own = 0, n_loc = 0, ne = 0, nnz = 0, slots = 0;
    shk_params params{};
    DevParams dp{};
    QuadArg quad{}, qpoly5{};
    HostPlan plan;
    int grid = 0;   // blocks used by grid-stride kernels == length of partial arrays
    // device memory
    std::vector<void*> allocs;
    int64_t device_bytes = 0;
    double2* d_xy = nullptr;
    int32_t* d_cells = nullptr;
    int32_t* d_perm = nullptr;
    double* d_io = nullptr;                   // 2*n_loc staging for permuted field I/O
    double* f[SHK_FIELD_COUNT] = {nullptr};   // SHK_Q slot unused (qx/qy are separate)
    double *d_melt_tmp = nullptr, *d_b_tmp = nullptr, *d_m0 = nullptr;
    uint8_t* d_bcflag = nullptr;
    uint32_t* d_slotsrc = nullptr;
    bool has_bc = false;
    double bc_value = 0.0;
    int32_t *d_sell_ptr = nullptr, *d_sell_col = nullptr, *d_lastcell = nullptr;
    uint8_t* d_rowlen = nullptr;
    int32_t *d_cbase = nullptr, *d_ptr16 = nullptr;
    uint16_t* d_col16 = nullptr;
    int32_t *d_blk_desc = nullptr, *d_blk_halo = nullptr, *d_incptr = nullptr;
    uint16_t *d_inccode = nullptr, *d_blk_cellv = nullptr;
    int nblk = 0;
    int64_t cells_staged = 0;  // cells computed per assembly incl. those shared between blocks
    size_t asm_lds = 0, asm_region_a = 0;
    size_t asm_lds_res = 0, asm_region_a_res = 0;   // the residual-only instance: 3 instead of 12 staged tensor entries per cell
    double *d_F = nullptr, *d_vals = nullptr, *d_vals_s = nullptr, *d_dinv = nullptr;
    float *d_vals32 = nullptr, *d_dinv32 = nullptr;   // float copies read by the multigrid preconditioner
    uint32_t* d_pk = nullptr;                         // ... and the packed copy its level-0 sweeps stream (DevSell::pk)
    // Krylov vectors
    double *d_r = nullptr, *d_rhat = nullptr, *d_p = nullptr, *d_v = nullptr, *d_s = nullptr, *d_t = nullptr,
           *d_y = nullptr, *d_ytot = nullptr, *d_rhs = nullptr;
    double cur_rtol2 = 0.0, cur_atol2 = 0.0;   // stopping rule of the inner solve being enqueued
    // launch_warm_start: solutions of Newton iteration k (< kWarmIts) of the last kWarmDepth time steps, newest
    // first (allocated on first use: shk_api.hip warm_alloc)
    static constexpr int kWarmIts = 3, kWarmDepth = 4, kWarmDots = kWarmDepth + kWarmDepth * (kWarmDepth + 1) / 2;
    double* d_guess[kWarmIts][kWarmDepth] = {};
    int n_guess[kWarmIts] = {};
    double *d_part_w = nullptr, *d_red_w = nullptr;   // kWarmDots partial arrays / all-reduced scalars
    int warm_its = kWarmIts;
    double* pending_keep = nullptr;   // where the Newton update that follows a solve also stores its solution (warm start)
    // multigrid preconditioner (empty when unavailable: subdomain contexts, tiny meshes)
    AmgHierarchy amg_local, amg_dist;
    AmgHierarchy* amg = nullptr;    // the active one when use_amg
    float *d_phat = nullptr, *d_shat = nullptr;   // M^-1 p, M^-1 s: float, like everything the cycle produces
    float *d_p32 = nullptr, *d_s32 = nullptr;     // float copies of p, s written by the kernels that produce them: the cycle's input
    bool use_amg = false;
    double* d_part = nullptr;  // 8 arrays of kMaxParts: this subdomain's partial sums
    double* d_red = nullptr;   // what the consumers read: d_part itself (one context), or P_COUNT scalars = this
                               // subdomain's partial arrays summed in a fixed order, then all-reduced (48 B per
                               // reduction point instead of 2-4 partial arrays of 16 KB)
    KrylovState* d_state = nullptr;
    KrylovState* h_state = nullptr;  // pinned, 2 slots
    double* h_part = nullptr;        // pinned, kMaxParts
    hipEvent_t poll_ev[3] = {nullptr, nullptr, nullptr};   // [0,1] stop-flag polls, [2] deadline waits
    // Interior / boundary split of the level-0 sweeps (several subdomains only, SplitSell below): the ghost exchange
    // runs on comm_stream while the main stream sweeps the slices that read no ghost column.
    bool overlap = false;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;   // vector complete (main) / ghosts arrived (comm_stream)
    uint8_t* d_slice_ghost = nullptr;   // nslice: 1 = the slice stores a ghost column
    int32_t* d_bslices = nullptr;       // those slices, ascending
    int n_bslices = 0;
    double* d_part_b = nullptr;         // P_COUNT arrays of kMaxParts: partial sums of the boundary passes
    bool assembled = false;   // d_F holds the residual of the current state
    bool jac_valid = false;   // ... and d_vals / d_dinv its Jacobian (false after a residual-only pass)
    int newton_prev = 0;      // Newton iterations of the previous shk_newton_solve (predicts the last iteration of this one)
    static constexpr int kNewtonHist = 16;
    double newton_ratio[kNewtonHist] = {};   // ||F_{k+1}|| / ||F_k|| of the previous shk_newton_solve (0: none): shk_params.krylov_forcing
    double newton_fk[kNewtonHist] = {};      // ... and its ||F_k||
    double newton_ratio2[kNewtonHist] = {};  // the ratios of the solve before that one (the rule wants a settled regime)
    double newton_hist_f0 = 0.0, newton_hist_dt = 0.0;   // ||F_0|| and dt of the solve that history belongs to
    double newton_hist_margin = 0.0;                     // its final ||F|| over its stopping threshold (0: did not converge)
    int64_t n_forced = 0;                    // linear solves stopped by the forcing rule (shk_solver_stats)
    int64_t n_asm_full = 0, n_asm_res = 0, n_asm_redo = 0;   // assembly passes: full, residual-only, full after a misprediction
    double assembled_dt = 0.0;
    bool poisoned = false;   // a host wait hit the RCCL deadline (or shk_comm_mark_stalled): the stream will never drain, so
                             // shk_destroy must neither synchronise nor free (both would block for ever)
    // profiling
    bool profiling = false;
    struct Ev { hipEvent_t a, b; int phase; double bytes; };
    double pending_bytes = 0.0;   // note_bytes() of the launches about to be timed
    double asm_bytes = 0.0;       // what one assembly pass streams (plan arrays + fields + outputs)
    int64_t slots16 = 0;          // SELL slots of the Jacobian with 16-bit columns
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    shk_profile prof{};

    DevSell sell() const {
        return DevSell{(int32_t)n_own, (int32_t)n_loc, plan.A.nslice, sell_fits_cache(slots), d_sell_ptr, d_sell_col,
                       d_rowlen, d_cbase, d_ptr16, d_col16};
    }
    DevSell sell32() const {   // same pattern, cache rule of the float copy (of the packed one where it exists)
        DevSell A = sell();
        A.xcd_local = sell_fits_cache(slots, d_pk ? kAmgPackedSlotBytes : kAmgSlotBytes);
        A.pk = d_pk;
        return A;
    }
};

// Setup-time uploads and zero fills are ordered ON THE CONTEXT'S STREAM: it is a non-blocking stream, which null-stream
// work is not ordered with (round 2 found a null-stream hipMemset overtaking a copy on it).  The upload waits for the
// copy, so the host array may die as soon as it returns.
hipError_t wait_stream(Ctx* c);          // hipStreamSynchronize with the RCCL deadline (shk_api.hip)
inline hipError_t upload_sync(Ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
    return e != hipSuccess ? e : wait_stream(c);
}
inline hipError_t zero_async(Ctx* c, void* p, size_t bytes) {
    return bytes ? hipMemsetAsync(p, 0, bytes, c->stream) : hipSuccess;
}

// Matrix stream loads: a matrix that cannot stay in the Infinity Cache is read non-temporally so that it does
// not evict the x vector (measured at 10M rows: -3 % time); a cache-resident one is read normally (non-temporal
// there costs +25 %: back-to-back products would re-fetch it from HBM).  See the two loops in k_spmv / k_amg_post.

// partial-array slots; [RR, RHV] and [TS, TT, RHT, RHS] are the two per-iteration reduction groups
enum { P_RR = 0, P_RHV = 1, P_TS = 2, P_TT = 3, P_RHT = 4, P_RHS = 5, P_AUX = 6, P_COUNT = 8 };

// error text of the calling thread (shk_api.hip: shk_last_error); returns -1
int set_error(const std::string& msg);

// launchers (shk_kernels.hip)
hipError_t prepare_kernels(Ctx* c);
size_t assemble_lds_bytes(const HostPlan& P, size_t* region_a, bool residual_only = false);
void launch_assemble(Ctx* c, double dt, bool residual_only = false);
void launch_slot_bc(Ctx* c);
void launch_scale(Ctx* c);
void launch_spmv_plain(Ctx* c, const double* vals, const double* x, double* y);
void launch_norm2(Ctx* c, const double* x, double* partials);
void launch_stream_read(Ctx* c);
void krylov_init(Ctx* c, const double* rhs);
void launch_accumulate(Ctx* c, bool first);
hipError_t launch_true_residual(Ctx* c);
hipError_t launch_warm_start(Ctx* c, int newton_it);
hipError_t krylov_iteration(Ctx* c, int it);
void launch_newton_update(Ctx* c, bool apply);
hipError_t launch_update_explicit(Ctx* c, double dt);
void launch_permute_in(Ctx* c, const double* io, double* dst);            // dst[i] = io[perm[i]]
void launch_permute_out(Ctx* c, const double* src, double* io);           // io[perm[i]] = src[i]
void launch_split_q(Ctx* c, const double* io);                            // io interleaved, external order
void launch_join_q(Ctx* c, double* io);

// communication (shk_comm.hip)
const char* rccl_load();
int rccl_unique_id(void* out128);
const char* rccl_init(Ctx* c, int rank, int nranks, const void* id128);
void comm_destroy(Ctx* c);
void comm_abort(Ctx* c);                 // poisoned context: ncclCommAbort if the library has it, else the communicator is leaked
const char* rccl_selftest(Ctx* c);
double rccl_time_round(Ctx* c, int kind, int64_t n, int reps);
hipError_t halo_exchange(Ctx* c, double* vec);                       // level 0
hipError_t halo_exchange_plan(Ctx* c, const HaloPlan& P, double* vec);
hipError_t halo_exchange_plan_f32(Ctx* c, const HaloPlan& P, float* vec);
// Overlapped form of the two level-0 exchanges: halo_begin* enqueues pack + exchange on comm_stream behind everything
// the main stream holds so far; the caller then launches its interior pass and calls halo_end, which makes the main
// stream wait for the ghosts.  (The host-staged transport blocks inside halo_begin*: launch the interior pass first.)
hipError_t halo_begin(Ctx* c, double* vec);
hipError_t halo_begin_f32(Ctx* c, float* vec);
hipError_t halo_end(Ctx* c);
hipError_t allreduce_buffer(Ctx* c, const double* src, double* dst, size_t n);  // element-wise sum over subdomains
hipError_t allgather_blocks(Ctx* c, void* buf, const std::vector<int64_t>& off);   // in place; off = R + 1 byte offsets
int amg_setup_distributed(Ctx* c, std::string& err);  // collective
hipError_t amg_numeric_setup(Ctx* c, AmgHierarchy& H, bool refresh_dense, bool decided = false, bool top_only = false);
hipError_t amg_vcycle(Ctx* c, AmgHierarchy& H, const double* rin, float* zout);
hipError_t amg_vcycle(Ctx* c, AmgHierarchy& H, const float* rin, float* zout, const double* rin_last = nullptr);
// upload one host hierarchy (shk_api.hip: owns the allocation helpers)
hipError_t amg_upload(Ctx* c, std::vector<AmgLevelPlan>& plans, AmgHierarchy& H, int64_t n_loc0,
                      const std::vector<int32_t>* krank0 = nullptr,    // krank0: k-d ranks of the top rows (default: the mesh's)
                      const SellPattern* top = nullptr);               // pattern of the top operator when it is itself a coarse
                                                                       // level of a larger cycle (replicated hierarchy)
hipError_t amg_upload_rep_top(Ctx* c, AmgHierarchy& R, const SellPattern& G, const std::vector<int32_t>& diag_slot);
hipError_t allreduce_parts(Ctx* c, int first, int nslots);
hipError_t allreduce_part_arrays(Ctx* c, const double* part, double* red, int nslots);   // any partial arrays -> scalars

// Byte accounting of the profiled launches (shk_profile.bytes): a launch site notes what its kernel has to move BEFORE
// it launches (launch_phase) or inside the PhaseTimer scope that times it.
inline void note_bytes(Ctx* c, double bytes) { if (c->profiling) c->pending_bytes += bytes; }
// bytes one sweep over a multigrid operator streams: packed slices 4 B per slot, the others float value + column
inline double amg_sell_bytes(int64_t slots, int64_t slots16, int64_t nslice, bool packed) {
    if (!packed) return (double)slots * 4 + 2.0 * (double)slots16 + 4.0 * (double)(slots - slots16) + 16.0 * (double)nslice;
    return 4.0 * (double)slots16 + 8.0 * (double)(slots - slots16) + 16.0 * (double)nslice;
}
inline double sell_bytes(int64_t slots, int64_t slots16, int64_t nslice, int value_bytes) {
    return (double)slots * value_bytes + 2.0 * (double)slots16 + 4.0 * (double)(slots - slots16) + 16.0 * (double)nslice;
}
struct PhaseTimer {  // RAII hipEvent pair when profiling is on
    Ctx* c;
    int idx = -1;
    PhaseTimer(Ctx* c, int phase);
    ~PhaseTimer();
};
int profile_slot(Ctx* c, int phase);   // next event pair of the pool, booked for `phase`

// Launch ONE kernel that is accounted to `phase`.  While profiling, its start / stop events ride on the dispatch
// packet itself (hipExtLaunchKernelGGL): they bracket the kernel's execution the way a profiler's timestamps do,
// without the dispatch latency that two separate hipEventRecord calls put between their records (measured: 190 us
// against 172 us in the rocprofv3 trace for k_spmv at 10M rows).
template <class F, class... Args>
inline void launch_phase(Ctx* c, int phase, F kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    if (!c->profiling) {
        hipLaunchKernelGGL(kernel, grid, block, shmem, c->stream, args...);
        return;
    }
    const Ctx::Ev& e = c->ev_pool[profile_slot(c, phase)];
    hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)shmem, c->stream, e.a, e.b, 0, args...);
}

}  // namespace shk
