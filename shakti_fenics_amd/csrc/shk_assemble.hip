// k_assemble: fused residual + Jacobian of the weak form at /root/reference/source/solvers.py:35-45 with the closures
// of constitutive.py:6-31 and DOLFINx's Dirichlet algebra, atomic-free and bitwise reproducible.
//
// One workgroup owns up to 4 SELL slices = 256 consecutive rows (a compact patch of the k-d order) and runs
//   phase 0  stages the 13 nodal doubles (x, y and the 11 fields) of its vertices in LDS: its own rows with fully
//            coalesced loads (they are consecutive), the halo vertices of its cells (~0.5 per own row) through a
//            per-block gather list.  Workgroups are dealt to the XCDs so that each XCD sweeps one contiguous range
//            of patches: neighbouring patches then share an L2, which serves most halo gathers.
//   phase 1  one thread per cell touching the owned rows (cells named by 3 x 16-bit LOCAL vertex ids: one coalesced
//            8-byte load per cell, no dependent global gathers): 15-point degree-7 rule for the transmissivity
//            integral, 7-point degree-5 rule for every polynomial term, both unrolled with their tables read as
//            scalar kernel arguments; the 3x3 + 3 element tensor stays in registers until every thread has read its
//            fields, then replaces them in the SAME LDS region (61 KB per workgroup instead of 100).
//   phase 2  one thread per SELL slot: an off-diagonal entry adds its <= 2 staged cells named by the plan (4 B per
//            slot, which also carries the slot's Dirichlet code), the diagonal and the residual row sum their
//            incidence list in ascending cell order.  Every value is written exactly once, coalesced.
#include "shk_device.h"

namespace shk {

#ifndef SHK_UNROLL_Q
#define SHK_UNROLL_Q 5   // independent quadrature points in flight (registers against ILP at 2 waves per SIMD)
#endif
#ifndef SHK_UNROLL_P
#define SHK_UNROLL_P 7
#endif
#ifndef SHK_ASM_WAVES
#define SHK_ASM_WAVES 2
#endif
// Timing ablations of the assembly kernel (skip the element computation / the slot phase / the field loads: results
// are WRONG by construction) exist only in probe builds (-DSHK_EXPERIMENTS, `make probe`): the shipping kernel has no
// such branch.
#ifdef SHK_EXPERIMENTS
#define SHK_ABLATE(bit) ((a.ablate & (bit)) != 0)
#else
#define SHK_ABLATE(bit) false
#endif
constexpr int kAsmFields = 13;   // x, y, N, N_n, b, qx, qy, z_b, z_s, G, melt_n, storage, inputs
enum { AF_X = 0, AF_Y, AF_N, AF_NN, AF_B, AF_QX, AF_QY, AF_ZB, AF_ZS, AF_G, AF_M, AF_S, AF_I };

struct QPoint { double f0, f1, f2, w; };   // barycentric weights of a quadrature point and 2 * its weight

// Difference of the head between two vertices from the differences of its coefficients (constitutive.py:6-9).
// Like FFCx, every coefficient is differenced on its own (reference gradient = nodal differences f1 - f0, f2 - f0):
// rounding is then relative to the differences, not to the head's magnitude (~1e3 m over cells of ~10 m), which
// puts the fp64 floor of ||F|| three orders of magnitude below what differencing nodal heads gives.
// (Divisions by constants of the run -- rho_w g, Lh -- are multiplications by their reciprocals here: an fp64 division is
//  a dozen instructions, a cell had thirteen of them, and the kernel is bound by fp64 issue.  One rounding differs.)
__device__ __forceinline__ double head_diff(double dzb, double dzs, double dN, const DevParams& p) {
    return dzb + p.ri_rw * (dzs - dzb) - dN * p.inv_rwg;
}

// sqrt(s) for s >= 0 well inside the double range (|q|^2): rsq, one coupled Goldschmidt step and one residual
// correction, without the compiler's range scaling (hipcc's sqrt: 15 instructions, this: 9).  <= 1 ulp on the
// inputs that matter; the result only enters through 1 + (omega / nu) sqrt(s).
__device__ __forceinline__ double sqrt_nn(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, s), h, g);
    return s > 0.0 ? g : 0.0;
}
// a / d for 1 <= d << 1e300: rcp + two Newton steps + one residual correction.  ~1 ulp.
__device__ __forceinline__ double div_ge1(double a, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-d, q, a), r, q);
}

// 1 / d for |d| well inside the double range (twice a cell's area): rcp + two Newton steps.  ~1 ulp.
__device__ __forceinline__ double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

// XCD-aware block map (MI355X: workgroups are dealt round-robin to 8 XCDs with private L2s): XCD x sweeps the
// contiguous block range [start(x), start(x+1)).  Placement changes speed only.
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int per = nb >> 3, rem = nb & 7, x = b & 7, i = b >> 3;
    return x * per + min(x, rem) + i;
}

struct CellOut { double K[9], F[3]; };

// JAC = false: the residual-only instance (the pass after a Newton update that is expected to be the last: shk_api.hip,
// shk_newton_solve): the element Jacobian is computed only for the few cells with a Dirichlet vertex, whose lifting
// needs it; o.K is otherwise left untouched.
template <int NQ, int NP, bool JAC>
__device__ __forceinline__ void cell_tensor(const AsmArgs& a, const double* __restrict__ fld, const uint8_t* __restrict__ bcf,
                                            const QPoint* __restrict__ qk, const QPoint* __restrict__ qp, int V,
                                            int l0, int l1, int l2, CellOut& o) {
    const DevParams& p = a.p;
#define LD(k, l) fld[(k) * V + (l)]
    const double x0 = LD(AF_X, l0), y0 = LD(AF_Y, l0);
    const double d1x = LD(AF_X, l1) - x0, d1y = LD(AF_Y, l1) - y0, d2x = LD(AF_X, l2) - x0, d2y = LD(AF_Y, l2) - y0;
    const double det = d1x * d2y - d1y * d2x;
    const double inv = rcp_nr(det);
    const double area = 0.5 * fabs(det);
    const double g1x = d2y * inv, g1y = -d2x * inv, g2x = -d1y * inv, g2y = d1x * inv;
    const double g0x = -(g1x + g2x), g0y = -(g1y + g2y);

    const double b0 = LD(AF_B, l0), b1 = LD(AF_B, l1), b2 = LD(AF_B, l2);
    const double qx0 = LD(AF_QX, l0), qx1 = LD(AF_QX, l1), qx2 = LD(AF_QX, l2);
    const double qy0 = LD(AF_QY, l0), qy1 = LD(AF_QY, l1), qy2 = LD(AF_QY, l2);
    // WaterFlux with the Reynolds switch, constitutive.py:11-20: q_w = -K grad(h); sK = int K dx
    double sK = 0.0;
    // (a P1 field at a quadrature point: f0 + (f1 - f0) phi1 + (f2 - f0) phi2 -- two FMAs per field and point)
    const double db1 = b1 - b0, db2 = b2 - b0, dqx1 = qx1 - qx0, dqx2 = qx2 - qx0, dqy1 = qy1 - qy0, dqy2 = qy2 - qy0;
    auto flux_point = [&](int k) {
        // built-in rule: k is a compile-time constant of the unrolled loop, so the point's weights come straight from the
        // kernel arguments through scalar loads and enter the FMAs as SGPR operands -- no LDS read and no vector
        // registers per point in flight (249 -> 201 VGPRs, -4 % time); a run-time table is broadcast from LDS
        const QPoint q = NQ > 0 ? QPoint{0.0, a.quad.phi1[k], a.quad.phi2[k], a.quad.w2[k]} : qk[k];
        const double bk = __builtin_fma(db2, q.f2, __builtin_fma(db1, q.f1, b0));
        const double qxk = __builtin_fma(dqx2, q.f2, __builtin_fma(dqx1, q.f1, qx0));
        const double qyk = __builtin_fma(dqy2, q.f2, __builtin_fma(dqy1, q.f1, qy0));
        const double qn = sqrt_nn(qxk * qxk + qyk * qyk);
        const double ab = fabs(bk);
        sK += div_ge1(q.w * (ab * ab * ab), 1.0 + p.om_nu * qn);
    };
    if constexpr (NQ > 0) {
#pragma unroll SHK_UNROLL_Q
        for (int k = 0; k < NQ; ++k) flux_point(k);
    } else {
        for (int k = 0; k < a.quad.nq; ++k) flux_point(k);
    }
    sK *= area * p.kcoef;

    const double N0 = LD(AF_N, l0), N1 = LD(AF_N, l1), N2 = LD(AF_N, l2);
    const double m0_ = LD(AF_M, l0), m1_ = LD(AF_M, l1), m2_ = LD(AF_M, l2);
    const double zb0 = LD(AF_ZB, l0), zs0 = LD(AF_ZS, l0);
    const double dh1 = head_diff(LD(AF_ZB, l1) - zb0, LD(AF_ZS, l1) - zs0, N1 - N0, p);
    const double dh2 = head_diff(LD(AF_ZB, l2) - zb0, LD(AF_ZS, l2) - zs0, N2 - N0, p);
    const double ghx = dh1 * g1x + dh2 * g2x, ghy = dh1 * g1y + dh2 * g2y;
    const double gbx = db1 * g1x + db2 * g2x, gby = db1 * g1y + db2 * g2y;
    const double gmx = (m1_ - m0_) * g1x + (m2_ - m0_) * g2x, gmy = (m1_ - m0_) * g1y + (m2_ - m0_) * g2y;
    const double gb2 = gbx * gbx + gby * gby;
    const double inv_den = div_ge1(1.0, 1.0 + gb2);
    const double gmgb = gmx * gbx + gmy * gby;
    const double Nn0 = LD(AF_NN, l0), Nn1 = LD(AF_NN, l1), Nn2 = LD(AF_NN, l2);
    const double G0 = LD(AF_G, l0), G1 = LD(AF_G, l1), G2 = LD(AF_G, l2);
    const double s0 = LD(AF_S, l0), s1 = LD(AF_S, l1), s2 = LD(AF_S, l2);
    const double i0 = LD(AF_I, l0), i1 = LD(AF_I, l1), i2 = LD(AF_I, l2);
#undef LD

    double F0 = 0.0, F1 = 0.0, F2 = 0.0;
    double T00 = 0.0, T01 = 0.0, T02 = 0.0, T11 = 0.0, T12 = 0.0, T22 = 0.0;
    const double qgh0 = p.rwg * (qx0 * ghx + qy0 * ghy), qgh1 = p.rwg * (qx1 * ghx + qy1 * ghy),
                 qgh2 = p.rwg * (qx2 * ghx + qy2 * ghy);  // rho_w g q.grad(h) is P1: interpolate its nodal values
    const double dN1 = N1 - N0, dN2 = N2 - N0, dNn1 = Nn1 - Nn0, dNn2 = Nn2 - Nn0, dG1 = G1 - G0, dG2 = G2 - G0;
    const double dm1 = m1_ - m0_, dm2 = m2_ - m0_, ds1 = s1 - s0, ds2 = s2 - s0, di1 = i1 - i0, di2 = i2 - i0;
    const double dg1 = qgh1 - qgh0, dg2 = qgh2 - qgh0;
    auto poly_point = [&](int k) {
        const QPoint q = NP > 0 ? QPoint{a.qpoly.phi0[k], a.qpoly.phi1[k], a.qpoly.phi2[k], a.qpoly.w2[k]} : qp[k];
        const double f0 = q.f0, f1 = q.f1, f2 = q.f2;
        const double w = q.w * area;
        const double Nk = __builtin_fma(dN2, f2, __builtin_fma(dN1, f1, N0));
        const double Nnk = __builtin_fma(dNn2, f2, __builtin_fma(dNn1, f1, Nn0));
        const double bk = __builtin_fma(db2, f2, __builtin_fma(db1, f1, b0));
        const double Gk = __builtin_fma(dG2, f2, __builtin_fma(dG1, f1, G0));
        const double mk = __builtin_fma(dm2, f2, __builtin_fma(dm1, f1, m0_));
        const double sk = __builtin_fma(ds2, f2, __builtin_fma(ds1, f1, s0));
        const double ik = __builtin_fma(di2, f2, __builtin_fma(di1, f1, i0));
        const double qghk = __builtin_fma(dg2, f2, __builtin_fma(dg1, f1, qgh0));
        // Melt, constitutive.py:22-27 (div of the cell-wise P1 product expanded)
        const double melt = (Gk - qghk) * p.inv_Lh + (mk * gb2 + bk * gmgb) * inv_den;
        const double pw = p.n_is_3 ? Nk * Nk : pow(fabs(Nk), p.n - 1.0);   // |N|^(n-1), constitutive.py:31
        const double closure = p.A * bk * Nk * pw;                 // constitutive.py:29-31
        const double stor = sk * (Nk - Nnk) * a.inv_rwg_dt;        // solvers.py:42
        const double ws = w * (p.c_m * melt - closure - stor - ik);
        F0 += ws * f0; F1 += ws * f1; F2 += ws * f2;
        if constexpr (JAC) {
            const double wd = w * (p.A * p.n * bk * pw + sk * a.inv_rwg_dt);
            T00 += wd * f0 * f0; T01 += wd * f0 * f1; T02 += wd * f0 * f2;
            T11 += wd * f1 * f1; T12 += wd * f1 * f2; T22 += wd * f2 * f2;
        }
    };
    if constexpr (NP > 0) {
#pragma unroll SHK_UNROLL_P
        for (int k = 0; k < NP; ++k) poly_point(k);
    } else {
        for (int k = 0; k < a.qpoly.nq; ++k) poly_point(k);
    }
    // flux term: K grad(h).grad(phi_i)
    double Fe0 = sK * (ghx * g0x + ghy * g0y) + F0;
    double Fe1 = sK * (ghx * g1x + ghy * g1y) + F1;
    double Fe2 = sK * (ghx * g2x + ghy * g2y) + F2;
    const int bc0 = bcf[l0], bc1 = bcf[l1], bc2 = bcf[l2];
    if constexpr (!JAC) {
        if (!(bc0 | bc1 | bc2)) { o.F[0] = Fe0; o.F[1] = Fe1; o.F[2] = Fe2; return; }
        // a cell with a Dirichlet vertex: its lifting needs the element Jacobian after all (a handful of cells per block)
        const int np_ = NP > 0 ? NP : a.qpoly.nq;
        for (int k = 0; k < np_; ++k) {
            const QPoint q = NP > 0 ? QPoint{a.qpoly.phi0[k], a.qpoly.phi1[k], a.qpoly.phi2[k], a.qpoly.w2[k]} : qp[k];
            const double f0 = q.f0, f1 = q.f1, f2 = q.f2, w = q.w * area;
            const double Nk = __builtin_fma(dN2, f2, __builtin_fma(dN1, f1, N0));
            const double bk = __builtin_fma(db2, f2, __builtin_fma(db1, f1, b0));
            const double sk = __builtin_fma(ds2, f2, __builtin_fma(ds1, f1, s0));
            const double pw = p.n_is_3 ? Nk * Nk : pow(fabs(Nk), p.n - 1.0);
            const double wd = w * (p.A * p.n * bk * pw + sk * a.inv_rwg_dt);
            T00 += wd * f0 * f0; T01 += wd * f0 * f1; T02 += wd * f0 * f2;
            T11 += wd * f1 * f1; T12 += wd * f1 * f2; T22 += wd * f2 * f2;
        }
    }
    const double kk = -sK * p.inv_rwg;
    const double d00 = g0x * g0x + g0y * g0y, d01 = g0x * g1x + g0y * g1y, d02 = g0x * g2x + g0y * g2y;
    const double d11 = g1x * g1x + g1y * g1y, d12 = g1x * g2x + g1y * g2y, d22 = g2x * g2x + g2y * g2y;
    // int phi_i q_x dx = area/12 (sum q_x + q_x,i): exact P1 mass matrix
    const double cq = p.cm_Lh * area * (1.0 / 12.0);
    const double sx = qx0 + qx1 + qx2, sy = qy0 + qy1 + qy2;
    const double Px0 = cq * (sx + qx0), Px1 = cq * (sx + qx1), Px2 = cq * (sx + qx2);
    const double Py0 = cq * (sy + qy0), Py1 = cq * (sy + qy1), Py2 = cq * (sy + qy2);
    o.K[0] = kk * d00 + (Px0 * g0x + Py0 * g0y) - T00;
    o.K[1] = kk * d01 + (Px0 * g1x + Py0 * g1y) - T01;
    o.K[2] = kk * d02 + (Px0 * g2x + Py0 * g2y) - T02;
    o.K[3] = kk * d01 + (Px1 * g0x + Py1 * g0y) - T01;
    o.K[4] = kk * d11 + (Px1 * g1x + Py1 * g1y) - T11;
    o.K[5] = kk * d12 + (Px1 * g2x + Py1 * g2y) - T12;
    o.K[6] = kk * d02 + (Px2 * g0x + Py2 * g0y) - T02;
    o.K[7] = kk * d12 + (Px2 * g1x + Py2 * g1y) - T12;
    o.K[8] = kk * d22 + (Px2 * g2x + Py2 * g2y) - T22;
    if (bc0 | bc1 | bc2) {
        // apply_lifting(alpha=-1): F_i += K_ij (g - N_j) over Dirichlet columns j
        const double e0 = bc0 ? a.bc_value - N0 : 0.0;
        const double e1 = bc1 ? a.bc_value - N1 : 0.0;
        const double e2 = bc2 ? a.bc_value - N2 : 0.0;
        Fe0 += o.K[0] * e0 + o.K[1] * e1 + o.K[2] * e2;
        Fe1 += o.K[3] * e0 + o.K[4] * e1 + o.K[5] * e2;
        Fe2 += o.K[6] * e0 + o.K[7] * e1 + o.K[8] * e2;
    }
    o.F[0] = Fe0; o.F[1] = Fe1; o.F[2] = Fe2;
}

// T threads per workgroup; NQ / NP: points of the two rules when they are the built-in ones (15 / 7: loops fully
// unrolled), 0 = run-time counts (a user table from shk_set_quadrature, or Glen's n != 3).
// (the residual-only instance stages 3 instead of 12 tensor entries per cell and needs 165 registers: three workgroups
//  per CU instead of two -- its LDS region is sized separately, Ctx::asm_lds_res)
template <int T, int NQ, int NP, bool JAC = true>
__global__ __launch_bounds__(T, (JAC || NQ == 0) ? SHK_ASM_WAVES : SHK_ASM_WAVES + 1) void k_assemble(const AsmArgs a) {
    constexpr int R = (kAsmCellsMax + T - 1) / T;   // cells per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = a.cells_max, V = a.verts_max;
    double* fld = reinterpret_cast<double*>(smem);                  // [13][V] staged fields ...
    double* et = fld;                                               // ... later [12][E] element tensors (same region)
    constexpr int kF = JAC ? 9 : 0;                                 // tensor row of the element residual (residual-only: [3][E])
    QPoint* qk = reinterpret_cast<QPoint*>(smem + a.lds_region_a);
    QPoint* qp = qk + kMaxQuad;
    int* sp = reinterpret_cast<int*>(qp + kMaxQuad);                // [slices_max+1] SELL ptr of owned slices
    int* ip = sp + (a.slices_max + 1);                              // [rows+1] incptr of owned rows
    uint16_t* ic = reinterpret_cast<uint16_t*>(ip + (a.slices_max * kSlice + 1));  // incidence codes
    uint8_t* bcf = reinterpret_cast<uint8_t*>(ic + a.inc_max);      // [V] Dirichlet flags of the staged vertices

    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    // A block has ~584 cells and ~2100 slots: the last trip of the cell loop (and of the slot loop) occupies only the first
    // one or two waves, i.e. always the SAME two SIMDs -- of both workgroups of a CU, whose waves are dealt to the SIMDs in
    // the same order.  Every other generation of workgroups (blockIdx / 256: the two residents of a CU come from adjacent
    // generations) therefore starts its loops two waves further on, so that the long and the short waves of the two
    // residents share SIMDs.  Which thread computes which cell / slot changes; no result does.
    const int rtid = (tid + (((blockIdx.x >> 8) & 1) << 7)) & (T - 1);
    // one descriptor (scalar loads) names everything this workgroup fetches: all its other loads depend on nothing else
    const int32_t* __restrict__ dsc = a.blk_desc + (size_t)kBlkDesc * blk;
    const int s0 = dsc[0], ns = dsc[1], c0 = dsc[2], ncell = dsc[3], h0 = dsc[4], nhalo = dsc[5];
    const int n0 = dsc[6], n1 = dsc[7], ip0 = dsc[8], ninc = dsc[9];
    const int r0 = s0 * kSlice;
    const int r1 = min(a.A.n_rows, (s0 + ns) * kSlice);
    const int nrows = r1 - r0;

    // ---- phase 0: stage plan slices, quadrature tables and the fields of the block's vertices ----
    // the cells' vertex ids are requested with the fields (they depend on the descriptor only), not after the barrier
    const ushort4* __restrict__ cellv = reinterpret_cast<const ushort4*>(a.blk_cellv) + c0;
    ushort4 cvw[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        cvw[r] = (!SHK_ABLATE(8) && rtid + r * T < ncell) ? cellv[rtid + r * T] : make_ushort4(0, 0, 0, 0);
    for (int i = tid; i <= ns; i += T) sp[i] = a.A.ptr[s0 + i];
    for (int i = tid; i <= nrows; i += T) ip[i] = a.incptr[r0 + i];
    for (int i = tid; i < ninc; i += T) ic[i] = a.inccode[ip0 + i];
    if constexpr (NQ == 0 || NP == 0) {   // run-time rules only: the built-in ones are read as scalar arguments
        if (tid < a.quad.nq) qk[tid] = QPoint{a.quad.phi0[tid], a.quad.phi1[tid], a.quad.phi2[tid], a.quad.w2[tid]};
        if (tid >= 64 && tid - 64 < a.qpoly.nq) {
            const int k = tid - 64;
            qp[k] = QPoint{a.qpoly.phi0[k], a.qpoly.phi1[k], a.qpoly.phi2[k], a.qpoly.w2[k]};
        }
    }
    // (all global loads of a thread -- its own row and up to kHaloPer halo vertices -- are issued before the first LDS
    //  write, so their latencies overlap instead of adding up)
    {
        constexpr int kStage = (kAsmVertsMax + T - 1) / T;
        double sv[kStage][kAsmFields];
        uint8_t sb[kStage];
#pragma unroll
        for (int r = 0; r < kStage; ++r) {
            const int i = tid + r * T;
            if (i < nrows + nhalo && !SHK_ABLATE(4)) {
                const int v = i < nrows ? r0 + i : a.blk_halo[h0 + (i - nrows)];   // own rows: consecutive -> coalesced
                const double2 xy = a.m.xy[v];
                sv[r][0] = xy.x;
                sv[r][1] = xy.y;
#pragma unroll
                for (int k = 0; k < kAsmFields - 2; ++k) sv[r][k + 2] = a.fld[k][v];
                sb[r] = a.bcflag ? a.bcflag[v] : (uint8_t)0;
            }
        }
#pragma unroll
        for (int r = 0; r < kStage; ++r) {
            const int i = tid + r * T;
            if (i < nrows + nhalo) {
#pragma unroll
                for (int k = 0; k < kAsmFields; ++k) fld[k * V + i] = SHK_ABLATE(4) ? 1.0 + 0.001 * k + 1e-5 * i : sv[r][k];
                bcf[i] = SHK_ABLATE(4) ? (uint8_t)0 : sb[r];
            }
        }
    }
    __syncthreads();

    // ---- phase 1: one thread per cell touching the owned rows; tensors stay in registers until all fields are read ----
    CellOut out[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = rtid + r * T;
        if (t < ncell) {
            const ushort4 cv = SHK_ABLATE(8) ? cellv[t] : cvw[r];
            if (SHK_ABLATE(1)) {
#pragma unroll
                for (int k = 0; k < 9; ++k) out[r].K[k] = fld[cv.x] + k;
#pragma unroll
                for (int k = 0; k < 3; ++k) out[r].F[k] = fld[cv.y] + fld[cv.z];
            } else {
                cell_tensor<NQ, NP, JAC>(a, fld, bcf, qk, qp, V, cv.x, cv.y, cv.z, out[r]);
            }
        }
    }
    // the plan words of this thread's slots (phase 2) are requested here -- after the element computation, whose
    // registers they would otherwise crowd (they cost 80 B per lane of scratch spills = 0.8 GB of traffic per pass when
    // requested at kernel start; with the 48 registers the scalar quadrature tables freed, still 1.74 against 1.70 ms)
    // -- and arrive while the tensors go to LDS
    constexpr int kSlotIt = JAC ? (kAsmSlotsMax + T - 1) / T : 1;
    uint32_t srcw[kSlotIt];
    if constexpr (JAC) {
#pragma unroll
        for (int r = 0; r < kSlotIt; ++r) {
            const int s = n0 + rtid + r * T;
            srcw[r] = s < n1 ? a.slotsrc[s] : kSrcEmpty;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = rtid + r * T;
        if (t < ncell) {
            if constexpr (JAC) {
#pragma unroll
                for (int k = 0; k < 9; ++k) et[k * E + t] = out[r].K[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) et[(kF + k) * E + t] = out[r].F[k];
        }
    }
    __syncthreads();
    auto tensor = [&](int k, int cell) -> double { return et[k * E + cell]; };

    // ---- phase 2b: one thread per owned residual row (first: it needs LDS only, while the plan words of phase 2a
    //      are still in flight) ----
    for (int i = tid; i < nrows; i += T) {
        const int v = r0 + i;
        double sum = 0.0;
        const int kb = ip[i] - ip0, ke = ip[i + 1] - ip0;
        for (int q = kb; q < ke; ++q) {
            const int code = ic[q];
            sum += tensor(kF + (code & 3), code >> 2);
        }
        if (bcf[i]) sum = a.fld[0][v] - a.bc_value;  // set_bc(b, bcs, x, -1): F = N - g
        a.F[v] = sum;
    }
    // ---- phase 2a: one thread per SELL slot of the owned slices (not in the residual-only instance) ----
    if constexpr (!JAC) return;
    if (!SHK_ABLATE(2)) {
        // first slot of the block's 2nd .. 4th slice (INT_MAX when absent): a slot's slice by three compares
        const int sp1 = ns > 1 ? sp[1] : 0x7FFFFFFF, sp2 = ns > 2 ? sp[2] : 0x7FFFFFFF, sp3 = ns > 3 ? sp[3] : 0x7FFFFFFF;
#pragma unroll
        for (int r = 0; r < kSlotIt; ++r) {
            const int s = n0 + rtid + r * T;
            if (s >= n1) break;
            const int j = (s >= sp1) + (s >= sp2) + (s >= sp3);
            const int off = s - (j == 0 ? n0 : j == 1 ? sp1 : j == 2 ? sp2 : sp3);
            const int k = off >> 6, lane = off & 63;
            const int v = (s0 + j) * kSlice + lane;
            const uint32_t src = srcw[r];
            double sum = 0.0;
            if (k == 0) {
                if (v < a.A.n_rows) {
                    const int i = v - r0;
                    const int kb = ip[i] - ip0, ke = ip[i + 1] - ip0;
                    for (int q = kb; q < ke; ++q) {  // ascending cell id: fixed summation order
                        const int code = ic[q];
                        sum += tensor(4 * (code & 3), code >> 2);   // K_ii of that cell: li*3 + li
                    }
                }
            } else {
                const uint32_t lo = src & 0x3FFFu, hi = (src >> 14) & 0x3FFFu;
                if ((lo >> 4) != kSrcNone) sum = tensor(lo & 15u, lo >> 4);
                if ((hi >> 4) != kSrcNone) sum += tensor(hi & 15u, hi >> 4);
            }
            const uint32_t bc = src >> 28;   // Dirichlet rows and columns zeroed, unit diagonal
            if (bc) sum = (bc == 2u) ? 1.0 : 0.0;
            if (k == 0 && v < a.A.n_rows) a.dinv[v] = (sum != 0.0) ? 1.0 / sum : 1.0;  // the diagonal is stored first
            a.vals[s] = sum;  // padding slots hold exact zeros
        }
    }
}

// Per-slot Dirichlet code (0 keep, 1 zero, 2 one) into bits 28-29 of the slot's plan word, so that the assembly
// neither gathers flags per entry nor streams a second per-slot array.
__global__ __launch_bounds__(kBlock) void k_slot_bc(DevSell A, const uint8_t* __restrict__ flag, uint32_t* __restrict__ slotsrc) {
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < A.n_rows; v += gridDim.x * kBlock) {
        const int s = v / kSlice, l = v % kSlice, base = A.ptr[s];
        const bool bv = flag[v];
        for (int k = 0; k < (int)A.rowlen[v]; ++k) {
            const int slot = base + k * kSlice + l;
            const int u = A.col[slot];
            const bool bu = flag[u];
            const uint32_t code = (bv | bu) ? ((u == v && bv) ? 2u : 1u) : 0u;
            slotsrc[slot] = (slotsrc[slot] & 0x0FFFFFFFu) | (code << 28);
        }
    }
}

void launch_slot_bc(Ctx* c) {
    const int g = (int)std::min<int64_t>((c->n_own + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(k_slot_bc, dim3(g), dim3(kBlock), 0, c->stream, c->sell(), c->d_bcflag, c->d_slotsrc);
}

static void fill_asm_args(Ctx* c, double dt, AsmArgs& a) {
    a.m.xy = c->d_xy;
    a.m.cells = c->d_cells;
    const int order[kAsmFields - 2] = {SHK_N, SHK_N_N, SHK_B, SHK_QX, SHK_QY, SHK_Z_B, SHK_Z_S, SHK_G, SHK_MELT_N,
                                       SHK_STORAGE, SHK_INPUTS};
    for (int k = 0; k < kAsmFields - 2; ++k) a.fld[k] = c->f[order[k]];
    a.bcflag = c->has_bc ? c->d_bcflag : nullptr;
    a.slotsrc = c->d_slotsrc;
    a.bc_value = c->bc_value;
    a.inv_rwg_dt = 1.0 / (c->dp.rwg * dt);
    a.A = c->sell();
    a.blk_desc = c->d_blk_desc; a.blk_halo = c->d_blk_halo; a.blk_cellv = c->d_blk_cellv;
    a.incptr = c->d_incptr; a.inccode = c->d_inccode;
    a.cells_max = c->plan.cells_max; a.slices_max = c->plan.slices_max; a.verts_max = c->plan.verts_max;
    a.inc_max = c->plan.max_inc_per_block;
    a.lds_region_a = (int)c->asm_region_a;   // (launch_assemble overrides it for the residual-only instance)
#ifdef SHK_EXPERIMENTS
    a.ablate = tunables().asm_ablate;   // probe builds only
#endif
    a.F = c->d_F; a.vals = c->d_vals; a.dinv = c->d_dinv;
    a.p = c->dp;
    a.quad = c->quad;
    a.qpoly = c->dp.n_is_3 ? c->qpoly5 : c->quad;
}

// LDS of one assembly workgroup: region A (staged fields, then the element tensors) + tables
size_t assemble_lds_bytes(const HostPlan& P, size_t* region_a, bool residual_only) {
    const size_t E = P.cells_max, S = P.slices_max, V = P.verts_max;
    size_t ra = std::max((size_t)kAsmFields * V, (residual_only ? 3 : 12) * E) * sizeof(double);
    ra = (ra + 15) & ~size_t(15);
    if (region_a) *region_a = ra;
    size_t lds = ra + 2 * kMaxQuad * sizeof(QPoint) + (S + 1) * sizeof(int) + (S * kSlice + 1) * sizeof(int) +
                 (size_t)P.max_inc_per_block * sizeof(uint16_t) + V;
    return (lds + 15) & ~size_t(15);
}

void launch_assemble(Ctx* c, double dt, bool residual_only) {
    AsmArgs a;
    fill_asm_args(c, dt, a);
    // host-side check of what the kernel's fixed loop counts assume (a violation would write outside its LDS)
    if (a.cells_max > kAsmCellsMax || a.verts_max > kAsmVertsMax) { set_error("assembly plan exceeds the kernel's staging limits"); return; }
    const bool builtin = a.quad.nq == 15 && a.qpoly.nq == 7;
    if (residual_only) {   // no plan words, no values, no 1/diag
        note_bytes(c, c->asm_bytes - 12.0 * (double)c->slots - 8.0 * (double)c->n_own);
        a.lds_region_a = (int)c->asm_region_a_res;
        if (builtin) launch_phase(c, SHK_PH_ASSEMBLE, k_assemble<kBlock, 15, 7, false>, dim3(c->nblk), dim3(kBlock), c->asm_lds_res, a);
        else launch_phase(c, SHK_PH_ASSEMBLE, k_assemble<kBlock, 0, 0, false>, dim3(c->nblk), dim3(kBlock), c->asm_lds_res, a);
        return;
    }
    note_bytes(c, c->asm_bytes);
    if (builtin) launch_phase(c, SHK_PH_ASSEMBLE, k_assemble<kBlock, 15, 7>, dim3(c->nblk), dim3(kBlock), c->asm_lds, a);
    else launch_phase(c, SHK_PH_ASSEMBLE, k_assemble<kBlock, 0, 0>, dim3(c->nblk), dim3(kBlock), c->asm_lds, a);
}

// Dynamic LDS above 64 KiB has to be requested per kernel.
hipError_t prepare_kernels(Ctx* c) {
    const void* fns[] = {reinterpret_cast<const void*>(&k_assemble<kBlock, 15, 7>), reinterpret_cast<const void*>(&k_assemble<kBlock, 0, 0>),
                         reinterpret_cast<const void*>(&k_assemble<kBlock, 15, 7, false>),
                         reinterpret_cast<const void*>(&k_assemble<kBlock, 0, 0, false>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->asm_lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace shk
