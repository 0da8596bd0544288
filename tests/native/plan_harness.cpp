// Host-side checks of shakti_fenics_amd/csrc/shk_plan.cpp (no GPU): built and run by tests/test_plan_host.py.
//   plan_harness mesh.bin  ->  one "key value" line per check on stdout, exit code 0 if every check holds
// mesh.bin: int64 nv, int64 ne, double xy[2 nv], int32 cells[3 ne]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "shk_plan.h"
using namespace shk;

static int fails = 0;
#define CHECK(name, cond)                                   \
    do {                                                    \
        const bool ok_ = (cond);                            \
        std::printf("%s %s\n", name, ok_ ? "ok" : "FAIL");  \
        if (!ok_) ++fails;                                  \
    } while (0)

// coarse operator of one transfer as a map (I, J) -> value, by applying its gather plan to fine values
static std::map<std::pair<int, int>, double> apply_plan(const AmgLevelPlan& L, const SellPattern& C,
                                                        const std::vector<double>& fine) {
    std::map<std::pair<int, int>, double> out;
    for (int I = 0; I < C.n_rows; ++I) {
        const int s = I / kSlice, l = I % kSlice, base = C.ptr[s];
        for (int k = 0; k < C.rowlen[I]; ++k) {
            const int slot = base + k * kSlice + l;
            double a = 0.0;
            for (int q = L.gptr[slot]; q < L.gptr[slot + 1]; ++q) a += fine[L.glist[q]];
            out[{I, C.col[slot]}] = a;
        }
    }
    return out;
}

// Mode 2 (plan_harness --parts part0.bin part1.bin ...): the host side of amg_setup_distributed's first level for a
// P-way decomposition, as shk_amg.hip drives it -- per-subdomain plans, the neighbours' aggregate ids (here looked up
// through the global vertex ids instead of a ghost exchange), 1024-aligned blocks of the replicated global level with
// dummy identity rows, coarse_rows of every subdomain concatenated into one SELL pattern, coarsen_onto_global per
// subdomain.  part file: int64 n_own, n_loc, ne, nglobal; double xy[2 n_loc]; int32 cells[3 ne]; int64 gid[n_loc].
static int run_parts(int argc, char** argv) {
    struct Part { int64_t n_own, n_loc, ne; std::vector<double> xy; std::vector<int32_t> cells; std::vector<int64_t> gid; HostPlan P; };
    const int R = argc - 2;
    std::vector<Part> parts(R);
    int64_t nglob = 0;
    for (int r = 0; r < R; ++r) {
        Part& q = parts[r];
        FILE* f = std::fopen(argv[2 + r], "rb");
        if (!f) return 2;
        int64_t h[4];
        if (std::fread(h, 8, 4, f) != 4) return 2;
        q.n_own = h[0]; q.n_loc = h[1]; q.ne = h[2]; nglob = h[3];
        q.xy.resize(2 * q.n_loc); q.cells.resize(3 * q.ne); q.gid.resize(q.n_loc);
        if (std::fread(q.xy.data(), 8, 2 * q.n_loc, f) != (size_t)(2 * q.n_loc) ||
            std::fread(q.cells.data(), 4, 3 * q.ne, f) != (size_t)(3 * q.ne) ||
            std::fread(q.gid.data(), 8, q.n_loc, f) != (size_t)q.n_loc) return 2;
        std::fclose(f);
        PlanOptions opt;
        const std::string err = build_plan(q.n_own, q.n_loc, q.ne, q.xy.data(), q.cells.data(), opt, q.P);
        if (!err.empty()) { std::printf("build_plan FAIL %s\n", err.c_str()); return 1; }
    }
    CHECK("build_plans", true);
    // owner rank and the owner's aggregate of every global vertex
    std::vector<int32_t> g_rank(nglob, -1), g_agg(nglob, -1);
    std::vector<int32_t> nc(R), offs(R + 1, 0);
    for (int r = 0; r < R; ++r) {
        const Part& q = parts[r];
        nc[r] = (int32_t)((q.n_own + 3) / 4);
        offs[r + 1] = offs[r] + ((nc[r] + 1023) / 1024) * 1024;
        for (int64_t e = 0; e < q.n_own; ++e) {   // external local id e (owned)
            const int32_t in = q.P.iperm[e];
            g_rank[q.gid[e]] = r;
            g_agg[q.gid[e]] = q.P.krank[in] / 4;
        }
    }
    bool owners = true;
    for (int64_t g = 0; g < nglob; ++g) owners = owners && g_rank[g] >= 0;
    CHECK("every_vertex_has_an_owner", owners);
    // every subdomain's rows of the global level, then the pattern all of them agree on
    std::vector<int32_t> grp(1, 0), gci;
    std::vector<std::vector<int32_t>> aggs(R), colmaps(R);
    bool rows_ok = true;
    for (int r = 0; r < R; ++r) {
        const Part& q = parts[r];
        aggs[r].resize(q.n_own);
        colmaps[r].assign(q.n_loc, -1);
        for (int64_t i = 0; i < q.n_own; ++i) { aggs[r][i] = q.P.krank[i] / 4; colmaps[r][i] = offs[r] + aggs[r][i]; }
        for (int64_t j = q.n_own; j < q.n_loc; ++j) {
            const int64_t g = q.gid[q.P.perm[j]];
            colmaps[r][j] = offs[g_rank[g]] + g_agg[g];
        }
        std::vector<int32_t> rp, ci;
        const std::string e = coarse_rows(q.P.A, aggs[r], colmaps[r], nc[r], offs[r], rp, ci);
        rows_ok = rows_ok && e.empty();
        if (!e.empty()) { std::printf("coarse_rows_error %s\n", e.c_str()); break; }
        for (int32_t I = 0; I < nc[r]; ++I) {
            for (int32_t k = rp[I]; k < rp[I + 1]; ++k) gci.push_back(ci[k]);
            grp.push_back((int32_t)gci.size());
        }
        for (int32_t I = offs[r] + nc[r]; I < offs[r + 1]; ++I) { gci.push_back(I); grp.push_back((int32_t)gci.size()); }   // dummy rows
    }
    CHECK("coarse_rows", rows_ok);
    if (!rows_ok) return 1;
    SellPattern G;
    std::vector<int32_t> gdiag;
    const std::string eg = sell_from_csr(offs[R], offs[R], grp, gci, G, gdiag);
    CHECK("global_level_pattern", eg.empty());
    if (!eg.empty()) return 1;
    bool diag_ok = true;
    for (int32_t I = 0; I < offs[R]; ++I) diag_ok = diag_ok && G.col[gdiag[I]] == I;
    CHECK("global_level_diag_slots", diag_ok);
    // transfers: every stored fine entry of every subdomain lands in exactly one slot of G, and with unit fine values the
    // global level sums to the number of entries of the whole fine matrix
    bool cover = true, tr_ok = true;
    double total = 0.0, expect = 0.0;
    std::vector<int> ghit(G.slots, 0);
    for (int r = 0; r < R && tr_ok; ++r) {
        const Part& q = parts[r];
        AmgLevelPlan L;
        L.with_ap = true;
        const std::string e = coarsen_onto_global(q.P.A, aggs[r], colmaps[r], nc[r], offs[r], G, L);
        tr_ok = e.empty();
        if (!tr_ok) { std::printf("coarsen_onto_global_error %s\n", e.c_str()); break; }
        std::vector<int> hit(q.P.A.slots, 0);
        for (size_t s2 = 0; s2 + 1 < L.gptr.size(); ++s2)
            for (int32_t k = L.gptr[s2]; k < L.gptr[s2 + 1]; ++k) { hit[L.glist[k]]++; ghit[s2]++; total += 1.0; }
        for (int i = 0; i < q.P.A.n_rows; ++i) {
            const int sl = i / kSlice, l = i % kSlice, base = q.P.A.ptr[sl];
            for (int k = 0; k < q.P.A.rowlen[i]; ++k) cover = cover && hit[base + k * kSlice + l] == 1;
        }
        expect += (double)q.P.A.nnz;
        cover = cover && (int64_t)L.gptr.size() == G.slots + 1;
        // A*P of the subdomain: columns are global coarse ids
        for (int32_t cidx : L.AP.col) cover = cover && cidx >= 0 && cidx < offs[R];
    }
    CHECK("onto_global_builds", tr_ok);
    CHECK("onto_global_covers_every_entry_once", cover);
    CHECK("global_level_sums_to_the_fine_matrix", total == expect);
    std::vector<int32_t> ident(offs[R]);
    for (int i = 0; i < offs[R]; ++i) ident[i] = i;
    std::vector<AmgLevelPlan> rep;
    PlanOptions o2;
    o2.amg_cost_nnz = expect;
    CHECK("replicated_hierarchy_builds", build_amg_levels(G, ident, o2, rep).empty() && !rep.empty() && rep.back().dense);
    std::printf("global_rows %d\nrep_levels %zu\n", offs[R], rep.size());
    return fails ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (std::string(argv[1]) == "--parts") return run_parts(argc, argv);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t nv = 0, ne = 0;
    if (std::fread(&nv, 8, 1, f) != 1 || std::fread(&ne, 8, 1, f) != 1) return 2;
    std::vector<double> xy(2 * nv);
    std::vector<int32_t> cells(3 * ne);
    if (std::fread(xy.data(), 8, 2 * nv, f) != (size_t)(2 * nv) || std::fread(cells.data(), 4, 3 * ne, f) != (size_t)(3 * ne)) return 2;
    std::fclose(f);

    PlanOptions opt;
    HostPlan P;
    const std::string err = build_plan(nv, nv, ne, xy.data(), cells.data(), opt, P);
    CHECK("build_plan", err.empty());
    if (!err.empty()) { std::printf("error %s\n", err.c_str()); return 1; }
    std::printf("nnz %lld\nslots %lld\nlevels %zu\nmax_row_len %d\n", (long long)P.A.nnz, (long long)P.A.slots, P.amg.size(),
                P.A.max_row_len);

    // permutation is a bijection; SELL rows hold the diagonal first and distinct columns
    {
        std::vector<int> seen(nv, 0);
        for (int64_t i = 0; i < nv; ++i) seen[P.perm[i]]++;
        bool bij = true;
        for (int64_t i = 0; i < nv; ++i) bij = bij && seen[i] == 1 && P.iperm[P.perm[i]] == i;
        CHECK("perm_bijection", bij);
        bool diag_first = true, distinct = true;
        for (int i = 0; i < P.A.n_rows; ++i) {
            const int s = i / kSlice, l = i % kSlice, base = P.A.ptr[s];
            diag_first = diag_first && P.A.col[base + l] == i;
            std::vector<int> c;
            for (int k = 0; k < P.A.rowlen[i]; ++k) c.push_back(P.A.col[base + k * kSlice + l]);
            std::sort(c.begin(), c.end());
            distinct = distinct && std::adjacent_find(c.begin(), c.end()) == c.end();
        }
        CHECK("sell_diag_first", diag_first);
        CHECK("sell_distinct_columns", distinct);
    }
    // order inside a 256-row window: rows by length, longest first, and within one length the rows with a column outside
    // the window (the rim: what other windows gather) before the interior ones
    {
        bool by_length = true, rim_first = true;
        const int W = 256;
        for (int w0 = 0; w0 < P.A.n_rows; w0 += W) {
            const int w1 = std::min(P.A.n_rows, w0 + W);
            int prev_len = 1 << 30, seen_interior_of_len = 0;
            for (int i = w0; i < w1; ++i) {
                const int s = i / kSlice, l = i % kSlice, base = P.A.ptr[s];
                const int len = P.A.rowlen[i];
                bool rim = false;
                for (int k = 0; k < len; ++k) {
                    const int c = P.A.col[base + k * kSlice + l];
                    rim = rim || c < w0 || c >= w1;
                }
                by_length = by_length && len <= prev_len;
                if (len != prev_len) seen_interior_of_len = 0;
                if (!rim) seen_interior_of_len = 1;
                else rim_first = rim_first && !seen_interior_of_len;
                prev_len = len;
            }
        }
        CHECK("window_rows_sorted_by_length", by_length);
        CHECK("window_rim_rows_first_within_a_length", rim_first);
    }
    // every fine slot of every transfer lands in exactly one coarse slot; aggregates have 1..4 members
    {
        bool cover = true, members = true;
        const SellPattern* Af = &P.A;
        for (const AmgLevelPlan& L : P.amg) {
            std::vector<int> hit(Af->slots, 0);
            for (int32_t g : L.glist) hit[g]++;
            for (int i = 0; i < Af->n_rows; ++i) {
                const int s = i / kSlice, l = i % kSlice, base = Af->ptr[s];
                for (int k = 0; k < Af->rowlen[i]; ++k) cover = cover && hit[base + k * kSlice + l] == 1;
            }
            for (int I = 0; I < L.n_coarse; ++I) {
                int cnt = 0;
                for (int m = 0; m < 4; ++m) cnt += L.members[4 * I + m] >= 0;
                members = members && cnt >= 1 && L.members[4 * I] >= 0;
            }
            if (L.dense) break;
            Af = &L.Ac;
        }
        CHECK("galerkin_plans_cover_every_entry_once", cover);
        CHECK("aggregates_have_members", members);
    }
    // fused multi-sweep smoother plans (build_sweep_plan): for every sparse level with a plan, (1) the local index maps
    // are consistent with the level's columns, (2) rings are closed (every column of a row that is recomputed is
    // addressable), and (3) the temporally blocked evaluation -- four damped-Jacobi sweeps computed block by block on
    // shrinking sets S3 > S2 > S1 > S0 from the plan's arrays alone -- equals four global sweeps (the first on A*P)
    {
        bool maps_ok = true, exact = true;
        int planned = 0;
        std::mt19937_64 rng(17);
        std::uniform_real_distribution<double> U(0.1, 1.0);
        for (size_t l = 0; l + 1 < P.amg.size(); ++l) {
            if (P.amg[l].dense || !P.amg[l + 1].with_ap) break;
            const SellPattern& A = P.amg[l].Ac;          // level l+1's operator
            const AmgLevelPlan& T = P.amg[l + 1];        // its transfer: A*P pattern + aggregates
            const SellPattern& Q = T.AP;
            SweepPlan S;
            CHECK("sweep_plan_builds", build_sweep_plan(A, Q, S).empty());
            if (S.nblk == 0) continue;
            ++planned;
            const int n = A.n_rows, nc = T.n_coarse_cols, W = S.width;
            // random level data; A made diagonally dominant so that the sweeps stay bounded
            std::vector<double> av(A.slots, 0.0), qv(Q.slots, 0.0), r(n), e(nc), dinv(n);
            for (int i = 0; i < n; ++i) {
                const int s = i / kSlice, li = i % kSlice, base = A.ptr[s];
                double off = 0.0;
                for (int k = 1; k < A.rowlen[i]; ++k) { av[base + k * kSlice + li] = -U(rng); off += std::fabs(av[base + k * kSlice + li]); }
                av[base + li] = off + 1.0;
                dinv[i] = 1.0 / av[base + li];
                const int qb = Q.ptr[s];
                for (int k = 0; k < Q.rowlen[i]; ++k) qv[qb + k * kSlice + li] = U(rng) - 0.5;
                r[i] = U(rng);
            }
            for (double& v : e) v = U(rng) - 0.5;
            const double w[4] = {0.4, 1.1, 0.5, 0.8}, alpha = 1.5;
            // reference: four global sweeps
            std::vector<double> x(n), y(n);
            auto arow = [&](int i, const std::vector<double>& xin) {
                const int s = i / kSlice, li = i % kSlice, base = A.ptr[s];
                double sum = 0.0;
                for (int k = 0; k < A.rowlen[i]; ++k) sum += av[base + k * kSlice + li] * xin[A.col[base + k * kSlice + li]];
                return sum;
            };
            for (int i = 0; i < n; ++i) {
                const int s = i / kSlice, li = i % kSlice, qb = Q.ptr[s];
                double sum = 0.0;
                for (int k = 0; k < Q.rowlen[i]; ++k) sum += qv[qb + k * kSlice + li] * e[Q.col[qb + k * kSlice + li]];
                x[i] = alpha * e[T.agg[i]] + w[0] * dinv[i] * (r[i] - alpha * sum);
            }
            for (int sw = 1; sw < 4; ++sw) {
                for (int i = 0; i < n; ++i) y[i] = x[i] + w[sw] * dinv[i] * (r[i] - arow(i, x));
                x.swap(y);
            }
            // blocked evaluation from the plan
            std::vector<double> out(n, 0.0);
            for (int b = 0; b < S.nblk && maps_ok; ++b) {
                const int r0 = b * kSweepRows, n0 = std::min(n - r0, kSweepRows);
                const int32_t* hd = &S.hdr[8 * (size_t)b];
                const int nS1 = kSweepRows + hd[2], nS2 = nS1 + hd[3], nS3 = nS2 + hd[4], nfix = hd[5];
                maps_ok = maps_ok && nS2 <= kSweepMaxS2 && nS3 <= kSweepMaxS3 && nS3 + nfix <= kSweepMaxLocal && nfix == 0;
                const int32_t* info = &S.ext_info[4 * (size_t)hd[0]];
                std::vector<int> grow(nS3, -1);                    // local id -> level row
                for (int t = 0; t < n0; ++t) grow[t] = r0 + t;
                for (int t = kSweepRows; t < nS3; ++t) grow[t] = info[4 * (t - kSweepRows)];
                std::vector<double> xa(nS3, 0.0), xb(nS3, 0.0);
                for (int t = 0; t < nS3; ++t) {
                    const int g = grow[t];
                    if (g < 0) continue;
                    const int pbase = t < n0 ? Q.ptr[g / kSlice] + g % kSlice : info[4 * (t - kSweepRows) + 2];
                    const int plen = t < n0 ? Q.rowlen[g] : info[4 * (t - kSweepRows) + 3] >> 8;
                    maps_ok = maps_ok && pbase == Q.ptr[g / kSlice] + g % kSlice && plen == Q.rowlen[g];
                    double sum = 0.0;
                    for (int k = 0; k < plen; ++k) sum += qv[pbase + k * kSlice] * e[Q.col[pbase + k * kSlice]];
                    xa[t] = alpha * e[T.agg[g]] + w[0] * dinv[g] * (r[g] - alpha * sum);
                }
                auto sweep = [&](const std::vector<double>& xin, std::vector<double>& xout, int upto, double ww) {
                    for (int t = 0; t < upto; ++t) {
                        const int g = grow[t];
                        if (g < 0) continue;
                        double sum = 0.0;
                        if (t < n0) {
                            const int base = A.ptr[g / kSlice] + g % kSlice, wid = (A.ptr[g / kSlice + 1] - A.ptr[g / kSlice]) / kSlice;
                            for (int k = 0; k < wid; ++k) {
                                const int lc = S.lcol_own[base + k * kSlice];
                                if (k < A.rowlen[g]) maps_ok = maps_ok && lc < nS3 && grow[lc] == A.col[base + k * kSlice];
                                sum += av[base + k * kSlice] * xin[lc];
                            }
                        } else {
                            const int base = info[4 * (t - kSweepRows) + 1], len = info[4 * (t - kSweepRows) + 3] & 255;
                            const uint16_t* lc = &S.ring_lcol[((size_t)hd[1] + (t - kSweepRows)) * W];
                            maps_ok = maps_ok && base == A.ptr[g / kSlice] + g % kSlice && len == A.rowlen[g];
                            for (int k = 0; k < len; ++k) {
                                maps_ok = maps_ok && lc[k] < nS3 && grow[lc[k]] == A.col[base + k * kSlice];
                                sum += av[base + k * kSlice] * xin[lc[k]];
                            }
                        }
                        xout[t] = xin[t] + ww * dinv[g] * (r[g] - sum);
                    }
                };
                sweep(xa, xb, nS2, w[1]);
                sweep(xb, xa, nS1, w[2]);
                sweep(xa, xb, n0, w[3]);
                for (int t = 0; t < n0; ++t) out[r0 + t] = xb[t];
            }
            double worst = 0.0;
            for (int i = 0; i < n; ++i) worst = std::max(worst, std::fabs(out[i] - x[i]));
            exact = exact && worst <= 1e-12;
        }
        CHECK("sweep_plan_maps_consistent", maps_ok);
        CHECK("blocked_sweeps_equal_global_sweeps", exact);
        std::printf("sweep_plans %d\n", planned);
    }
    // the replicated-level path on ONE "subdomain": coarse_rows + sell_from_csr + coarsen_onto_global must produce
    // the same coarse operator as the ordinary transfer, entry by entry
    if (!P.amg.empty() && !P.amg[0].dense) {
        const int32_t n0 = P.A.n_rows, nc = (n0 + 3) / 4;
        std::vector<int32_t> agg(n0), colmap(P.A.n_cols);
        for (int i = 0; i < n0; ++i) agg[i] = P.krank[i] / 4;
        for (int j = 0; j < P.A.n_cols; ++j) colmap[j] = agg[j];
        std::vector<int32_t> rp, ci, diag;
        SellPattern G;
        AmgLevelPlan Lg, Lr;
        std::string e1 = coarse_rows(P.A, agg, colmap, nc, 0, rp, ci);
        std::string e2 = e1.empty() ? sell_from_csr(nc, nc, rp, ci, G, diag) : e1;
        Lg.with_ap = true;
        std::string e3 = e2.empty() ? coarsen_onto_global(P.A, agg, colmap, nc, 0, G, Lg) : e2;
        Lr.with_ap = true;
        std::string e4 = coarsen(P.A, agg, colmap, nc, nc, false, Lr);
        CHECK("onto_global_builds", e3.empty() && e4.empty());
        if (e3.empty() && e4.empty()) {
            std::mt19937_64 rng(5);
            std::uniform_real_distribution<double> U(-1.0, 1.0);
            std::vector<double> fine(P.A.slots);
            for (double& v : fine) v = U(rng);
            const auto a = apply_plan(Lg, G, fine), b = apply_plan(Lr, Lr.Ac, fine);
            bool same = a.size() == b.size();
            for (const auto& kv : a) {
                auto it = b.find(kv.first);
                same = same && it != b.end() && std::fabs(it->second - kv.second) <= 1e-13;
            }
            CHECK("onto_global_equals_ordinary_transfer", same);
            CHECK("onto_global_ap_pattern_equal", Lg.AP.nnz == Lr.AP.nnz && Lg.AP.slots == Lr.AP.slots && Lg.ap_glist == Lr.ap_glist);
            bool diag_ok = true;
            for (int I = 0; I < nc; ++I) diag_ok = diag_ok && G.col[diag[I]] == I;
            CHECK("global_level_diag_slots", diag_ok);
        }
        // a hierarchy built on that global level (the replicated hierarchy)
        std::vector<int32_t> ident(nc);
        for (int i = 0; i < nc; ++i) ident[i] = i;
        std::vector<AmgLevelPlan> rep;
        PlanOptions o2;
        o2.amg_cost_nnz = (double)P.A.nnz;
        CHECK("replicated_hierarchy_builds", build_amg_levels(G, ident, o2, rep).empty() && !rep.empty() && rep.back().dense);
        std::printf("rep_levels %zu\nrep_dense_rows %d\n", rep.size(), rep.empty() ? 0 : rep.back().n_coarse);
    }
    return fails ? 1 : 0;
}
