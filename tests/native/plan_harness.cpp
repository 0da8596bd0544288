// Host-side checks of shakti_fenics_amd/csrc/shk_plan.cpp (no GPU): built and run by tests/test_plan_host.py.
//   plan_harness mesh.bin  ->  one "key value" line per check on stdout, exit code 0 if every check holds
// mesh.bin: int64 nv, int64 ne, double xy[2 nv], int32 cells[3 ne]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
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

int main(int argc, char** argv) {
    if (argc < 2) return 2;
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
