// Static plan construction on the host (see shk_plan.h).  Everything here is O(ne) with small
// constants and runs once in shk_create().
#include "shk_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace shk {

void SellPattern::build_col16() {
    cbase.assign(nslice, -1);
    ptr16.assign(nslice + 1, 0);
    col16.clear();
    for (int32_t s = 0; s < nslice; ++s) {
        const int32_t base = ptr[s], w = (ptr[s + 1] - base) / kSlice;
        int32_t lo = INT32_MAX, hi = -1;
        for (int l = 0; l < kSlice; ++l) {
            const int64_t row = (int64_t)s * kSlice + l;
            if (row >= n_rows) continue;
            for (int k = 0; k < rowlen[row]; ++k) {
                const int32_t c = col[base + k * kSlice + l];
                lo = std::min(lo, c);
                hi = std::max(hi, c);
            }
        }
        if (hi >= lo && hi - lo <= 65535) {
            cbase[s] = lo;
            for (int k = 0; k < w; ++k)
                for (int l = 0; l < kSlice; ++l) {
                    const int64_t row = (int64_t)s * kSlice + l;
                    const bool real = row < n_rows && k < rowlen[row];
                    // padding entries carry a zero value: any in-range column will do
                    const int32_t c = real ? col[base + k * kSlice + l] : lo;
                    col16.push_back((uint16_t)(c - lo));
                }
        }
        ptr16[s + 1] = (int32_t)col16.size();
    }
}

// k-d tree order: recursive splits along the longer side of the bounding box.  Split positions are
// multiples of the largest power of 4 that keeps both parts non-empty, so every aligned run of 4, 16, 64,
// ... consecutive vertices is a compact k-d cell: those runs are the multigrid aggregates of every level.
struct KdPoint { double x, y; int32_t id; };

static void kd_order(std::vector<KdPoint>& pts) {
    struct Node { int64_t a, n; };
    std::vector<Node> stack;
    stack.push_back({0, (int64_t)pts.size()});
    while (!stack.empty()) {
        const Node nd = stack.back();
        stack.pop_back();
        if (nd.n <= 4) continue;
        int64_t G = 1;
        while (G * 4 <= nd.n / 2) G *= 4;
        int64_t s = 0;
        for (;;) {
            s = ((2 * nd.a + nd.n + G) / (2 * G)) * G - nd.a;  // multiple of G nearest to the middle
            if (s > 0 && s < nd.n) break;
            G /= 4;
            if (G < 1) { s = nd.n / 2; break; }
        }
        double x0 = pts[nd.a].x, x1 = x0, y0 = pts[nd.a].y, y1 = y0;
        for (int64_t i = nd.a; i < nd.a + nd.n; ++i) {
            x0 = std::min(x0, pts[i].x); x1 = std::max(x1, pts[i].x);
            y0 = std::min(y0, pts[i].y); y1 = std::max(y1, pts[i].y);
        }
        auto b = pts.begin() + nd.a;
        if (x1 - x0 >= y1 - y0)
            std::nth_element(b, b + s, b + nd.n, [](const KdPoint& p, const KdPoint& q) {
                return p.x < q.x || (p.x == q.x && p.id < q.id); });
        else
            std::nth_element(b, b + s, b + nd.n, [](const KdPoint& p, const KdPoint& q) {
                return p.y < q.y || (p.y == q.y && p.id < q.id); });
        stack.push_back({nd.a, s});
        stack.push_back({nd.a + s, nd.n - s});
    }
}

// sorted unique vertex set of the cells incident to v, diagonal first
static int row_columns(int32_t v, const int32_t* v2c_ptr, const int32_t* v2c, const int32_t* cells,
                       std::vector<int32_t>& tmp) {
    tmp.clear();
    for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
        const int32_t* cv = cells + 3 * (int64_t)v2c[k];
        tmp.push_back(cv[0]); tmp.push_back(cv[1]); tmp.push_back(cv[2]);
    }
    std::sort(tmp.begin(), tmp.end());
    int n = (int)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
    tmp.resize(n);
    auto it = std::lower_bound(tmp.begin(), tmp.end(), v);
    std::rotate(tmp.begin(), it, it + 1);  // move the diagonal to the front, rest stays ascending
    return n;
}

static void build_v2c(int64_t nverts, int64_t ne, const int32_t* cells, std::vector<int32_t>& ptr,
                      std::vector<int32_t>& lst) {
    ptr.assign(nverts + 1, 0);
    for (int64_t i = 0; i < 3 * ne; ++i) ptr[cells[i] + 1]++;
    for (int64_t v = 0; v < nverts; ++v) ptr[v + 1] += ptr[v];
    lst.resize(3 * ne);
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t c = 0; c < ne; ++c)
        for (int k = 0; k < 3; ++k) lst[fill[cells[3 * c + k]]++] = (int32_t)c;
}

std::string build_plan(int64_t n_own, int64_t n_loc, int64_t ne, const double* xy, const int32_t* cells_ext,
                       const PlanOptions& opt, HostPlan& P) {
    if (n_own <= 0 || ne <= 0 || n_loc < n_own) return "empty mesh";
    if (n_loc > INT32_MAX / 2 || ne > INT32_MAX / 4) return "mesh too large for int32 indexing";
    P.n_own = n_own;
    P.n_loc = n_loc;
    P.ne = ne;
    for (int64_t c = 0; c < ne; ++c) {
        const int32_t* cv = cells_ext + 3 * c;
        for (int k = 0; k < 3; ++k)
            if (cv[k] < 0 || cv[k] >= n_loc) return "cell references a vertex outside [0, nv)";
        if (cv[0] == cv[1] || cv[1] == cv[2] || cv[0] == cv[2]) return "degenerate cell (repeated vertex)";
        if (cv[0] >= n_own && cv[1] >= n_own && cv[2] >= n_own) return "cell touches no owned vertex";
    }

    std::vector<int32_t> v2c_ptr, v2c, tmp;
    tmp.reserve(64);
    // ---- internal numbering of the owned vertices ----
    P.perm.resize(n_loc);
    std::iota(P.perm.begin(), P.perm.end(), 0);
    if (opt.reorder) {
        std::vector<KdPoint> pts(n_own);
        for (int64_t v = 0; v < n_own; ++v) pts[v] = {xy[2 * v], xy[2 * v + 1], (int32_t)v};
        kd_order(pts);
        std::vector<std::pair<int32_t, int32_t>> key(n_own);  // (k-d rank, external id)
        for (int64_t i = 0; i < n_own; ++i) key[i] = {(int32_t)i, pts[i].id};
        pts = std::vector<KdPoint>();
        // row lengths (in the caller's numbering), then sort each window by length, longest first
        build_v2c(n_loc, ne, cells_ext, v2c_ptr, v2c);
        std::vector<uint8_t> len(n_own);
        for (int64_t v = 0; v < n_own; ++v) {
            if (v2c_ptr[v + 1] == v2c_ptr[v]) return "mesh has a vertex that belongs to no cell";
            int n = row_columns((int32_t)v, v2c_ptr.data(), v2c.data(), cells_ext, tmp);
            if (n > 255) return "a vertex has more than 254 neighbours";
            len[v] = (uint8_t)n;
        }
        const int64_t W = std::max(kSlice, opt.sort_window / kSlice * kSlice);
        // Rim rows first WITHIN a length class (PlanOptions::rim_first): a window is a compact patch of the k-d order, and
        // the only entries of a vector that OTHER windows gather are those of its rim (~60 of 256 rows).  Kept together
        // they occupy fewer cache lines of the vector, so a neighbouring window's gathers touch fewer lines.  (Rim rows
        // first regardless of length was measured: +8-10 % bandwidth on the gathering kernels, but a second length-mixed
        // slice per window -- the bench mesh alternates rows of 5 and 9 entries -- took the SELL padding from 5.4 to 14.3 %
        // and the step from 47.9 to 48.5 ms.  Length stays the primary key.)
        std::vector<uint8_t> rim;
        if (opt.rim_first) {
            std::vector<int32_t> rank_of(n_own);
            for (int64_t i = 0; i < n_own; ++i) rank_of[key[i].second] = (int32_t)i;
            rim.assign(n_own, 0);
            for (int64_t v = 0; v < n_own; ++v) {
                const int64_t w = rank_of[v] / W;
                row_columns((int32_t)v, v2c_ptr.data(), v2c.data(), cells_ext, tmp);
                for (int32_t u : tmp)
                    if (u >= n_own || rank_of[u] / W != w) { rim[v] = 1; break; }
            }
        }
        for (int64_t w0 = 0; w0 < n_own; w0 += W) {
            const int64_t w1 = std::min(n_own, w0 + W);
            std::stable_sort(key.begin() + w0, key.begin() + w1,
                             [&](const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) {
                                 if (len[a.second] != len[b.second]) return len[a.second] > len[b.second];
                                 return !rim.empty() && rim[a.second] > rim[b.second];
                             });
        }
        P.krank.resize(n_own);
        for (int64_t i = 0; i < n_own; ++i) { P.perm[i] = key[i].second; P.krank[i] = key[i].first; }
    } else {
        P.krank.resize(n_own);
        std::iota(P.krank.begin(), P.krank.end(), 0);
    }
    P.iperm.resize(n_loc);
    for (int64_t i = 0; i < n_loc; ++i) P.iperm[P.perm[i]] = (int32_t)i;
    P.xy.resize(2 * n_loc);
    for (int64_t i = 0; i < n_loc; ++i) {
        P.xy[2 * i] = xy[2 * (int64_t)P.perm[i]];
        P.xy[2 * i + 1] = xy[2 * (int64_t)P.perm[i] + 1];
    }
    P.cells.resize(3 * ne);
    for (int64_t i = 0; i < 3 * ne; ++i) P.cells[i] = P.iperm[cells_ext[i]];
    const int32_t* cells = P.cells.data();

    // ---- vertex -> incident cells (ascending cell id), internal numbering ----
    build_v2c(n_loc, ne, cells, v2c_ptr, v2c);
    for (int64_t v = 0; v < n_own; ++v)
        if (v2c_ptr[v + 1] == v2c_ptr[v]) return "mesh has a vertex that belongs to no cell";

    // ---- "last cell wins": highest cell index containing v ----
    P.lastcell.resize(n_own);
    for (int64_t v = 0; v < n_own; ++v) P.lastcell[v] = v2c[v2c_ptr[v + 1] - 1];

    // ---- SELL-64 pattern of the owned rows ----
    SellPattern& A = P.A;
    A.n_rows = (int32_t)n_own;
    A.n_cols = (int32_t)n_loc;
    A.nslice = (int32_t)((n_own + kSlice - 1) / kSlice);
    A.rowlen.assign((size_t)A.nslice * kSlice, 0);
    A.ptr.assign(A.nslice + 1, 0);
    A.nnz = 0;
    for (int64_t v = 0; v < n_own; ++v) {
        int n = row_columns((int32_t)v, v2c_ptr.data(), v2c.data(), cells, tmp);
        if (n > 255) return "a vertex has more than 254 neighbours";
        A.rowlen[v] = (uint8_t)n;
        A.max_row_len = std::max(A.max_row_len, n);
        A.nnz += n;
    }
    int64_t slots = 0;
    for (int32_t s = 0; s < A.nslice; ++s) {
        int w = 0;
        for (int l = 0; l < kSlice; ++l) w = std::max(w, (int)A.rowlen[(size_t)s * kSlice + l]);
        slots += (int64_t)w * kSlice;
        if (slots > INT32_MAX) return "padded nnz exceeds int32";
        A.ptr[s + 1] = (int32_t)slots;
    }
    A.slots = slots;
    A.col.assign(slots, 0);
    for (int32_t s = 0; s < A.nslice; ++s) {
        const int32_t base = A.ptr[s];
        const int w = (A.ptr[s + 1] - base) / kSlice;
        for (int l = 0; l < kSlice; ++l) {
            const int64_t v = (int64_t)s * kSlice + l;
            int n = 0;
            if (v < n_own) n = row_columns((int32_t)v, v2c_ptr.data(), v2c.data(), cells, tmp);
            for (int k = 0; k < w; ++k)
                A.col[base + k * kSlice + l] = (k < n) ? tmp[k] : (v < n_own ? (int32_t)v : 0);
        }
    }

    A.build_col16();

    // ---- assembly blocks: runs of slices whose incident cells fit the LDS budget ----
    P.slices_max = std::max(1, opt.slices_max);
    P.cells_max = std::min(opt.cells_max, (int)kSrcNone - 1);   // slotsrc keeps the cell slot in 10 bits
    P.slotsrc.assign((size_t)A.slots, kSrcEmpty);
    P.blk_haloptr.assign(1, 0);
    P.blk_halo.clear();
    P.blk_cellv.clear();
    P.verts_max = 0;
    std::vector<int32_t> hlocal(n_loc, -1);
    std::vector<int32_t> mark(ne, -1), slot(ne, 0);
    P.blk_slice0.assign(1, 0);
    P.blk_cellptr.assign(1, 0);
    P.blk_cells.clear();
    P.incptr.assign(v2c_ptr.begin(), v2c_ptr.begin() + n_own + 1);
    P.inccode.assign(v2c_ptr[n_own], 0);
    int32_t blk = 0, slices_in = 0, cells_in = 0, first_slice = 0;
    std::string plan_error;
    auto close_block = [&](int32_t slice_end) {
        const size_t c0 = P.blk_cells.size();
        const int64_t ra = (int64_t)first_slice * kSlice, rb = std::min<int64_t>(n_own, (int64_t)slice_end * kSlice);
        for (int64_t v = ra; v < rb; ++v)
            for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
                const int32_t c = v2c[k];
                if (mark[c] == blk) { mark[c] = -2 - blk; P.blk_cells.push_back(c); }
            }
        std::sort(P.blk_cells.begin() + c0, P.blk_cells.end());
        for (size_t i = c0; i < P.blk_cells.size(); ++i) slot[P.blk_cells[i]] = (int32_t)(i - c0);
        // halo vertices of the block (ascending ids) and the cells' local vertex ids
        {
            const size_t h0 = P.blk_halo.size();
            for (size_t i = c0; i < P.blk_cells.size(); ++i)
                for (int k = 0; k < 3; ++k) {
                    const int32_t u = cells[3 * (int64_t)P.blk_cells[i] + k];
                    if ((u < ra || u >= rb) && hlocal[u] < 0) { hlocal[u] = 0; P.blk_halo.push_back(u); }
                }
            std::sort(P.blk_halo.begin() + h0, P.blk_halo.end());
            const int32_t nrows = (int32_t)(rb - ra);
            for (size_t i = h0; i < P.blk_halo.size(); ++i) hlocal[P.blk_halo[i]] = nrows + (int32_t)(i - h0);
            if (nrows + (int64_t)(P.blk_halo.size() - h0) > 65535) plan_error = "an assembly block touches more than 65535 vertices";
            for (size_t i = c0; i < P.blk_cells.size(); ++i) {
                for (int k = 0; k < 3; ++k) {
                    const int32_t u = cells[3 * (int64_t)P.blk_cells[i] + k];
                    P.blk_cellv.push_back((uint16_t)((u >= ra && u < rb) ? u - ra : hlocal[u]));
                }
                P.blk_cellv.push_back(0);
            }
            P.verts_max = std::max(P.verts_max, nrows + (int)(P.blk_halo.size() - h0));
            for (size_t i = h0; i < P.blk_halo.size(); ++i) hlocal[P.blk_halo[i]] = -1;
            P.blk_haloptr.push_back((int32_t)P.blk_halo.size());
        }
        int inc = 0;
        for (int64_t v = ra; v < rb; ++v)
            for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
                const int32_t c = v2c[k];
                const int32_t* cv = cells + 3 * (int64_t)c;
                const int li = (cv[0] == v) ? 0 : (cv[1] == v) ? 1 : 2;
                P.inccode[k] = (uint16_t)((slot[c] << 2) | li);
                ++inc;
            }
        // direct sources of the off-diagonal entries of the block's rows (ascending cell id)
        for (int64_t v = ra; v < rb; ++v) {
            const int32_t sl = (int32_t)(v / kSlice), ln = (int32_t)(v % kSlice), base = A.ptr[sl];
            for (int k = 1; k < A.rowlen[v]; ++k) {
                const int32_t slot_id = base + k * kSlice + ln, u = A.col[slot_id];
                uint32_t code[2] = {kSrcNone << 4, kSrcNone << 4};
                int nsrc = 0;
                for (int32_t q = v2c_ptr[v]; q < v2c_ptr[v + 1]; ++q) {
                    const int32_t c = v2c[q];
                    const int32_t* cv = cells + 3 * (int64_t)c;
                    const int lj = (cv[0] == u) ? 0 : (cv[1] == u) ? 1 : (cv[2] == u) ? 2 : -1;
                    if (lj < 0) continue;
                    const int li = (cv[0] == v) ? 0 : (cv[1] == v) ? 1 : 2;
                    if (nsrc < 2) code[nsrc] = ((uint32_t)slot[c] << 4) | (uint32_t)(3 * li + lj);
                    ++nsrc;
                }
                if (nsrc < 1 || nsrc > 2) { plan_error = "an edge belongs to more than two cells (non-manifold mesh)"; }
                P.slotsrc[slot_id] = code[0] | (code[1] << 14);
            }
        }
        P.max_inc_per_block = std::max(P.max_inc_per_block, inc);
        P.blk_slice0.push_back(slice_end);
        P.blk_cellptr.push_back((int32_t)P.blk_cells.size());
        ++blk;
        slices_in = 0;
        cells_in = 0;
        first_slice = slice_end;
    };
    std::vector<int32_t> seen(ne, -1);
    for (int32_t s = 0; s < A.nslice; ++s) {
        const int64_t ra = (int64_t)s * kSlice, rb = std::min<int64_t>(n_own, ra + kSlice);
        int fresh = 0;  // cells of this slice not yet staged by the open block
        for (int64_t v = ra; v < rb; ++v)
            for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
                const int32_t c = v2c[k];
                if (mark[c] != blk && seen[c] != s) { seen[c] = s; ++fresh; }
            }
        if (fresh > P.cells_max) return "one 64-row slice touches more cells than the assembly LDS budget";
        const int64_t slice_slots = A.ptr[s + 1] - A.ptr[s];
        if (slice_slots > opt.slots_max) return "a vertex has more than 47 neighbours: SELL slices wider than 48 entries exceed the assembly kernel's slot budget";
        if (slices_in > 0 && (slices_in >= P.slices_max || cells_in + fresh > P.cells_max ||
                              A.ptr[s + 1] - A.ptr[first_slice] > opt.slots_max))
            close_block(s);
        for (int64_t v = ra; v < rb; ++v)
            for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
                const int32_t c = v2c[k];
                if (mark[c] != blk) { mark[c] = blk; ++cells_in; }
            }
        ++slices_in;
    }
    close_block(A.nslice);
    if (!plan_error.empty()) return plan_error;
    {
        const int32_t nblk = (int32_t)P.blk_slice0.size() - 1;
        P.blk_desc.assign((size_t)kBlkDesc * nblk, 0);
        for (int32_t b = 0; b < nblk; ++b) {
            int32_t* d = &P.blk_desc[(size_t)kBlkDesc * b];
            const int32_t s0 = P.blk_slice0[b], s1 = P.blk_slice0[b + 1];
            const int64_t r0 = (int64_t)s0 * kSlice, r1 = std::min<int64_t>(n_own, (int64_t)s1 * kSlice);
            d[0] = s0; d[1] = s1 - s0;
            d[2] = P.blk_cellptr[b]; d[3] = P.blk_cellptr[b + 1] - P.blk_cellptr[b];
            d[4] = P.blk_haloptr[b]; d[5] = P.blk_haloptr[b + 1] - P.blk_haloptr[b];
            d[6] = A.ptr[s0]; d[7] = A.ptr[s1];
            d[8] = P.incptr[r0]; d[9] = P.incptr[r1] - P.incptr[r0];
        }
    }
    if (opt.amg) {
        std::string err = build_amg(P, opt);
        if (!err.empty()) return err;
    }
    return std::string();
}

// A*P of one transfer: pattern (fine rows x coarse columns colmap[.]) and the gather plan of its values.
static void build_ap_plan(const SellPattern& Af, const std::vector<int32_t>& colmap, int32_t n_coarse_cols,
                          AmgLevelPlan& L) {
    const int32_t nf = Af.n_rows;
    // A*P pattern: per fine row the sorted unique coarse columns of its entries (own aggregate first)
    SellPattern& Q = L.AP;
    Q.n_rows = nf;
    Q.n_cols = n_coarse_cols;
    Q.nslice = Af.nslice;
    Q.rowlen.assign((size_t)Q.nslice * kSlice, 0);
    Q.ptr.assign(Q.nslice + 1, 0);
    std::vector<int32_t> rp2(nf + 1, 0), ci2, tmp2;
    ci2.reserve((size_t)Af.nnz * 2 / 3);
    for (int32_t i = 0; i < nf; ++i) {
        const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s];
        tmp2.clear();
        for (int k = 0; k < Af.rowlen[i]; ++k) {
            const int32_t J = colmap[Af.col[base + k * kSlice + l]];
            if (J >= 0) tmp2.push_back(J);
        }
        std::sort(tmp2.begin(), tmp2.end());
        tmp2.erase(std::unique(tmp2.begin(), tmp2.end()), tmp2.end());
        Q.rowlen[i] = (uint8_t)tmp2.size();
        Q.max_row_len = std::max(Q.max_row_len, (int)tmp2.size());
        ci2.insert(ci2.end(), tmp2.begin(), tmp2.end());
        rp2[i + 1] = (int32_t)ci2.size();
    }
    Q.nnz = rp2[nf];
    int64_t qs = 0;
    for (int32_t s = 0; s < Q.nslice; ++s) {
        int w = 0;
        for (int l = 0; l < kSlice; ++l) w = std::max(w, (int)Q.rowlen[(size_t)s * kSlice + l]);
        qs += (int64_t)w * kSlice;
        Q.ptr[s + 1] = (int32_t)qs;
    }
    Q.slots = qs;
    Q.col.assign(qs, 0);
    std::vector<int32_t> target2((size_t)Af.slots, -1);
    for (int32_t i = 0; i < nf; ++i) {
        const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s], qb = Q.ptr[s];
        const int w = (Q.ptr[s + 1] - qb) / kSlice;
        const int32_t* row = ci2.data() + rp2[i];
        const int len = rp2[i + 1] - rp2[i];
        for (int k = 0; k < w; ++k) Q.col[qb + k * kSlice + l] = (k < len) ? row[k] : row[0];  // padding: any valid column
        for (int k = 0; k < Af.rowlen[i]; ++k) {
            const int32_t slot = base + k * kSlice + l;
            const int32_t J = colmap[Af.col[slot]];
            if (J < 0) continue;
            const int kk = (int)(std::lower_bound(row, row + len, J) - row);
            target2[slot] = qb + kk * kSlice + l;
        }
    }
    for (int32_t i = nf; i < Q.nslice * kSlice; ++i) {  // tail padding rows
        const int32_t s = i / kSlice, l = i % kSlice, qb = Q.ptr[s];
        const int w = (Q.ptr[s + 1] - qb) / kSlice;
        for (int k = 0; k < w; ++k) Q.col[qb + k * kSlice + l] = 0;
    }
    Q.build_col16();
    L.ap_gptr.assign(qs + 1, 0);
    for (int64_t s = 0; s < Af.slots; ++s)
        if (target2[s] >= 0) L.ap_gptr[target2[s] + 1]++;
    for (int64_t s = 0; s < qs; ++s) L.ap_gptr[s + 1] += L.ap_gptr[s];
    L.ap_glist.resize(L.ap_gptr[qs]);
    std::vector<int32_t> fill2(L.ap_gptr.begin(), L.ap_gptr.end() - 1);
    for (int64_t s = 0; s < Af.slots; ++s)
        if (target2[s] >= 0) L.ap_glist[fill2[target2[s]]++] = (int32_t)s;
}

// ---- aggregation multigrid hierarchy (static patterns) ----
// One coarsening step.  `agg` maps the fine rows to coarse rows; `colmap` maps EVERY fine column (owned
// and ghost) to a coarse column, or -1 to drop it (ghost couplings of a block-local hierarchy).  Sparse result:
// coarse columns are local ids (owned rows first, then coarse ghosts); dense result: `n_coarse_cols` columns
// (global ids when the coarsest operator is shared by all subdomains), row-major targets I * n_coarse_cols + J.
std::string coarsen(const SellPattern& Af, const std::vector<int32_t>& agg, const std::vector<int32_t>& colmap,
                    int32_t n_coarse, int32_t n_coarse_cols, bool dense, AmgLevelPlan& L) {
    const int32_t nf = Af.n_rows;
    if ((int32_t)agg.size() != nf || (int32_t)colmap.size() != Af.n_cols) return "coarsen: map sizes";
    L.n_fine = nf;
    L.n_coarse = n_coarse;
    L.n_coarse_cols = n_coarse_cols;
    L.agg = agg;
    L.dense = dense;
    L.members.assign((size_t)4 * n_coarse, -1);
    {
        std::vector<uint8_t> cnt(n_coarse, 0);
        for (int32_t i = 0; i < nf; ++i) {
            const int32_t I = agg[i];
            if (I < 0 || I >= n_coarse) return "aggregate id out of range";
            if (cnt[I] >= 4) return "aggregate with more than 4 members";
            L.members[(size_t)4 * I + cnt[I]++] = i;
        }
        for (int32_t I = 0; I < n_coarse; ++I) {
            if (cnt[I] == 0) return "empty aggregate";
            const int32_t g0 = L.members[(size_t)4 * I] / 256;
            for (int m = 1; m < cnt[I]; ++m)
                if (L.members[(size_t)4 * I + m] / 256 != g0) return "aggregate straddles a 256-row group";
        }
    }
    std::vector<int32_t> target((size_t)Af.slots, -1);  // coarse slot of every fine slot
    int64_t nslots_c;
    if (dense) {
        nslots_c = (int64_t)n_coarse * n_coarse_cols;
        for (int32_t i = 0; i < nf; ++i) {
            const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s], I = agg[i];
            for (int k = 0; k < Af.rowlen[i]; ++k) {
                const int32_t slot = base + k * kSlice + l;
                const int32_t J = colmap[Af.col[slot]];
                if (J >= 0) target[slot] = I * n_coarse_cols + J;
            }
        }
    } else {
        // coarse rows: sorted unique coarse columns of the members' entries, diagonal first
        std::vector<int32_t> rp(n_coarse + 1, 0), ci, tmp;
        ci.reserve((size_t)Af.nnz / 2);
        for (int32_t I = 0; I < n_coarse; ++I) {
            tmp.clear();
            for (int m = 0; m < 4; ++m) {
                const int32_t i = L.members[(size_t)4 * I + m];
                if (i < 0) continue;
                const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s];
                for (int k = 0; k < Af.rowlen[i]; ++k) {
                    const int32_t J = colmap[Af.col[base + k * kSlice + l]];
                    if (J >= 0) tmp.push_back(J);
                }
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            auto it = std::lower_bound(tmp.begin(), tmp.end(), I);
            if (it == tmp.end() || *it != I) return "coarse row without a diagonal";
            std::rotate(tmp.begin(), it, it + 1);
            if (tmp.size() > 255) return "coarse row longer than 255";
            ci.insert(ci.end(), tmp.begin(), tmp.end());
            rp[I + 1] = (int32_t)ci.size();
        }
        SellPattern& C = L.Ac;
        C.n_rows = n_coarse;
        C.n_cols = n_coarse_cols;
        C.nslice = (n_coarse + kSlice - 1) / kSlice;
        C.rowlen.assign((size_t)C.nslice * kSlice, 0);
        C.ptr.assign(C.nslice + 1, 0);
        C.nnz = rp[n_coarse];
        for (int32_t I = 0; I < n_coarse; ++I) {
            C.rowlen[I] = (uint8_t)(rp[I + 1] - rp[I]);
            C.max_row_len = std::max(C.max_row_len, (int)C.rowlen[I]);
        }
        int64_t slots = 0;
        for (int32_t s = 0; s < C.nslice; ++s) {
            int w = 0;
            for (int l = 0; l < kSlice; ++l) w = std::max(w, (int)C.rowlen[(size_t)s * kSlice + l]);
            slots += (int64_t)w * kSlice;
            C.ptr[s + 1] = (int32_t)slots;
        }
        C.slots = slots;
        C.col.assign(slots, 0);
        L.diag_slot.resize(n_coarse);
        for (int32_t I = 0; I < n_coarse; ++I) {
            const int32_t s = I / kSlice, l = I % kSlice, base = C.ptr[s];
            const int w = (C.ptr[s + 1] - base) / kSlice;
            for (int k = 0; k < w; ++k)
                C.col[base + k * kSlice + l] = (k < C.rowlen[I]) ? ci[rp[I] + k] : I;
            L.diag_slot[I] = base + l;
        }
        C.build_col16();
        nslots_c = slots;
        for (int32_t i = 0; i < nf; ++i) {
            const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s], I = agg[i];
            const int32_t cs = I / kSlice, cl = I % kSlice, cbase = C.ptr[cs];
            const int32_t* row = ci.data() + rp[I];
            const int len = rp[I + 1] - rp[I];
            for (int k = 0; k < Af.rowlen[i]; ++k) {
                const int32_t slot = base + k * kSlice + l;
                const int32_t J = colmap[Af.col[slot]];
                if (J < 0) continue;
                int kk = 0;
                while (kk < len && row[kk] != J) ++kk;
                target[slot] = cbase + kk * kSlice + cl;
            }
        }
    }
    if (L.with_ap) build_ap_plan(Af, colmap, n_coarse_cols, L);
    // invert: coarse slot -> ascending list of fine slots
    L.gptr.assign(nslots_c + 1, 0);
    for (int64_t s = 0; s < Af.slots; ++s)
        if (target[s] >= 0) L.gptr[target[s] + 1]++;
    for (int64_t s = 0; s < nslots_c; ++s) L.gptr[s + 1] += L.gptr[s];
    L.glist.resize(L.gptr[nslots_c]);
    std::vector<int32_t> fill(L.gptr.begin(), L.gptr.end() - 1);
    for (int64_t s = 0; s < Af.slots; ++s)
        if (target[s] >= 0) L.glist[fill[target[s]]++] = (int32_t)s;
    return std::string();
}

// Same, but the coarse rows are first renumbered inside 256-row windows by decreasing length (like the finest
// level), which removes most of the SELL padding of the coarse operators (47 % -> a few % on level 1).
// `krank_out[I]` is the k-d position of coarse row I: the next level aggregates runs of 4 of THOSE, which stay
// inside one window.  Only for hierarchies without ghost columns (neighbours would need the new numbering).
static std::string coarsen_sorted(const SellPattern& Af, const std::vector<int32_t>& agg,
                                  const std::vector<int32_t>& colmap, int32_t n_coarse, AmgLevelPlan& L,
                                  std::vector<int32_t>& krank_out) {
    const int32_t nf = Af.n_rows;
    std::vector<std::vector<int32_t>> rows(n_coarse);
    for (int32_t i = 0; i < nf; ++i) {
        const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s];
        auto& r = rows[agg[i]];
        for (int k = 0; k < Af.rowlen[i]; ++k) {
            const int32_t J = colmap[Af.col[base + k * kSlice + l]];
            if (J >= 0) r.push_back(J);
        }
    }
    std::vector<int32_t> len(n_coarse);
    for (int32_t I = 0; I < n_coarse; ++I) {
        auto& r = rows[I];
        std::sort(r.begin(), r.end());
        len[I] = (int32_t)(std::unique(r.begin(), r.end()) - r.begin());
        std::vector<int32_t>().swap(r);
    }
    krank_out.resize(n_coarse);
    std::iota(krank_out.begin(), krank_out.end(), 0);
    for (int32_t w0 = 0; w0 < n_coarse; w0 += 256)
        std::stable_sort(krank_out.begin() + w0, krank_out.begin() + std::min(n_coarse, w0 + 256),
                         [&](int32_t a, int32_t b) { return len[a] > len[b]; });
    std::vector<int32_t> inv(n_coarse);
    for (int32_t I = 0; I < n_coarse; ++I) inv[krank_out[I]] = I;
    std::vector<int32_t> agg2(agg.size()), colmap2(colmap.size());
    for (size_t i = 0; i < agg.size(); ++i) agg2[i] = inv[agg[i]];
    for (size_t c = 0; c < colmap.size(); ++c) colmap2[c] = colmap[c] >= 0 ? inv[colmap[c]] : -1;
    std::string err = coarsen(Af, agg2, colmap2, n_coarse, n_coarse, false, L);
    L.kd_pos = inv;
    return err;
}

// Block-local hierarchy on the square part of a SELL operator (columns >= n_rows dropped): needs no communication.
std::string build_amg_levels(const SellPattern& A0, const std::vector<int32_t>& krank, const PlanOptions& opt,
                             std::vector<AmgLevelPlan>& out) {
    out.clear();
    out.reserve(40);  // `Af` below points into this vector: no reallocation (4^40 rows is out of reach)
    const int32_t n0 = A0.n_rows;
    if ((int32_t)krank.size() != n0) return "amg: k-d ranks do not match the operator";
    // dense coarsest level: as large as its O(n^3) inversion stays small next to the fine-level work
    // (n^3 <= 250 nnz: 2441 rows at 10M vertices, 977 at 1M), and never so large that a small mesh gets no
    // hierarchy at all.  Measured at 10M rows: 38 -> 153 -> 610 dense rows = 254 -> 218 -> 165 iterations.
    const int by_cost = (int)std::cbrt(250.0 * (opt.amg_cost_nnz > 0 ? opt.amg_cost_nnz : (double)A0.nnz));
    const int coarsest = std::min(std::min(std::min(4096, std::max(4, opt.amg_coarsest)), std::max(64, by_cost)),
                                  (int)std::max<int64_t>(64, n0 / 16));
    const SellPattern* Af = &A0;
    std::vector<int32_t> agg(n0), colmap;
    for (int32_t i = 0; i < n0; ++i) agg[i] = krank[i] / 4;
    while (Af->n_rows > coarsest) {
        const int32_t nc = (Af->n_rows + 3) / 4;
        out.emplace_back();
        colmap.assign(Af->n_cols, -1);
        std::copy(agg.begin(), agg.end(), colmap.begin());
        const bool dense = nc <= coarsest;
        out.back().with_ap = Af->n_rows > kTailRows;  // levels handled by launches (the one-workgroup tail prolongates)
        std::vector<int32_t> kr;
        std::string err = dense ? coarsen(*Af, agg, colmap, nc, nc, true, out.back())
                                : coarsen_sorted(*Af, agg, colmap, nc, out.back(), kr);
        if (!err.empty()) { out.clear(); return "amg: " + err; }
        if (dense) break;
        Af = &out.back().Ac;
        agg.resize(nc);
        for (int32_t I = 0; I < nc; ++I) agg[I] = kr[I] / 4;
    }
    return std::string();
}

std::string build_amg(HostPlan& P, const PlanOptions& opt) {
    return build_amg_levels(P.A, P.krank, opt, P.amg);
}

static std::string aggregate_members(const std::vector<int32_t>& agg, int32_t n_coarse, std::vector<int32_t>& members) {
    members.assign((size_t)4 * n_coarse, -1);
    std::vector<uint8_t> cnt(n_coarse, 0);
    for (int32_t i = 0; i < (int32_t)agg.size(); ++i) {
        const int32_t I = agg[i];
        if (I < 0 || I >= n_coarse) return "aggregate id out of range";
        if (cnt[I] >= 4) return "aggregate with more than 4 members";
        members[(size_t)4 * I + cnt[I]++] = i;
    }
    for (int32_t I = 0; I < n_coarse; ++I) {
        if (cnt[I] == 0) return "empty aggregate";
        const int32_t g0 = members[(size_t)4 * I] / 256;
        for (int m = 1; m < cnt[I]; ++m)
            if (members[(size_t)4 * I + m] / 256 != g0) return "aggregate straddles a 256-row group";
    }
    return std::string();
}

std::string coarse_rows(const SellPattern& Af, const std::vector<int32_t>& agg, const std::vector<int32_t>& colmap,
                        int32_t n_coarse, int32_t diag0, std::vector<int32_t>& rp, std::vector<int32_t>& ci) {
    if ((int32_t)agg.size() != Af.n_rows || (int32_t)colmap.size() != Af.n_cols) return "coarse_rows: map sizes";
    std::vector<int32_t> members, tmp;
    std::string err = aggregate_members(agg, n_coarse, members);
    if (!err.empty()) return err;
    rp.assign(n_coarse + 1, 0);
    ci.clear();
    for (int32_t I = 0; I < n_coarse; ++I) {
        tmp.clear();
        for (int m = 0; m < 4; ++m) {
            const int32_t i = members[(size_t)4 * I + m];
            if (i < 0) continue;
            const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s];
            for (int k = 0; k < Af.rowlen[i]; ++k) {
                const int32_t J = colmap[Af.col[base + k * kSlice + l]];
                if (J >= 0) tmp.push_back(J);
            }
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        auto it = std::lower_bound(tmp.begin(), tmp.end(), diag0 + I);
        if (it == tmp.end() || *it != diag0 + I) return "coarse row without a diagonal";
        std::rotate(tmp.begin(), it, it + 1);
        if (tmp.size() > 255) return "coarse row longer than 255";
        ci.insert(ci.end(), tmp.begin(), tmp.end());
        rp[I + 1] = (int32_t)ci.size();
    }
    return std::string();
}

std::string sell_from_csr(int32_t n_rows, int32_t n_cols, const std::vector<int32_t>& rp,
                          const std::vector<int32_t>& ci, SellPattern& C, std::vector<int32_t>& diag_slot) {
    if ((int32_t)rp.size() != n_rows + 1 || rp[n_rows] != (int32_t)ci.size()) return "sell_from_csr: bad row pointers";
    C = SellPattern();
    C.n_rows = n_rows;
    C.n_cols = n_cols;
    C.nslice = (n_rows + kSlice - 1) / kSlice;
    C.rowlen.assign((size_t)C.nslice * kSlice, 0);
    C.ptr.assign(C.nslice + 1, 0);
    C.nnz = rp[n_rows];
    for (int32_t I = 0; I < n_rows; ++I) {
        const int len = rp[I + 1] - rp[I];
        if (len < 1 || len > 255) return "sell_from_csr: row length out of range";
        if (ci[rp[I]] != I) return "sell_from_csr: the diagonal must come first";
        C.rowlen[I] = (uint8_t)len;
        C.max_row_len = std::max(C.max_row_len, len);
    }
    int64_t slots = 0;
    for (int32_t s = 0; s < C.nslice; ++s) {
        int w = 0;
        for (int l = 0; l < kSlice; ++l) w = std::max(w, (int)C.rowlen[(size_t)s * kSlice + l]);
        slots += (int64_t)w * kSlice;
        if (slots > INT32_MAX) return "sell_from_csr: too many slots";
        C.ptr[s + 1] = (int32_t)slots;
    }
    C.slots = slots;
    C.col.assign(slots, 0);
    diag_slot.resize(n_rows);
    for (int32_t I = 0; I < n_rows; ++I) {
        const int32_t s = I / kSlice, l = I % kSlice, base = C.ptr[s];
        const int w = (C.ptr[s + 1] - base) / kSlice;
        for (int k = 0; k < w; ++k) C.col[base + k * kSlice + l] = (k < C.rowlen[I]) ? ci[rp[I] + k] : I;
        diag_slot[I] = base + l;
    }
    C.build_col16();
    return std::string();
}

std::string coarsen_onto_global(const SellPattern& Af, const std::vector<int32_t>& agg,
                                const std::vector<int32_t>& colmap, int32_t n_coarse, int32_t row0,
                                const SellPattern& G, AmgLevelPlan& L) {
    const int32_t nf = Af.n_rows;
    if ((int32_t)agg.size() != nf || (int32_t)colmap.size() != Af.n_cols) return "coarsen_onto_global: map sizes";
    if (row0 < 0 || row0 + n_coarse > G.n_rows) return "coarsen_onto_global: rows outside the global level";
    L.n_fine = nf;
    L.n_coarse = n_coarse;
    L.n_coarse_cols = G.n_rows;
    L.agg = agg;
    L.dense = false;
    L.onto_global = true;
    std::string err = aggregate_members(agg, n_coarse, L.members);
    if (!err.empty()) return err;
    std::vector<int32_t> target((size_t)Af.slots, -1);
    for (int32_t i = 0; i < nf; ++i) {
        const int32_t s = i / kSlice, l = i % kSlice, base = Af.ptr[s], I = row0 + agg[i];
        const int32_t gs = I / kSlice, gl = I % kSlice, gbase = G.ptr[gs];
        const int len = G.rowlen[I];
        for (int k = 0; k < Af.rowlen[i]; ++k) {
            const int32_t slot = base + k * kSlice + l;
            const int32_t J = colmap[Af.col[slot]];
            if (J < 0) continue;
            int kk = 0;
            while (kk < len && G.col[gbase + kk * kSlice + gl] != J) ++kk;
            if (kk == len) return "coarsen_onto_global: entry missing from the global pattern";
            target[slot] = gbase + kk * kSlice + gl;
        }
    }
    L.gptr.assign(G.slots + 1, 0);
    for (int64_t s = 0; s < Af.slots; ++s)
        if (target[s] >= 0) L.gptr[target[s] + 1]++;
    for (int64_t s = 0; s < G.slots; ++s) L.gptr[s + 1] += L.gptr[s];
    L.glist.resize(L.gptr[G.slots]);
    std::vector<int32_t> fill(L.gptr.begin(), L.gptr.end() - 1);
    for (int64_t s = 0; s < Af.slots; ++s)
        if (target[s] >= 0) L.glist[fill[target[s]]++] = (int32_t)s;
    if (L.with_ap) build_ap_plan(Af, colmap, G.n_rows, L);   // columns = global rows of the replicated level
    return std::string();
}

std::string build_sweep_plan(const SellPattern& A, const SellPattern& AP, SweepPlan& out) {
    out = SweepPlan();
    const int32_t n = A.n_rows;
    // no plan (the cycle keeps one launch per sweep) unless the rows fit the kernel's register rows
    if (n < 1 || A.max_row_len > kSweepMaxWidth || AP.n_rows != n || AP.max_row_len > kSweepMaxWidthAP) return std::string();
    const int32_t nblk = (n + kSweepRows - 1) / kSweepRows;
    const int W = A.max_row_len;
    std::vector<int32_t> stamp((size_t)A.n_cols, -1), lid((size_t)A.n_cols, 0);
    std::vector<int32_t> ring[3], fixed;
    SweepPlan P;
    P.width = W;
    P.hdr.reserve((size_t)8 * nblk);
    P.lcol_own.assign((size_t)A.slots, 0);
    auto columns = [&](int32_t i, auto&& f) {
        const int32_t s = i / kSlice, l = i % kSlice, base = A.ptr[s];
        for (int k = 0; k < A.rowlen[i]; ++k) f(k, A.col[base + k * kSlice + l]);
    };
    for (int32_t b = 0; b < nblk; ++b) {
        const int32_t r0 = b * kSweepRows, r1 = std::min(n, r0 + kSweepRows);
        for (auto& v : ring) v.clear();
        fixed.clear();
        for (int32_t i = r0; i < r1; ++i) { stamp[i] = b; lid[i] = i - r0; }
        // ring k+1 = columns of the rows of ring k (ring 0 = the block) not seen yet; ghost columns are constants
        for (int k = 0; k < 3; ++k) {
            auto visit = [&](int, int32_t c) {
                if (stamp[c] == b) return;
                stamp[c] = b;
                if (c >= n) fixed.push_back(c); else ring[k].push_back(c);
            };
            if (k == 0) for (int32_t i = r0; i < r1; ++i) columns(i, visit);
            else for (int32_t i : ring[k - 1]) columns(i, visit);
        }
        // (the columns of ring-3 rows are never read: their first-sweep value comes from the A*P operator alone)
        // local indices: the block's rows 0 .. n0-1, ring rows from kSweepRows on (also in a last, shorter block: thread
        // tid of the kernel owns local rows tid, tid + 256, tid + 512), fixed entries behind them
        int32_t next = kSweepRows;
        for (auto& v : ring) for (int32_t i : v) lid[i] = next++;
        const int32_t nS2 = kSweepRows + (int32_t)(ring[0].size() + ring[1].size()), nS3 = next;
        for (int32_t c : fixed) lid[c] = next++;
        if (next > kSweepMaxLocal || nS2 > kSweepMaxS2 || nS3 > kSweepMaxS3 || (int32_t)fixed.size() > kSweepRows)
            return std::string();   // too irregular for the kernel's fixed row slots: no plan
        P.max_local = std::max(P.max_local, next);
        const int32_t hdr[8] = {(int32_t)(P.ext_info.size() / 4), (int32_t)(P.ring_lcol.size() / W), (int32_t)ring[0].size(),
                                (int32_t)ring[1].size(), (int32_t)ring[2].size(), (int32_t)fixed.size(), 0, 0};
        P.hdr.insert(P.hdr.end(), hdr, hdr + 8);
        for (auto& v : ring)
            for (int32_t i : v) {
                const int32_t info[4] = {i, A.ptr[i / kSlice] + i % kSlice, AP.ptr[i / kSlice] + i % kSlice,
                                         (int32_t)A.rowlen[i] | ((int32_t)AP.rowlen[i] << 8)};
                P.ext_info.insert(P.ext_info.end(), info, info + 4);
            }
        for (int32_t c : fixed) { const int32_t info[4] = {c, 0, 0, 0}; P.ext_info.insert(P.ext_info.end(), info, info + 4); }
        P.ring_rows += (int64_t)(ring[0].size() + ring[1].size() + ring[2].size());
        // block-local column indices: own rows per SELL slot (padding slots point at the row itself, value 0) ...
        for (int32_t i = r0; i < r1; ++i) {
            const int32_t s = i / kSlice, l = i % kSlice, base = A.ptr[s];
            const int w = (A.ptr[s + 1] - base) / kSlice;
            for (int k = 0; k < w; ++k) {
                const int32_t slot = base + k * kSlice + l;
                P.lcol_own[slot] = (uint16_t)(k < A.rowlen[i] ? lid[A.col[slot]] : lid[i]);
            }
        }
        // ... and the rows of rings 1 and 2 (the ones later sweeps recompute), `width` entries each
        for (int k = 0; k < 2; ++k)
            for (int32_t i : ring[k]) {
                const size_t at = P.ring_lcol.size();
                P.ring_lcol.resize(at + W, (uint16_t)lid[i]);
                columns(i, [&](int kk, int32_t c) { P.ring_lcol[at + kk] = (uint16_t)lid[c]; });
            }
        if (P.ring_lcol.size() / W > (size_t)INT32_MAX / 2 || P.ext_info.size() / 4 > (size_t)INT32_MAX / 2) return std::string();
    }
    P.nblk = nblk;
    out = std::move(P);
    return std::string();
}

void sell_to_csr(const HostPlan& P, const double* sell_vals, std::vector<int32_t>& rowptr,
                 std::vector<int32_t>& colidx, std::vector<double>* vals) {
    const SellPattern& A = P.A;
    const int64_t n = P.n_own;
    rowptr.assign(n + 1, 0);
    for (int64_t e = 0; e < n; ++e) rowptr[e + 1] = rowptr[e] + A.rowlen[P.iperm[e]];
    colidx.resize(A.nnz);
    if (vals) vals->resize(A.nnz);
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t e = 0; e < n; ++e) {
        const int32_t i = P.iperm[e];
        const int32_t s = i / kSlice, l = i % kSlice, base = A.ptr[s];
        const int len = A.rowlen[i];
        row.resize(len);
        for (int k = 0; k < len; ++k) {
            const int32_t slot = base + k * kSlice + l;
            row[k] = {P.perm[A.col[slot]], sell_vals ? sell_vals[slot] : 0.0};
        }
        std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) {
            return a.first < b.first;
        });
        for (int k = 0; k < len; ++k) {
            colidx[rowptr[e] + k] = row[k].first;
            if (vals) (*vals)[rowptr[e] + k] = row[k].second;
        }
    }
}

}  // namespace shk
