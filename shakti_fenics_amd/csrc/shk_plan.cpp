// Static plan construction on the host (see shk_plan.h).  Everything here is O(ne) with small
// constants; it runs once in shk_create().
#include "shk_plan.h"

#include <algorithm>
#include <cstring>

namespace shk {

std::string build_plan(int64_t nv, int64_t ne, const int32_t* cells, const PlanOptions& opt, HostPlan& P) {
    if (nv <= 0 || ne <= 0) return "empty mesh";
    if (nv > INT32_MAX / 2 || ne > INT32_MAX / 4) return "mesh too large for int32 indexing";
    P.nv = nv;
    P.ne = ne;
    for (int64_t i = 0; i < 3 * ne; ++i)
        if (cells[i] < 0 || cells[i] >= nv) return "cell references a vertex outside [0, nv)";

    // ---- vertex -> incident cells (ascending cell id) ----
    std::vector<int32_t> v2c_ptr(nv + 1, 0);
    for (int64_t i = 0; i < 3 * ne; ++i) v2c_ptr[cells[i] + 1]++;
    for (int64_t v = 0; v < nv; ++v) {
        if (v2c_ptr[v + 1] == 0) return "mesh has a vertex that belongs to no cell";
        v2c_ptr[v + 1] += v2c_ptr[v];
    }
    std::vector<int32_t> v2c(3 * ne);
    {
        std::vector<int32_t> fill(v2c_ptr.begin(), v2c_ptr.end() - 1);
        for (int64_t c = 0; c < ne; ++c)
            for (int k = 0; k < 3; ++k) v2c[fill[cells[3 * c + k]]++] = (int32_t)c;
    }

    // ---- "last cell wins": highest cell index containing v ----
    P.lastcell.resize(nv);
    for (int64_t v = 0; v < nv; ++v) P.lastcell[v] = v2c[v2c_ptr[v + 1] - 1];

    // ---- CSR pattern: row v = sorted unique vertices of v's incident cells ----
    P.rowptr.assign(nv + 1, 0);
    std::vector<int32_t> tmp;
    tmp.reserve(64);
    // pass 1: counts
    for (int64_t v = 0; v < nv; ++v) {
        tmp.clear();
        for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
            const int32_t* cv = cells + 3 * (int64_t)v2c[k];
            tmp.push_back(cv[0]); tmp.push_back(cv[1]); tmp.push_back(cv[2]);
        }
        std::sort(tmp.begin(), tmp.end());
        int32_t n = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        P.rowptr[v + 1] = n;
        P.max_row_len = std::max(P.max_row_len, (int)n);
    }
    int64_t nnz = 0;
    for (int64_t v = 0; v < nv; ++v) {
        nnz += P.rowptr[v + 1];
        if (nnz > INT32_MAX) return "nnz exceeds int32";
        P.rowptr[v + 1] = (int32_t)nnz;
    }
    P.nnz = nnz;
    P.colidx.resize(nnz);
    P.diagpos.resize(nv);
    for (int64_t v = 0; v < nv; ++v) {
        tmp.clear();
        for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
            const int32_t* cv = cells + 3 * (int64_t)v2c[k];
            tmp.push_back(cv[0]); tmp.push_back(cv[1]); tmp.push_back(cv[2]);
        }
        std::sort(tmp.begin(), tmp.end());
        int32_t n = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        int32_t* dst = P.colidx.data() + P.rowptr[v];
        for (int32_t i = 0; i < n; ++i) {
            dst[i] = tmp[i];
            if (tmp[i] == v) P.diagpos[v] = P.rowptr[v] + i;
        }
    }

    // ---- assembly blocks: contiguous row ranges whose incident cells fit the LDS budget ----
    P.rows_max = opt.rows_max;
    P.cells_max = std::min(opt.cells_max, 16383);  // inccode keeps the cell slot in 14 bits
    std::vector<int32_t> mark(ne, -1), slot(ne, 0);
    P.blk_row0.clear();
    P.blk_cellptr.clear();
    P.blk_cells.clear();
    P.incptr.assign(nv + 1, 0);
    P.inccode.resize(3 * ne);
    P.blk_row0.push_back(0);
    P.blk_cellptr.push_back(0);
    int32_t blk = 0, rows_in = 0, cells_in = 0;
    int64_t blk_first_row = 0;
    auto close_block = [&](int64_t row_end) {
        // cells of this block, ascending; assign slots; encode incidences of its rows
        size_t c0 = P.blk_cells.size();
        for (int64_t v = blk_first_row; v < row_end; ++v)
            for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
                int32_t c = v2c[k];
                if (mark[c] == blk) { mark[c] = -2 - blk; P.blk_cells.push_back(c); }
            }
        std::sort(P.blk_cells.begin() + c0, P.blk_cells.end());
        for (size_t i = c0; i < P.blk_cells.size(); ++i) slot[P.blk_cells[i]] = (int32_t)(i - c0);
        int inc = 0;
        for (int64_t v = blk_first_row; v < row_end; ++v) {
            for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k) {
                int32_t c = v2c[k];
                const int32_t* cv = cells + 3 * (int64_t)c;
                int li = (cv[0] == v) ? 0 : (cv[1] == v) ? 1 : 2;
                P.inccode[k] = (uint16_t)((slot[c] << 2) | li);
                ++inc;
            }
            P.incptr[v + 1] = v2c_ptr[v + 1];
        }
        P.max_inc_per_block = std::max(P.max_inc_per_block, inc);
        P.blk_row0.push_back((int32_t)row_end);
        P.blk_cellptr.push_back((int32_t)P.blk_cells.size());
        ++blk;
        rows_in = 0;
        cells_in = 0;
        blk_first_row = row_end;
    };
    for (int64_t v = 0; v < nv; ++v) {
        int32_t fresh = 0;
        for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k)
            if (mark[v2c[k]] != blk) ++fresh;
        if (fresh > P.cells_max) return "a vertex has more incident cells than the assembly LDS budget";
        if (rows_in > 0 && (rows_in >= P.rows_max || cells_in + fresh > P.cells_max)) close_block(v);
        for (int32_t k = v2c_ptr[v]; k < v2c_ptr[v + 1]; ++k)
            if (mark[v2c[k]] != blk) { mark[v2c[k]] = blk; ++cells_in; }
        ++rows_in;
    }
    close_block(nv);
    // a cell may not be listed twice in one vertex's incidence (degenerate cell)
    for (int64_t c = 0; c < ne; ++c) {
        const int32_t* cv = cells + 3 * c;
        if (cv[0] == cv[1] || cv[1] == cv[2] || cv[0] == cv[2]) return "degenerate cell (repeated vertex)";
    }

    // ---- SpMV row blocks (CSR-stream): contiguous rows with <= spmv_nnz stored entries ----
    P.sp_row0.clear();
    P.sp_row0.push_back(0);
    int32_t r0 = 0;
    if (P.max_row_len > opt.spmv_nnz) return "a row is longer than the SpMV LDS budget";
    for (int64_t v = 0; v < nv; ++v) {
        int32_t n_with = P.rowptr[v + 1] - P.rowptr[r0];
        if ((v - r0) >= opt.spmv_rows || n_with > opt.spmv_nnz) {
            P.sp_row0.push_back((int32_t)v);
            r0 = (int32_t)v;
        }
    }
    P.sp_row0.push_back((int32_t)nv);
    for (size_t b = 0; b + 1 < P.sp_row0.size(); ++b) {
        P.sp_max_rows = std::max(P.sp_max_rows, P.sp_row0[b + 1] - P.sp_row0[b]);
        P.sp_max_nnz = std::max(P.sp_max_nnz, P.rowptr[P.sp_row0[b + 1]] - P.rowptr[P.sp_row0[b]]);
    }
    return std::string();
}

}  // namespace shk
