// Host-side static plans built once per mesh (the mesh never changes during a run):
//  - P1 CSR sparsity, what DOLFINx preallocates at /root/reference/source/solvers.py:51-52;
//  - "last cell wins" table for the interpolations at solvers.py:186-192;
//  - the atomic-free assembly plan (row-owning blocks + their cell lists + vertex->cell
//    incidence in block-local numbering) and the CSR-stream SpMV row blocks.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace shk {

struct PlanOptions {
    int rows_max = 256;    // rows owned by one assembly block
    int cells_max = 640;   // cells staged in LDS by one assembly block
    int spmv_nnz = 2048;   // products staged in LDS by one SpMV row block
    int spmv_rows = 512;   // row cap of one SpMV row block
};

struct HostPlan {
    int64_t nv = 0, ne = 0, nnz = 0;
    std::vector<int32_t> rowptr, colidx, diagpos;
    std::vector<int32_t> lastcell;  // T*(v)
    // assembly
    int rows_max = 0, cells_max = 0;
    std::vector<int32_t> blk_row0;     // nblk+1 : rows [blk_row0[b], blk_row0[b+1])
    std::vector<int32_t> blk_cellptr;  // nblk+1 into blk_cells
    std::vector<int32_t> blk_cells;    // global cell ids, ascending inside a block
    std::vector<int32_t> incptr;       // nv+1 into inccode
    std::vector<uint16_t> inccode;     // (block-local cell slot << 2) | local vertex index
    int max_inc_per_block = 0;         // max over blocks of sum of incidences (LDS sizing)
    // spmv
    std::vector<int32_t> sp_row0;      // nsb+1
    int sp_max_nnz = 0, sp_max_rows = 0;
    int max_row_len = 0;
};

// Returns empty string on success, else an error message.
std::string build_plan(int64_t nv, int64_t ne, const int32_t* cells, const PlanOptions& opt, HostPlan& out);

}  // namespace shk
