// Host-side static plans built once per mesh (the mesh never changes during a run):
//  - an internal vertex numbering (k-d tree order of the coordinates whose leaves are runs of 4
//    vertices, then rows sorted by length inside 256-row windows) that makes every kernel's access
//    pattern independent of the numbering the caller uses and gives the multigrid its aggregates;
//  - the P1 sparsity (what DOLFINx preallocates at /root/reference/source/solvers.py:51-52) stored as
//    SELL-64: rows are grouped in slices of 64 (one wavefront), each slice padded to its longest row
//    and stored column-major, diagonal first, so a wavefront streams values and column indices with
//    fully coalesced loads and keeps its row sums in registers;
//  - "last cell wins" table for the interpolations at solvers.py:186-192;
//  - the atomic-free assembly plan: blocks own whole slices, list every cell touching their rows,
//    and know, per owned row, which staged cells contribute to it.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace shk {

constexpr int kSlice = 64;  // rows per SELL slice = wavefront width
// Multigrid levels of at most this many rows run inside ONE workgroup (k_amg_tail: restrictions, sweeps and prolongations
// separated by workgroup barriers); larger ones by launches, with an A*P operator for their first sweep.  Round 3: 4096 ->
// 512.  One workgroup walks a level of thousands of rows as a chain of dependent loads -- at 1M DOF the tail (one level
// of 3 905 rows) took 50 us per cycle, a quarter of the step -- where the fused four-sweep launch (k_amg_sweeps) takes ~10.
constexpr int kTailRows = 512;
constexpr int kBlkDesc = 12;               // ints per assembly-block descriptor (10 used, 48-byte stride)
constexpr uint32_t kSrcNone = 0x3FFu;      // slotsrc: cell slot meaning "no source"
constexpr uint32_t kSrcEmpty = (kSrcNone << 4) | (kSrcNone << 18);   // a slot without sources, Dirichlet code 0

struct PlanOptions {
    int slices_max = 4;    // slices (of 64 rows) owned by one assembly block
    int cells_max = 640;   // cells staged in LDS by one assembly block
    int slots_max = 3072;  // SELL slots one assembly block may own (its threads preload their plan words)
    int sort_window = 256; // rows per row-length sorting window (multiple of 64)
    bool rim_first = true;  // among the rows of one length inside a window: those with a neighbour outside it first
    bool reorder = true;   // internal k-d order + window sort (false: keep the caller's numbering)
    bool amg = true;       // also build the aggregation-multigrid hierarchy (of the owned diagonal block)
    int amg_coarsest = 4096; // cap of the dense coarsest level of a local hierarchy (inverted by Gauss-Jordan)
    double amg_cost_nnz = 0; // entries of the finest operator that the dense level's size is weighed against
                             // (0: those of the operator the hierarchy is built on)
};

// SELL-64 sparsity of the owned rows; columns index owned + ghost vertices.
struct SellPattern {
    int32_t n_rows = 0, n_cols = 0, nslice = 0;
    int64_t nnz = 0, slots = 0;
    std::vector<int32_t> ptr;      // nslice+1 : first slot of each slice
    std::vector<int32_t> col;      // slots   : column of slot ptr[s] + k*64 + lane (padding: the row itself / 0)
    std::vector<uint8_t> rowlen;   // nslice*64 : stored entries of each row (0 for tail padding rows)
    int max_row_len = 0;
    // 16-bit column offsets: a slice whose columns span < 65536 stores them as col = cbase[s] + col16[...]
    // (2 B instead of 4 B per entry in the SpMV stream; k-d order makes > 90 % of the slices qualify)
    std::vector<int32_t> cbase;    // nslice   : smallest column of the slice, or -1 = use the 32-bit `col`
    std::vector<int32_t> ptr16;    // nslice+1 : first entry of each slice in col16
    std::vector<uint16_t> col16;
    void build_col16();
};

// One coarsening step of the static-pattern aggregation multigrid: 4 children per aggregate
// (consecutive leaves of the k-d order), piecewise-constant prolongation, Galerkin coarse operator
// A_c = P^T A P whose sparsity never changes, so its values are a fixed gather-sum of the finer values.
struct AmgLevelPlan {
    int32_t n_fine = 0, n_coarse = 0, n_coarse_cols = 0;
    std::vector<int32_t> agg;       // n_fine          : aggregate (= coarse row) of each fine row
    std::vector<int32_t> members;   // 4*n_coarse      : fine rows of each aggregate, -1 padded; all inside one
                                    //                   256-row group of the fine level
    SellPattern Ac;                 // coarse pattern (n_coarse rows), empty for the dense coarsest level
    std::vector<int32_t> gptr;      // coarse slots+1 (or n_coarse^2+1 when dense) into glist
    std::vector<int32_t> glist;     // fine slots summed into each coarse slot, ascending
    std::vector<int32_t> diag_slot; // n_coarse        : slot of the coarse diagonal (sparse levels)
    std::vector<int32_t> kd_pos;    // n_coarse : coarse row (storage position) of k-d rank k; empty = identity
    bool dense = false;
    bool onto_global = false;       // the coarse level is a GLOBAL sparse level replicated on every subdomain: gptr /
                                    // glist target its slots, n_coarse = my rows of it, n_coarse_cols = all of its rows
    // A*P (fine rows x coarse columns, ~4 entries per row instead of ~7): lets the first smoothing sweep after
    // the prolongation read a thinner operator and skip the prolongated vector altogether:
    //   x1 = alpha P e + w D^-1 (r - alpha (A P) e)
    SellPattern AP;                 // same slicing as the fine level; columns = coarse ids
    std::vector<int32_t> ap_gptr;   // AP slots+1 into ap_glist
    std::vector<int32_t> ap_glist;  // fine slots summed into each AP slot, ascending
    bool with_ap = false;
    // decomposed levels: the coarse COLUMN (own aggregate, ghost aggregate, or global row of the replicated level) of
    // each ghost column of the fine level -- a ghost's value inside the fine level's sweeps is the prolongated coarse
    // correction alpha * e[ghost_col[g]] (filled by amg_setup_distributed)
    std::vector<int32_t> ghost_col;
};

// Plan of the fused multi-sweep smoother of one sparse multigrid level (shk_amg.hip: k_amg_sweeps).  The level's four
// damped-Jacobi sweeps run inside ONE launch: a workgroup owns a block of 256 consecutive rows and computes sweep j on
// the rows within graph distance 4 - j of the block (S3 > S2 > S1 > S0 = the block), so that every value a later sweep
// reads was produced inside the workgroup -- redundant work on the rings instead of a launch boundary (~4.5 us each on
// levels whose whole sweep takes less than that) between the sweeps.  Ghost columns of a decomposed level are constants
// of the sweeps (frozen-ghost smoothing), listed per block as `fixed` entries.
constexpr int kSweepRows = 256;        // rows of a block (4 SELL slices) = threads of its workgroup
constexpr int kSweepMaxS2 = 2 * kSweepRows;   // a thread recomputes at most 2 rows (its own + one ring row) in sweeps 2-4 ...
constexpr int kSweepMaxS3 = 3 * kSweepRows;   // ... and at most 3 in the first sweep
constexpr int kSweepMaxLocal = 1024;   // most local entries (S3 + fixed) a block may address; beyond it the level falls back
constexpr int kSweepMaxWidth = 16;     // longest row of A the kernel keeps in registers (12- and 16-wide instances)
constexpr int kSweepMaxWidthAP = 8;    // longest row of A*P
struct SweepPlan {
    int32_t nblk = 0;                  // 0: no plan (level too irregular, or rows longer than the kernel's register rows)
    int32_t width = 0;                 // stride of ring_lcol = longest row of the level
    int32_t max_local = 0;             // largest S3 + fixed over the blocks
    std::vector<int32_t> hdr;          // 8 per block: first entry in ext_info, first row in ring_lcol, rows of ring 1, 2, 3,
                                       // fixed entries, 0, 0
    std::vector<int32_t> ext_info;     // 4 per extended entry of a block, rings 1, 2, 3 then the fixed columns:
                                       // ring row {row, its first slot in A, its first slot in A*P, len(A) | len(A*P) << 8};
                                       // fixed {column, 0, 0, 0} -- ONE 16-byte load tells a thread everything about its row
    std::vector<uint16_t> lcol_own;    // per SELL slot of the level: block-local index of the slot's column
    std::vector<uint16_t> ring_lcol;   // width per ring-1 / ring-2 row: block-local indices of its columns
    int64_t ring_rows = 0;             // sum over blocks of rings 1..3 (the redundant first-sweep rows)
};
// AP: the A*P operator of the level's first sweep (same rows as A)
std::string build_sweep_plan(const SellPattern& A, const SellPattern& AP, SweepPlan& out);

struct HostPlan {
    int64_t n_own = 0, n_loc = 0, ne = 0;
    std::vector<int32_t> krank;        // k-d rank of each internal owned vertex (aggregate = krank / 4)
    std::vector<AmgLevelPlan> amg;     // amg[l]: level l -> l+1
    std::vector<int32_t> perm;     // internal -> external local vertex id (n_loc)
    std::vector<int32_t> iperm;    // external -> internal
    std::vector<double> xy;        // internal order, 2*n_loc
    std::vector<int32_t> cells;    // 3*ne, internal vertex ids, caller's cell order
    SellPattern A;
    std::vector<int32_t> lastcell; // T*(v) for owned v
    // assembly
    int cells_max = 0, slices_max = 0;
    std::vector<int32_t> blk_slice0;   // nblk+1 : slices [blk_slice0[b], blk_slice0[b+1])
    std::vector<int32_t> blk_cellptr;  // nblk+1 into blk_cells
    std::vector<int32_t> blk_cells;    // cell ids, ascending inside a block
    std::vector<int32_t> incptr;       // n_own+1 into inccode
    std::vector<uint16_t> inccode;     // (block-local cell slot << 2) | local vertex index
    std::vector<uint32_t> slotsrc;     // per SELL slot: the <= 2 staged cells of an off-diagonal entry, 14 bits each
                                       // (bits 0-13 and 14-27): (cell slot << 4) | (3 li + lj); cell slot kSrcNone = none
                                       // (diagonal / padding).  Bits 28-29 are the slot's Dirichlet code, written on
                                       // the device by shk_set_dirichlet (0 keep, 1 zero, 2 one).
    int max_inc_per_block = 0;
    // staging of the nodal fields: a block's vertices are its owned rows (local ids 0 .. rows-1, loaded coalesced)
    // followed by its halo vertices (the other vertices of its cells, gathered)
    std::vector<int32_t> blk_haloptr;  // nblk+1 into blk_halo
    std::vector<int32_t> blk_halo;     // internal vertex ids, ascending inside a block
    std::vector<uint16_t> blk_cellv;   // 4 per staged cell: local ids of its three vertices, 0
    int verts_max = 0;                 // largest rows + halo of a block (LDS stride of the staged fields)
    std::vector<int32_t> blk_desc;     // kBlkDesc ints per block: first slice, slices, first staged cell, cells, first
                                       // halo entry, halo vertices, first SELL slot, end slot, first incidence entry,
                                       // incidence entries -- ONE load tells a workgroup everything it fetches next
};

// xy: (n_loc,2) external order; cells: (ne,3) external local ids, every cell must touch at least one
// owned vertex (id < n_own); ghosts are ids [n_own, n_loc).  Returns "" or an error message.
std::string build_plan(int64_t n_own, int64_t n_loc, int64_t ne, const double* xy, const int32_t* cells,
                       const PlanOptions& opt, HostPlan& out);

std::string build_amg(HostPlan& P, const PlanOptions& opt);
std::string coarsen(const SellPattern& Af, const std::vector<int32_t>& agg, const std::vector<int32_t>& colmap,
                    int32_t n_coarse, int32_t n_coarse_cols, bool dense, AmgLevelPlan& L);

// Replicated coarse levels of a decomposed hierarchy (shk_amg.hip: amg_setup_distributed).
//   coarse_rows          the rows of the coarse operator that `agg` / `colmap` (global column ids) produce from Af:
//                        CSR, sorted unique columns, the diagonal (column diag0 + I) first
//   sell_from_csr        SELL-64 pattern of such rows (all subdomains' rows concatenated = the global level)
//   coarsen_onto_global  the transfer plan of a subdomain's level onto rows [row0, row0 + n_coarse) of G
//   build_amg_levels     a block-local hierarchy (like build_amg) on any SELL operator; krank = k-d rank of its rows
std::string coarse_rows(const SellPattern& Af, const std::vector<int32_t>& agg, const std::vector<int32_t>& colmap,
                        int32_t n_coarse, int32_t diag0, std::vector<int32_t>& rp, std::vector<int32_t>& ci);
std::string sell_from_csr(int32_t n_rows, int32_t n_cols, const std::vector<int32_t>& rp,
                          const std::vector<int32_t>& ci, SellPattern& C, std::vector<int32_t>& diag_slot);
std::string coarsen_onto_global(const SellPattern& Af, const std::vector<int32_t>& agg,
                                const std::vector<int32_t>& colmap, int32_t n_coarse, int32_t row0,
                                const SellPattern& G, AmgLevelPlan& L);
std::string build_amg_levels(const SellPattern& A0, const std::vector<int32_t>& krank, const PlanOptions& opt,
                             std::vector<AmgLevelPlan>& out);

// External CSR (rows = external owned ids, columns external local ids ascending) of a SELL pattern.
void sell_to_csr(const HostPlan& P, const double* sell_vals, std::vector<int32_t>& rowptr,
                 std::vector<int32_t>& colidx, std::vector<double>* vals);

}  // namespace shk
