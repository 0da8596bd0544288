// Experiment switches of the library: every SHK_* environment variable it honours is read HERE, once per process,
// into one struct; `overrides` lists the ones that were set ("NAME=value, ..."), which shk_env_overrides() reports and
// the Python mirror's md.solve() logs -- a stray variable can no longer change a run unnoticed.  Defaults are what
// every committed measurement was taken with; none of these switches can change WHAT is computed (only block shapes,
// placements, schedules and the preconditioner's tuning).  The assembly kernel's timing ablations, which do produce
// wrong results by construction, exist only in builds compiled with -DSHK_EXPERIMENTS (`make probe`).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>

namespace shk {

struct Tunables {
    // assembly plan / numbering
    int asm_slices = 4;           // SHK_ASM_SLICES   slices of 64 rows per assembly workgroup, 1..4 (the kernel's limit)
    int asm_cells = 640;          // SHK_ASM_CELLS    cells one assembly workgroup stages
    int sort_window = 256;        // SHK_SORT_WINDOW  rows per row-length sorting window
    bool reorder = true;          // SHK_REORDER      0: keep the caller's numbering (disables the hierarchy)
    int xcd = -1;                 // SHK_XCD          force the sweep placement (0 streaming, 1 XCD-contiguous); -1 = by size
    // multigrid hierarchy and cycle
    bool amg = true;              // SHK_AMG
    int amg_coarsest = 4096;      // SHK_AMG_COARSEST cap of the dense coarsest level
    double amg_alpha = 0.0;       // SHK_AMG_ALPHA    over-correction (0 = the hierarchy's default 1.5)
    int amg_coarse4 = -1;         // SHK_AMG_COARSE4  0: two sweeps on every level
    int amg_coarse4_from = 0;     // SHK_AMG_COARSE4_FROM  first level with the four-sweep sequence (0 = default 1)
    int amg_dense_period = 0;     // SHK_AMG_DENSE_PERIOD  steps between rebuilds of the dense inverse (0 = default 8)
    int64_t amg_w_rows = 0;       // SHK_AMG_W_ROWS   cycle doubling at the first level of at most that many rows
    double amg_damp_scale = 0.0;  // SHK_AMG_DAMP_SCALE  every damping x f (robustness tests)
    int amg_lanczos = 32;         // SHK_AMG_LANCZOS  Lanczos steps behind the damping caps
    bool amg_reuse = true;        // SHK_AMG_REUSE    0: every Newton iteration refreshes the whole hierarchy
    int amg_lambda_period = 4;    // SHK_AMG_LAMBDA_PERIOD  dense refreshes between renewals of the spectral estimates
    bool fused_restrict = true;   // SHK_FUSED_RESTRICT  0: one launch per restriction
    bool amg_fused_sweeps = true; // SHK_AMG_FUSED_SWEEPS  0: one launch per smoothing sweep on the small levels
    double amg_w1 = 0.0, amg_w2 = 0.0;   // SHK_AMG_W1 / W2  absolute dampings of the two finest-level sweeps
    int amg_halo_levels = -1;     // SHK_AMG_HALO_LEVELS  decomposed levels whose sweeps see their neighbours (-1 = all)
    int64_t amg_rep_rows = -1;    // SHK_AMG_REP_ROWS global size from which the coarse levels are replicated (-1 = default)
    int amg_ghost_exchange = -1;  // SHK_AMG_GHOST_EXCHANGE  1: exchange ghosts after the first sweep of a decomposed level
                                  //                  (rounds 1-2); 0: ghosts hold the prolongated coarse correction (default)
    int gal_ilp[2] = {1, 4}, gal_grid[2] = {1024, 2048}, gal_contig[2] = {1, 0};   // SHK_GAL_ILP0/1, _GRID0/1, _CONTIG0/1
    bool gj_pivotwise = false;    // SHK_GJ_PIVOTWISE round 1's dense inverse (cross-checks)
    // Krylov driver
    int warm_its = 3;             // SHK_WARM_ITS     Newton iterations of a step that are warm-started (1..3)
    int krylov_chunk = 0;         // SHK_KRYLOV_CHUNK iterations enqueued per stop-flag poll (0 = automatic)
    double krylov_near = 0.0;     // SHK_KRYLOV_NEAR  stop queueing ahead within that factor of the target (0 = off)
    int predict_last = 1;         // SHK_PREDICT_LAST 0: every Newton iteration ends with a full residual + Jacobian pass
    // communication
    double comm_timeout_s = 300.0;   // SHK_COMM_TIMEOUT_S  deadline of host waits while RCCL collectives are in flight
    int overlap = -1;             // SHK_OVERLAP      interior / boundary overlap of the finest level's exchanges (-1 = default off)
    bool debug = false;           // SHK_DEBUG        print the smoother's spectral estimates and damping caps
    int asm_ablate = 0;           // SHK_ASM_ABLATE   honoured by -DSHK_EXPERIMENTS builds only
    std::string overrides;        // "NAME=value, ..." of every variable above that is set in the environment
    int n_overrides = 0;
};

inline const Tunables& tunables() {
    static const Tunables T = [] {
        Tunables t;
        auto raw = [&](const char* name) -> const char* {
            const char* s = std::getenv(name);
            if (s) {
                if (!t.overrides.empty()) t.overrides += ", ";
                t.overrides += std::string(name) + "=" + s;
                ++t.n_overrides;
            }
            return s;
        };
        auto geti = [&](const char* name, int& v) { if (const char* s = raw(name)) v = std::atoi(s); };
        auto getl = [&](const char* name, int64_t& v) { if (const char* s = raw(name)) v = std::atoll(s); };
        auto getd = [&](const char* name, double& v) { if (const char* s = raw(name)) v = std::atof(s); };
        auto getb = [&](const char* name, bool& v) { if (const char* s = raw(name)) v = std::atoi(s) != 0; };
        geti("SHK_ASM_SLICES", t.asm_slices); geti("SHK_ASM_CELLS", t.asm_cells); geti("SHK_SORT_WINDOW", t.sort_window);
        getb("SHK_REORDER", t.reorder); geti("SHK_XCD", t.xcd);
        getb("SHK_AMG", t.amg); geti("SHK_AMG_COARSEST", t.amg_coarsest); getd("SHK_AMG_ALPHA", t.amg_alpha);
        geti("SHK_AMG_COARSE4", t.amg_coarse4); geti("SHK_AMG_COARSE4_FROM", t.amg_coarse4_from);
        geti("SHK_AMG_DENSE_PERIOD", t.amg_dense_period); getl("SHK_AMG_W_ROWS", t.amg_w_rows);
        getd("SHK_AMG_DAMP_SCALE", t.amg_damp_scale); geti("SHK_AMG_LANCZOS", t.amg_lanczos); getb("SHK_AMG_REUSE", t.amg_reuse);
        geti("SHK_AMG_LAMBDA_PERIOD", t.amg_lambda_period); getb("SHK_FUSED_RESTRICT", t.fused_restrict);
        getb("SHK_AMG_FUSED_SWEEPS", t.amg_fused_sweeps);
        getd("SHK_AMG_W1", t.amg_w1); getd("SHK_AMG_W2", t.amg_w2); geti("SHK_AMG_HALO_LEVELS", t.amg_halo_levels);
        getl("SHK_AMG_REP_ROWS", t.amg_rep_rows); geti("SHK_AMG_GHOST_EXCHANGE", t.amg_ghost_exchange);
        geti("SHK_GAL_ILP0", t.gal_ilp[0]); geti("SHK_GAL_ILP1", t.gal_ilp[1]);
        geti("SHK_GAL_GRID0", t.gal_grid[0]); geti("SHK_GAL_GRID1", t.gal_grid[1]);
        geti("SHK_GAL_CONTIG0", t.gal_contig[0]); geti("SHK_GAL_CONTIG1", t.gal_contig[1]);
        getb("SHK_GJ_PIVOTWISE", t.gj_pivotwise);
        geti("SHK_WARM_ITS", t.warm_its); geti("SHK_KRYLOV_CHUNK", t.krylov_chunk); getd("SHK_KRYLOV_NEAR", t.krylov_near);
        geti("SHK_PREDICT_LAST", t.predict_last);
        getd("SHK_COMM_TIMEOUT_S", t.comm_timeout_s); geti("SHK_OVERLAP", t.overlap); getb("SHK_DEBUG", t.debug);
#ifdef SHK_EXPERIMENTS
        geti("SHK_ASM_ABLATE", t.asm_ablate);
        t.overrides += std::string(t.overrides.empty() ? "" : ", ") + "[probe build: -DSHK_EXPERIMENTS]";
#endif
        if (t.amg_lambda_period < 1) t.amg_lambda_period = 1;
        if (t.warm_its < 1) t.warm_its = 1;
        return t;
    }();
    return T;
}

}  // namespace shk
