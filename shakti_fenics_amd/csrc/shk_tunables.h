// Experiment switches of the library: every SHK_* environment variable it honours is read HERE, once per process,
// into one struct; overrides() lists the ones that are set ("NAME=value, ..."), which shk_env_overrides() reports and
// the Python mirror's md.solve() logs -- a stray variable can no longer change a run unnoticed.  After that first read
// the set only changes through the explicit call shk_tunable_set (tests, probes), which is recorded the same way.  Defaults are what
// every committed measurement was taken with; none of these switches can change WHAT is computed (only block shapes,
// placements, schedules and the preconditioner's tuning).  The assembly kernel's timing ablations, which do produce
// wrong results by construction, exist only in builds compiled with -DSHK_EXPERIMENTS (`make probe`).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <string>

namespace shk {

struct Tunables {
    // assembly plan / numbering
    int asm_slices = 4;           // SHK_ASM_SLICES   slices of 64 rows per assembly workgroup, 1..4 (the kernel's limit)
    int asm_cells = 640;          // SHK_ASM_CELLS    cells one assembly workgroup stages
    int sort_window = 256;        // SHK_SORT_WINDOW  rows per row-length sorting window
    bool sort_rim = true;         // SHK_SORT_RIM     0: rows of one length inside a window keep their k-d order (rounds 1-2)
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
    int64_t amg_fused_rows = 320000;   // SHK_AMG_FUSED_ROWS  largest level (rows) whose four sweeps run in one launch
    int64_t amg_bf16_rows = 500000;    // SHK_AMG_BF16_ROWS  levels of at least that many rows smooth on the packed bfloat16 copy
                                       //                  of their operator (DevSell::pk); 0: float values everywhere
    double amg_w1 = 0.0, amg_w2 = 0.0;   // SHK_AMG_W1 / W2  absolute dampings of the two finest-level sweeps
    int amg_halo_levels = -1;     // SHK_AMG_HALO_LEVELS  decomposed levels whose sweeps see their neighbours (-1 = all)
    int64_t amg_rep_rows = -1;    // SHK_AMG_REP_ROWS global size from which the coarse levels are replicated (-1 = default)
    int amg_ghost_exchange = -1;  // SHK_AMG_GHOST_EXCHANGE  bit mask of the decomposed levels that exchange ghosts after their
                                  //                  first sweep (-1 = default 1: the finest level only; 7 = rounds 1-2)
    int amg_e_exchange = -1;      // SHK_AMG_E_EXCHANGE  bit mask of frozen levels whose result is still exchanged for the finer
                                  //                  level's A*P sweep (-1 = all)
    int gal_ilp[2] = {1, 4}, gal_grid[2] = {1024, 2048}, gal_contig[2] = {1, 0};   // SHK_GAL_ILP0/1, _GRID0/1, _CONTIG0/1
    bool gj_pivotwise = false;    // SHK_GJ_PIVOTWISE round 1's dense inverse (cross-checks)
    // Krylov driver
    int warm_its = 3;             // SHK_WARM_ITS     Newton iterations of a step that are warm-started (1..3)
    int krylov_chunk = 0;         // SHK_KRYLOV_CHUNK iterations enqueued per stop-flag poll (0 = automatic)
    double krylov_near = 0.0;     // SHK_KRYLOV_NEAR  stop queueing ahead within that factor of the target (0 = off)
    bool amg_warm_s = true;           // SHK_AMG_WARM_S  0: the cycle on s reads only the float copy (k_spmv<2> then finds s cold)
    bool krylov_early_check = true;   // SHK_KRYLOV_EARLY_CHECK 0: the stop test only inside k_bicg_s (one cycle + product later)
    int predict_last = 1;         // SHK_PREDICT_LAST 0: every Newton iteration ends with a full residual + Jacobian pass
    // communication
    double comm_timeout_s = 300.0;   // SHK_COMM_TIMEOUT_S  deadline of host waits while RCCL collectives are in flight
    int overlap = -1;             // SHK_OVERLAP      interior / boundary overlap of the finest level's exchanges (-1 = default off)
    bool debug = false;           // SHK_DEBUG        print the smoother's spectral estimates and damping caps
    int asm_ablate = 0;           // SHK_ASM_ABLATE   honoured by -DSHK_EXPERIMENTS builds only
    // every switch that is set: by the environment at first use, or by shk_tunable_set since ("NAME=value, ...")
    std::map<std::string, std::string> set;
    std::string overrides() const {
        std::string o;
        for (const auto& kv : set) o += (o.empty() ? "" : ", ") + kv.first + "=" + kv.second;
#ifdef SHK_EXPERIMENTS
        o += std::string(o.empty() ? "" : ", ") + "[probe build: -DSHK_EXPERIMENTS]";
#endif
        return o;
    }
};

// name -> field.  Returns false for a name the library does not honour.
inline bool tunable_apply(Tunables& t, const std::string& name, const char* s) {
    auto I = [&](int& v) { v = std::atoi(s); return true; };
    auto L = [&](int64_t& v) { v = std::atoll(s); return true; };
    auto D = [&](double& v) { v = std::atof(s); return true; };
    auto B = [&](bool& v) { v = std::atoi(s) != 0; return true; };
    if (name == "SHK_ASM_SLICES") return I(t.asm_slices);
    if (name == "SHK_ASM_CELLS") return I(t.asm_cells);
    if (name == "SHK_SORT_WINDOW") return I(t.sort_window);
    if (name == "SHK_SORT_RIM") return B(t.sort_rim);
    if (name == "SHK_REORDER") return B(t.reorder);
    if (name == "SHK_XCD") return I(t.xcd);
    if (name == "SHK_AMG") return B(t.amg);
    if (name == "SHK_AMG_COARSEST") return I(t.amg_coarsest);
    if (name == "SHK_AMG_ALPHA") return D(t.amg_alpha);
    if (name == "SHK_AMG_COARSE4") return I(t.amg_coarse4);
    if (name == "SHK_AMG_COARSE4_FROM") return I(t.amg_coarse4_from);
    if (name == "SHK_AMG_DENSE_PERIOD") return I(t.amg_dense_period);
    if (name == "SHK_AMG_W_ROWS") return L(t.amg_w_rows);
    if (name == "SHK_AMG_DAMP_SCALE") return D(t.amg_damp_scale);
    if (name == "SHK_AMG_LANCZOS") return I(t.amg_lanczos);
    if (name == "SHK_AMG_REUSE") return B(t.amg_reuse);
    if (name == "SHK_AMG_LAMBDA_PERIOD") { I(t.amg_lambda_period); if (t.amg_lambda_period < 1) t.amg_lambda_period = 1; return true; }
    if (name == "SHK_FUSED_RESTRICT") return B(t.fused_restrict);
    if (name == "SHK_AMG_FUSED_SWEEPS") return B(t.amg_fused_sweeps);
    if (name == "SHK_AMG_FUSED_ROWS") return L(t.amg_fused_rows);
    if (name == "SHK_AMG_BF16_ROWS") return L(t.amg_bf16_rows);
    if (name == "SHK_AMG_W1") return D(t.amg_w1);
    if (name == "SHK_AMG_W2") return D(t.amg_w2);
    if (name == "SHK_AMG_HALO_LEVELS") return I(t.amg_halo_levels);
    if (name == "SHK_AMG_REP_ROWS") return L(t.amg_rep_rows);
    if (name == "SHK_AMG_GHOST_EXCHANGE") return I(t.amg_ghost_exchange);
    if (name == "SHK_AMG_E_EXCHANGE") return I(t.amg_e_exchange);
    if (name == "SHK_GAL_ILP0") return I(t.gal_ilp[0]);
    if (name == "SHK_GAL_ILP1") return I(t.gal_ilp[1]);
    if (name == "SHK_GAL_GRID0") return I(t.gal_grid[0]);
    if (name == "SHK_GAL_GRID1") return I(t.gal_grid[1]);
    if (name == "SHK_GAL_CONTIG0") return I(t.gal_contig[0]);
    if (name == "SHK_GAL_CONTIG1") return I(t.gal_contig[1]);
    if (name == "SHK_GJ_PIVOTWISE") return B(t.gj_pivotwise);
    if (name == "SHK_WARM_ITS") { I(t.warm_its); if (t.warm_its < 1) t.warm_its = 1; return true; }
    if (name == "SHK_KRYLOV_CHUNK") return I(t.krylov_chunk);
    if (name == "SHK_KRYLOV_NEAR") return D(t.krylov_near);
    if (name == "SHK_AMG_WARM_S") return B(t.amg_warm_s);
    if (name == "SHK_KRYLOV_EARLY_CHECK") return B(t.krylov_early_check);
    if (name == "SHK_PREDICT_LAST") return I(t.predict_last);
    if (name == "SHK_COMM_TIMEOUT_S") return D(t.comm_timeout_s);
    if (name == "SHK_OVERLAP") return I(t.overlap);
    if (name == "SHK_DEBUG") return B(t.debug);
#ifdef SHK_EXPERIMENTS
    if (name == "SHK_ASM_ABLATE") return I(t.asm_ablate);
#endif
    return false;
}

inline const char* const* tunable_names(int* n) {
    static const char* const names[] = {
        "SHK_ASM_SLICES", "SHK_ASM_CELLS", "SHK_SORT_WINDOW", "SHK_SORT_RIM", "SHK_REORDER", "SHK_XCD", "SHK_AMG", "SHK_AMG_COARSEST",
        "SHK_AMG_ALPHA", "SHK_AMG_COARSE4", "SHK_AMG_COARSE4_FROM", "SHK_AMG_DENSE_PERIOD", "SHK_AMG_W_ROWS",
        "SHK_AMG_DAMP_SCALE", "SHK_AMG_LANCZOS", "SHK_AMG_REUSE", "SHK_AMG_LAMBDA_PERIOD", "SHK_FUSED_RESTRICT",
        "SHK_AMG_FUSED_SWEEPS", "SHK_AMG_FUSED_ROWS", "SHK_AMG_BF16_ROWS", "SHK_AMG_W1", "SHK_AMG_W2", "SHK_AMG_HALO_LEVELS", "SHK_AMG_REP_ROWS",
        "SHK_AMG_GHOST_EXCHANGE", "SHK_AMG_E_EXCHANGE", "SHK_GAL_ILP0", "SHK_GAL_ILP1", "SHK_GAL_GRID0", "SHK_GAL_GRID1", "SHK_GAL_CONTIG0",
        "SHK_GAL_CONTIG1", "SHK_GJ_PIVOTWISE", "SHK_WARM_ITS", "SHK_KRYLOV_CHUNK", "SHK_KRYLOV_NEAR", "SHK_AMG_WARM_S", "SHK_KRYLOV_EARLY_CHECK", "SHK_PREDICT_LAST",
        "SHK_COMM_TIMEOUT_S", "SHK_OVERLAP", "SHK_DEBUG", "SHK_ASM_ABLATE"};
    *n = (int)(sizeof(names) / sizeof(names[0]));
    return names;
}

// The process-wide set: initialised from the environment at first use, changed afterwards only through
// shk_tunable_set (tests and probes: an explicit call, recorded like an environment override).
inline Tunables& tunables_mut() {
    static Tunables T = [] {
        Tunables t;
        int n = 0;
        const char* const* names = tunable_names(&n);
        for (int i = 0; i < n; ++i)
            if (const char* s = std::getenv(names[i]))
                if (tunable_apply(t, names[i], s)) t.set[names[i]] = s;
        return t;
    }();
    return T;
}
inline const Tunables& tunables() { return tunables_mut(); }

// value == nullptr: back to the default.  Returns false for an unknown name.
inline bool tunable_set(const std::string& name, const char* value) {
    Tunables& T = tunables_mut();
    Tunables probe;
    if (!tunable_apply(probe, name, value ? value : "0")) return false;
    std::map<std::string, std::string> keep = T.set;
    if (value) keep[name] = value; else keep.erase(name);
    Tunables fresh;
    for (const auto& kv : keep) tunable_apply(fresh, kv.first, kv.second.c_str());
    fresh.set = keep;
    T = fresh;
    return true;
}

}  // namespace shk
