// dppr_host_state.hpp -- host side of libdppr_hip.so, part 1 of 4: what an engine OWNS (replaces gpu/DeviceMemory.cuh,
// gpu/GPUEdgeBatch.cuh and the members of gpu/SlidingGraphBuilder.cuh) -- the resident graph epochs, the per-source slots and source
// groups, the two HIP streams (solver / builder) and their scratch -- plus the small helpers every other part uses (error
// reporting, the bounded stream wait, the id lookahead of dppr_hint_next_batch).
// Included by dppr_engine.hip only, after the kernel headers (one translation unit: the kernels are templates).
#pragma once

static constexpr int MAX_CHUNK = 64;
// Slot::cnt: [0..2] rotating frontier counters, [3] phase-1 candidates, [4] scratch, [5..6] big-row
// counters, [7] list scratch / status word of a resident launch; the log of a launch (up to
// 2 x MAX_CHUNK entries: a whole batch) follows the header
static constexpr int CNT_HDR = 16;
static constexpr int PERSIST_RETRY_BATCHES = 64; // after a failed roll-call: batches on per-iteration launches before the next try
static constexpr int GMULTI_MAX = 2 * MAX_CHUNK; // sweeps a multi-sweep launch of a source group may run
static constexpr int GQ_PAD = 32;               // ints between the rotating group counters of k_gsweep (own 128-byte line each)
static constexpr int BIN_MAX_BLOCKS = 1 << 16, BIN_MAX_BIG = 4096, BIN_SMALL_INTS = BIN_MAX_BLOCKS + BIN_MAX_BIG + 1 + 64; // bin_cut scratch
static constexpr int SU_SPLIT_MIN = 1 << 16;  // batch records from which IncrementalBatchUpdate runs as k_su_terms + k_su_apply (stream_update)
static constexpr int MERGE_MISS_WORD = 44; // word of hub_hist (64 ints) that counts the retired keys a slide's merge did not find
static constexpr int RESIDENT_MARGIN = 8; // sweeps a resident launch is given beyond what the last batch needed

namespace {

struct Epoch {
    int *row_ptr = nullptr; // V+1   in-CSR (push)
    Adj *adj = nullptr;     // Ed
    int *out_row_ptr = nullptr; // V+1  out-CSR (pull)
    int *out_col = nullptr;     // Ed
    int Ed = 0;
    // batch that produced this epoch (empty for epoch 0)
    int *b1 = nullptr, *b2 = nullptr, *deg_after = nullptr; // 4c each
    uint8_t *ins = nullptr;
    int L = 0;
    uint32_t *sk = nullptr, *sv = nullptr; // the batch's records grouped by tail at slide time: tails ascending, record indices (stable)
    bool grouped = false;
    int id = -1; // global epoch number stored in this ring entry
    // sweep groups: tiles [grp_tile[g], grp_tile[g+1]) per workgroup, about equal edges each
    int *grp_tile = nullptr; // V/64 + 2
    int n_groups = 0;
    int grp_n_int = 0;       // internal ids covered by the table
    // the same for the source-group sweeps (k_gsweep), cut once a source group exists: many small groups --
    // two workgroups per CU and an even spread matter there, a bound on the group count does not; at
    // most 512 vertices each when a 16-wide source group exists
    int *ggrp_tile = nullptr;
    int n_ggroups = 0;
    int ggrp_max_tiles = 0;
    int *gtab = nullptr;     // row tables of those groups (k_gtables): GT_STRIDE(512 | 1024) ints per group
    size_t gtab_cap = 0;     // ints allocated
    // hub directory of this epoch (vertices whose pushes are aggregated in LDS)
    int *hub_v = nullptr, *hub_degp1 = nullptr;
    int n_hubs = 0;
    // binned sweep (dppr_binned.hpp): block cuts (first tile of every A- / B-block), per edge the head index inside
    // its A-block + B-major position (A-major order) and the row index inside its B-block (B-major order)
    int *acut = nullptr, *bcut = nullptr; // first vertex of every A- / B-block (one allocation; bcut points into it)
    size_t bin_tab_cap = 0;
    int n_a = 0, n_b = 0;
    BinChunk *chunks = nullptr;           // work items of k_bin_scatter
    size_t chunk_cap = 0;
    int n_chunks = 0;
    uint16_t *hl = nullptr, *dl = nullptr; // hl: per RUN (A-major), head index + first-of-tile bit; dl: per edge (B-major), row index + first-of-run bit
    int *vb = nullptr, *tb = nullptr;      // per aligned block of 64 edges: run that holds its first edge; per 64 runs: tile that holds the first run
    int *tdelta = nullptr;                 // per tile (A-major order): B-major run index - A-major run index of its runs
    size_t tdelta_cap = 0;
    int n_runs = 0, n_tiles = 0;
    bool bin_valid = false;
    int bin_n_int = 0; // internal ids the tables cover (<= grp_n_int: later ids have no edge in this epoch)
    // slot table of the resident sweep (dppr_resident.hpp: k_res_slots), rebuilt with every group cut
    uint32_t *res_pk = nullptr; // Ed entries, group by group, sorted by gather position
    size_t res_pk_cap = 0;
    bool res_valid = false;
    // the batch's records (sk / sv) cut into the sweep groups' ranges, for IncrementalBatchUpdate inside a resident launch
    int *su_rng = nullptr;      // n_groups + 1 first-record indices, then two result words of k_res_rec_ranges
    size_t su_rng_cap = 0;
    bool su_inline = false;     // every group's range fits the launch's workgroup and no tail lies beyond the groups
};

struct Slot {
    int source = 0;     // internal id
    int source_ext = 0; // id the caller gave
    double *p = nullptr, *r = nullptr;
    double *x = nullptr, *x2 = nullptr; // dense per-iteration push amounts (x) and pull output (x2)
    uint32_t *act[2] = {nullptr, nullptr}; // activity bitmaps of x / x2 for sweeps on windows that cannot run resident
    size_t act_bytes = 0;
    int *ft[2] = {nullptr, nullptr};
    int *neg = nullptr;     // phase-1 candidates
    int *status = nullptr;  // status-array duplicate filter (variants EAGER / VANILLA): launch number that queued the vertex last; allocated on first use
    int *cnt = nullptr;     // [0..2] rotating frontier counters, [3] neg candidates, [4] scratch, [5..6] big-row counters
    BigItem *big = nullptr; // deferred big rows of the current iteration
    int *log = nullptr;     // per-chunk log: frontier size seen by each enqueued iteration
    long long iter_seq = 0; // running iteration number (selects the big-row counter)
    double sweep_us = 0;       // binned windows: running mean of a sweep's time (the push / sweep decision)
    double atomic_ns = 1.0 / 23.5; // ... and of a push iteration's time per in-edge (starts at the chip's rate of returning f64 atomics)
    int iter_hint[2] = {0, 0}; // iterations the last loop of each phase took (sizes the next chunks)
    int iter_hist[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // ... and the last four
    bool start_dense[2] = {false, false}; // the last loop of each phase began with a frontier worth a sweep
    int last_F0[2] = {0, 0};   // ... and its size
    IterStats *dstats = nullptr; // two blocks: [0] push iterations (and resident launches), [1] dense sweeps -- the roofline of the sweep kernel counts its own edges
    bool converged = false; // |r| <= eps everywhere (state after a completed solve)
    double conv_eps = 0.0;
    double park_eps = 0.0;  // parked rows satisfy |r| <= park_eps (0: they are exactly zero)
    int last_epoch = -2;    // epoch whose batch was applied last (-2: unknown, e.g. after dppr_write: anything goes)
    bool seed_lists_valid = false; // ft[0]/cnt[0] and neg/cnt[3] hold the lists of the last dppr_incremental_batch_update
    bool phase0_done = false; // ExecuteMainLoop(0) completed since the last modification
    double phase0_eps = 0.0;
    dppr_stats_t st{};
    bool trace = false;
    std::vector<int64_t> trace_off;
    std::vector<int32_t> trace_ids;
};

// f2: up to 16 sources solved together on interleaved state (dppr_multi.hpp)
struct Group {
    int n = 0;                 // sources in use (1..16)
    int spl = 1;               // doubles per lane of an octet: 1 (rows of <= 8 doubles) or 2
    int gw = OCT;              // doubles per vertex = row_width(n): 2, 4, .. 16 (dppr_multi.hpp)
    int src_ext[GS_MAX] = {0}; // ids the caller gave
    SrcN src{};                // internal ids, -1 = unused lane
    double *p = nullptr, *r = nullptr, *x = nullptr, *x2 = nullptr; // [V][gw]
    uint32_t *act[2] = {nullptr, nullptr}; // activity bitmaps that go with x / x2
    size_t act_bytes = 0;
    int *cnt = nullptr;        // [3][GS_MAX] rotating frontier sizes, then the per-chunk log [MAX][GS_MAX]
    int *gq = nullptr;         // one-sweep launches: three rotating group counters (a launch takes tickets from one and zeroes the next), GQ_PAD ints apart
    unsigned gq_seq = 0;       // one-sweep launches enqueued so far
    int *mlog = nullptr;       // multi-sweep launches: [GS_MAX] status word + padding, then one row of frontier sizes per sweep
    IterStats *dstats = nullptr; // two blocks: [0] push iterations (and resident launches), [1] dense sweeps -- the roofline of the sweep kernel counts its own edges
    dppr_stats_t st{};
    int iter_hint[2] = {0, 0};
    int iter_hist[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // sweeps the last four loops of each phase took
    int dense_hist[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // ... before the frontier was small enough for the push form
    // the tail of a loop as pushes (dppr_gpush.hpp): vertex lists, scan, control block; allocated on first use
    int *plist[2] = {nullptr, nullptr};
    int *ppre = nullptr;
    GPushCtl *pctl = nullptr;
    int plist_cap = 0;
    bool converged = false;    // |r| <= conv_eps for every source (state after a completed solve)
    double conv_eps = 0.0;
    double park_eps = 0.0;     // parked rows satisfy |r| <= park_eps
    int last_epoch = -2;       // epoch whose batch was applied last (-2: unknown)
};

} // namespace

struct dppr_engine : dppr::IdSpace { // (the id maps, the parked zone and the pending row moves: dppr_idspace.hpp)
    int device = 0;
    int V = 0, W = 0, c = 0, directed = 1, n_epochs = 1;
    int Ed = 0;   // directed edges in the window
    int bits = 1; // bits of a vertex id
    int schedule = DPPR_SCHEDULE_EAGER;
    // the reference's variants (-o, gpu/PPRRevPushGPUVariants.cuh) as mechanisms of the push iterations: dppr_set_variant
    bool status_dedup = false; // duplicate filter of a push iteration: status array (EAGER 2, VANILLA 3) instead of the threshold crossing
    bool pre_extract = false;  // synchronous push iterations zero residual[u] at the snapshot (InspectExtra: FAST_FRONTIER 1, VANILLA 3) instead of repairing
    bool group_at_slide = false; // dppr_set_batch_grouping(1): the batch's records are grouped by tail (and CopyOutDegree done) when the batch is uploaded
                                 // (dppr_slide); default since round 5: inside dppr_update, where the reference times them (gpu/PPRGPU.cuh:138-164)
    int merge_miss_host = 0;        // retired keys the last slide's merge did not find (read back with the build's synchronisations)
    long long merge_fallbacks = 0;  // slides that re-sorted the window because of that
    bool test_force_merge_miss = false; // (test hook, DPPR_TEST_MERGE_MISS=1: every incremental slide takes the fallback)
    bool launch_called_off = false; // batch_ahead: the last whole-batch launch changed nothing (roll-call failed, or a group had too many records)
    int raw_backoff = 0;         // batches for which a resident launch does not take the records raw (after one called itself off: a group with more records than threads)
    bool merge_phases = false; // dppr_set_phase_merge: one loop for residuals of both signs (eager schedule only)
    int merge_div = 4;         // ... run to eps / merge_div
    hipStream_t stream = nullptr; // the SOLVER's stream: IncrementalBatchUpdate, the frontier loops, reads and writes of p / r
    hipStream_t bs = nullptr;     // the graph BUILDER's stream (lowest priority): window ring, key merge, CSRs, group cuts and tables, binned tables,
                                  // id-space row moves. Every builder entry point ends with a host synchronisation of bs, every solver call with one of
                                  // `stream`, so calls made one after the other need no cross-stream event; dppr_slide_concurrent runs beside a solver call.
    bool build_concurrent = false; // (builder thread only) the slide in progress may run beside dppr_update / dppr_group_update on an OLDER epoch
    std::mutex err_mu;             // `err` is written by whichever of the two threads fails
    std::mutex map_mu;             // the id maps and the pending row moves: held by the builder from the first id it assigns to the end of the row
                                   // moves (dppr_set_batch; the id-assigning head of a slide), and by a reader on the solver thread (dppr_read,
                                   // dppr_group_read, settle_parked) from its copy of the map to the end of its gathers -- a read beside a
                                   // concurrent slide sees either the maps and rows of before a revival or those of after it
    unsigned map_gen_on_device = 0; // IdSpace::map_gen the device copy of ext2int was taken at
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evpool[2 * 64] = {};
    bool profiling = false;
    int pull_block = 0;   // sweep workgroup size pinned by dppr_set_tuning (0: 1024)
    int chunk_iters = 24; // iterations enqueued between two host read-backs of the frontier size
    bool chunk_explicit = false; // set by dppr_set_tuning: then it also caps what a resident launch is given
    // resident sweeps (dppr_resident.hpp)
    int persist_mode = 1;              // 1: use resident sweeps when an epoch's groups fit the chip at once
    bool persist_ok = true;            // cleared when a roll-call gives up: per-iteration launches until re-armed
    int persist_retry = 0;             // dppr_update calls until resident launches are tried again (0: not pending)
    int persist_cap = 0;               // co-resident workgroups of k_pull_resident at the sweep's block size
    int res_slots = 1;                 // 1: resident launches take their edge slots from the sorted slot table (0: CSR order)
    int res_update = 1;                // 1: a whole-batch resident launch applies the batch's records itself (PLAN_UPDATE)
    double *res_arena = nullptr;       // snapshot vectors of a resident launch (resident_arena)
    long long res_arena_stride = 0;    // doubles per vector
    unsigned long long persist_ticks = 5000000ull; // roll-call time limit in 100 MHz ticks (50 ms)
    int persist_rollcall_extra = 0;    // tests: the roll-call waits for a workgroup that does not exist
    GridBar *bar = nullptr;
    // window ring, stream order
    int *w1 = nullptr, *w2 = nullptr;
    int head = 0;
    bool loaded = false;
    bool broken = false; // a renumbering failed half way (HIP error after the host maps changed): every call but dppr_destroy is refused
    int *outdeg = nullptr;
    int *hub_slot_of = nullptr; // V, scratch of the CSR build (k_assign_hubs: ~hub slot, or out-degree + 1)
    int *hub_hist = nullptr;    // 32 + 1 ints (histogram, hub counter)
    int hub_min_degree = HUB_MIN_DEGREE_DEFAULT;
    int big_row = BIG_ROW_DEFAULT;
    int pull_min_frontier = 0; // 0: auto (max(1024, Ed/192)); < 0: never pull; > 0: pull when F >= value
    // CSR build: persistent sorted key arrays (in-orientation dst<<bits|src, out-orientation
    // src<<bits|dst; undirected graphs share one) + scratch of the same size
    uint64_t *in_sorted = nullptr, *out_sorted = nullptr;
    uint64_t *keys_a = nullptr, *keys_b = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int *su_grp = nullptr; // hand-written grouping of a large batch (dppr_update.hpp k_su_grp_*): histogram | cursors | bucket starts + slice starts
    // incremental maintenance (f1): batch keys, positions of the retired keys
    uint64_t *bk[4] = {nullptr, nullptr, nullptr, nullptr}; // del-in, ins-in, del-out, ins-out (unsorted)
    uint64_t *bks[4] = {nullptr, nullptr, nullptr, nullptr}; // the same, sorted
    int *delpos = nullptr; // positions of a slide's retired keys in the persistent sorted keys (2 * max_batch)
    struct Pre { // dppr_hint_next_batch
        std::future<bool> task;
        const int32_t *src[4] = {nullptr, nullptr, nullptr, nullptr};
        int n[4] = {0, 0, 0, 0};
        std::vector<int32_t> out[4];
        std::vector<uint32_t> miss[4]; // positions the lookup left at -1 (no id yet / parked), in array order
        unsigned long long epoch = 0;
        bool armed = false, ok = false;
    } pre;
    long long pre_hits = 0, pre_misses = 0; // id arrays that dppr_set_batch / dppr_slide took from the lookahead; entries resolved at the call
    bool incremental = true; // dppr_slide merges the batch into the sorted keys (false: full re-sort)
    bool sweep_bits = false;        // per-iteration single-source sweeps test an activity bitmap before each gather (dppr_set_sweep_bitmap;
                                    // measured slower on every stand-in, DESIGN.md section 6: off unless asked for)
    bool hot_blocks = true;         // vertex numbering in blocks of falling in-degree (DPPR_HOT_BLOCKS=0: two blocks, hot | rest)
    int gsweep_hot_rows = 24576;    // k_gsweep: rows below this id are gathered with the default cache policy, the others non-temporal (DPPR_GSWEEP_HOT)
    int gsweep_grid_cap = 0;        // workgroups of a one-sweep launch of k_gsweep: 0 = two per CU, what is resident at once (the groups beyond
                                    // the grid are dealt by a device counter; LiveJournal stand-in, 10 sources: 12.40 ms per batch at 512, 13.1-13.2 at
                                    // 768 / 1024 / 2048; DPPR_GSWEEP_GRID: tuning runs)
    bool group_resident = true;     // source groups on windows whose sweep groups are all resident run a loop as multi-sweep launches
    int gmulti_cap[2] = {-1, -1};   // co-resident workgroups of k_gsweep<.., true> per state width (-1: not queried yet)
    bool any_groups = false;        // a source group exists: epochs carry the second group table
    int ggroups_min = 256;          // ... of at least this many groups (DPPR_GGROUPS_MIN: tuning runs; 512 / 1008 measured slower on
                                    // the configs[1] stand-in, equal on the LiveJournal one)
    bool force_radix_grouping = false; // (A/B, tests: DPPR_GROUPING_RADIX=1 -- the in-region grouping always as key kernel + device radix sort)
    bool cost_model = true;         // binned windows: push or sweep by estimated cost (DPPR_COST_MODEL=0: by the vertex-count threshold)
    bool group_full_rows = false;   // (A/B, DPPR_GROUP_FULL_ROWS=1: rows of 64 / 128 bytes whatever the source count, as until round 3)
    bool wide_groups = false;       // ... one of more than 8 sources: its groups hold at most 512 vertices
    bool group_tail_seeding = true; // source groups seed from the batch tails after a converged solve (false: dense Inspect)
    int gpush_enter_pairs = -1;     // a group's loop switches to the push form below this many frontier pairs (-1: automatic,
                                    // max(64, gpush_auto_factor pairs per sweep group); 0: never) -- dppr_set_group_push
    int gpush_list_cap = 1 << 20;   // vertices a frontier list of that form holds
    long long gpush_max_edges = 0;  // in-edges one iteration of that form may have (0: from the sweep's floor, 200 per sweep group)
    int gpush_auto_factor = 2;      // automatic threshold: this many pairs per sweep group (DPPR_GROUP_PUSH_FACTOR: tuning runs)
    // binned sweep of single-source loops on windows far beyond the L2s (dppr_binned.hpp, dppr_set_binned_sweep)
    int bin_mode = 1;               // 0: never, 1: when a source slot exists and the window has >= bin_min_ids vertices, 2: always
    int bin_ha_tiles = 128, bin_hb_tiles = 60; // an A-block holds at most 64 x ha_tiles heads (8 B of LDS each), a B-block 64 x hb_tiles rows (20 B each: 77 KB, two workgroups per CU)
    long long bin_target = 0;       // edges a B-block is cut for (one workgroup of k_bin_reduce); 0: from the window, clamp(Ed / 256, 16 K, 384 K)
                                    // (measured: LiveJournal stand-in best at 16-32 K, twitter / friendster at 192 K)
    long long bin_target_a = 4ll << 20; // ... an A-block (its edges are dealt to workgroups of k_bin_scatter in chunks: large, so that tiles are long runs)
    long long bin_min_ids = 1ll << 19; // (smaller windows run resident or gather: R-MAT window of 2 M edges, ~0.65 M ids, single source: binned 2.66 ms per batch
                                       // against 3.26 gathering; window of 1 M edges, ~0.38 M ids: 2.45 against 1.47 -- tools/r04/midsize_probe.sh)
    int *bin_vblk_a = nullptr;      // vertex -> A-block (V ints; the B-block of a row is found by bisection, k_bin_keys)
    unsigned long long *bin_scan = nullptr; // counts and scans of the tables' tail (4 x (Ed / 64 + 3) words)
    int *bin_small = nullptr;       // quantile vertices | big rows | counter (bin_cut)
    long long bin_chunk = 32768;    // edges per workgroup of k_bin_scatter
    double *bin_vals = nullptr;     // the values in B-major order: what pass 1 hands to pass 2 (one loop runs at a time)
    // the tables PATCHED per slide (dppr_binned.hpp, round 5): both orders persistent as sorted words under frozen block cuts
    uint64_t *bin_wb = nullptr, *bin_wa = nullptr; // B-major / A-major words of the NEWEST epoch (rotate with keys_a / keys_b at a merge)
    int *bin_first = nullptr;       // B-major run index of every block pair's first run: n_b x n_a ints (scratch of the tables' tail)
    size_t bin_first_cap = 0;
    std::vector<int32_t> bin_cut_a, bin_cut_b; // the cuts in use (first vertex of every block; frozen between re-cuts, extended by new ids)
    int bin_abits = 0, bin_bbits = 0;   // width of the block-number fields of the words (with room for appended blocks)
    int bin_cut_ids = 0;            // ids the cuts cover
    bool bin_words_valid = false;   // bin_wb / bin_wa describe the newest epoch under bin_cut_a / bin_cut_b
    int bin_slides_since_cut = 0;
    bool bin_incremental = true;    // DPPR_BIN_INCREMENTAL=0: every epoch's tables by the two sorts (rounds 3-4)
    int bin_recut_every = 32;       // slides between two fresh cuts (DPPR_BIN_RECUT_EVERY)
    bool bin_frozen_rebuild = false; // (tests, DPPR_BIN_FROZEN_REBUILD=1: the sorts, but under the frozen cuts -- what the patched tables must equal bit for bit)
    bool bin_force_full = false;    // the next build sorts afresh (a merge missed a key)
    int bin_miss_host = 0;
    long long bin_patched = 0, bin_rebuilt = 0; // epochs whose tables were patched / built by the sorts
    void *bin_tmp = nullptr;
    size_t bin_tmp_bytes = 0;
    bool bin_ready = false;         // scratch allocated, kernels' LDS sizes registered
    std::vector<int32_t> h_tiles_in;
    // stream-update scratch
    uint32_t *su_k[2] = {nullptr, nullptr}, *su_v[2] = {nullptr, nullptr};
    double *su_term = nullptr;
    uint8_t *su_ins = nullptr;
    void *su_tmp = nullptr;
    size_t su_tmp_bytes = 0;
    // staged batch (set_batch before slide)
    std::vector<int32_t> st_b1, st_b2;
    std::vector<uint8_t> st_ins;
    bool batch_staged = false;
    // epochs
    std::vector<Epoch> epochs;
    int newest = -1; // global id of newest epoch
    std::vector<Slot> slots;
    std::vector<Group> groups;
    int *pinned = nullptr; // host-pinned readback words
    std::atomic<unsigned long long> heartbeat{0}; // bumped at every read-back of a frontier loop and every stage of a graph build (dppr_heartbeat)
    char *dump_pin = nullptr;     // host-pinned landing area of dppr_debug_dump's device reads, owned by the engine for its whole life
    static constexpr size_t DUMP_PIN_BYTES = sizeof(GridBar) + 4096;
    // vertex compaction: external id <-> internal id (assigned on first appearance), live zone [0, n_int) and parked
    // zone [V - n_parked, V): IdSpace. Renumbering (dppr_builder.hpp) is decided here:
    bool renumber_on = true;       // dppr_set_renumbering
    int renumber_growth_pct = 15;  // a slide considers it once n_int has grown by this much since the last numbering ...
    int renumber_min_parked = 1024; // ... and does it if at least this many ids (and growth_pct / 2 % of the live ones) would be parked
    int renumber_next = 0;         // n_int at which the next slide looks at the live count
    int renumberings = 0;
    std::vector<int32_t> mv_src, mv_dst, mv_zero; // row moves of revived vertices being applied (flush_moves)
    int *mv_idx = nullptr;         // device: src | dst | zero lists
    size_t mv_idx_cap = 0;
    double *mv_tmp = nullptr;      // device: the rows in flight
    size_t mv_tmp_cap = 0;
    int *d_ext2int = nullptr;  // device copy of ext2int, refreshed on demand
    double *d_xfer = nullptr;  // V doubles: staging of p / r in external order
    std::vector<int32_t> h_tmp1, h_tmp2;
    std::vector<int32_t> h_tiles; // host copy of the tile edge prefix / group table
    int max_iters = 1 << 20;
    std::string err;
};

namespace {

inline void set_err(dppr_engine *e, const char *msg) {
    std::lock_guard<std::mutex> lk(e->err_mu);
    e->err = msg;
}

#define HIP_TRY(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            char _b[512];                                                                               \
            snprintf(_b, sizeof(_b), "%s in %s at line %d", hipGetErrorString(_e), __FILE__, __LINE__); \
            set_err(e, _b);                                                                             \
            return _e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP;                           \
        }                                                                                               \
    } while (0)

inline int grid_for(int64_t n, int per_block = BLOCK, int cap = 2048) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// Wait for the engine's stream from inside a frontier loop (the read-back at the end of a chunk of iterations): polling the
// stream's completion instead of a blocking hipStreamSynchronize, whose wake-up is part of every chunk boundary's gap
// (DPPR_SYNC_SPIN=0: the blocking call, for A/B runs).
// The poll is bounded: a chunk of sweeps is over within a millisecond or two; after LOOP_SPIN_US the
// thread gives its core back and blocks -- N engines driven by N host threads (./pagerank -g N, two engines on one
// device) must not hold N cores at 100 % for a wait that has turned long.
constexpr long LOOP_SPIN_US = 2000; // (covers a resident launch of a configs[1]-size batch, ~0.4 ms: at 200 the blocking call's wake-up cost that path 6 %)
inline hipError_t loop_sync(hipStream_t st) {
    static const bool spin = !(getenv("DPPR_SYNC_SPIN") && atoi(getenv("DPPR_SYNC_SPIN")) == 0);
    if (!spin) return hipStreamSynchronize(st);
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    hipError_t r;
    unsigned polls = 0;
    while ((r = hipStreamQuery(st)) == hipErrorNotReady) {
        if ((++polls & 15u) == 0) {
            timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            if ((now.tv_sec - t0.tv_sec) * 1000000L + (now.tv_nsec - t0.tv_nsec) / 1000L > LOOP_SPIN_US) return hipStreamSynchronize(st);
        }
        __builtin_ia32_pause();
    }
    return r;
}

inline hipError_t loop_wait(dppr_engine *e) { // a read-back of a frontier loop: a sign of life for a watchdog (dppr_heartbeat)
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    return loop_sync(e->stream);
}

int fail(dppr_engine *e, int code, const char *msg) {
    if (e) set_err(e, msg);
    return code;
}

// dppr_hint_next_batch: the lookups of the NEXT batch's ids run on helper threads while dppr_update waits for the device. They
// only read the id maps; every path that changes the maps waits for them first (pre_join).
inline void pre_join(dppr_engine *e) {
    if (e->pre.task.valid()) e->pre.ok = e->pre.task.get();
}

inline int to_int(dppr_engine *e, int ext) { // (IdSpace: assigns, or revives a parked vertex)
    pre_join(e);
    return e->to_int(ext);
}

// a finished, still valid lookahead for exactly this array? (its index in e->pre, or -1)
int pre_find(dppr_engine *e, const int32_t *src, int n) {
    pre_join(e);
    if (!e->pre.armed || !e->pre.ok || e->pre.epoch != e->renumber_epoch || n <= 0) return -1;
    for (int k = 0; k < 4; ++k)
        if (e->pre.src[k] == src && e->pre.n[k] == n) {
            // same pointer, same length -- and still the same CONTENTS? (a caller that refilled the buffer without hinting again would get
            // the ids of the old contents: ADVICE r04.) First, last and strided samples: an id the lookup resolved maps back to src[i].
            const std::vector<int32_t> &o = e->pre.out[k];
            const int step = std::max(1, n / 64);
            for (int i = 0; i < n; i = (i + step < n || i == n - 1) ? i + step : n - 1) {
                const int m = o[(size_t)i];
                if (m >= 0 && (m >= (int)e->int2ext.size() || e->int2ext[(size_t)m] != src[i])) {
                    e->pre.armed = false; // stale: everything the hint holds is dropped
                    return -1;
                }
                if (i == n - 1) break;
            }
            return k;
        }
    return -1;
}

// ids inside [0, V)? (no side effect: a rejected call must not assign ids, revive parked vertices or queue row moves)
bool ids_in_range(dppr_engine *e, const int32_t *src, int n) {
    if (pre_find(e, src, n) >= 0) return true; // (the lookahead checked every id)
    for (int i = 0; i < n; ++i)
        if (src[i] < 0 || src[i] >= e->V) return false;
    return true;
}

// translate an id array; returns false (nothing changed) if any id is outside [0, V)
bool translate(dppr_engine *e, const int32_t *src, int n, std::vector<int32_t> &dst) {
    // looked up ahead of time (dppr_hint_next_batch) and still valid: no renumbering since (ids assigned or revived in between only
    // concern entries the lookup left at -1: those are resolved now, in array order, exactly as IdSpace::translate does)
    const int k = pre_find(e, src, n);
    if (k >= 0) {
        e->pre.src[k] = nullptr; // (consumed)
        dst.swap(e->pre.out[k]);
        e->resolve(src, dst.data(), e->pre.miss[k]);
        e->pre_hits++;
        e->pre_misses += (long long)e->pre.miss[k].size();
        return true;
    }
    // an array the lookahead does not cover: whatever it still holds is for calls that did not come -- drop it rather than let a
    // later call match a reused buffer by pointer and length alone
    if (n > 0) e->pre.armed = false;
    dst.resize((size_t)std::max(n, 1));
    return e->translate(src, (size_t)std::max(n, 0), dst.data());
}

} // namespace
