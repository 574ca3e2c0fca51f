// dppr_engine.hip -- host side of libdppr_hip.so: the C ABI of include/dppr.h.
//
// Owns device memory (replaces gpu/DeviceMemory.cuh, gpu/GPUEdgeBatch.cuh,
// gpu/SlidingGraphBuilder.cuh), drives the frontier loop (replaces
// PPRRevPushGPU::ExecuteOptimized, gpu/PPRRevPushGPU.cuh:97-131) and times the
// region the reference times (gpu/PPRGPU.cuh:138-164).
//
// There is NO CPU fallback: without a HIP device dppr_create fails with
// DPPR_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <ctime>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp> // device radix sort only (CSR rebuild, batch grouping); no CUB/Thrust in kernels

#include "../../include/dppr.h"
#include "dppr_cut.hpp"
#include "dppr_idspace.hpp"
#include "dppr_kernels.hpp"
#include "dppr_multi.hpp"
#include "dppr_gpush.hpp"
#include "dppr_binned.hpp"
#include "dppr_calib.hpp"

using namespace dppr;

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
    uint16_t *hl = nullptr, *dl = nullptr;
    int *apos = nullptr;
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
    int bin_ha_tiles = 128, bin_hb_tiles = 48; // an A-block holds at most 64 x ha_tiles heads (8 B of LDS each), a B-block 64 x hb_tiles rows (20 B each)
    long long bin_target = 0;       // edges a B-block is cut for (one workgroup of k_bin_reduce); 0: from the window, clamp(Ed / 256, 16 K, 192 K)
                                    // (measured: LiveJournal stand-in best at 16-32 K, twitter / friendster at 192 K)
    long long bin_target_a = 4ll << 20; // ... an A-block (its edges are dealt to workgroups of k_bin_scatter in chunks: large, so that tiles are long runs)
    long long bin_min_ids = 1ll << 19; // (smaller windows run resident or gather: R-MAT window of 2 M edges, ~0.65 M ids, single source: binned 2.66 ms per batch
                                       // against 3.26 gathering; window of 1 M edges, ~0.38 M ids: 2.45 against 1.47 -- tools/r04/midsize_probe.sh)
    int *bin_vblk_b = nullptr;      // vertex -> B-block (V ints; the A-block of a head is found by bisection, k_bin_keys)
    int *bin_small = nullptr;       // quantile vertices | big rows | counter (bin_cut)
    long long bin_chunk = 32768;    // edges per workgroup of k_bin_scatter
    double *bin_vals = nullptr;     // the values in B-major order: what pass 1 hands to pass 2 (one loop runs at a time)
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

int cut_sweep_groups(dppr_engine *e, Epoch &ep);
int build_bins(dppr_engine *e, Epoch &ep);
bool resident_arena(dppr_engine *e, const Epoch &ep);
int res_record_ranges(dppr_engine *e, Epoch &ep);

// A vertex that got its internal id AFTER an epoch was built (a source outside the window, a
// dppr_write to an unseen vertex) is not covered by that epoch's sweep groups: re-cut them.
int recut_stale_groups(dppr_engine *e) {
    for (auto &ep : e->epochs)
        if (ep.id >= 0 && (ep.grp_n_int != e->n_int || (e->any_groups && ep.n_ggroups == 0) ||
                           (e->wide_groups && ep.ggrp_max_tiles > 512 / WAVE))) {
            int rc = cut_sweep_groups(e, ep);
            if (rc) return rc;
            // (binned tables stay valid: k_bin_reduce takes the ids beyond bin_n_int, which have no edge in this epoch, on the side)
        }
    return DPPR_OK;
}

int sync_map(dppr_engine *e) {
    const unsigned gen = e->map_gen.load(std::memory_order_acquire); // (read BEFORE the copy: an id assigned during it leaves the copy stale)
    if (gen == e->map_gen_on_device) return DPPR_OK;
    HIP_TRY(hipMemcpyAsync(e->d_ext2int, e->ext2int.data(), sizeof(int) * (size_t)e->V, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->map_gen_on_device = gen;
    return DPPR_OK;
}

// Apply the row moves that revivals queued (to_int): every solver state's p / r rows, in one gather + scatter + zero
// per array. States that lag behind the newest epoch may be moved too: a parked row is not touched by any epoch,
// and the fresh id lies beyond the ids every older epoch sweeps.
int flush_moves(dppr_engine *e) {
    if (e->mv_origin.empty()) return DPPR_OK;
    e->take_moves(e->mv_src, e->mv_dst, e->mv_zero);
    const int n = (int)e->mv_src.size(), nz = (int)e->mv_zero.size();
    if (e->slots.empty() && e->groups.empty()) return DPPR_OK;
    const size_t need_idx = (size_t)2 * n + nz + 1;
    if (need_idx > e->mv_idx_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(e->mv_idx);
        e->mv_idx = nullptr;
        e->mv_idx_cap = 0;
        HIP_TRY(hipMalloc((void **)&e->mv_idx, sizeof(int) * (need_idx * 2 + 1024)));
        e->mv_idx_cap = need_idx * 2 + 1024;
    }
    const size_t need_tmp = (size_t)std::max(n, 1) * GS_MAX;
    if (need_tmp > e->mv_tmp_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(e->mv_tmp);
        e->mv_tmp = nullptr;
        e->mv_tmp_cap = 0;
        HIP_TRY(hipMalloc((void **)&e->mv_tmp, sizeof(double) * (need_tmp * 2 + 4096)));
        e->mv_tmp_cap = need_tmp * 2 + 4096;
    }
    int *d_src = e->mv_idx, *d_dst = e->mv_idx + n, *d_zero = e->mv_idx + 2 * n;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(d_src, e->mv_src.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
        HIP_TRY(hipMemcpyAsync(d_dst, e->mv_dst.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
    }
    if (nz > 0) HIP_TRY(hipMemcpyAsync(d_zero, e->mv_zero.data(), sizeof(int) * (size_t)nz, hipMemcpyHostToDevice, e->bs));
    auto move = [&](double *a, int w) -> int {
        if (n > 0) {
            hipLaunchKernelGGL(k_rows_gather<double>, dim3(grid_for((int64_t)n * w)), dim3(BLOCK), 0, e->bs, e->mv_tmp, a, d_src, n, w);
            hipLaunchKernelGGL(k_rows_scatter<double>, dim3(grid_for((int64_t)n * w)), dim3(BLOCK), 0, e->bs, a, e->mv_tmp, d_dst, n, w);
        }
        if (nz > 0)
            hipLaunchKernelGGL(k_rows_zero<double>, dim3(grid_for((int64_t)nz * w)), dim3(BLOCK), 0, e->bs, a, d_zero, nz, w);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    };
    for (auto &s : e->slots) {
        if (int rc = move(s.p, 1)) return rc;
        if (int rc = move(s.r, 1)) return rc;
    }
    for (auto &g : e->groups) {
        if (int rc = move(g.p, g.gw)) return rc;
        if (int rc = move(g.r, g.gw)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->bs)); // the host index vectors are reused
    return DPPR_OK;
}

// Parked rows were inert under the eps they were parked with; a solve with a smaller one pushes them first.
int settle_parked(dppr_engine *e, double *p, double *r, int w, double eps, double *park_eps, dppr_stats_t *st) {
    if (e->n_parked == 0 || !(eps < *park_eps)) return DPPR_OK;
    const size_t base = (size_t)(e->V - e->n_parked) * (size_t)w;
    const int64_t n = (int64_t)e->n_parked * w;
    int *cnt = e->hub_hist + 41; // scratch word
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL(k_settle_parked, dim3(grid_for(n)), dim3(BLOCK), 0, e->stream, p + base, r + base, n, eps, cnt);
    HIP_TRY(hipGetLastError());
    int pushed = 0;
    HIP_TRY(hipMemcpyAsync(&pushed, cnt, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (st) st->sum_F += pushed;
    *park_eps = eps;
    return DPPR_OK;
}

// n_int at which a slide looks at the live count again: grown by growth_pct % (64-bit: n * pct overflows an int at friendster scale)
inline int renumber_threshold(int n, int growth_pct) {
    const long long t = (long long)n + std::max<long long>((long long)n * growth_pct / 100, 1);
    return (int)std::min<long long>(t, 0x7fffffff);
}

// The live vertices of a renumbering in the order they are to be numbered: hashed, in blocks of falling in-degree on
// large windows -- what dppr::numbering_order does for dppr_load_window on the host, here as device keys and one
// radix sort (a host sort of a million pairs was most of a renumbering slide; of thirty million it would stall the
// stream). live[v] for the old ids v < n_old; order receives the n_live ids.
int device_numbering_order(dppr_engine *e, const std::vector<uint8_t> &live, int n_old, int n_live, std::vector<int32_t> &order) {
    order.clear();
    if (n_live <= 0) return DPPR_OK;
    const size_t n = (size_t)n_old;
    uint8_t *d_live = nullptr;
    int *d_i2e = nullptr, *d_vals = nullptr, *d_vals2 = nullptr, *d_deg2 = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_live); (void)hipFree(d_i2e); (void)hipFree(d_vals); (void)hipFree(d_vals2); (void)hipFree(d_deg2);
        (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_tmp);
    };
#define NO_TRY(call)                                                          \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) {                                               \
            cleanup();                                                        \
            e->err = std::string("renumbering order: ") + hipGetErrorString(_e); \
            return _e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP; \
        }                                                                     \
    } while (0)
    NO_TRY(hipMalloc((void **)&d_live, n));
    NO_TRY(hipMalloc((void **)&d_i2e, sizeof(int) * n));
    NO_TRY(hipMalloc((void **)&d_vals, sizeof(int) * n));
    NO_TRY(hipMalloc((void **)&d_vals2, sizeof(int) * n));
    NO_TRY(hipMalloc((void **)&d_keys, sizeof(uint64_t) * n));
    NO_TRY(hipMalloc((void **)&d_keys2, sizeof(uint64_t) * n));
    NO_TRY(hipMemcpyAsync(d_live, live.data(), n, hipMemcpyHostToDevice, e->bs));
    NO_TRY(hipMemcpyAsync(d_i2e, e->int2ext.data(), sizeof(int) * n, hipMemcpyHostToDevice, e->bs));
    HotThresholds ht{};
    int *d_deg = e->hub_slot_of; // (scratch of the CSR build, V ints)
    if ((size_t)n_live > HOT_WINDOW_MIN) {
        NO_TRY(hipMalloc((void **)&d_deg2, sizeof(int) * n));
        NO_TRY(hipMemsetAsync(d_deg, 0, sizeof(int) * n, e->bs));
        hipLaunchKernelGGL(k_in_degree, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, e->W, e->directed, d_deg);
        hipLaunchKernelGGL(k_live_degree, dim3(grid_for(n_old)), dim3(BLOCK), 0, e->bs, d_live, d_deg, n_old, d_vals);
        size_t tb = 0;
        NO_TRY(rocprim::radix_sort_keys_desc(nullptr, tb, d_vals, d_deg2, n, 0u, 32u, e->bs));
        NO_TRY(hipMalloc(&d_tmp, tb));
        NO_TRY(rocprim::radix_sort_keys_desc(d_tmp, tb, d_vals, d_deg2, n, 0u, 32u, e->bs));
        // in-degree of rank k among the live vertices (the non-live ones sorted last as -1)
        for (size_t k = HOT_SET; k >= (e->hot_blocks ? HOT_MIN : HOT_SET); k >>= 1) {
            if (k >= (size_t)n_live) continue;
            NO_TRY(hipMemcpyAsync(&ht.thr[ht.n], d_deg2 + k, sizeof(int), hipMemcpyDeviceToHost, e->bs));
            ht.n++;
        }
        NO_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(d_tmp);
        d_tmp = nullptr;
    }
    hipLaunchKernelGGL(k_number_keys, dim3(grid_for(n_old)), dim3(BLOCK), 0, e->bs, d_live, d_i2e, d_deg, ht, n_old, d_keys, d_vals);
    {
        size_t tb = 0;
        NO_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0u, 64u, e->bs));
        NO_TRY(hipMalloc(&d_tmp, tb));
        NO_TRY(rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0u, 64u, e->bs));
    }
    order.resize((size_t)n_live);
    NO_TRY(hipMemcpyAsync(order.data(), d_vals2, sizeof(int) * (size_t)n_live, hipMemcpyDeviceToHost, e->bs));
    NO_TRY(hipStreamSynchronize(e->bs));
    NO_TRY(hipGetLastError());
#undef NO_TRY
    cleanup();
    return DPPR_OK;
}

// Renumber the internal ids (dppr_builder.hpp has the why). Called by dppr_slide before anything of the new batch is
// looked at; does nothing unless every solver state is converged on the newest epoch (older epochs and their CSRs
// are in the old numbering: nothing may still need them) and enough ids would be parked. On success every epoch is
// invalidated, the ring, the out-degrees, the id maps, the staged batch and every state row are in the new
// numbering, and *did tells the caller to sort the whole window for the epoch it is about to build.
int compact_ids(dppr_engine *e, bool *did) {
    *did = false;
    pre_join(e);
    if (e->build_concurrent) return DPPR_OK; // (a renumbering moves every state row: only an exclusive slide may; dppr_renumbering_due tells the host)
    if (!e->renumber_on || e->W == 0 || e->n_int < e->renumber_next) return DPPR_OK;
    if (e->slots.empty() && e->groups.empty()) return DPPR_OK;
    for (const auto &s : e->slots)
        if (!s.converged || s.last_epoch != e->newest) return DPPR_OK;
    for (const auto &g : e->groups)
        if (!g.converged || g.last_epoch != e->newest) return DPPR_OK;
    if (int rc = flush_moves(e)) return rc;
    const int V = e->V, n_old = e->n_int;
    static const bool trace = getenv("DPPR_RENUMBER_TRACE") != nullptr; // (diagnostic: where a renumbering's time goes)
    timespec t_mark;
    clock_gettime(CLOCK_MONOTONIC, &t_mark);
    auto mark = [&](const char *what) {
        if (!trace) return;
        timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        fprintf(stderr, "[renumber] %-28s %8.2f ms\n", what, (now.tv_sec - t_mark.tv_sec) * 1e3 + (now.tv_nsec - t_mark.tv_nsec) * 1e-6);
        t_mark = now;
    };
    // which ids have an edge in the window
    uint8_t *d_live = nullptr;
    HIP_TRY(hipMalloc((void **)&d_live, (size_t)std::max(n_old, 1)));
    HIP_TRY(hipMemsetAsync(d_live, 0, (size_t)std::max(n_old, 1), e->bs));
    hipLaunchKernelGGL(k_mark_live, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, e->W, d_live);
    std::vector<uint8_t> live((size_t)std::max(n_old, 1));
    hipError_t herr = hipMemcpyAsync(live.data(), d_live, (size_t)n_old, hipMemcpyDeviceToHost, e->bs);
    if (herr == hipSuccess) herr = hipStreamSynchronize(e->bs);
    (void)hipFree(d_live);
    HIP_TRY(herr);
    if (e->batch_staged) { // the staged records are in internal ids already: their vertices stay where they are
        for (int v : e->st_b1) live[(size_t)v] = 1;
        for (int v : e->st_b2) live[(size_t)v] = 1;
    }
    for (const auto &s : e->slots) live[(size_t)s.source] = 1;
    for (const auto &g : e->groups)
        for (int k = 0; k < g.n; ++k) live[(size_t)g.src.s[k]] = 1;
    int n_live = 0;
    for (int v = 0; v < n_old; ++v) n_live += live[(size_t)v];
    mark("live flags");
    const int to_park = n_old - n_live;
    if (to_park < e->renumber_min_parked || (long long)to_park * 200 < (long long)n_live * e->renumber_growth_pct) {
        e->renumber_next = renumber_threshold(n_old, 12); // look again after some more growth
        return DPPR_OK;
    }
    // old position -> new position. Live vertices are numbered afresh the way dppr_load_window numbers a window
    // (hashed, hot blocks on large windows): arrivals are appended in arrival order between two renumberings, and a
    // tail of low-degree late-comers next to each other unbalances the sweep groups (configs[1] stand-in in step,
    // survivors kept in their old order instead: 0.52 ms per batch at the start, 0.61 after 400 batches of the
    // same work). The order is computed on the device (hash + in-degree blocks as keys, one radix sort).
    // Renumbering is an optimisation: whatever can fail for lack of memory is obtained BEFORE anything is changed, and
    // then the slide simply goes on in the old numbering (and looks again after some more growth).
    int *d_perm = nullptr;
    double *tmp = nullptr;
    int maxw = 1;
    for (const auto &g : e->groups) maxw = std::max(maxw, g.gw);
    auto cleanup = [&]() {
        (void)hipFree(d_perm);
        (void)hipFree(tmp);
    };
    if (hipMalloc((void **)&d_perm, sizeof(int) * (size_t)V) != hipSuccess ||
        hipMalloc((void **)&tmp, sizeof(double) * (size_t)V * (size_t)maxw) != hipSuccess) {
        (void)hipGetLastError();
        cleanup();
        e->renumber_next = renumber_threshold(n_old, 12);
        return DPPR_OK;
    }
    mark("scratch allocation");
    std::vector<int32_t> perm, order;
    if (int rc = device_numbering_order(e, live, n_old, n_live, order)) {
        cleanup();
        if (rc != DPPR_ERR_NOMEM) return rc;
        e->renumber_next = renumber_threshold(n_old, 12);
        return DPPR_OK;
    }
    mark("numbering order (device)");
    e->renumber(live, order, perm); // (IdSpace: perm, the maps, n_int, n_parked)
    mark("host maps");
    // From here on the host maps are in the NEW numbering: a failure below leaves ring, degrees and state rows part old,
    // part new -- the engine refuses all further work (`broken`).
#define RN_TRY(call)                                                     \
    do {                                                                 \
        hipError_t _e = (call);                                          \
        if (_e != hipSuccess) {                                          \
            cleanup();                                                   \
            e->broken = true;                                            \
            for (auto &ep : e->epochs) ep.id = -1;                       \
            e->err = std::string("renumbering failed half way (") + hipGetErrorString(_e) + "): the engine is unusable, destroy it"; \
            return DPPR_ERR_HIP;                                         \
        }                                                                \
    } while (0)
    RN_TRY(hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * (size_t)V, hipMemcpyHostToDevice, e->bs));
    hipLaunchKernelGGL(k_remap_ids, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w1, e->W, d_perm);
    hipLaunchKernelGGL(k_remap_ids, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w2, e->W, d_perm);
    { // out-degrees (ints) through the row scratch
        int *itmp = reinterpret_cast<int *>(tmp);
        RN_TRY(hipMemsetAsync(itmp, 0, sizeof(int) * (size_t)V, e->bs));
        hipLaunchKernelGGL(k_permute_rows<int>, dim3(grid_for(V)), dim3(BLOCK), 0, e->bs, itmp, e->outdeg, d_perm, V, 1);
        RN_TRY(hipMemcpyAsync(e->outdeg, itmp, sizeof(int) * (size_t)V, hipMemcpyDeviceToDevice, e->bs));
    }
    auto permute = [&](double *&a, int w) -> hipError_t { // a's rows in the new order; the old array becomes the scratch
        hipError_t r = hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)V * (size_t)w, e->bs);
        if (r != hipSuccess) return r;
        hipLaunchKernelGGL(k_permute_rows<double>, dim3(grid_for((int64_t)V * w)), dim3(BLOCK), 0, e->bs, tmp, a, d_perm, V, w);
        r = hipMemcpyAsync(a, tmp, sizeof(double) * (size_t)V * (size_t)w, hipMemcpyDeviceToDevice, e->bs);
        return r != hipSuccess ? r : hipGetLastError();
    };
    for (auto &s : e->slots) {
        RN_TRY(permute(s.p, 1));
        RN_TRY(permute(s.r, 1));
        // between two loops the snapshot vectors are all zero and the lists empty: nothing to carry over
        RN_TRY(hipMemsetAsync(s.x, 0, sizeof(double) * (size_t)V, e->bs));
        RN_TRY(hipMemsetAsync(s.x2, 0, sizeof(double) * (size_t)V, e->bs));
        RN_TRY(hipMemsetAsync(s.act[0], 0, s.act_bytes, e->bs));
        RN_TRY(hipMemsetAsync(s.act[1], 0, s.act_bytes, e->bs));
        s.source = perm[(size_t)s.source];
        s.seed_lists_valid = false;
        s.phase0_done = false;
        s.park_eps = std::max(s.park_eps, s.conv_eps);
    }
    for (auto &g : e->groups) {
        RN_TRY(permute(g.p, g.gw));
        RN_TRY(permute(g.r, g.gw));
        // (snapshot rows mean something only where an activity bit is set, and between loops none is)
        RN_TRY(hipMemsetAsync(g.act[0], 0, g.act_bytes, e->bs));
        RN_TRY(hipMemsetAsync(g.act[1], 0, g.act_bytes, e->bs));
        for (int k = 0; k < g.n; ++k) g.src.s[k] = perm[(size_t)g.src.s[k]];
        g.park_eps = std::max(g.park_eps, g.conv_eps);
    }
    RN_TRY(hipStreamSynchronize(e->bs));
#undef RN_TRY
    mark("ring, degrees, state rows");
    cleanup();
    mark("scratch release");
    if (e->batch_staged) {
        for (auto &v : e->st_b1) v = perm[(size_t)v];
        for (auto &v : e->st_b2) v = perm[(size_t)v];
    }
    for (auto &ep : e->epochs) ep.id = -1; // CSRs, group tables and batch records of the old numbering
    e->renumber_next = renumber_threshold(n_live, e->renumber_growth_pct);
    e->renumberings++;
    *did = true;
    return DPPR_OK;
}

// A state that has seen the batches up to epoch `last` can only take epoch last + 1 next: skipping or
// replaying one would leave the batch delta of a whole epoch out of (or twice in) p / r and still
// "converge" (n_epochs > 1 keeps many epochs resident, so nothing else would notice).
bool epoch_in_sequence(int last, int id) { return last < 0 || id == last + 1; }

Epoch *find_epoch(dppr_engine *e, int epoch) {
    if (e->newest < 0) return nullptr;
    if (epoch < 0) epoch = e->newest;
    Epoch &ep = e->epochs[epoch % e->n_epochs];
    return ep.id == epoch ? &ep : nullptr;
}

// Sort the whole window into the persistent key arrays (load_window; also the non-incremental
// slide = what gpu/SlidingGraphBuilder.cuh:203-221 does every batch).
int sort_window_full(dppr_engine *e) {
    const int W = e->W, Ed = e->Ed;
    if (W == 0) return DPPR_OK;
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, W, e->directed, e->bits,
                       e->keys_a);
    HIP_TRY(hipGetLastError());
    size_t tmp = e->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, e->keys_a, e->in_sorted, (size_t)Ed, 0u, (unsigned)(2 * e->bits),
                                     e->bs));
    if (e->directed) { // undirected: the out-orientation is the same multiset, out_sorted aliases in_sorted
        hipLaunchKernelGGL(k_make_out_keys, dim3(grid_for(W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, W, e->directed,
                           e->bits, e->keys_a);
        tmp = e->sort_tmp_bytes;
        HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, e->keys_a, e->out_sorted, (size_t)Ed, 0u,
                                         (unsigned)(2 * e->bits), e->bs));
    }
    return DPPR_OK;
}

// One orientation of the incremental update: sorted' = (sorted minus deleted instances) merged with inserted.
int merge_batch_keys(dppr_engine *e, uint64_t *&sorted, uint64_t *del_unsorted, uint64_t *del_sorted, int nd,
                     uint64_t *ins_unsorted, uint64_t *ins_sorted, int ni) {
    const int Ed = e->Ed;
    size_t tmp = e->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, del_unsorted, del_sorted, (size_t)nd, 0u, (unsigned)(2 * e->bits),
                                     e->bs));
    tmp = e->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, ins_unsorted, ins_sorted, (size_t)ni, 0u, (unsigned)(2 * e->bits),
                                     e->bs));
    // retired positions, then one pass: every kept and every inserted key straight to its place (dppr_builder.hpp k_merge_tiles)
    hipLaunchKernelGGL(k_del_positions, dim3(grid_for(nd)), dim3(BLOCK), 0, e->bs, sorted, Ed, del_sorted, nd, e->delpos,
                       e->hub_hist + MERGE_MISS_WORD);
    const int n_tiles = (Ed + CMP_TILE - 1) / CMP_TILE;
    hipLaunchKernelGGL(k_merge_tiles, dim3(n_tiles), dim3(BLOCK), 0, e->bs, sorted, Ed, e->delpos, nd, ins_sorted, ni, e->keys_b,
                       (size_t)Ed - (size_t)nd + (size_t)ni);
    HIP_TRY(hipGetLastError());
    std::swap(sorted, e->keys_b); // the merged array is the new persistent one; the old becomes scratch
    return DPPR_OK;
}

// Workgroup size of the sweeps: 1024 unless pinned (dppr_set_tuning; 512 was measured on the LiveJournal
// and twitter stand-ins and is not better once two 1024-thread workgroups fit a CU).
int sweep_block(const dppr_engine *e) { return e->pull_block ? e->pull_block : 1024; }

// workgroups of the resident sweep that the device holds at once (0: resident sweeps are off)
int persist_capacity(const dppr_engine *e) {
    const int pb = sweep_block(e);
    if (!e->persist_mode || !e->persist_ok || (pb != 256 && pb != 512 && pb != 1024)) return 0;
    return e->persist_cap;
}

// How many workgroups of the resident sweep the device holds at once, from the runtime's occupancy
// figure for the instantiation the engine will launch.
int query_persist_cap(dppr_engine *e) {
    e->persist_cap = 0;
    int per_cu = 0;
    const int pb = sweep_block(e);
    if (pb != 256 && pb != 512 && pb != 1024) return DPPR_OK; // other block sizes (tuning only): per-iteration launches
    if (pb == 256) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pull_resident<256>, 256, 0));
    else if (pb == 512) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pull_resident<512>, 512, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pull_resident<1024>, 1024, 0));
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
    e->persist_cap = std::min(per_cu * cus, STAT_SLOTS);
    return DPPR_OK;
}

// Cut the vertex range into sweep groups of at most (workgroup size / 64) consecutive tiles with
// about equal weight (edges + a per-vertex term), so that no workgroup of k_pull_iter is the
// straggler because a hub's long row happens to sit in its range. Host greedy over the tile
// prefix; part of the (untimed) graph build.
int cut_sweep_groups(dppr_engine *e, Epoch &ep) {
    const int NV = e->n_int;
    const int n_tiles = (NV + WAVE - 1) / WAVE;
    const int max_tiles = sweep_block(e) / WAVE;
    e->h_tiles.resize((size_t)n_tiles + 2);
    if (n_tiles > 0) {
        int *scratch = reinterpret_cast<int *>(e->keys_a); // Ed * 8 bytes >= (n_tiles + 1) * 4 unless the graph is tiny
        const bool fits = (size_t)e->Ed * sizeof(uint64_t) >= ((size_t)n_tiles + 1) * sizeof(int);
        if (!fits) scratch = e->hub_slot_of;               // V ints: always large enough
        hipLaunchKernelGGL(k_tile_prefix, dim3(grid_for(n_tiles + 1)), dim3(BLOCK), 0, e->bs, ep.out_row_ptr, NV,
                           n_tiles, scratch);
        HIP_TRY(hipMemcpyAsync(e->h_tiles.data(), scratch, sizeof(int) * ((size_t)n_tiles + 1), hipMemcpyDeviceToHost,
                               e->bs));
        HIP_TRY(hipStreamSynchronize(e->bs));
    }
    std::vector<int32_t> cut;
    const int32_t *prefix = e->h_tiles.data();
    // A window small enough for one workgroup per group to be resident at once gets at most that
    // many groups (then runs of dense iterations are single launches, dppr_resident.hpp). A resident
    // workgroup's time is its edge count (every iteration all workgroups wait for the slowest one's
    // values), so this cut MINIMISES THE LARGEST group (dppr_cut.hpp); per tile the per-vertex work of a
    // resident workgroup is small and fixed (weight 8). Otherwise: many groups of about equal weight.
    const int cap = persist_capacity(e);
    bool fitted = false;
    if (cap > 0 && (long long)n_tiles <= (long long)cap * max_tiles * 7 / 8) fitted = cut_minmax(prefix, n_tiles, max_tiles, cap, 8, cut);
    if (!fitted)
        cut_greedy(prefix, n_tiles, max_tiles,
                   std::max<long long>(252, (n_tiles + max_tiles * 3 / 4 - 1) / std::max(1, max_tiles * 3 / 4)), 2 * WAVE, cut);
    ep.n_groups = (int)cut.size() - 1;
    ep.grp_n_int = NV;
    HIP_TRY(hipMemcpyAsync(ep.grp_tile, cut.data(), sizeof(int) * cut.size(), hipMemcpyHostToDevice, e->bs));
    // slot tables for resident launches (a window that got the resident cut; every group must fit the table build's sort)
    ep.res_valid = false;
    // (a single-source slot exists: its launches will want the arena -- grown here unless a solver call may be using it right now:
    // dppr_update grows it itself before its first resident launch)
    if (fitted && !e->slots.empty() && !e->build_concurrent) (void)resident_arena(e, ep);
    if (fitted && e->res_slots && ep.Ed > 0 && NV <= RES_ID_LIMIT) {
        long long largest = 0;
        for (size_t g = 0; g + 1 < cut.size(); ++g) largest = std::max<long long>(largest, (long long)prefix[cut[g + 1]] - prefix[cut[g]]);
        if (largest <= RES_SORT_MAX) {
            if ((size_t)ep.Ed > ep.res_pk_cap) {
                HIP_TRY(hipStreamSynchronize(e->bs));
                (void)hipFree(ep.res_pk);
                ep.res_pk = nullptr;
                ep.res_pk_cap = 0;
                HIP_TRY(hipMalloc((void **)&ep.res_pk, sizeof(uint32_t) * ((size_t)ep.Ed + (size_t)ep.Ed / 8 + 1024)));
                ep.res_pk_cap = (size_t)ep.Ed + (size_t)ep.Ed / 8 + 1024;
            }
            hipLaunchKernelGGL(k_res_slots, dim3(ep.n_groups), dim3(1024), 0, e->bs, NV, ep.grp_tile, ep.out_row_ptr, ep.out_col, ep.res_pk);
            HIP_TRY(hipGetLastError());
            ep.res_valid = true;
        }
    }
    HIP_TRY(hipStreamSynchronize(e->bs)); // `cut` is a local
    ep.su_inline = false;
    if (fitted)
        if (int rrc = res_record_ranges(e, ep)) return rrc;
    ep.n_ggroups = 0;
    if (e->any_groups) { // groups of at most 16 (8) tiles for k_gsweep<1, 1024> (<2, 512>)
        const int gmax = (e->wide_groups ? 512 : 1024) / WAVE;
        const long long want = std::max<long long>(e->ggroups_min, (n_tiles + gmax * 3 / 4 - 1) / std::max(1, gmax * 3 / 4));
        ep.ggrp_max_tiles = gmax;
        cut_greedy(prefix, n_tiles, gmax, want, 2 * WAVE, cut);
        ep.n_ggroups = (int)cut.size() - 1;
        HIP_TRY(hipMemcpyAsync(ep.ggrp_tile, cut.data(), sizeof(int) * cut.size(), hipMemcpyHostToDevice, e->bs));
        // the groups' row tables, once per epoch (every sweep of every source group of this epoch loads them)
        const int nvx = gmax * WAVE;
        const size_t need = (size_t)ep.n_ggroups * (size_t)GT_STRIDE(nvx);
        if (need > ep.gtab_cap) {
            HIP_TRY(hipStreamSynchronize(e->bs));
            (void)hipFree(ep.gtab);
            ep.gtab = nullptr;
            ep.gtab_cap = 0;
            HIP_TRY(hipMalloc((void **)&ep.gtab, sizeof(int) * (need + need / 8 + 1024)));
            ep.gtab_cap = need + need / 8 + 1024;
        }
        if (ep.n_ggroups <= 0) {
            // (no vertex has an id yet: nothing to sweep)
        } else if (nvx == 512)
            hipLaunchKernelGGL(k_gtables<512>, dim3(std::min(ep.n_ggroups, 1024)), dim3(GNT), 0, e->bs, NV, ep.ggrp_tile,
                               ep.n_ggroups, ep.out_row_ptr, ep.gtab);
        else
            hipLaunchKernelGGL(k_gtables<1024>, dim3(std::min(ep.n_ggroups, 1024)), dim3(GNT), 0, e->bs, NV, ep.ggrp_tile,
                               ep.n_ggroups, ep.out_row_ptr, ep.gtab);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(e->bs));
    }
    return DPPR_OK;
}


// ---- binned sweep tables of an epoch (dppr_binned.hpp). Part of the (untimed) graph build; needs the sorted
// out-orientation keys, i.e. `ep` must be the epoch the persistent key arrays describe (the newest one).
bool bin_wanted(const dppr_engine *e) {
    if (e->bin_mode == 2) return true;
    return e->bin_mode == 1 && !e->slots.empty() && (long long)e->n_int >= e->bin_min_ids;
}

// The largest block shapes dppr_set_binned_sweep admits -- ONE pair of constants for the validation, its message and the
// kernels' dynamic-LDS attribute (the attribute said 272 tiles, the validation 288).
constexpr int BIN_MAX_HA_TILES = 272, BIN_MAX_HB_TILES = 120;
static_assert(BIN_MAX_HA_TILES * WAVE * (int)sizeof(double) + 4096 <= 160 * 1024, "k_bin_scatter: the largest A-block's slice of x + static LDS fits a gfx950 CU");
static_assert(BIN_MAX_HB_TILES * WAVE * 20 + 4096 <= 160 * 1024, "k_bin_reduce: the largest B-block's rows + static LDS fit a gfx950 CU");
static_assert(BIN_MAX_HB_TILES * WAVE <= (1 << BIN_RL) && BIN_MAX_HA_TILES * WAVE <= (1 << BIN_HL), "a row / head index inside its block fits its field of the sort words");
static_assert(BIN_RL + BIN_HL + 32 <= 64 && BIN_W2_A + 16 <= 64, "sort words: two block numbers of <= 32 bits together, an A-block number of <= 16 bits (BIN_MAX_BLOCKS)");

// An allocation of the (optional) binned-sweep tables that fails is not an error of the call that wanted them: the
// partial allocations are released, the sticky HIP error is cleared and the epoch sweeps with k_pull_iter.
static bool bin_alloc(void **p, size_t bytes) {
    if (*p) return true;
    if (getenv("DPPR_TEST_BIN_OOM")) { // (test hook: these allocations fail as if the device were out of memory)
        *p = nullptr;
        return false;
    }
    if (hipMalloc(p, bytes) == hipSuccess) return true;
    *p = nullptr;
    (void)hipGetLastError();
    return false;
}

int bin_prepare(dppr_engine *e, bool *have) { // engine-level scratch, once (idempotent per pointer: a failed attempt may be repeated)
    *have = false;
    if (e->bin_ready) {
        *have = true;
        return DPPR_OK;
    }
    const size_t Edn = (size_t)std::max(e->Ed, 1);
    bool ok = true;
    ok = ok && bin_alloc((void **)&e->bin_vblk_b, sizeof(int) * (size_t)e->V);
    ok = ok && bin_alloc((void **)&e->bin_small, sizeof(int) * BIN_SMALL_INTS);
    ok = ok && bin_alloc((void **)&e->bin_vals, sizeof(double) * (Edn + 64));
    if (ok && !e->bin_tmp) {
        HIP_TRY(rocprim::radix_sort_keys(nullptr, e->bin_tmp_bytes, e->keys_a, e->keys_b, Edn, 0u, 64u, e->bs));
        ok = bin_alloc(&e->bin_tmp, std::max<size_t>(e->bin_tmp_bytes, 16));
    }
    if (!ok) { // out of memory: nothing half-built stays behind, the sweeps of this engine gather (k_pull_iter)
        (void)hipFree(e->bin_vblk_b); (void)hipFree(e->bin_small); (void)hipFree(e->bin_vals); (void)hipFree(e->bin_tmp);
        e->bin_vblk_b = e->bin_small = nullptr;
        e->bin_vals = nullptr;
        e->bin_tmp = nullptr;
        return DPPR_OK;
    }
    // (the attribute belongs to the kernel, not to this engine: the largest shapes dppr_set_binned_sweep admits, so that engines
    // with different block shapes can share a process)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BIN_MAX_HA_TILES * WAVE * (int)sizeof(double)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_reduce), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BIN_MAX_HB_TILES * WAVE * 20));
    e->bin_ready = true;
    *have = true;
    return DPPR_OK;
}

// One cut (dppr_binned.hpp: every multiple of `cap` vertices, the first vertex behind every `target` edges, both sides of
// every row of >= target / 4 edges); device searches, the merge of the few thousand boundaries on the host.
int bin_cut(dppr_engine *e, const int *row_ptr, int NV, int cap, long long target, std::vector<int32_t> &cut) {
    const int Ed = e->Ed;
    target = std::max<long long>(target, 64);
    const int K = (int)std::min<long long>((Ed + target - 1) / target, BIN_MAX_BLOCKS);
    int *d_q = e->bin_small, *d_big = e->bin_small + BIN_MAX_BLOCKS, *d_cnt = d_big + BIN_MAX_BIG;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int), e->bs));
    if (K > 1) hipLaunchKernelGGL(k_bin_quantiles, dim3(grid_for(K)), dim3(BLOCK), 0, e->bs, row_ptr, NV, target, K, d_q);
    hipLaunchKernelGGL(k_bin_big_rows, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, row_ptr, NV, (int)std::max<long long>(target / 4, 1),
                       BIN_MAX_BIG, d_big, d_cnt);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h((size_t)BIN_MAX_BLOCKS + BIN_MAX_BIG + 1);
    HIP_TRY(hipMemcpyAsync(h.data(), e->bin_small, sizeof(int) * h.size(), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipStreamSynchronize(e->bs));
    cut.clear();
    for (long long v = 0; v < NV; v += cap) cut.push_back((int32_t)v);
    for (int k = 0; k + 1 < K; ++k) cut.push_back(h[(size_t)k]);
    const int nbig = std::min(h[(size_t)BIN_MAX_BLOCKS + BIN_MAX_BIG], BIN_MAX_BIG);
    for (int k = 0; k < nbig; ++k) {
        cut.push_back(h[(size_t)BIN_MAX_BLOCKS + k]);
        cut.push_back(h[(size_t)BIN_MAX_BLOCKS + k] + 1);
    }
    cut.push_back(NV);
    std::sort(cut.begin(), cut.end());
    cut.erase(std::unique(cut.begin(), cut.end()), cut.end());
    while (!cut.empty() && cut.back() > NV) cut.pop_back();
    if (cut.empty() || cut.front() != 0) cut.insert(cut.begin(), 0);
    if (cut.back() != NV) cut.push_back(NV);
    return DPPR_OK;
}

int build_bins(dppr_engine *e, Epoch &ep) {
    ep.bin_valid = false;
    if (!bin_wanted(e) || e->Ed <= 0 || ep.grp_n_int <= 0) return DPPR_OK;
    bool have = false;
    if (int rc = bin_prepare(e, &have)) return rc;
    if (!have) return DPPR_OK;
    const int Ed = e->Ed, NV = ep.grp_n_int;
    if (!ep.hl || !ep.dl || !ep.apos) { // all three or none (a partial set from a failed attempt is released first)
        const size_t Edn = (size_t)Ed;
        const bool ok = bin_alloc((void **)&ep.hl, sizeof(uint16_t) * Edn) && bin_alloc((void **)&ep.dl, sizeof(uint16_t) * Edn) &&
                        bin_alloc((void **)&ep.apos, sizeof(int) * Edn);
        if (!ok) {
            (void)hipFree(ep.hl); (void)hipFree(ep.dl); (void)hipFree(ep.apos);
            ep.hl = ep.dl = nullptr;
            ep.apos = nullptr;
            return DPPR_OK; // (bin_valid stays false: this epoch's sweeps gather)
        }
    }
    std::vector<int32_t> cut_a, cut_b;
    if (int rc = bin_cut(e, ep.row_ptr, NV, e->bin_ha_tiles * WAVE, e->bin_target_a, cut_a)) return rc;
    if (int rc = bin_cut(e, ep.out_row_ptr, NV, e->bin_hb_tiles * WAVE,
                         e->bin_target > 0 ? e->bin_target : std::min<long long>(std::max<long long>(Ed / 256, 16384), 196608), cut_b)) return rc;
    ep.n_a = (int)cut_a.size() - 1;
    ep.n_b = (int)cut_b.size() - 1;
    int abits = 1, bbits = 1;
    while ((1 << abits) < ep.n_a) abits++;
    while ((1 << bbits) < ep.n_b) bbits++;
    if (abits + bbits > 32 || ep.n_a + 2 > BIN_MAX_BLOCKS || ep.n_b + 2 > BIN_MAX_BLOCKS) return DPPR_OK; // (a window of that many blocks: the sweep stays k_pull_iter)
    // per epoch: acut | astart | bcut (block tables), then the chunk table
    const size_t tab_ints = (size_t)2 * (ep.n_a + 1) + (ep.n_b + 1);
    if (tab_ints > ep.bin_tab_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(ep.acut);
        ep.acut = nullptr;
        ep.bin_tab_cap = 0;
        HIP_TRY(hipMalloc((void **)&ep.acut, sizeof(int) * (tab_ints + tab_ints / 4 + 1024)));
        ep.bin_tab_cap = tab_ints + tab_ints / 4 + 1024;
    }
    int *d_astart = ep.acut + (ep.n_a + 1);
    ep.bcut = d_astart + (ep.n_a + 1);
    HIP_TRY(hipMemcpyAsync(ep.acut, cut_a.data(), sizeof(int) * cut_a.size(), hipMemcpyHostToDevice, e->bs));
    HIP_TRY(hipMemcpyAsync(ep.bcut, cut_b.data(), sizeof(int) * cut_b.size(), hipMemcpyHostToDevice, e->bs));
    hipLaunchKernelGGL(k_bin_vertex_block, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, ep.acut, ep.n_a, NV, ep.row_ptr, (int *)nullptr, d_astart);
    int *d_bstart = e->bin_small; // (not kept: a B-block's edges are out_row_ptr[bcut[b]] .. out_row_ptr[bcut[b + 1]])
    hipLaunchKernelGGL(k_bin_vertex_block, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, ep.bcut, ep.n_b, NV, ep.out_row_ptr, e->bin_vblk_b, d_bstart);
    const uint64_t *out_keys = e->directed ? e->out_sorted : e->in_sorted;
    const bool cuts_in_lds = (size_t)(ep.n_a + 1) * sizeof(int) <= 48 * 1024;
    hipLaunchKernelGGL(k_bin_keys, dim3(grid_for(Ed)), dim3(BLOCK), cuts_in_lds ? (size_t)(ep.n_a + 1) * sizeof(int) : 0, e->bs, out_keys, Ed,
                       e->bits, ep.acut, ep.n_a, cuts_in_lds ? 1 : 0, e->bin_vblk_b, ep.bcut, abits, e->keys_b);
    HIP_TRY(hipGetLastError());
    // chunks of the A-major runs (a block of many edges is dealt to several workgroups of k_bin_scatter)
    std::vector<int32_t> astart((size_t)ep.n_a + 1);
    HIP_TRY(hipMemcpyAsync(astart.data(), d_astart, sizeof(int) * astart.size(), hipMemcpyDeviceToHost, e->bs));
    size_t tmp = e->bin_tmp_bytes; // B-major: stable by (B-block, A-block); the words are in (row, head) order
    const char *placement = getenv("DPPR_BIN_PLACEMENT"); // (tests / A-B runs: "counting" wherever it can run -- small windows never qualify by themselves --, "radix" never)
    const bool cs_force = placement && !strcmp(placement, "counting"), cs_never = placement && !strcmp(placement, "radix");
    // every B-block's segment grouped by A-block in one pass (k_bin_bmajor) where the radix sort would need FOUR passes over its
    // 8-bit digits (friendster stand-in, 26 bits: 6.0 ms against 8.9; with three -- twitter, 23 bits -- the sort wins, 3.0 against 3.8:
    // the single pass scatters 8-byte words over thousands of runs, a radix pass over 256)
    if (ep.n_a <= BIN_CS_MAX_A && (abits + bbits > 24 || cs_force) && !cs_never) {
        int n_pad = WAVE;
        while (n_pad < ep.n_a) n_pad *= 2;
        hipLaunchKernelGGL(k_bin_bmajor, dim3(ep.n_b), dim3(BIN_CS_NT), sizeof(int) * (size_t)n_pad, e->bs, e->keys_b, d_bstart, ep.bcut, ep.n_a,
                           n_pad, abits, e->keys_a);
    } else {
        HIP_TRY(rocprim::radix_sort_keys(e->bin_tmp, tmp, e->keys_b, e->keys_a, (size_t)Ed, (unsigned)(BIN_RL + BIN_HL),
                                         (unsigned)(BIN_RL + BIN_HL + abits + bbits), e->bs));
    }
    hipLaunchKernelGGL(k_bin_fill_b, dim3(grid_for(Ed)), dim3(BLOCK), 0, e->bs, e->keys_a, Ed, abits, ep.dl, e->keys_b);
    HIP_TRY(hipGetLastError());
    tmp = e->bin_tmp_bytes;        // A-major: the B-major sequence, stable by A-block
    HIP_TRY(rocprim::radix_sort_keys(e->bin_tmp, tmp, e->keys_b, e->keys_a, (size_t)Ed, (unsigned)BIN_W2_A, (unsigned)(BIN_W2_A + abits), e->bs));
    hipLaunchKernelGGL(k_bin_fill_a, dim3(grid_for(Ed)), dim3(BLOCK), 0, e->bs, e->keys_a, Ed, ep.hl, ep.apos);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->bs)); // astart has arrived; the cuts are locals
    std::vector<BinChunk> chunks;
    const int csize = (int)std::max<long long>(e->bin_chunk, 64);
    for (int a = 0; a < ep.n_a; ++a) {
        const int j0 = astart[(size_t)a], j1 = astart[(size_t)a + 1];
        const int pieces = (j1 - j0 + csize - 1) / csize; // (a block without an edge: no workgroup)
        for (int k = 0; k < pieces; ++k) {
            const long long lo = j0 + (long long)(j1 - j0) * k / pieces, hi = j0 + (long long)(j1 - j0) * (k + 1) / pieces;
            chunks.push_back(BinChunk{a, (int)lo, (int)hi});
        }
    }
    ep.n_chunks = (int)chunks.size();
    if (chunks.size() > ep.chunk_cap) {
        (void)hipFree(ep.chunks);
        ep.chunks = nullptr;
        ep.chunk_cap = 0;
        HIP_TRY(hipMalloc((void **)&ep.chunks, sizeof(BinChunk) * (chunks.size() + chunks.size() / 4 + 256)));
        ep.chunk_cap = chunks.size() + chunks.size() / 4 + 256;
    }
    if (!chunks.empty()) HIP_TRY(hipMemcpy(ep.chunks, chunks.data(), sizeof(BinChunk) * chunks.size(), hipMemcpyHostToDevice));
    ep.bin_n_int = NV;
    ep.bin_valid = true;
    return DPPR_OK;
}

// Hub directory + in-CSR + out-CSR of `ep` from the persistent sorted keys and outdeg.
int build_epoch(dppr_engine *e, Epoch &ep) {
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    const int Ed = e->Ed;
    const int NV = e->n_int; // only vertices that ever had an edge (or are a source) exist internally
    // hub directory: the (at most HUB_CAP) vertices of largest out-degree, at least hub_min_degree
    {
        HIP_TRY(hipMemsetAsync(e->hub_hist, 0, sizeof(int) * 33, e->bs));
        hipLaunchKernelGGL(k_deg_hist, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, e->outdeg, NV, e->hub_min_degree,
                           e->hub_hist);
        int hist[32];
        HIP_TRY(hipMemcpyAsync(hist, e->hub_hist, sizeof(hist), hipMemcpyDeviceToHost, e->bs));
        HIP_TRY(hipStreamSynchronize(e->bs));
        long long above = 0;
        int k = 31;
        for (; k >= 0; --k) {
            if (above + hist[k] > HUB_CAP) break;
            above += hist[k];
        }
        // every bucket > k fits; threshold = lower edge of bucket k+1
        const long long thresh = (long long)e->hub_min_degree << (k + 1);
        const int th = (int)std::min<long long>(thresh, 0x7fffffff);
        hipLaunchKernelGGL(k_assign_hubs, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, e->outdeg, NV, th,
                           e->hub_slot_of, ep.hub_v, ep.hub_degp1, e->hub_hist + 32);
        HIP_TRY(hipGetLastError());
        ep.n_hubs = (int)above;
    }
    // row pointers are filled for the whole id capacity: ids assigned later read as empty rows
    hipLaunchKernelGGL(k_build_csr, dim3(grid_for(std::max(Ed, e->V + 1))), dim3(BLOCK), 0, e->bs, e->in_sorted, Ed,
                       e->V, e->bits, e->hub_slot_of, ep.row_ptr, ep.adj);
    hipLaunchKernelGGL(k_build_out_csr, dim3(grid_for(std::max(Ed, e->V + 1))), dim3(BLOCK), 0, e->bs,
                       e->directed ? e->out_sorted : e->in_sorted, Ed, e->V, e->bits, ep.out_row_ptr, ep.out_col);
    HIP_TRY(hipGetLastError());
    ep.Ed = Ed;
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    if (int rc = cut_sweep_groups(e, ep)) return rc;
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    const int brc = build_bins(e, ep);
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    return brc;
}

int read_count(dppr_engine *e, const int *dptr, int *out) {
    HIP_TRY(hipMemcpyAsync(e->pinned, dptr, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(loop_wait(e));
    *out = e->pinned[0];
    return DPPR_OK;
}

// Frontier loop: PPRRevPushGPU::ExecuteOptimized's while(1) (gpu/PPRRevPushGPU.cuh:106-130).
// On entry s.ft[buf] holds the frontier and s.cnt[cur] its size; cnt[(cur+1)%3] is zero and
// the dense vectors s.x / s.x2 are all zero (no snapshot taken yet) -- unless `entry` says
// otherwise.
//
// The reference reads the frontier count back after EVERY iteration (blocking 4-byte D2H,
// :107). Here iterations are enqueued in CHUNKS: every kernel takes F from device memory,
// rotates the three counters itself and exits at once when F == 0, so the host only reads
// the count (and the per-iteration log of F) once per chunk. The host also picks, per chunk,
// how the iterations are evaluated: SPARSE (push kernels, atomics) or DENSE (pull sweep, no
// atomics; as ONE resident launch for the whole chunk when the epoch's sweep groups fit the chip,
// dppr_resident.hpp) -- the same sums either way.
//
// `entry` describes a loop that is picked up in the middle (after a launch of batch_ahead that
// ended before the loop did): iterations already done, the frontier size if the host knows it,
// and whether s.x already holds the frontier's dense snapshot.
struct LoopEntry {
    int it = 0;
    int F = -1; // -1: read cnt[cur]
    bool dense = false;
    bool any_pull = false;
};

int pull_min_frontier(const dppr_engine *e) {
    return e->pull_min_frontier > 0 ? e->pull_min_frontier : e->pull_min_frontier < 0 ? 0x7fffffff : std::max(1024, e->Ed / 192);
}

// The batch's records, grouped by tail at slide time, cut into the sweep groups' ranges (dppr_resident.hpp, PLAN_UPDATE). Needs
// both the grouping and a resident-size group cut: called by whichever of the two is made last. Untimed (graph build / slide).
int res_record_ranges(dppr_engine *e, Epoch &ep) {
    ep.su_inline = false;
    const int pb = sweep_block(e);
    if (!e->res_update || !ep.grouped || ep.L <= 0 || ep.L >= SU_SPLIT_MIN || ep.n_groups <= 0 || ep.n_groups > persist_capacity(e)) return DPPR_OK;
    const size_t need = (size_t)ep.n_groups + 3;
    if (need > ep.su_rng_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(ep.su_rng);
        ep.su_rng = nullptr;
        ep.su_rng_cap = 0;
        HIP_TRY(hipMalloc((void **)&ep.su_rng, sizeof(int) * (need + 1024)));
        ep.su_rng_cap = need + 1024;
    }
    int *stat = ep.su_rng + ep.n_groups + 1;
    HIP_TRY(hipMemsetAsync(stat, 0, sizeof(int) * 2, e->bs));
    hipLaunchKernelGGL(k_res_rec_ranges, dim3((ep.n_groups + 256) / 256), dim3(256), 0, e->bs, ep.sk, ep.L, ep.grp_tile, ep.n_groups,
                       ep.su_rng, stat);
    HIP_TRY(hipGetLastError());
    int h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, stat, sizeof(h), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipStreamSynchronize(e->bs));
    ep.su_inline = h[0] <= pb && h[1] == ep.L; // (a tail beyond the last group: an id assigned after the cut -- the cut is redone then)
    return DPPR_OK;
}

// The arena of a resident launch (dppr_resident.hpp, FRESH VECTORS): RES_VECTORS vectors of `stride` doubles, scratch between
// launches, one per engine (the engine's launches are serial on its stream).
// Grown when a larger window is cut (graph build) or, failing that, before the first launch that needs it; without it (out of
// memory) the window's sweeps simply run as per-iteration launches.
bool resident_arena(dppr_engine *e, const Epoch &ep) {
    const long long stride = ((long long)ep.grp_n_int + 1023) / 1024 * 1024;
    if (stride <= e->res_arena_stride) return true;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    (void)hipFree(e->res_arena);
    e->res_arena = nullptr;
    e->res_arena_stride = 0;
    const long long want = std::min<long long>(((long long)e->V + 1023) / 1024 * 1024, stride + stride / 4);
    if (hipMalloc((void **)&e->res_arena, sizeof(double) * (size_t)want * RES_VECTORS) != hipSuccess) {
        (void)hipGetLastError();
        e->res_arena = nullptr;
        return false;
    }
    e->res_arena_stride = want;
    return true;
}

int run_frontier_loop(dppr_engine *e, Slot &s, const Epoch &ep, int phase, double eps, int buf, int cur,
                      LoopEntry entry = LoopEntry()) {
    const int hp = phase == PHASE_BOTH ? 0 : phase; // (loop histories: the merged loop uses slot 0)
    const int pull_min = pull_min_frontier(e);
    const bool sync_sched = e->schedule == DPPR_SCHEDULE_SYNC;
    const HubTable hubs{ep.hub_v, ep.hub_degp1, ep.n_hubs};
    // the sparse grid must cover the largest frontier a push chunk can meet
    const int push_grid = pull_min == 0x7fffffff ? 2048 : std::min(2048, std::max(64, (pull_min * 4 / WAVE + 3) / 4));
    // Sweeps on a window that cannot run resident carry the activity bitmap of their snapshot (k_pull_iter<.., true>)
    const int pcap0 = persist_capacity(e);
    const bool binned = ep.bin_valid && ep.bin_n_int <= ep.grp_n_int && (pcap0 <= 0 || ep.n_groups > pcap0 || e->bin_mode == 2);
    const bool use_bits = e->sweep_bits && !binned && !entry.dense && (pcap0 <= 0 || ep.n_groups > pcap0);
    // The merged loop always filters through the status array: adds of both signs can take a residual across the threshold more
    // than once per iteration, and with the crossing test every crossing would append -- the next-frontier list (V entries) could
    // overflow. One entry per vertex and launch keeps it bounded.
    const bool use_status = e->status_dedup || phase == PHASE_BOTH;
    if (use_status && !s.status) { // (first use: -1 everywhere = "never queued")
        HIP_TRY(hipMalloc((void **)&s.status, sizeof(int) * (size_t)e->V));
        HIP_TRY(hipMemsetAsync(s.status, 0xff, sizeof(int) * (size_t)e->V, e->stream));
    }
    bool extracted = false;         // ... and that snapshot zeroed the residuals it took (InspectExtra): the push needs no repair
    bool dense_valid = entry.dense; // s.x holds the snapshot of the current frontier (p already updated)
    bool list_valid = !entry.dense; // s.ft[buf] holds the frontier as a list (sweeps only count it)
    bool any_pull = entry.any_pull;
    bool x_clean = false;           // a resident launch ended the loop and left s.x / s.x2 all zero
    auto make_list = [&]() -> int { // dense snapshot -> sparse list (after a sweep)
        HIP_TRY(hipMemsetAsync(s.cnt + 7, 0, sizeof(int), e->stream));
        hipLaunchKernelGGL(k_list_from_dense, dim3(grid_for(ep.grp_n_int, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream,
                           s.x, ep.grp_n_int, s.cnt + cur, s.ft[buf], s.cnt + 7);
        HIP_TRY(hipGetLastError());
        list_valid = true;
        return DPPR_OK;
    };
    int F = entry.F, prevF = 0, active_iters = entry.it;
    long long D = -1; // in-edges of the current frontier (binned windows), -1 = not counted
    unsigned long long *dsum = reinterpret_cast<unsigned long long *>(s.cnt + 8); // three slots beside the rotating counters
    int follow = 4; // size of the next follow-up chunk of per-iteration sweeps
    int rc = DPPR_OK;
    if (F < 0 && (rc = read_count(e, s.cnt + cur, &F))) return rc;
    if (entry.it == 0) {
        s.start_dense[hp] = F >= pull_min;
        s.last_F0[hp] = F;
    }
    for (int it = entry.it; F > 0;) {
        if (it >= e->max_iters) return fail(e, DPPR_ERR_NOT_CONVERGED, "iteration cap hit");
        if (s.trace) {
            if (!list_valid && (rc = make_list())) return rc;
            size_t old = s.trace_ids.size();
            s.trace_ids.resize(old + (size_t)F);
            HIP_TRY(hipMemcpyAsync(s.trace_ids.data() + old, s.ft[buf], sizeof(int) * (size_t)F,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(loop_wait(e));
            for (size_t i = old; i < s.trace_ids.size(); ++i) s.trace_ids[i] = e->int2ext[(size_t)s.trace_ids[i]];
            s.trace_off.push_back((int64_t)s.trace_ids.size());
        }
        bool pull = F >= pull_min;
        // a window whose iterations cost hundreds of microseconds and more (twitter / friendster size): decisions per iteration
        const bool costly = binned && !e->chunk_explicit && (s.sweep_us > 0 ? s.sweep_us : 6.5e-6 * (double)ep.Ed) >= 300.0;
        if (costly && e->cost_model && !sync_sched && !s.trace && e->pull_min_frontier == 0) {
            // Push or sweep by what each would cost. A push is one returning atomic per in-edge of the
            // frontier, executed at the memory side at ~23.5 G/s chip-wide whatever the locality (profiles/r03_atomics_probe.json);
            // a sweep of this window costs what the last ones did. The frontier's in-edges are counted by the sweep that left it
            // (k_bin_reduce) or, for a list, by k_front_degree. (Round 3 switched on the vertex count: a late frontier of 1.7 M
            // low-degree vertices is pushed in 0.23 ms and was swept for 2.6, the 156 K batch tails -- hubs -- cost a sweep's time.)
            if (D < 0 && F >= 1024) {
                if (!list_valid && (rc = make_list())) return rc;
                HIP_TRY(hipMemsetAsync(dsum + cur, 0, sizeof(unsigned long long), e->stream));
                hipLaunchKernelGGL(k_front_degree, dim3(grid_for(F)), dim3(BLOCK), 0, e->stream, s.ft[buf], s.cnt + cur, ep.row_ptr, dsum + cur);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(e->pinned, dsum + cur, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(loop_wait(e));
                unsigned long long d;
                memcpy(&d, e->pinned, sizeof(d));
                D = (long long)d;
            }
            if (D >= 0 || F < 1024) {
                const double sweep_us = s.sweep_us > 0 ? s.sweep_us : 6.5e-6 * (double)ep.Ed; // (no sweep timed yet: ~6.5 ps per edge)
                const double push_us = 15.0 + (double)std::max<long long>(D, 0) * s.atomic_ns * 1e-3; // (measured on this slot's own pushes)
                pull = F >= 1024 && push_us > 0.9 * sweep_us;
            }
        }
        int n;
        if (s.trace || e->chunk_iters <= 1) n = 1;
        else if (costly)
            // A window on binned sweeps: an iteration costs milliseconds (friendster stand-in: 2.6 ms a sweep, 11-13 ms the push
            // of a 3-10 M-vertex frontier), a read-back tens of microseconds. Nothing is enqueued blind: round 3 ran the second
            // iteration of every loop as a push of ten million vertices (decided at 156 K) and ended every loop with three to
            // seven sweeps over frontiers of a few hundred vertices (enqueued from the last batches' lengths) -- 40 of 183 ms.
            n = (!pull && F < 4096 && F <= prevF) ? e->chunk_iters : 1;
        else if (pull) // consecutive batches take almost the same number of iterations: aim just past the end
            n = s.iter_hint[hp] > it ? s.iter_hint[hp] - it + 1 : e->chunk_iters;
        else if ((long long)F * 4 >= pull_min) n = 1;          // about to turn dense: re-decide next iteration
        else n = F > prevF ? 2 : e->chunk_iters;                // growing: short chunks; decaying tail: long
        const int pcap = persist_capacity(e);
        const bool resident = pull && n >= 2 && !s.trace && pcap > 0 && ep.n_groups > 0 && ep.n_groups <= pcap && resident_arena(e, ep);
        if (resident && s.iter_hint[hp] > it) n += RESIDENT_MARGIN - 1;
        if (!resident && pull && n > 1) {
            // per-iteration sweeps: a launch that finds the frontier empty is still a dispatch, a chunk boundary (read-back
            // + relaunch) costs about three of them -- go as far as the SHORTEST of the last four loops of this phase went
            // (almost surely needed in full), then in chunks that double from 4 (group_loop sizes its chunks the same way)
            int lo = 0;
            for (int h : s.iter_hist[hp]) lo = h > 0 && (lo == 0 || h < lo) ? h : lo;
            if (lo > it) n = lo - it;
            else if (lo > 0) {
                n = std::min(follow, e->chunk_iters);
                follow *= 2;
            }
        }
        n = std::min(n, MAX_CHUNK);
        if (e->chunk_explicit) n = std::min(n, std::max(e->chunk_iters, 1));
        if (!pull && !list_valid && (rc = make_list())) return rc;
        if (resident) {
            // ---- a run of dense iterations as ONE resident launch (dppr_resident.hpp)
            if (!dense_valid) {
                hipLaunchKernelGGL(k_snapshot_dense, dim3(std::min(grid_for(std::max(F, 1 << 14)), 1024)), dim3(BLOCK), 0,
                                   e->stream, s.ft[buf], s.cnt + cur, s.r, s.p, s.x, (uint32_t *)nullptr, phase == PHASE_BOTH ? 1 : 0, 0);
                dense_valid = true;
            }
            HIP_TRY(hipMemsetAsync(e->bar, 0, sizeof(GridBar), e->stream));
            n = std::min(n, RES_MAX_SWEEPS);
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[0], e->stream));
#define DPPR_LAUNCH_PERSIST(PB)                                                                                       \
    hipLaunchKernelGGL(k_pull_resident<PB>, dim3(ep.n_groups), dim3(PB), 0, e->stream, ep.grp_n_int, ep.grp_tile,       \
                       ep.out_row_ptr, ep.out_col, s.x, e->res_arena, e->res_arena_stride, s.r, s.p, s.cnt, cur, phase, eps, s.dstats, s.log,  \
                       n, e->bar, s.cnt + 7, e->persist_ticks, e->persist_rollcall_extra, 0,                          \
                       ep.res_valid ? ep.res_pk : nullptr, ResUpdate{})
            switch (sweep_block(e)) {
            case 256: DPPR_LAUNCH_PERSIST(256); break;
            case 512: DPPR_LAUNCH_PERSIST(512); break;
            default: DPPR_LAUNCH_PERSIST(1024); break;
            }
#undef DPPR_LAUNCH_PERSIST
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[1], e->stream));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(e->pinned, s.cnt, sizeof(int) * (size_t)(CNT_HDR + n), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(loop_wait(e));
            const int status = e->pinned[7];
            s.st.persist_launches++;
            if (status & PERSIST_FAULT) return fail(e, DPPR_ERR_HIP, "grid barrier of the resident sweep timed out");
            if (status & PERSIST_ABORTED) {
                // the roll-call failed (the grid was not co-resident): nothing was changed; this engine
                // goes on with per-iteration launches
                s.st.persist_aborts++;
                e->persist_ok = false;
                e->persist_retry = PERSIST_RETRY_BATCHES;
                continue;
            }
            for (int k = 0; k < n; ++k) {
                const int f = e->pinned[CNT_HDR + k];
                if (f <= 0) continue;
                s.st.iterations++;
                s.st.pull_iterations++;
                s.st.sum_F += f;
                active_iters = it + k + 1;
            }
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
                s.st.push_ms += ms;
                s.st.push_launches++;
            }
            // (s.x holds the snapshot the last sweep wrote)
            cur = 0;                              // the launch leaves the live count in cnt[0]
            list_valid = false;
            any_pull = true;
            x_clean = (status & PERSIST_CONVERGED) != 0;
            prevF = F;
            F = e->pinned[0];
            it += n;
            continue;
        }
        for (int k = 0; k < n; ++k) {
            const int nxt = (cur + 1) % 3, zer = (cur + 2) % 3;
            int *log_slot = s.log + k;
            if ((pull || sync_sched) && !dense_valid) {
                // grid-stride over a frontier whose size is only known on the device (k > 0): sized for
                // the last size the host saw, capped
                const bool bm = use_bits && pull;
                if (bm) HIP_TRY(hipMemsetAsync(s.act[0], 0, s.act_bytes, e->stream));
                extracted = e->pre_extract && !pull; // (a sweep repairs by itself: rn -= x[v])
                hipLaunchKernelGGL(k_snapshot_dense, dim3(std::min(grid_for(std::max(F, 1 << 14)), 1024)), dim3(BLOCK), 0,
                                   e->stream, s.ft[buf], s.cnt + cur, s.r, s.p, s.x, bm ? s.act[0] : (uint32_t *)nullptr, phase == PHASE_BOTH ? 1 : 0,
                                   extracted ? 1 : 0);
                dense_valid = true;
            }
            if (costly) HIP_TRY(hipMemsetAsync(dsum + nxt, 0, sizeof(unsigned long long), e->stream));
            const bool timed = e->profiling || (costly && n == 1); // (the push / sweep decision prices both by what the last ones took)
            if (timed) HIP_TRY(hipEventRecord(e->evpool[2 * k], e->stream));
            if (pull && binned) {
                // the sweep as two streaming passes over the epoch's binned edge layout (dppr_binned.hpp)
                if (ep.n_chunks > 0)
                    hipLaunchKernelGGL(k_bin_scatter, dim3(ep.n_chunks), dim3(BIN_NT), (size_t)e->bin_ha_tiles * WAVE * sizeof(double), e->stream,
                                       ep.bin_n_int, s.cnt + cur, ep.acut, ep.chunks, ep.hl, ep.apos, s.x, e->bin_vals);
                const int rows_cap = e->bin_hb_tiles * WAVE;
                hipLaunchKernelGGL(k_bin_reduce, dim3(ep.n_b + (ep.grp_n_int - ep.bin_n_int + rows_cap - 1) / rows_cap), dim3(BIN_NT),
                                   (size_t)rows_cap * 20, e->stream, ep.grp_n_int, ep.bin_n_int, ep.n_b, s.cnt + cur, ep.bcut, rows_cap,
                                   ep.out_row_ptr, ep.dl, e->bin_vals, s.x,
                                   s.x2, s.r, s.p, s.cnt + nxt, s.cnt + zer, phase, eps, s.dstats + 1, log_slot, e->directed ? ep.row_ptr : (const int *)nullptr,
                                   costly ? dsum + nxt : (unsigned long long *)nullptr);
                std::swap(s.x, s.x2);
                dense_valid = true;
                extracted = false;
                list_valid = false;
                any_pull = true;
            } else if (pull) {
                // workgroup size = max tiles per group x 64 (the groups themselves were cut by the builder)
                const int pb = sweep_block(e);
#define DPPR_LAUNCH_PULL(PB, BITS)                                                                                    \
    hipLaunchKernelGGL((k_pull_iter<PB, BITS>), dim3(std::min(std::max(ep.n_groups, 1), 1024)), dim3(PB), 0, e->stream, \
                       ep.grp_n_int, ep.grp_tile, ep.n_groups, s.cnt + cur, ep.out_row_ptr, ep.out_col, s.x, s.x2, s.r, \
                       s.p, s.cnt + nxt, s.cnt + zer, phase, eps, s.dstats + 1, log_slot,                               \
                       std::min(e->big_row, PULL_BIG_ROW_DEFAULT), s.act[0], s.act[1])
                if (use_bits) {
                    switch (pb) {
                    case 256: DPPR_LAUNCH_PULL(256, true); break;
                    case 384: DPPR_LAUNCH_PULL(384, true); break;
                    case 512: DPPR_LAUNCH_PULL(512, true); break;
                    case 576: DPPR_LAUNCH_PULL(576, true); break;
                    case 640: DPPR_LAUNCH_PULL(640, true); break;
                    case 768: DPPR_LAUNCH_PULL(768, true); break;
                    case 896: DPPR_LAUNCH_PULL(896, true); break;
                    default: DPPR_LAUNCH_PULL(1024, true); break;
                    }
                    std::swap(s.act[0], s.act[1]);
                } else { // (block sizes that are not 256 / 512 / 1024 never run resident: they always take the form above)
                    switch (pb) {
                    case 256: DPPR_LAUNCH_PULL(256, false); break;
                    case 512: DPPR_LAUNCH_PULL(512, false); break;
                    default: DPPR_LAUNCH_PULL(1024, false); break;
                    }
                }
#undef DPPR_LAUNCH_PULL
                std::swap(s.x, s.x2); // the sweep wrote every entry of x2: it is the next snapshot
                dense_valid = true;
                extracted = false;
                list_valid = false;
                any_pull = true;
            } else {
                int *big_cnt = s.cnt + 5 + (int)(s.iter_seq & 1), *big_zero = s.cnt + 5 + (int)((s.iter_seq + 1) & 1);
                s.iter_seq++;
                const Dedup dd{use_status ? s.status : nullptr, (int)(s.iter_seq & 0x3fffffff)};
                if (dense_valid)
                    hipLaunchKernelGGL(k_push_iter<true>, dim3(push_grid), dim3(BLOCK), 0, e->stream, s.ft[buf],
                                       s.cnt + cur, s.ft[buf ^ 1], s.cnt + nxt, s.cnt + zer, s.x, ep.row_ptr, ep.adj, hubs,
                                       s.big, big_cnt, big_zero, e->big_row, s.r, s.p, phase, eps, s.dstats, log_slot, dd, extracted ? 1 : 0);
                else
                    hipLaunchKernelGGL(k_push_iter<false>, dim3(push_grid), dim3(BLOCK), 0, e->stream, s.ft[buf],
                                       s.cnt + cur, s.ft[buf ^ 1], s.cnt + nxt, s.cnt + zer, s.x, ep.row_ptr, ep.adj, hubs,
                                       s.big, big_cnt, big_zero, e->big_row, s.r, s.p, phase, eps, s.dstats, log_slot, dd, 0);
                hipLaunchKernelGGL(k_push_big, dim3(512), dim3(BLOCK), 0, e->stream, s.big, big_cnt, s.ft[buf ^ 1],
                                   s.cnt + nxt, ep.adj, hubs, s.r, phase, eps, s.dstats, dd);
                dense_valid = false; // the push consumed (and zeroed) the snapshot
                extracted = false;
                list_valid = true;
            }
            if (timed) HIP_TRY(hipEventRecord(e->evpool[2 * k + 1], e->stream));
            buf ^= 1;
            cur = nxt;
        }
        HIP_TRY(hipGetLastError());
        // one read-back per chunk: the new frontier size and the F of each iteration just run
        HIP_TRY(hipMemcpyAsync(e->pinned, s.cnt, sizeof(int) * (size_t)(CNT_HDR + n), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(loop_wait(e));
        for (int k = 0; k < n; ++k) {
            const int f = e->pinned[CNT_HDR + k];
            if (f <= 0) continue; // the frontier emptied inside the chunk: the rest were no-ops
            s.st.iterations++;
            s.st.sum_F += f;
            if (pull) s.st.pull_iterations++;
            if (pull) s.st.sweep_F += f;
            if (pull && binned) s.st.binned_sweeps++;
            active_iters = it + k + 1;
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[2 * k], e->evpool[2 * k + 1]));
                s.st.push_ms += ms;
                s.st.push_launches++;
                if (pull) {
                    s.st.sweep_ms += ms;
                    s.st.sweep_launches++;
                }
                static const bool trace = getenv("DPPR_LOOP_TRACE") != nullptr; // (diagnostic: one line per iteration of a profiled batch)
                if (trace)
                    fprintf(stderr, "[loop  ] phase %d iteration %3d  %-6s frontier %9d  %8.1f us\n", phase, it + k,
                            pull ? (binned ? "binned" : "sweep") : "push", f, ms * 1e3);
            }
        }
        if (costly && n == 1 && e->pinned[CNT_HDR] > 0) { // what a sweep of this window costs / what an atomic of a push does (running means)
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
            if (pull) s.sweep_us = s.sweep_us > 0 ? 0.75 * s.sweep_us + 0.25 * ms * 1e3 : ms * 1e3;
            else if (D >= (1 << 20)) s.atomic_ns = 0.75 * s.atomic_ns + 0.25 * std::min(1.0, std::max(0.02, (ms * 1e6 - 15e3) / (double)D));
        }
        prevF = F;
        F = e->pinned[cur];
        if (binned && pull) { // the sweep counted the in-edges of the frontier it left
            unsigned long long d;
            memcpy(&d, e->pinned + 8 + 2 * cur, sizeof(d));
            D = (long long)d;
        } else {
            D = -1;
        }
        it += n;
    }
    s.iter_hint[hp] = active_iters;
    for (int k = 3; k > 0; --k) s.iter_hist[hp][k] = s.iter_hist[hp][k - 1];
    s.iter_hist[hp][0] = active_iters;
    if (any_pull && !x_clean) { // leave both dense vectors all-zero for the next loop
        // only internal ids below n_int are ever written
        HIP_TRY(hipMemsetAsync(s.x, 0, sizeof(double) * (size_t)ep.grp_n_int, e->stream));
        HIP_TRY(hipMemsetAsync(s.x2, 0, sizeof(double) * (size_t)ep.grp_n_int, e->stream));
    }
    return DPPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Both frontier loops of one batch as ONE resident launch, without a read-back in between.
//
// When consecutive batches behave alike (both phases start with a frontier worth a sweep -- the
// steady state of a sliding-window stream), the host knows what it will launch before it has seen
// any count. After a converged solve the frontier of a phase is {v : legal(residual[v])}, which the
// resident kernel reads off its registers (PLAN_SEED), and when phase 0 is over it seeds phase 1
// the same way and goes on (PLAN_BOTH): Inspect / snapshot / phase 0 / Inspect / snapshot / phase 1
// of gpu/PPRGPU.cuh:138-164 are one kernel. One copy of the counters, the status word and the log
// comes back at the end. Whatever did not go as expected (a phase needed more sweeps than the
// launch was given, the roll-call failed) leaves the state at a well-defined point from which the
// ordinary host-driven loop resumes (`stage`, `en0`, `en1`).
// The reference pays a blocking read-back per ITERATION (gpu/PPRRevPushGPU.cuh:107).
// ---------------------------------------------------------------------------------------------
bool can_batch_ahead(const dppr_engine *e, const Slot &s, const Epoch &ep) {
    const int cap = persist_capacity(e);
    if (e->persist_mode != 1 || cap <= 0 || ep.n_groups <= 0 || ep.n_groups > cap || s.trace || e->chunk_iters <= 1 || ep.L <= 0)
        return false;
    // A resident sweep costs the same ~5 us whatever the frontier size, less than one push iteration's
    // launches: with the automatic push/pull threshold a window that can run resident always does.
    // With an explicit threshold (tests) only if the last batch's phases both started above it.
    if (e->merge_phases && e->schedule == DPPR_SCHEDULE_EAGER) // (the merged loop keeps its history in slot 0)
        return e->pull_min_frontier == 0 || (s.iter_hint[0] > 0 && s.start_dense[0]);
    return e->pull_min_frontier == 0 ||
           (s.iter_hint[0] > 0 && s.iter_hint[1] > 0 && s.start_dense[0] && s.start_dense[1]);
}

// stage (out): 0 = phase 0 still open (resume with en0), 1 = phase 0 done, phase 1 open (resume with
// en1; *p1_seeded tells whether its snapshot exists), 2 = both phases done
int batch_ahead(dppr_engine *e, Slot &s, const Epoch &ep, double eps, int *stage, LoopEntry *en0, LoopEntry *en1,
                bool *p1_seeded, bool merged = false, bool inline_update = false) {
    // merged (dppr_set_phase_merge): ONE loop over residuals of both signs -- the launch seeds it (PLAN_SEED) and runs it to the
    // end; stage 0 + en0 if it ran out of sweeps, stage 2 when it converged (histories in slot 0)
    const int pull_min = pull_min_frontier(e);
    // a resident launch stops by itself when the frontier empties: a generous allowance costs nothing,
    // a short one costs a read-back and another launch (+1: the step that seeds phase 1)
    int n = merged ? (s.iter_hint[0] > 0 ? std::min(s.iter_hint[0] + 2 * RESIDENT_MARGIN, 2 * MAX_CHUNK) : 2 * MAX_CHUNK)
            : s.iter_hint[0] > 0 && s.iter_hint[1] > 0
                      ? std::min(s.iter_hint[0] + s.iter_hint[1] + 1 + 2 * RESIDENT_MARGIN, 2 * MAX_CHUNK)
                      : 2 * MAX_CHUNK; // no history yet
    if (e->chunk_explicit) n = std::min(n, e->chunk_iters); // (tests: launches that stop mid-phase and are resumed)
    n = std::min(n, RES_MAX_SWEEPS);
    int *status = s.cnt + 7; // (the GridBar was zeroed by the batch's first kernel, k_su_keys)
    const ResUpdate upd = !inline_update ? ResUpdate{}
                          : ep.grouped   ? ResUpdate{ep.su_rng, ep.sk, ep.sv, ep.b2, ep.ins, ep.deg_after, s.source, nullptr, 0}
                                         : ResUpdate{nullptr, nullptr, nullptr, ep.b2, ep.ins, nullptr, s.source, ep.b1, ep.L}; // raw records
    const int plan = (merged ? PLAN_SEED : (PLAN_SEED | PLAN_BOTH)) | (inline_update ? PLAN_UPDATE : 0);
    if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[0], e->stream));
#define DPPR_LAUNCH_PERSIST(PB)                                                                                        \
    hipLaunchKernelGGL(k_pull_resident<PB>, dim3(ep.n_groups), dim3(PB), 0, e->stream, ep.grp_n_int, ep.grp_tile,        \
                       ep.out_row_ptr, ep.out_col, s.x, e->res_arena, e->res_arena_stride, s.r, s.p, s.cnt, 0,                  \
                       merged ? PHASE_BOTH : 0, eps, s.dstats,                                                             \
                       s.log, n, e->bar, status, e->persist_ticks, e->persist_rollcall_extra,                             \
                       plan, ep.res_valid ? ep.res_pk : nullptr, upd)
    switch (sweep_block(e)) {
    case 256: DPPR_LAUNCH_PERSIST(256); break;
    case 512: DPPR_LAUNCH_PERSIST(512); break;
    default: DPPR_LAUNCH_PERSIST(1024); break;
    }
#undef DPPR_LAUNCH_PERSIST
    if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[1], e->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e->pinned, s.cnt, sizeof(int) * (size_t)(CNT_HDR + n), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(loop_wait(e));

    const int st = e->pinned[7];
    *stage = 0;
    *p1_seeded = false;
    *en0 = LoopEntry();
    *en1 = LoopEntry();
    s.st.persist_launches++;
    if (st & PERSIST_FAULT) return fail(e, DPPR_ERR_HIP, "a wait inside the resident sweep timed out");
    e->launch_called_off = false;
    if ((st & PERSIST_ABORTED) && inline_update && !ep.grouped && e->pinned[4] == 1) {
        // a sweep group owns more of the batch's records than it has threads: the launch called itself off before anything was
        // changed -- not a residency problem. The caller applies the update with its own kernels; the next batches do so at once.
        e->launch_called_off = true;
        e->raw_backoff = 16;
        return DPPR_OK;
    }
    if (st & PERSIST_ABORTED) { // roll-call failed: nothing was changed, the lists of the stream update stand
        s.st.persist_aborts++;
        e->launch_called_off = true;
        e->persist_ok = false;
        e->persist_retry = PERSIST_RETRY_BATCHES;
        return DPPR_OK;
    }
    if (e->profiling) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
        s.st.push_ms += ms;
        s.st.push_launches++;
    }
    // the log: frontier sizes of phase 0, a zero (phase 0 over), those of phase 1, a zero
    const int *log = e->pinned + CNT_HDR;
    const int pos = st & PERSIST_SWEEPS; // loop position the launch stopped at
    int act[2] = {0, 0}, ph = 0;
    for (int k = 0; k < std::min(pos + 1, n) && ph < 2; ++k) {
        if (log[k] <= 0) {
            ++ph;
            continue;
        }
        if (act[ph] == 0) {
            s.start_dense[ph] = log[k] >= pull_min;
            s.last_F0[ph] = log[k];
        }
        s.st.iterations++;
        s.st.pull_iterations++;
        s.st.sum_F += log[k];
        act[ph]++;
    }
    if (merged) {
        if (!(st & PERSIST_CONVERGED)) { // out of sweeps: the host-driven loop goes on from here
            en0->it = act[0];
            en0->F = e->pinned[0];
            en0->dense = true;
            en0->any_pull = true;
            return DPPR_OK;
        }
        s.iter_hint[0] = act[0];
        for (int k = 3; k > 0; --k) s.iter_hist[0][k] = s.iter_hist[0][k - 1];
        s.iter_hist[0][0] = act[0];
        if (act[0] == 0) s.start_dense[0] = false;
        *stage = 2;
        return DPPR_OK;
    }
    if (!(st & PERSIST_PHASE1)) { // phase 0 needs more sweeps than the launch had; phase 1 has not started
        en0->it = act[0];
        en0->F = e->pinned[0];
        en0->dense = true;
        en0->any_pull = true;
        return DPPR_OK;
    }
    s.iter_hint[0] = act[0];
    if (act[0] == 0) s.start_dense[0] = false;
    *stage = 1;
    *p1_seeded = true;
    if (!(st & PERSIST_CONVERGED)) {
        en1->it = act[1];
        en1->F = e->pinned[0];
        en1->dense = true;
        en1->any_pull = true;
        return DPPR_OK;
    }
    s.iter_hint[1] = act[1];
    if (act[1] == 0) s.start_dense[1] = false;
    *stage = 2;
    return DPPR_OK;
}

// full Inspect seeding + loop = ExecuteMainLoop(phase)
int main_loop_inspect(dppr_engine *e, Slot &s, const Epoch &ep, int phase, double eps) {
    s.seed_lists_valid = false;
    HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * 3, e->stream));
    hipLaunchKernelGGL(k_inspect, dim3(grid_for(ep.grp_n_int, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream, s.r,
                       ep.grp_n_int, phase, eps, s.ft[0], s.cnt + 0);
    HIP_TRY(hipGetLastError());
    s.st.inspected += ep.grp_n_int;
    return run_frontier_loop(e, s, ep, phase, eps, 0, 0);
}

// Stable grouping of the epoch's batch records by tail: su_k[1] = tails ascending, su_v[1] = record indices
// (ascending inside a tail): key extraction + the device radix sort. `zero` / `zero_ints` are cleared on the
// way (the counters of what follows).
inline const uint32_t *batch_tails(const dppr_engine *e, const Epoch &ep) { return ep.grouped ? ep.sk : e->su_k[1]; }
inline const uint32_t *batch_order(const dppr_engine *e, const Epoch &ep) { return ep.grouped ? ep.sv : e->su_v[1]; }

// The grouping is a function of the batch's records alone (not of any solver state): by default it is done once, when the batch
// is uploaded (dppr_slide -> epoch_group_records; the reference uploads its GPUEdgeBatch untimed as well, gpu/PPRGPU.cuh:131-135),
// and the timed region starts with a kernel that only clears the loop's counters. dppr_set_batch_grouping(e, 0) keeps it inside
// dppr_update (the accounting of rounds 1-2: + 5 dispatches of the device radix sort per batch).
int epoch_group_records(dppr_engine *e, Epoch &ep) {
    ep.grouped = false;
    ep.su_inline = false;
    if (!e->group_at_slide || ep.L <= 0) return DPPR_OK;
    if (ep.id >= 0) { // an epoch built while the default accounting was on, grouped outside the bracket after all (prepare_epoch): its degrees too
        hipLaunchKernelGGL(k_copy_out_degree, dim3(grid_for(ep.L)), dim3(BLOCK), 0, e->bs, ep.b1, ep.L, ep.out_row_ptr, ep.deg_after);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_su_keys, dim3(grid_for(ep.L)), dim3(BLOCK), 0, e->bs, ep.b1, ep.L, e->su_k[0], e->su_v[0],
                       (unsigned long long *)nullptr, 0, (int *)nullptr, 0);
    size_t tmp = e->su_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(e->su_tmp, tmp, e->su_k[0], ep.sk, e->su_v[0], ep.sv, (size_t)ep.L, 0u, (unsigned)e->bits, e->bs));
    ep.grouped = true;
    return res_record_ranges(e, ep);
}

int group_records_by_tail(dppr_engine *e, const Epoch &ep, unsigned long long *zero, int nz, int *zero_ints, int nzi) {
    const int L = ep.L;
    if (ep.grouped) { // only the counters (and the GridBar of a resident launch enqueued ahead) are cleared here
        if (nz > 0 || nzi > 0)
            hipLaunchKernelGGL(k_su_keys, dim3(1), dim3(BLOCK), 0, e->stream, ep.b1, 0, e->su_k[0], e->su_v[0], zero, nz, zero_ints, nzi);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    }
    // inside the timed region (default): CopyOutDegree (gpu/StreamUpdate.cuh:7-17; a tail's post-batch out-degree = the length of
    // its row in this epoch's out-CSR) and the grouping by tail -- ranked in one launch up to SU_RANK_MAX records, radix-sorted beyond
    if (L <= SU_RANK_MAX && !e->force_radix_grouping) {
        hipLaunchKernelGGL(k_su_group_rank, dim3((L + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, e->stream, ep.b1, L, ep.out_row_ptr, e->su_k[1],
                           e->su_v[1], ep.deg_after, zero, nz, zero_ints, nzi);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    }
    hipLaunchKernelGGL(k_copy_out_degree, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, ep.b1, L, ep.out_row_ptr, ep.deg_after);
    hipLaunchKernelGGL(k_su_keys, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, ep.b1, L, e->su_k[0], e->su_v[0], zero, nz,
                       zero_ints, nzi);
    size_t tmp = e->su_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(e->su_tmp, tmp, e->su_k[0], e->su_k[1], e->su_v[0], e->su_v[1], (size_t)L, 0u,
                                      (unsigned)e->bits, e->stream));
    return DPPR_OK;
}

// dppr_set_batch_grouping(1) after epochs were built: their records are grouped now, BEFORE the caller's event bracket opens
inline int prepare_epoch(dppr_engine *e, Epoch &ep) {
    if (e->group_at_slide && !ep.grouped && ep.L > 0) return epoch_group_records(e, ep);
    return DPPR_OK;
}

int stream_update(dppr_engine *e, Slot &s, const Epoch &ep, double eps, bool seed, bool zero_bars = false) {
    const int L = ep.L;
    if (L == 0) {
        HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * 5, e->stream));
        return DPPR_OK;
    }
    // (the batch's first kernel also clears cnt[0..4] and, for a resident launch enqueued ahead, its GridBar)
    int rc = group_records_by_tail(e, ep, zero_bars ? reinterpret_cast<unsigned long long *>(e->bar) : nullptr,
                                   zero_bars ? (int)(sizeof(GridBar) / sizeof(unsigned long long)) : 0, s.cnt, 5);
    if (rc) return rc;
    // without seeding the lists go to scratch space (cnt[4] / neg) and are ignored
    if (L >= SU_SPLIT_MIN) {
        // Large batches: a hub's tail owns thousands of records, and the fused kernel's leader walks what lies beyond its
        // 1 024-record LDS window through three dependent gathers per record (twitter stand-in, 2.9 M records: 3.0 ms of a batch).
        // The terms of ALL records are computed in parallel first; the leaders then walk contiguous arrays (same expressions,
        // same order: bit-identical, the form source groups use).
        SuSources srcs{};
        srcs.s[0] = s.source;
        hipLaunchKernelGGL(k_su_terms, dim3(grid_for(L), 1), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), ep.b2, ep.ins, L, s.p, 1,
                           e->su_term, e->su_ins);
        hipLaunchKernelGGL(k_su_apply, dim3(grid_for(L), 1), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), e->su_term, e->su_ins,
                           ep.deg_after, L, s.r, 1, srcs, seed ? eps : 1e300, s.ft[0], s.cnt + 0, s.neg, s.cnt + 3);
    } else {
        hipLaunchKernelGGL(k_su_apply_fused, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), ep.b2, ep.ins,
                           ep.deg_after, L, s.p, s.r, s.source, seed ? eps : 1e300, s.ft[0], s.cnt + 0, s.neg, s.cnt + 3);
    }
    HIP_TRY(hipGetLastError());
    s.st.records += L;
    return DPPR_OK;
}

int pull_device_stats(dppr_engine *e, Slot &s) {
    static thread_local IterStats h[2];
    HIP_TRY(hipMemcpyAsync(h, s.dstats, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long t = 0, ts = 0;
    for (int i = 0; i < STAT_SLOTS; ++i) {
        t += h[0].blk_E[i];
        ts += h[1].blk_E[i];
    }
    s.st.sum_E = (int64_t)(t + ts);
    s.st.sweep_E = (int64_t)ts;
    return DPPR_OK;
}

// ---------------------------------------------------------------------------- f2: groups
// The group kernels are instantiated per row width (dppr_multi.hpp: GW = 2, 4, .. 16 doubles; one double per lane of
// an octet up to 8, two beyond): f(SPL, GW) is called with the two as compile-time constants.
template <int N> using IC = std::integral_constant<int, N>;
template <class F>
void with_row(int gw, F &&f) {
    switch (gw) {
    case 2: f(IC<1>{}, IC<2>{}); break;
    case 4: f(IC<1>{}, IC<4>{}); break;
    case 6: f(IC<1>{}, IC<6>{}); break;
    case 8: f(IC<1>{}, IC<8>{}); break;
    case 10: f(IC<2>{}, IC<10>{}); break;
    case 12: f(IC<2>{}, IC<12>{}); break;
    case 14: f(IC<2>{}, IC<14>{}); break;
    default: f(IC<2>{}, IC<16>{}); break;
    }
}

// workgroups of the multi-sweep form of k_gsweep that the device holds at once
int group_multi_capacity(dppr_engine *e, int spl) {
    int &cap = e->gmulti_cap[spl - 1];
    if (cap < 0) {
        int per_cu = 0, cus = 0; // (the widest row of each lane split: narrower ones need no more)
        hipError_t rc = spl == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gsweep<1, 8, 1024, true, 2>, GNT, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gsweep<2, 16, 512, true, 2>, GNT, 0);
        if (rc != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess)
            per_cu = 0;
        cap = std::min(per_cu * cus, STAT_SLOTS);
    }
    return cap;
}

// One frontier loop of a source group. `tails`: the state was converged before the batch's stream
// update, so only the batch tails (sorted in su_k[1]) can be legal -- no pass over all vertices.
// The tail of a group's loop in push form (dppr_gpush.hpp). Called between two chunks of sweeps when the frontier is
// small: g.act[0] / g.x hold the frontier the last sweep left. Returns with *converged set (the loop is over; state as
// a finished loop leaves it) or cleared (the mode gave up -- an iteration too large for it -- and put the frontier back
// in sweep form: g.act[0], g.x, frontier sizes in row 0 of g.cnt, the other rows zero), or with *entered false if it
// did not start (nothing changed). Iterations run are added to *iters and to the group's statistics.
// *owed: the handed-over snapshot's pagerank share is still to be credited (the last sweep was a deferring one,
// dppr_multi.hpp); on a return in sweep form it says the same about the snapshot handed back.
int group_push_tail(dppr_engine *e, Group &g, const Epoch &ep, int phase, double eps, long long pairs_at_entry, int *iters, bool *entered,
                    bool *converged, bool *owed) {
    *entered = false;
    *converged = false;
    const int GWM = GS_MAX;
    const int cap = std::max(1024, std::min(e->gpush_list_cap, e->V));
    if (g.plist_cap != cap) {
        HIP_TRY(loop_wait(e));
        (void)hipFree(g.plist[0]); (void)hipFree(g.plist[1]); (void)hipFree(g.ppre); (void)hipFree(g.pctl);
        g.plist[0] = g.plist[1] = g.ppre = nullptr;
        g.pctl = nullptr;
        g.plist_cap = 0;
        HIP_TRY(hipMalloc((void **)&g.plist[0], sizeof(int) * (size_t)cap));
        HIP_TRY(hipMalloc((void **)&g.plist[1], sizeof(int) * (size_t)cap));
        HIP_TRY(hipMalloc((void **)&g.ppre, sizeof(int) * ((size_t)cap + 1)));
        HIP_TRY(hipMalloc((void **)&g.pctl, sizeof(GPushCtl)));
        g.plist_cap = cap;
    }
    static thread_local GPushCtl h;
    // no host round trip on the way in: a list that does not fit (overflow) moves nothing and makes the first scan call
    // the mode off, which the read-back of the first chunk shows
    HIP_TRY(hipMemsetAsync(g.pctl, 0, sizeof(GPushCtl), e->stream));
    const int n_words = (ep.grp_n_int + 31) / 32;
    hipLaunchKernelGGL(k_gpush_list, dim3(grid_for(n_words)), dim3(BLOCK), 0, e->stream, g.act[0], n_words, g.plist[0], cap, g.pctl);
    // the frontier's rows move from the snapshot back to residual[]; its bits stay set (they queue it for iteration 0)
    const int rows_grid = grid_for(std::min<long long>(pairs_at_entry, cap), BLOCK / OCT);
    with_row(g.gw, [&](auto spl, auto gw) {
        hipLaunchKernelGGL((k_gpush_rows<decltype(spl)::value, decltype(gw)::value>), dim3(rows_grid), dim3(BLOCK), 0, e->stream, g.plist[0],
                           g.pctl, 0, g.x, g.r, false);
    });
    HIP_TRY(hipGetLastError());
    *entered = true;
    const int credit_first = *owed ? 1 : 0; // (iteration 0 of this mode settles it; every later one credits as it snapshots)
    // what an iteration may cost here: a sweep's floor is ~0.02 us per sweep group, a returning f64 atomic ~1 / 20 000 us
    const long long max_edges = e->gpush_max_edges > 0 ? e->gpush_max_edges : std::max<long long>(4096, 200ll * std::max(ep.n_ggroups, 1));
    const int grid = 256;
    static const bool trace = getenv("DPPR_GROUP_TRACE") != nullptr;
    int it_done = 0;
    long long known_n = pairs_at_entry; // (an upper bound of the frontier's vertices until the first read-back)
    bool tiny_declined = false;
    long long last_adds = pairs_at_entry <= 64 ? 0 : -1; // edge x source adds of the last iteration run (-1: not known yet)
    for (;;) {
        if (known_n <= TINY_N && last_adds >= 0 && last_adds <= TINY_E / 2 && !tiny_declined) {
            // a frontier of a few hundred vertices: a run of iterations as ONE single-workgroup launch
            with_row(g.gw, [&](auto spl, auto gw) {
                hipLaunchKernelGGL((k_gpush_tiny<decltype(spl)::value, decltype(gw)::value>), dim3(1), dim3(1024), 0, e->stream, g.pctl, g.plist[0],
                                   g.plist[1], ep.row_ptr, ep.adj, ep.hub_degp1, g.r, g.p, g.act[0], phase, eps, g.dstats, GPUSH_LOG, credit_first);
            });
        } else {
        // iterations per chunk (<= GPUSH_LOG): down here the frontier about halves per iteration, so the first chunk is
        // sized to reach the single-workgroup form (an iteration that finds nothing is three empty dispatches)
        int m = 2;
        if (it_done == 0)
            for (long long f = pairs_at_entry; f > 128 && m < GPUSH_LOG; f >>= 2) ++m;
        tiny_declined = false;
        for (int k = 0; k < m; ++k) {
            hipLaunchKernelGGL(k_gpush_scan, dim3(1), dim3(1024), 0, e->stream, g.pctl, g.plist[0], g.plist[1], ep.row_ptr, g.ppre, cap - 1, max_edges);
            with_row(g.gw, [&](auto spl, auto gw) {
                constexpr int SPL = decltype(spl)::value, GW = decltype(gw)::value;
                hipLaunchKernelGGL((k_gpush_snap<SPL, GW>), dim3(grid), dim3(BLOCK), 0, e->stream, g.plist[0], g.plist[1], g.pctl, g.x, g.r, g.p,
                                   g.act[0], phase, eps, credit_first);
                hipLaunchKernelGGL((k_gpush_expand<SPL, GW>), dim3(grid), dim3(BLOCK), 0, e->stream, g.plist[0], g.plist[1], g.pctl, g.ppre,
                                   ep.row_ptr, ep.adj, ep.hub_degp1, g.x, g.r, g.act[0], g.plist[0], g.plist[1], cap, phase, eps, g.dstats);
            });
        }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h, g.pctl, sizeof(GPushCtl), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(loop_wait(e));
        for (int i = it_done; i < h.it; ++i) {
            long long F = 0;
            for (int s = 0; s < GWM; ++s) F += h.F[i & (GPUSH_LOG - 1)][s];
            if (F == 0) continue;
            g.st.iterations++;
            g.st.sum_F += F;
            ++*iters;
            if (trace)
                fprintf(stderr, "[gpush ] phase %d iteration +%d  frontier pairs %9lld  adds %lld\n", phase, i, F, h.atomics[i & (GPUSH_LOG - 1)]);
        }
        if (h.it == it_done && !h.stop && known_n <= TINY_N) tiny_declined = true; // (too many vertices or in-edges for one workgroup)
        if (h.it > it_done) last_adds = h.atomics[(h.it - 1) & (GPUSH_LOG - 1)];
        it_done = h.it;
        known_n = h.n[h.it & 1];
        if (h.stop && h.it == 0 && h.overflow) { // the frontier did not fit the lists: nothing was moved, the sweeps go on
            *entered = false;
            return DPPR_OK;
        }
        if (h.stop) { // an iteration too large for this form: the queued vertices go back to sweep form
            if (trace) fprintf(stderr, "[gpush ] phase %d: an iteration of %d vertices called itself off after %d iterations\n", phase, h.n[h.it & 1], h.it);
            HIP_TRY(hipMemsetAsync(g.cnt, 0, sizeof(int) * 3 * GWM, e->stream));
            with_row(g.gw, [&](auto spl, auto gw) {
                hipLaunchKernelGGL((k_gpush_leave<decltype(spl)::value, decltype(gw)::value>), dim3(grid_for(ep.grp_n_int, BLOCK / OCT)), dim3(BLOCK), 0,
                                   e->stream, ep.grp_n_int, g.act[0], g.x, g.r, g.p, h.it > 0 ? 1 : 0, phase, eps, g.cnt);
            });
            HIP_TRY(hipGetLastError());
            if (h.it > 0) *owed = false; // (iteration 0 settled the hand-over, k_gpush_leave credited what it queued)
            return DPPR_OK;
        }
        if (h.n[h.it & 1] == 0) {
            *converged = true;
            return DPPR_OK;
        }
        if (it_done >= e->max_iters) return fail(e, DPPR_ERR_NOT_CONVERGED, "iteration cap hit");
    }
}

int group_loop(dppr_engine *e, Group &g, const Epoch &ep, int phase, double eps, bool tails) {
    const int hp = phase == PHASE_BOTH ? 0 : phase; // (loop histories: the merged loop uses slot 0)
    int cur = 0;
    const int GWM = GS_MAX;
    HIP_TRY(hipMemsetAsync(g.cnt, 0, sizeof(int) * 3 * GWM, e->stream));
    if (tails) {
        HIP_TRY(hipMemsetAsync(g.act[0], 0, g.act_bytes, e->stream));
        if (ep.L > 0) {
            with_row(g.gw, [&](auto spl, auto gw) {
                hipLaunchKernelGGL((k_gseed_tails<decltype(spl)::value, decltype(gw)::value>), dim3(grid_for(ep.L, BLOCK / OCT)), dim3(BLOCK), 0,
                                   e->stream, batch_tails(e, ep), ep.L, g.r, g.x, g.p, g.act[0], phase, eps, g.cnt + cur * GWM);
            });
        }
    } else {
        // dense seeding: every legal vertex of every source enters, snapshot taken
        with_row(g.gw, [&](auto spl, auto gw) {
            hipLaunchKernelGGL((k_gseed_dense<decltype(spl)::value, decltype(gw)::value>), dim3(grid_for(ep.grp_n_int, BLOCK / OCT)), dim3(BLOCK), 0,
                               e->stream, ep.grp_n_int, g.r, g.x, g.p, g.act[0], phase, eps, g.cnt + cur * GWM);
        });
        g.st.inspected += (int64_t)ep.grp_n_int * g.n;
    }
    HIP_TRY(hipGetLastError());
    int *log = g.cnt + 5 * GWM;
    auto any_left = [&](const int *c) {
        for (int s = 0; s < GWM; ++s)
            if (c[s] > 0) return true;
        return false;
    };
    HIP_TRY(hipMemcpyAsync(e->pinned, g.cnt + cur * GWM, sizeof(int) * GWM, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(loop_wait(e));
    bool more = any_left(e->pinned);
    int active_iters = 0;
    if (e->gsweep_grid_cap <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0) cus = 256;
        e->gsweep_grid_cap = std::min(2 * cus, STAT_SLOTS);
    }
    const int sweep_grid = std::min(std::max(ep.n_ggroups, 1), e->gsweep_grid_cap);
    int follow = 4; // size of the next follow-up chunk of one-sweep launches
    // pagerank is credited every other sweep (dppr_multi.hpp): the seeding credited its snapshot, so the first sweep defers;
    // `owed` = the live snapshot's share has not been added yet, the next sweep is a crediting one
    bool owed = false;
    // the tail of the loop as pushes (dppr_gpush.hpp): below push_thr frontier pairs, one-sweep launches only
    long long push_thr = e->gpush_enter_pairs == 0 ? 0 : e->gpush_enter_pairs > 0 ? e->gpush_enter_pairs : std::max(64, ep.n_ggroups * e->gpush_auto_factor);
    bool push_gave_up = false;
    int dense_len = -1; // sweeps of this loop before the frontier was that small
    const int nvx = ep.ggrp_max_tiles * WAVE; // vertices per sweep group of this epoch's tables: 1024, or 512 once a 16-wide group exists
    for (int it = 0; more;) {
        if (it >= e->max_iters) return fail(e, DPPR_ERR_NOT_CONVERGED, "iteration cap hit");
        // ---- a window whose sweep groups are all resident at once: a run of sweeps as ONE launch (k_gsweep<.., true>)
        const int mcap = e->group_resident && e->persist_mode && e->persist_ok && e->chunk_iters > 1 ? group_multi_capacity(e, g.spl) : 0;
        if (mcap > 0 && ep.n_ggroups > 0 && ep.n_ggroups <= mcap) {
            int n = g.iter_hint[hp] > it ? g.iter_hint[hp] - it + RESIDENT_MARGIN : 2 * e->chunk_iters;
            n = std::max(2, std::min(n, GMULTI_MAX));
            if (e->chunk_explicit) n = std::min(n, std::max(e->chunk_iters, 2)); // (tests: launches that stop mid-loop and are resumed)
            HIP_TRY(hipMemsetAsync(g.mlog, 0, sizeof(int) * (size_t)(n + 2) * GWM, e->stream));
            HIP_TRY(hipMemsetAsync(e->bar, 0, sizeof(GridBar), e->stream));
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[0], e->stream));
            int *status = g.mlog, *rows = g.mlog + GWM;
#define DPPR_LAUNCH_GMULTI(SPL, GW, NVX)                                                                                 \
    hipLaunchKernelGGL((k_gsweep<SPL, GW, NVX, true, 2>), dim3(ep.n_ggroups), dim3(GNT), 0, e->stream, ep.grp_n_int, ep.gtab,   \
                       ep.n_ggroups, g.cnt + cur * GWM, e->gsweep_hot_rows, ep.out_col, g.x, g.x2, g.act[0], g.act[1], g.r, g.p, \
                       g.cnt + 3 * GWM, g.cnt + 4 * GWM, phase, eps, g.dstats + 1, rows, n, e->bar, status, e->persist_ticks,    \
                       e->persist_rollcall_extra, owed ? 1 : 0, (int *)nullptr, (int *)nullptr)
            with_row(g.gw, [&](auto spl, auto gw) {
                constexpr int SPL = decltype(spl)::value, GW = decltype(gw)::value;
                if constexpr (SPL == 2) DPPR_LAUNCH_GMULTI(2, GW, 512);
                else if (nvx == 512) DPPR_LAUNCH_GMULTI(1, GW, 512); // (a narrow group on an engine that also has a wide one)
                else DPPR_LAUNCH_GMULTI(1, GW, 1024);
            });
#undef DPPR_LAUNCH_GMULTI
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[1], e->stream));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(e->pinned, g.mlog, sizeof(int) * (size_t)(n + 2) * GWM, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(loop_wait(e));
            const int st = e->pinned[0];
            g.st.persist_launches++;
            if (st & GSM_FAULT) return fail(e, DPPR_ERR_HIP, "a grid barrier of the multi-sweep group launch timed out");
            if (st & GSM_ABORTED) { // not co-resident: nothing was changed; one-sweep launches from here on (re-armed later)
                g.st.persist_aborts++;
                e->persist_ok = false;
                e->persist_retry = PERSIST_RETRY_BATCHES;
                continue;
            }
            const int sweeps = st & GSM_SWEEPS;
            for (int k = 0; k < sweeps; ++k) {
                const int *f = e->pinned + GWM + k * GWM;
                g.st.iterations++;
                g.st.pull_iterations++;
                for (int s = 0; s < GWM; ++s) g.st.sum_F += f[s];
                for (int s = 0; s < GWM; ++s) g.st.sweep_F += f[s];
                active_iters = it + k + 1;
            }
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
                g.st.push_ms += ms;
                g.st.push_launches++;
            }
            if (sweeps & 1) {
                std::swap(g.x, g.x2);
                std::swap(g.act[0], g.act[1]);
                owed = !owed;
            }
            it += sweeps;
            if (st & GSM_CONVERGED) break;
            // out of sweeps: the live frontier sizes are in row `sweeps`; the launch left them in cnt[3] -- make them cnt[0]
            HIP_TRY(hipMemcpyAsync(g.cnt, g.cnt + 3 * GWM, sizeof(int) * GWM, hipMemcpyDeviceToDevice, e->stream));
            HIP_TRY(hipMemsetAsync(g.cnt + GWM, 0, sizeof(int) * 2 * GWM, e->stream));
            cur = 0;
            more = any_left(e->pinned + GWM + sweeps * GWM);
            continue;
        }
        // One-sweep launches are enqueued in chunks; a launch that finds every frontier empty returns at once, but it
        // still costs a dispatch (~4 us + gap). Consecutive batches take about the same number of sweeps, so the first
        // chunk is the SHORTEST of the last four loops of this phase (almost surely needed in full), and what follows
        // doubles from 4: a boundary (read-back + relaunch) costs about three empty dispatches.
        int n;
        if (it == 0) {
            int lo = 0;
            for (int h : (push_thr > 0 ? g.dense_hist : g.iter_hist)[hp]) lo = h > 0 && (lo == 0 || h < lo) ? h : lo;
            n = lo > 0 ? lo : e->chunk_iters;
            follow = 4;
        } else if (push_thr > 0 && !push_gave_up) {
            // the push form takes over below push_thr pairs and a sweep of the tail costs its floor whatever it finds: go
            // only as far as the frontier is sure to stay above the threshold (it shrinks by <= ~4x per sweep down there)
            long long F = 0;
            for (int s = 0; s < GWM; ++s) F += e->pinned[cur * GWM + s];
            n = 1;
            for (long long f = F / 4; f > push_thr && n < e->chunk_iters; f /= 4) ++n;
        } else {
            n = std::min(follow, e->chunk_iters);
            follow *= 2;
        }
        if (e->chunk_explicit) n = std::min(n, std::max(e->chunk_iters, 1));
        n = std::max(1, std::min(n, MAX_CHUNK));
        for (int k = 0; k < n; ++k) {
            const int nxt = (cur + 1) % 3, zer = (cur + 2) % 3;
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[2 * k], e->stream));
#define DPPR_LAUNCH_GSWEEP(SPL, GW, NVX) do { if (owed) DPPR_LAUNCH_GSWEEP_CM(SPL, GW, NVX, 1); else DPPR_LAUNCH_GSWEEP_CM(SPL, GW, NVX, 0); } while (0)
#define DPPR_LAUNCH_GSWEEP_CM(SPL, GW, NVX, CM)                                                                          \
    hipLaunchKernelGGL((k_gsweep<SPL, GW, NVX, false, CM>), dim3(sweep_grid), dim3(GNT), 0, e->stream, ep.grp_n_int, ep.gtab,  \
                       ep.n_ggroups, g.cnt + cur * GWM, e->gsweep_hot_rows, ep.out_col, g.x, g.x2, g.act[0], g.act[1], g.r, g.p,  \
                       g.cnt + nxt * GWM, g.cnt + zer * GWM, phase, eps, g.dstats + 1, log + k * GWM, 1, (GridBar *)nullptr,      \
                       (int *)nullptr, 0ull, 0, owed ? 1 : 0, g.gq + (g.gq_seq % 3) * GQ_PAD, g.gq + ((g.gq_seq + 1) % 3) * GQ_PAD)
            with_row(g.gw, [&](auto spl, auto gw) {
                constexpr int SPL = decltype(spl)::value, GW = decltype(gw)::value;
                if constexpr (SPL == 2) DPPR_LAUNCH_GSWEEP(2, GW, 512);
                else if (nvx == 512) DPPR_LAUNCH_GSWEEP(1, GW, 512);
                else DPPR_LAUNCH_GSWEEP(1, GW, 1024);
            });
#undef DPPR_LAUNCH_GSWEEP
#undef DPPR_LAUNCH_GSWEEP_CM
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[2 * k + 1], e->stream));
            std::swap(g.x, g.x2);
            std::swap(g.act[0], g.act[1]);
            cur = nxt;
            g.gq_seq++;
            owed = !owed; // (if the frontier emptied on the way, the later launches do nothing and nothing is owed: `more` is false below)
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(e->pinned, g.cnt, sizeof(int) * (size_t)(5 * GWM + n * GWM), hipMemcpyDeviceToHost,
                               e->stream));
        HIP_TRY(loop_wait(e));
        for (int k = 0; k < n; ++k) {
            const int *f = e->pinned + 5 * GWM + k * GWM;
            if (!any_left(f)) continue;
            if (push_thr > 0 && dense_len < 0) { // (the sweep that FOUND the frontier this small could have been a push iteration)
                long long F = 0;
                for (int s = 0; s < GWM; ++s) F += f[s];
                if (F <= push_thr) dense_len = it + k;
            }
            g.st.iterations++;
            g.st.pull_iterations++;
            for (int s = 0; s < GWM; ++s) g.st.sum_F += f[s];
            for (int s = 0; s < GWM; ++s) g.st.sweep_F += f[s];
            active_iters = it + k + 1;
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[2 * k], e->evpool[2 * k + 1]));
                g.st.push_ms += ms;
                g.st.push_launches++;
                g.st.sweep_ms += ms;
                g.st.sweep_launches++;
                static const bool trace = getenv("DPPR_GROUP_TRACE") != nullptr; // (diagnostic: one line per sweep)
                if (trace) {
                    long long F = 0;
                    for (int s = 0; s < GWM; ++s) F += f[s];
                    fprintf(stderr, "[gsweep] phase %d sweep %3d  frontier pairs %9lld  %7.1f us\n", phase, it + k, F, ms * 1e3);
                }
            }
        }
        more = any_left(e->pinned + cur * GWM);
        it += n;
        if (more && push_thr > 0 && !push_gave_up) {
            long long F = 0;
            for (int s = 0; s < GWM; ++s) F += e->pinned[cur * GWM + s];
            if (F <= push_thr) {
                if (dense_len < 0) dense_len = it;
                int pushed = 0;
                bool entered = false, conv = false;
                int rc = group_push_tail(e, g, ep, phase, eps, F, &pushed, &entered, &conv, &owed);
                if (rc) return rc;
                if (entered) {
                    active_iters = it + pushed;
                    it += pushed;
                    if (conv) more = false;
                    else { // back in sweep form: frontier sizes in row 0; the next try waits for a much smaller frontier
                        cur = 0;
                        HIP_TRY(hipMemcpyAsync(e->pinned, g.cnt, sizeof(int) * GWM, hipMemcpyDeviceToHost, e->stream));
                        HIP_TRY(loop_wait(e));
                        more = any_left(e->pinned);
                        push_thr = std::max<long long>(F / 8, 1);
                        dense_len = -1;
                    }
                } else { // (the frontier did not fit the lists)
                    push_thr = std::max<long long>(F / 8, 1);
                    dense_len = -1;
                }
            }
        }
    }
    if (push_thr > 0) {
        for (int k = 3; k > 0; --k) g.dense_hist[hp][k] = g.dense_hist[hp][k - 1];
        g.dense_hist[hp][0] = dense_len >= 0 ? std::max(dense_len, 1) : std::max(active_iters, 1);
    }
    g.iter_hint[hp] = active_iters;
    for (int k = 3; k > 0; --k) g.iter_hist[hp][k] = g.iter_hist[hp][k - 1];
    g.iter_hist[hp][0] = active_iters;
    return DPPR_OK;
}

int group_stream_update(dppr_engine *e, Group &g, const Epoch &ep) {
    const int L = ep.L;
    if (L == 0) return DPPR_OK;
    int rc = group_records_by_tail(e, ep, nullptr, 0, nullptr, 0);
    if (rc) return rc;
    SuSources srcs{};
    for (int s = 0; s < GS_MAX; ++s) srcs.s[s] = g.src.s[s];
    // blockIdx.y = source lane; state element (v, lane) at base[v * gw + lane]
    hipLaunchKernelGGL(k_su_terms, dim3(grid_for(L), g.n), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), ep.b2,
                       ep.ins, L, g.p, g.gw, e->su_term, e->su_ins);
    hipLaunchKernelGGL(k_su_apply, dim3(grid_for(L), g.n), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), e->su_term,
                       e->su_ins, ep.deg_after, L, g.r, g.gw, srcs, 0.0, (int *)nullptr, (int *)nullptr, (int *)nullptr,
                       (int *)nullptr);
    HIP_TRY(hipGetLastError());
    g.st.records += (int64_t)L * g.n;
    return DPPR_OK;
}

} // namespace

extern "C" {

int dppr_abi_version(void) { return DPPR_ABI_VERSION; }

const char *dppr_strerror(int status) {
    switch (status) {
    case DPPR_OK: return "ok";
    case DPPR_ERR_INVALID: return "invalid argument or call order";
    case DPPR_ERR_HIP: return "HIP runtime error";
    case DPPR_ERR_NOMEM: return "out of device memory";
    case DPPR_ERR_NO_DEVICE: return "no usable HIP device (the HIP path is mandatory; there is no CPU fallback)";
    case DPPR_ERR_NOT_CONVERGED: return "iteration cap hit before the frontier emptied";
    default: return "unknown status";
    }
}

const char *dppr_last_error(const dppr_engine *e) { return e ? e->err.c_str() : ""; }

int dppr_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int dppr_create(dppr_engine **out, int device, int32_t V, int32_t W, int directed, int32_t c, int32_t n_epochs) {
    if (!out || V <= 0 || W < 0 || c < 0 || n_epochs < 1) return DPPR_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    dppr_engine *e = new dppr_engine();
    dppr::g_live_engines.fetch_add(1, std::memory_order_relaxed);
    auto bail = [&](int code) {
        dppr_destroy(e);
        return code;
    };
#define HIP_TRY_C(call)                                                                 \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            fprintf(stderr, "dppr_create: %s at line %d\n", hipGetErrorString(_e), __LINE__); \
            return bail(_e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP);     \
        }                                                                               \
    } while (0)
    if (const char *v = getenv("DPPR_SWEEP_BITS")) e->sweep_bits = atoi(v) != 0; // diagnostic A/B switches
    if (const char *v = getenv("DPPR_HOT_BLOCKS")) e->hot_blocks = atoi(v) != 0;
    if (const char *v = getenv("DPPR_GSWEEP_HOT")) e->gsweep_hot_rows = std::max(0, atoi(v));
    if (const char *v = getenv("DPPR_GSWEEP_GRID")) e->gsweep_grid_cap = std::max(1, std::min(atoi(v), STAT_SLOTS));
    if (const char *v = getenv("DPPR_RENUMBER")) e->renumber_on = atoi(v) != 0;
    if (const char *v = getenv("DPPR_RENUMBER_PCT")) e->renumber_growth_pct = std::max(1, atoi(v));
    if (const char *v = getenv("DPPR_RENUMBER_MIN")) e->renumber_min_parked = std::max(1, atoi(v));
    if (const char *v = getenv("DPPR_GROUP_PUSH")) e->gpush_enter_pairs = std::max(-1, atoi(v));
    if (const char *v = getenv("DPPR_GROUP_PUSH_FACTOR")) e->gpush_auto_factor = std::max(1, atoi(v));
    if (const char *v = getenv("DPPR_GGROUPS_MIN")) e->ggroups_min = std::max(1, atoi(v));
    if (const char *v = getenv("DPPR_COST_MODEL")) e->cost_model = atoi(v) != 0;
    if (const char *v = getenv("DPPR_GROUPING_RADIX")) e->force_radix_grouping = atoi(v) != 0;
    if (const char *v = getenv("DPPR_TEST_MERGE_MISS")) e->test_force_merge_miss = atoi(v) != 0;
    if (const char *v = getenv("DPPR_GROUP_AT_SLIDE")) e->group_at_slide = atoi(v) != 0;
    if (const char *v = getenv("DPPR_GROUP_FULL_ROWS")) e->group_full_rows = atoi(v) != 0;
    e->device = device;
    e->V = V;
    e->W = W;
    e->c = c;
    e->directed = directed ? 1 : 0;
    e->n_epochs = n_epochs;
    e->Ed = directed ? W : 2 * W;
    e->bits = 1;
    while ((1ll << e->bits) < (long long)V) e->bits++;
    HIP_TRY_C(hipSetDevice(device));
    HIP_TRY_C(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    {
        int least = 0, greatest = 0; // (numerically: least priority >= greatest priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        HIP_TRY_C(hipStreamCreateWithPriority(&e->bs, hipStreamNonBlocking, least));
    }
    HIP_TRY_C(hipEventCreate(&e->ev0));
    HIP_TRY_C(hipEventCreate(&e->ev1));
    for (auto &ev : e->evpool) HIP_TRY_C(hipEventCreate(&ev));
    HIP_TRY_C(hipHostMalloc((void **)&e->dump_pin, dppr_engine::DUMP_PIN_BYTES, hipHostMallocDefault));
    HIP_TRY_C(hipHostMalloc((void **)&e->pinned, sizeof(int) * ((GMULTI_MAX + 2) * GS_MAX + 3 * GS_MAX + MAX_CHUNK * GS_MAX + 16), hipHostMallocDefault));
    const size_t Wn = (size_t)std::max(W, 1), Edn = (size_t)std::max(e->Ed, 1), Ln = (size_t)std::max(4 * c, 1);
    HIP_TRY_C(hipMalloc((void **)&e->w1, sizeof(int) * Wn));
    HIP_TRY_C(hipMalloc((void **)&e->w2, sizeof(int) * Wn));
    HIP_TRY_C(hipMalloc((void **)&e->outdeg, sizeof(int) * (size_t)V));
    // every memset / copy of the engine goes on ITS stream: the stream is non-blocking, so work on
    // the null stream (plain hipMemset / hipMemcpy) is not ordered with it
    HIP_TRY_C(hipMemsetAsync(e->outdeg, 0, sizeof(int) * (size_t)V, e->stream));
    HIP_TRY_C(hipMalloc((void **)&e->hub_slot_of, sizeof(int) * (size_t)V));
    HIP_TRY_C(hipMalloc((void **)&e->d_ext2int, sizeof(int) * (size_t)V));
    HIP_TRY_C(hipMalloc((void **)&e->d_xfer, sizeof(double) * (size_t)V));
    e->init_ids(V);
    HIP_TRY_C(hipMalloc((void **)&e->hub_hist, sizeof(int) * 64));
    HIP_TRY_C(hipMalloc((void **)&e->bar, sizeof(GridBar)));
    HIP_TRY_C(hipMalloc((void **)&e->keys_a, sizeof(uint64_t) * Edn));
    HIP_TRY_C(hipMalloc((void **)&e->keys_b, sizeof(uint64_t) * Edn));
    HIP_TRY_C(hipMalloc((void **)&e->in_sorted, sizeof(uint64_t) * Edn));
    if (e->directed) HIP_TRY_C(hipMalloc((void **)&e->out_sorted, sizeof(uint64_t) * Edn));
    HIP_TRY_C(hipMalloc((void **)&e->delpos, sizeof(int) * ((size_t)2 * (size_t)std::max(c, 1) + 16)));
    {
        const size_t bn = (size_t)std::max(2 * c, 1);
        for (int k = 0; k < 4; ++k) {
            HIP_TRY_C(hipMalloc((void **)&e->bk[k], sizeof(uint64_t) * bn));
            HIP_TRY_C(hipMalloc((void **)&e->bks[k], sizeof(uint64_t) * bn));
        }
    }
    HIP_TRY_C(rocprim::radix_sort_keys(nullptr, e->sort_tmp_bytes, e->keys_a, e->keys_b, Edn, 0u,
                                       (unsigned)(2 * e->bits), e->stream));
    HIP_TRY_C(hipMalloc(&e->sort_tmp, std::max<size_t>(e->sort_tmp_bytes, 16)));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY_C(hipMalloc((void **)&e->su_k[k], sizeof(uint32_t) * Ln));
        HIP_TRY_C(hipMalloc((void **)&e->su_v[k], sizeof(uint32_t) * Ln));
    }
    HIP_TRY_C(hipMalloc((void **)&e->su_term, sizeof(double) * Ln * GS_MAX)); // one term array per source lane of a group
    HIP_TRY_C(hipMalloc((void **)&e->su_ins, Ln));
    HIP_TRY_C(rocprim::radix_sort_pairs(nullptr, e->su_tmp_bytes, e->su_k[0], e->su_k[1], e->su_v[0], e->su_v[1], Ln,
                                        0u, (unsigned)e->bits, e->stream));
    HIP_TRY_C(hipMalloc(&e->su_tmp, std::max<size_t>(e->su_tmp_bytes, 16)));
    e->epochs.resize((size_t)n_epochs);
    for (auto &ep : e->epochs) {
        HIP_TRY_C(hipMalloc((void **)&ep.row_ptr, sizeof(int) * ((size_t)V + 1)));
        HIP_TRY_C(hipMalloc((void **)&ep.adj, sizeof(Adj) * Edn));
        HIP_TRY_C(hipMalloc((void **)&ep.out_row_ptr, sizeof(int) * ((size_t)V + 1)));
        HIP_TRY_C(hipMalloc((void **)&ep.out_col, sizeof(int) * Edn));
        HIP_TRY_C(hipMalloc((void **)&ep.b1, sizeof(int) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.b2, sizeof(int) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.deg_after, sizeof(int) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.ins, Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.sk, sizeof(uint32_t) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.sv, sizeof(uint32_t) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.grp_tile, sizeof(int) * ((size_t)V / WAVE + 3)));
        HIP_TRY_C(hipMalloc((void **)&ep.ggrp_tile, sizeof(int) * ((size_t)V / WAVE + 3)));
        HIP_TRY_C(hipMalloc((void **)&ep.hub_v, sizeof(int) * HUB_CAP));
        HIP_TRY_C(hipMalloc((void **)&ep.hub_degp1, sizeof(int) * HUB_CAP));
    }
    HIP_TRY_C(hipStreamSynchronize(e->stream)); // (the builder's stream is another one: nothing of this set-up may still be in flight)
#undef HIP_TRY_C
    *out = e;
    return DPPR_OK;
}

void dppr_destroy(dppr_engine *e) {
    if (!e) return;
    if (e->pre.task.valid()) e->pre.task.wait();
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->bs) (void)hipStreamSynchronize(e->bs);
    for (auto &s : e->slots) {
        (void)hipFree(s.p); (void)hipFree(s.r); (void)hipFree(s.x); (void)hipFree(s.x2);
        (void)hipFree(s.ft[0]); (void)hipFree(s.ft[1]); (void)hipFree(s.neg); (void)hipFree(s.status); (void)hipFree(s.act[0]); (void)hipFree(s.act[1]);
        (void)hipFree(s.cnt); (void)hipFree(s.dstats); (void)hipFree(s.big);
    }
    for (auto &g : e->groups) {
        (void)hipFree(g.p); (void)hipFree(g.r); (void)hipFree(g.x); (void)hipFree(g.x2);
        (void)hipFree(g.act[0]); (void)hipFree(g.act[1]);
        (void)hipFree(g.cnt); (void)hipFree(g.mlog); (void)hipFree(g.dstats); (void)hipFree(g.gq);
    }
    for (auto &ep : e->epochs) {
        (void)hipFree(ep.row_ptr); (void)hipFree(ep.adj); (void)hipFree(ep.out_row_ptr); (void)hipFree(ep.out_col); (void)hipFree(ep.b1); (void)hipFree(ep.b2);
        (void)hipFree(ep.deg_after); (void)hipFree(ep.ins); (void)hipFree(ep.sk); (void)hipFree(ep.sv); (void)hipFree(ep.hub_v); (void)hipFree(ep.hub_degp1); (void)hipFree(ep.grp_tile); (void)hipFree(ep.ggrp_tile); (void)hipFree(ep.gtab);
        (void)hipFree(ep.acut); (void)hipFree(ep.chunks); (void)hipFree(ep.hl); (void)hipFree(ep.dl); (void)hipFree(ep.apos);
        (void)hipFree(ep.res_pk); (void)hipFree(ep.su_rng);
    }
    (void)hipFree(e->bin_vblk_b); (void)hipFree(e->bin_small); (void)hipFree(e->bin_vals); (void)hipFree(e->bin_tmp);
    (void)hipFree(e->w1); (void)hipFree(e->w2); (void)hipFree(e->outdeg);
    (void)hipFree(e->bar);
    (void)hipFree(e->res_arena);
    (void)hipFree(e->hub_slot_of); (void)hipFree(e->hub_hist); (void)hipFree(e->d_ext2int); (void)hipFree(e->d_xfer);
    (void)hipFree(e->mv_idx); (void)hipFree(e->mv_tmp);
    (void)hipFree(e->keys_a); (void)hipFree(e->keys_b); (void)hipFree(e->sort_tmp);
    (void)hipFree(e->in_sorted); (void)hipFree(e->out_sorted); (void)hipFree(e->delpos);
    for (int k = 0; k < 4; ++k) { (void)hipFree(e->bk[k]); (void)hipFree(e->bks[k]); }
    for (int k = 0; k < 2; ++k) { (void)hipFree(e->su_k[k]); (void)hipFree(e->su_v[k]); }
    (void)hipFree(e->su_term); (void)hipFree(e->su_ins); (void)hipFree(e->su_tmp);
    if (e->pinned) (void)hipHostFree(e->pinned);
    if (e->dump_pin) (void)hipHostFree(e->dump_pin);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    for (auto &ev : e->evpool)
        if (ev) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->bs) (void)hipStreamDestroy(e->bs);
    dppr::g_live_engines.fetch_sub(1, std::memory_order_relaxed);
    delete e;
}

int dppr_set_schedule(dppr_engine *e, int schedule) {
    if (!e || (schedule != DPPR_SCHEDULE_EAGER && schedule != DPPR_SCHEDULE_SYNC)) return DPPR_ERR_INVALID;
    e->schedule = schedule;
    return DPPR_OK;
}

int dppr_set_batch_grouping(dppr_engine *e, int at_slide) {
    if (!e) return DPPR_ERR_INVALID;
    e->group_at_slide = at_slide != 0; // (applies to the epochs dppr_slide builds from now on)
    return DPPR_OK;
}

int dppr_set_variant(dppr_engine *e, int variant) {
    if (!e || variant < 0 || variant > 3) return fail(e, DPPR_ERR_INVALID, "set_variant: 0 OPTIMIZED, 1 FAST_FRONTIER, 2 EAGER, 3 VANILLA");
    e->schedule = (variant == 1 || variant == 3) ? DPPR_SCHEDULE_SYNC : DPPR_SCHEDULE_EAGER; // pre-extracted residuals = the synchronous schedule
    e->pre_extract = variant == 1 || variant == 3;
    e->status_dedup = variant == 2 || variant == 3;
    return DPPR_OK;
}

int dppr_set_phase_merge(dppr_engine *e, int on, int eps_divisor) {
    if (!e || eps_divisor < 0 || eps_divisor > 1024) return fail(e, DPPR_ERR_INVALID, "set_phase_merge: eps_divisor 1..1024 (0 keeps it)");
    e->merge_phases = on != 0;
    if (eps_divisor > 0) e->merge_div = eps_divisor;
    return DPPR_OK;
}

int dppr_set_profiling(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->profiling = on != 0;
    return DPPR_OK;
}

int dppr_set_persistent(dppr_engine *e, int mode, int64_t timeout_us) {
    if (!e || mode < 0 || mode > 2 || e->loaded || !e->slots.empty())
        return fail(e, DPPR_ERR_INVALID, "set_persistent: call right after dppr_create, mode 0, 1 or 2");
    e->persist_mode = mode;
    if (timeout_us > 0) e->persist_ticks = (unsigned long long)timeout_us * 100ull; // wall_clock64 runs at 100 MHz
    if (timeout_us < 0) { // tests: a roll-call that cannot succeed, given up after 200 us
        e->persist_ticks = 20000ull;
        e->persist_rollcall_extra = 1;
    }
    return DPPR_OK;
}

int dppr_set_tuning(dppr_engine *e, int hub_min_degree, int big_row_edges, int pull_min_frontier, int chunk_iters,
                    int pull_block) {
    if (!e || hub_min_degree < 1 || big_row_edges < 1 || e->loaded || !e->slots.empty())
        return fail(e, DPPR_ERR_INVALID, "set_tuning: call right after dppr_create, values >= 1");
    e->hub_min_degree = hub_min_degree;
    e->big_row = big_row_edges;
    e->pull_min_frontier = pull_min_frontier;
    if (chunk_iters > 0) {
        e->chunk_iters = std::min(chunk_iters, MAX_CHUNK);
        e->chunk_explicit = true;
    }
    if (pull_block >= 256 && pull_block <= 1024 && pull_block % 64 == 0) e->pull_block = pull_block;
    return DPPR_OK;
}

int dppr_set_sweep_bitmap(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->sweep_bits = on != 0;
    return DPPR_OK;
}

int dppr_set_binned_sweep(dppr_engine *e, int mode, int ha_tiles, int hb_tiles, int64_t target_edges, int64_t min_ids, int64_t chunk_edges,
                          int64_t target_a_edges) {
    if (!e || mode < 0 || mode > 2 || ha_tiles < 0 || ha_tiles > BIN_MAX_HA_TILES || hb_tiles < 0 || hb_tiles > BIN_MAX_HB_TILES || target_edges < 0 || min_ids < 0 ||
        chunk_edges < 0 || target_a_edges < 0 || e->bin_ready || e->loaded)
        return fail(e, DPPR_ERR_INVALID, "set_binned_sweep: call right after dppr_create; mode 0..2, ha_tiles <= 272, hb_tiles <= 120");
    e->bin_mode = mode;
    if (ha_tiles > 0) e->bin_ha_tiles = ha_tiles;
    if (hb_tiles > 0) e->bin_hb_tiles = hb_tiles;
    if (target_edges > 0) e->bin_target = target_edges;
    if (min_ids > 0) e->bin_min_ids = min_ids;
    if (chunk_edges > 0) e->bin_chunk = chunk_edges;
    if (target_a_edges > 0) e->bin_target_a = target_a_edges;
    return DPPR_OK;
}

int dppr_set_resident_slots(dppr_engine *e, int sorted) {
    if (!e) return DPPR_ERR_INVALID;
    e->res_slots = sorted != 0;
    // epochs already cut keep their tables until the next cut; switching OFF takes effect at once
    if (!e->res_slots)
        for (auto &ep : e->epochs) ep.res_valid = false;
    return DPPR_OK;
}

int dppr_set_resident_update(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->res_update = on != 0;
    if (!e->res_update)
        for (auto &ep : e->epochs) ep.su_inline = false;
    return DPPR_OK;
}

int dppr_set_group_resident(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->group_resident = on != 0;
    return DPPR_OK;
}

int dppr_set_group_push(dppr_engine *e, int enter_pairs, int list_cap, int64_t max_edges) {
    if (!e || enter_pairs < -1 || list_cap < 0 || max_edges < 0) return fail(e, DPPR_ERR_INVALID, "set_group_push: bad argument");
    e->gpush_enter_pairs = enter_pairs;
    if (list_cap > 0) e->gpush_list_cap = list_cap;
    e->gpush_max_edges = max_edges;
    return DPPR_OK;
}

int dppr_set_group_seeding(dppr_engine *e, int from_tails) {
    if (!e) return DPPR_ERR_INVALID;
    e->group_tail_seeding = from_tails != 0;
    return DPPR_OK;
}

int dppr_set_renumbering(dppr_engine *e, int on, int growth_pct, int min_parked) {
    if (!e || growth_pct < 0 || min_parked < 0) return fail(e, DPPR_ERR_INVALID, "set_renumbering: bad argument");
    e->renumber_on = on != 0;
    if (growth_pct > 0) e->renumber_growth_pct = growth_pct;
    if (min_parked > 0) e->renumber_min_parked = min_parked;
    e->renumber_next = std::min(e->renumber_next, renumber_threshold(e->n_int, e->renumber_growth_pct));
    return DPPR_OK;
}

int dppr_id_space(dppr_engine *e, int32_t *n_ids, int32_t *n_parked, int32_t *renumberings, int64_t *revivals) {
    if (!e) return DPPR_ERR_INVALID;
    if (n_ids) *n_ids = e->n_int;
    if (n_parked) *n_parked = e->n_parked;
    if (renumberings) *renumberings = e->renumberings;
    if (revivals) *revivals = e->revivals;
    return DPPR_OK;
}

int dppr_set_incremental_graph(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->incremental = on != 0;
    return DPPR_OK;
}

int dppr_synchronize(dppr_engine *e) {
    if (!e) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_load_window(dppr_engine *e, const int32_t *e1, const int32_t *e2, int32_t n) {
    if (!e || n != e->W || (n > 0 && (!e1 || !e2))) return fail(e, DPPR_ERR_INVALID, "load_window: n must equal W");
    HIP_TRY(hipSetDevice(e->device));
    {
        // Number the window's vertices in a pseudo-random order (hash of the external id), not by
        // first appearance: high-degree vertices show up early in a stream, and packing them into
        // the first tiles would serialise the sweeps on a few workgroups.
        std::vector<std::pair<uint64_t, int32_t>> fresh;
        for (int k = 0; k < 2; ++k) {
            const int32_t *a = k ? e2 : e1;
            for (int i = 0; i < n; ++i) {
                const int v = a[i];
                if (v < 0 || v >= e->V) return fail(e, DPPR_ERR_INVALID, "load_window: vertex id out of range");
                if (e->ext2int[(size_t)v] == -1) {
                    e->ext2int[(size_t)v] = -2; // seen, not numbered yet
                    fresh.emplace_back(id_hash(v), v);
                }
            }
        }
        std::vector<int32_t> indeg;
        if (fresh.size() > HOT_WINDOW_MIN) {
            indeg.assign((size_t)e->V, 0);
            for (int i = 0; i < n; ++i) {
                indeg[(size_t)e2[i]]++;
                if (!e->directed) indeg[(size_t)e1[i]]++;
            }
        }
        numbering_order(fresh, indeg.empty() ? nullptr : indeg.data(), e->hot_blocks);
        for (auto &kv : fresh) {
            e->ext2int[(size_t)kv.second] = -1;
            (void)to_int(e, kv.second);
        }
    }
    if (!translate(e, e1, n, e->h_tmp1) || !translate(e, e2, n, e->h_tmp2))
        return fail(e, DPPR_ERR_INVALID, "load_window: vertex id out of range");
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(e->w1, e->h_tmp1.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
        HIP_TRY(hipMemcpyAsync(e->w2, e->h_tmp2.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
    }
    e->head = 0;
    HIP_TRY(hipMemsetAsync(e->outdeg, 0, sizeof(int) * (size_t)e->V, e->bs));
    if (n > 0) {
        hipLaunchKernelGGL(k_deg_update, dim3(grid_for(n)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, n, e->directed, 1,
                           e->outdeg);
        HIP_TRY(hipGetLastError());
    }
    for (auto &ep : e->epochs) ep.id = -1;
    Epoch &ep = e->epochs[0];
    ep.L = 0;
    ep.grouped = false;
    ep.su_inline = false;
    int rc = query_persist_cap(e);
    if (rc) return rc;
    rc = sort_window_full(e);
    if (rc) return rc;
    rc = build_epoch(e, ep);
    if (rc) return rc;
    ep.id = 0;
    e->newest = 0;
    e->renumber_next = renumber_threshold(e->n_int, e->renumber_growth_pct);
    e->loaded = true;
    e->batch_staged = false;
    HIP_TRY(hipStreamSynchronize(e->bs));
    return DPPR_OK;
}

int dppr_hint_next_batch(dppr_engine *e, const int32_t *b1, const int32_t *b2, int32_t L, const int32_t *n1, const int32_t *n2, int32_t c) {
    // (no message through fail(): this call may come from a helper thread while the engine's own thread writes e->err -- ADVICE r04)
    if (!e || e->broken || L < 0 || L > 4 * e->c || c < 0 || c > e->c || (L > 0 && (!b1 || !b2)) || (c > 0 && (!n1 || !n2))) return DPPR_ERR_INVALID;
    pre_join(e);
    dppr_engine::Pre &pr = e->pre;
    const int32_t *src[4] = {b1, b2, n1, n2};
    const int n[4] = {L, L, c, c};
    for (int k = 0; k < 4; ++k) {
        pr.src[k] = src[k];
        pr.n[k] = n[k];
        pr.out[k].resize((size_t)std::max(n[k], 1));
    }
    pr.epoch = e->renumber_epoch;
    pr.armed = true;
    pr.ok = false;
    pr.task = std::async(std::launch::async, [e] {
        bool ok = true;
        for (int k = 0; k < 4; ++k)
            if (e->pre.n[k] > 0) ok = e->lookup_only(e->pre.src[k], (size_t)e->pre.n[k], e->pre.out[k].data(), e->pre.miss[k]) && ok;
        return ok;
    });
    return DPPR_OK;
}

int dppr_set_batch(dppr_engine *e, const int32_t *b1, const int32_t *b2, const uint8_t *ins, int32_t L) {
    if (!e || e->broken || L < 0 || L > 4 * e->c || (L > 0 && (!b1 || !b2 || !ins)))
        return fail(e, DPPR_ERR_INVALID, "set_batch: length exceeds 4*max_batch");
    if (!ids_in_range(e, b1, L) || !ids_in_range(e, b2, L) || !translate(e, b1, L, e->st_b1) || !translate(e, b2, L, e->st_b2))
        return fail(e, DPPR_ERR_INVALID, "set_batch: vertex id out of range");
    e->st_b1.resize((size_t)L);
    e->st_b2.resize((size_t)L);
    e->st_ins.assign(ins, ins + L);
    e->batch_staged = true;
    HIP_TRY(hipSetDevice(e->device));
    return flush_moves(e); // (a record may have named a parked vertex)
}

static int slide_impl(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch);

int dppr_slide(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch) {
    if (e) e->build_concurrent = false;
    return slide_impl(e, n1, n2, c, out_epoch);
}

int dppr_slide_concurrent(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch) {
    if (!e || e->n_epochs < 2) return fail(e, DPPR_ERR_INVALID, "slide_concurrent: needs n_epochs >= 2 (the epoch being built must not be the one being solved)");
    e->build_concurrent = true;
    const int rc = slide_impl(e, n1, n2, c, out_epoch);
    e->build_concurrent = false;
    return rc;
}

int dppr_renumbering_due(const dppr_engine *e) {
    return e && e->renumber_on && e->W > 0 && e->n_int >= e->renumber_next ? 1 : 0;
}

static int slide_impl(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch) {
    // c is bounded by max_batch of dppr_create: the batch key buffers (2 * max_batch keys each) and the
    // merge scratch are sized for it
    if (!e || e->broken || !e->loaded || c < 0 || c > e->W || c > e->c || (c > 0 && (!n1 || !n2)))
        return fail(e, DPPR_ERR_INVALID, "slide: window not loaded, or c exceeds the window / max_batch of dppr_create");
    if (!ids_in_range(e, n1, c) || !ids_in_range(e, n2, c)) // before anything is touched: a rejected slide is a no-op
        return fail(e, DPPR_ERR_INVALID, "slide: vertex id out of range");
    HIP_TRY(hipSetDevice(e->device));
    const int W = e->W;
    static const bool slide_trace = getenv("DPPR_SLIDE_TRACE") != nullptr; // (diagnostic: phases of a slide; each mark synchronises)
    timespec t_mark;
    clock_gettime(CLOCK_MONOTONIC, &t_mark);
    auto mark = [&](const char *what) {
        if (!slide_trace) return;
        (void)hipStreamSynchronize(e->bs);
        timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        fprintf(stderr, "[slide] %-34s %8.1f us\n", what, (now.tv_sec - t_mark.tv_sec) * 1e6 + (now.tv_nsec - t_mark.tv_nsec) * 1e-3);
        t_mark = now;
    };
    bool renumbered = false;
    if (int rc = compact_ids(e, &renumbered)) return rc;
    mark("renumbering check");
    if (!translate(e, n1, c, e->h_tmp1) || !translate(e, n2, c, e->h_tmp2))
        return fail(e, DPPR_ERR_INVALID, "slide: vertex id out of range");
    if (int rc = flush_moves(e)) return rc;
    mark("translate new edges");
    n1 = e->h_tmp1.data();
    n2 = e->h_tmp2.data();
    // the c oldest edges sit at ring positions head .. head+c (mod W): retire their degrees (and
    // note their keys), overwrite them with the new edges, add the new degrees (and note those keys)
    const bool inc = e->incremental && c > 0 && 2 * c <= e->Ed && !renumbered; // (renumbered: the sorted keys are stale)
    const int per = e->directed ? 1 : 2; // keys per stream edge in the in-orientation array
    int done = 0;
    while (done < c) {
        const int pos = (e->head + done) % W;
        const int len = std::min(c - done, W - pos);
        hipLaunchKernelGGL(k_deg_update, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos, len,
                           e->directed, -1, e->outdeg);
        if (inc)
            hipLaunchKernelGGL(k_make_keys_seg, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos,
                               len, e->directed, e->bits, e->bk[0] + (size_t)done * per, e->bk[2] + done);
        HIP_TRY(hipMemcpyAsync(e->w1 + pos, n1 + done, sizeof(int) * (size_t)len, hipMemcpyHostToDevice, e->bs));
        HIP_TRY(hipMemcpyAsync(e->w2 + pos, n2 + done, sizeof(int) * (size_t)len, hipMemcpyHostToDevice, e->bs));
        hipLaunchKernelGGL(k_deg_update, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos, len,
                           e->directed, 1, e->outdeg);
        if (inc)
            hipLaunchKernelGGL(k_make_keys_seg, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos,
                               len, e->directed, e->bits, e->bk[1] + (size_t)done * per, e->bk[3] + done);
        HIP_TRY(hipGetLastError());
        done += len;
    }
    if (W > 0) e->head = (e->head + c) % W;
    mark("ring, degrees, batch keys");
    const int id = e->newest + 1;
    Epoch &ep = e->epochs[id % e->n_epochs];
    ep.id = -1;
    ep.L = 0;            // (before build_epoch: its group cut looks at the epoch's records, and the ring entry still holds
    ep.grouped = false;  //  the previous occupant's -- possibly in an older numbering)
    ep.su_inline = false;
    int rc;
    e->merge_miss_host = 0;
    if (inc) { // f1: merge the batch into the previous sorted keys
        HIP_TRY(hipMemsetAsync(e->hub_hist + MERGE_MISS_WORD, 0, sizeof(int), e->bs));
        rc = merge_batch_keys(e, e->in_sorted, e->bk[0], e->bks[0], c * per, e->bk[1], e->bks[1], c * per);
        if (!rc && e->directed) rc = merge_batch_keys(e, e->out_sorted, e->bk[2], e->bks[2], c, e->bk[3], e->bks[3], c);
        // (read with the build's own synchronisations below: no extra wait on the path that finds every key)
        if (!rc) HIP_TRY(hipMemcpyAsync(&e->merge_miss_host, e->hub_hist + MERGE_MISS_WORD, sizeof(int), hipMemcpyDeviceToHost, e->bs));
    } else {
        rc = sort_window_full(e);
    }
    if (rc) return rc;
    mark("sorted keys (merge / full sort)");
    rc = build_epoch(e, ep);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->bs));
    if (inc && (e->merge_miss_host != 0 || e->test_force_merge_miss)) {
        // a retired key was not among the kept sorted keys (an inconsistent window: never seen): the merged arrays
        // cannot be trusted -- the ring itself is right, so sort it afresh and build the epoch again
        e->merge_fallbacks++;
        rc = sort_window_full(e);
        if (rc) return rc;
        rc = build_epoch(e, ep);
        if (rc) return rc;
    }
    mark("hubs, CSRs, group cut + tables");
    ep.L = 0;
    ep.grouped = false;
    if (e->batch_staged) {
        const int L = (int)e->st_b1.size();
        ep.L = L;
        if (L > 0) {
            HIP_TRY(hipMemcpyAsync(ep.b1, e->st_b1.data(), sizeof(int) * (size_t)L, hipMemcpyHostToDevice, e->bs));
            HIP_TRY(hipMemcpyAsync(ep.b2, e->st_b2.data(), sizeof(int) * (size_t)L, hipMemcpyHostToDevice, e->bs));
            HIP_TRY(hipMemcpyAsync(ep.ins, e->st_ins.data(), (size_t)L, hipMemcpyHostToDevice, e->bs));
            if (e->group_at_slide) {
                // CopyOutDegree (gpu/StreamUpdate.cuh:7-17): post-batch out-degree of every tail
                hipLaunchKernelGGL(k_gather_deg, dim3(grid_for(L)), dim3(BLOCK), 0, e->bs, ep.b1, L, e->outdeg,
                                   ep.deg_after);
                HIP_TRY(hipGetLastError());
                if (int grc = epoch_group_records(e, ep)) return grc; // the records grouped by tail, for IncrementalBatchUpdate
            } // (default: both are part of the timed region -- group_records_by_tail, or the resident launch itself)
        }
    }
    HIP_TRY(hipStreamSynchronize(e->bs)); // staged host vectors may be reused now
    mark("batch records");
    e->batch_staged = false;
    ep.id = id;
    e->newest = id;
    if (out_epoch) *out_epoch = id;
    return DPPR_OK;
}

int dppr_add_source(dppr_engine *e, int32_t source, int32_t *out_slot) {
    if (!e || e->broken || source < 0 || source >= e->V) return fail(e, DPPR_ERR_INVALID, "add_source: vertex out of range");
    HIP_TRY(hipSetDevice(e->device));
    Slot s;
    s.source_ext = source;
    s.source = to_int(e, source);
    source = s.source;
    if (int rc = flush_moves(e)) return rc;
    const size_t V = (size_t)e->V;
    HIP_TRY(hipMalloc((void **)&s.p, sizeof(double) * V));
    HIP_TRY(hipMalloc((void **)&s.r, sizeof(double) * V));
    HIP_TRY(hipMalloc((void **)&s.x, sizeof(double) * V));
    HIP_TRY(hipMalloc((void **)&s.x2, sizeof(double) * V));
    HIP_TRY(hipMemsetAsync(s.x, 0, sizeof(double) * V, e->stream));
    HIP_TRY(hipMemsetAsync(s.x2, 0, sizeof(double) * V, e->stream));
    s.act_bytes = (V / 32 + 64) * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void **)&s.act[0], s.act_bytes));
    HIP_TRY(hipMalloc((void **)&s.act[1], s.act_bytes));
    HIP_TRY(hipMemsetAsync(s.act[0], 0, s.act_bytes, e->stream));
    HIP_TRY(hipMemsetAsync(s.act[1], 0, s.act_bytes, e->stream));
    HIP_TRY(hipMalloc((void **)&s.ft[0], sizeof(int) * V));
    HIP_TRY(hipMalloc((void **)&s.ft[1], sizeof(int) * V));
    HIP_TRY(hipMalloc((void **)&s.neg, sizeof(int) * (size_t)std::max(4 * e->c, 1)));
    HIP_TRY(hipMalloc((void **)&s.cnt, sizeof(int) * (CNT_HDR + 2 * MAX_CHUNK)));
    s.log = s.cnt + CNT_HDR; // the per-chunk log sits right behind the counters: one read-back fetches both
    // a row is deferred only if it has >= big_row edges, so at most Ed / big_row of them exist
    // (pieces of <= 1024 edges of rows of >= big_row edges: a row of d edges has ceil(d / 1024) <= d / min(big_row, 512) of them)
    HIP_TRY(hipMalloc((void **)&s.big, sizeof(BigItem) * ((size_t)e->Ed / (size_t)std::min(std::max(e->big_row, 1), 512) + 64)));
    HIP_TRY(hipMalloc((void **)&s.dstats, 2 * sizeof(IterStats)));
    HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * (CNT_HDR + 2 * MAX_CHUNK), e->stream));
    HIP_TRY(hipMemsetAsync(s.dstats, 0, 2 * sizeof(IterStats), e->stream));
    hipLaunchKernelGGL(k_init, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, s.p, s.r, e->V, source);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->slots.push_back(std::move(s));
    if (int rc = recut_stale_groups(e)) return rc; // the source may have received a fresh internal id
    if (Epoch *nw = find_epoch(e, -1)) // a window that sweeps binned: the newest epoch's tables, if it was built before any slot existed
        if (!nw->bin_valid && bin_wanted(e))
            if (int rc = build_bins(e, *nw)) return rc;
    if (out_slot) *out_slot = (int)e->slots.size() - 1;
    return DPPR_OK;
}

#define GET_SLOT(e, slot)                                                                       \
    if (!(e) || (slot) < 0 || (slot) >= (int)(e)->slots.size()) return fail((e), DPPR_ERR_INVALID, "bad slot"); \
    if ((e)->broken) return fail((e), DPPR_ERR_INVALID, "engine unusable after a failed renumbering");            \
    Slot &s = (e)->slots[(size_t)(slot)]
#define GET_EPOCH(e, epoch)                                                       \
    Epoch *epp = find_epoch((e), (epoch));                                        \
    if (!epp) return fail((e), DPPR_ERR_INVALID, "epoch not resident (evicted or never built)"); \
    Epoch &ep = *epp

int dppr_time_batch_grouping(dppr_engine *e, int32_t epoch, int32_t reps, float *out_ms) {
    if (!e || e->broken || reps < 1 || !out_ms) return fail(e, DPPR_ERR_INVALID, "time_batch_grouping: reps >= 1");
    GET_EPOCH(e, epoch);
    HIP_TRY(hipSetDevice(e->device));
    *out_ms = 0.0f;
    const int L = ep.L;
    if (L <= 0) return DPPR_OK;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    int *deg_scratch = reinterpret_cast<int *>(e->su_term);
    for (int k = 0; k < reps; ++k) {
        // CopyOutDegree (gpu/StreamUpdate.cuh:7-17) + the stable grouping by tail, into scratch, as group_records_by_tail runs them
        // inside the timed region (one ranking launch up to SU_RANK_MAX records; degree gather + keys + radix sort beyond)
        if (L <= SU_RANK_MAX && !e->force_radix_grouping) {
            hipLaunchKernelGGL(k_su_group_rank, dim3((L + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, e->stream, ep.b1, L, ep.out_row_ptr, e->su_k[1],
                               e->su_v[1], deg_scratch, (unsigned long long *)nullptr, 0, (int *)nullptr, 0);
            continue;
        }
        hipLaunchKernelGGL(k_copy_out_degree, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, ep.b1, L, ep.out_row_ptr, deg_scratch);
        hipLaunchKernelGGL(k_su_keys, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, ep.b1, L, e->su_k[0], e->su_v[0],
                           (unsigned long long *)nullptr, 0, (int *)nullptr, 0);
        size_t tmp = e->su_tmp_bytes;
        HIP_TRY(rocprim::radix_sort_pairs(e->su_tmp, tmp, e->su_k[0], e->su_k[1], e->su_v[0], e->su_v[1], (size_t)L, 0u, (unsigned)e->bits, e->stream));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *out_ms = ms / (float)reps;
    return DPPR_OK;
}

int dppr_debug_dump(dppr_engine *e, char *buf, int32_t cap) {
    if (!e || !buf || cap < 2) return 0;
    std::string o;
    char line[512];
#define add(...)                                  \
    do {                                          \
        snprintf(line, sizeof(line), __VA_ARGS__); \
        o += line;                                \
    } while (0)
    add("dppr engine %p device %d: V %d W %d c %d directed %d n_int %d newest epoch %d broken %d\n", (void *)e, e->device, e->V, e->W, e->c,
        e->directed, e->n_int, e->newest, (int)e->broken);
    add("last error: %s\n", e->err.empty() ? "(none)" : e->err.c_str());
    add("slides that discarded their key merge and re-sorted the window (a retired key was missing): %lld\n", e->merge_fallbacks);
    add("id lookahead (dppr_hint_next_batch): %lld id arrays taken from it so far (%lld entries resolved at the call), renumberings %llu\n", e->pre_hits, e->pre_misses, e->renumber_epoch);
    add("resident launches: mode %d ok %d retry %d time limit %llu ticks (100 MHz) rollcall_extra %d; schedule %d merge %d\n", e->persist_mode,
        (int)e->persist_ok, e->persist_retry, e->persist_ticks, e->persist_rollcall_extra, e->schedule, (int)e->merge_phases);
    // Device words through a stream of their own, waited for at most ~2 s in total. The copies land in a PINNED buffer the engine
    // owns for its whole life (a copy into pageable memory is staged and may block inside the call on a wedged device,
    // and one that completes after its stack destination is gone writes into dead memory); once a copy has not completed in time no
    // further one is issued and the side stream is abandoned, not destroyed (hipStreamDestroy would wait for it).
    hipStream_t side = nullptr;
    const bool have_side = e->dump_pin && hipSetDevice(e->device) == hipSuccess && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess;
    bool side_stuck = false;
    int polls_left = 2000; // x 1 ms, shared by all fetches of this dump
    auto fetch = [&](void *dst, const void *src, size_t bytes) -> bool {
        if (!have_side || side_stuck || !src || bytes > dppr_engine::DUMP_PIN_BYTES) return false;
        if (hipMemcpyAsync(e->dump_pin, src, bytes, hipMemcpyDeviceToHost, side) != hipSuccess) return false;
        while (polls_left-- > 0) {
            const hipError_t q = hipStreamQuery(side);
            if (q == hipSuccess) {
                memcpy(dst, e->dump_pin, bytes);
                return true;
            }
            if (q != hipErrorNotReady) return false;
            timespec ts{0, 1000000};
            nanosleep(&ts, nullptr);
        }
        side_stuck = true; // the copy stays queued: its destination outlives it
        return false;
    };
    add("engine stream: %s\n", hipStreamQuery(e->stream) == hipSuccess ? "idle" : "BUSY (work enqueued or running)");
    {
        static thread_local GridBar hb;
        if (fetch(&hb, e->bar, sizeof(GridBar))) {
            unsigned long long roll = 0, sub[2] = {0, 0};
            for (int s = 0; s < BAR_SUBS; ++s) roll += hb.roll[s].w >> 32;
            for (int par = 0; par < 2; ++par)
                for (int s = 0; s < BAR_SUBS; ++s) sub[par] += hb.sub[par][0][s].w >> 32;
            add("GridBar: gen %llu (0 pending, %llu ready, %llu abort) roll-call check-ins %llu, arrivals (replica 0) even sweeps %llu odd sweeps %llu\n",
                hb.gen.w, (unsigned long long)BAR_READY, (unsigned long long)BAR_ABORT, roll, sub[0], sub[1]);
        } else {
            add("GridBar: not readable (copy did not complete within 2 s)\n");
        }
    }
    for (size_t i = 0; i < e->slots.size(); ++i) {
        const Slot &s = e->slots[i];
        int h[CNT_HDR] = {0};
        const bool ok = fetch(h, s.cnt, sizeof(h));
        add("slot %zu: source %d (internal %d) converged %d last_epoch %d iterations %lld persist launches %lld aborts %lld binned sweeps %lld; "
            "device counters %s[%d %d %d | cand %d | big %d %d | status 0x%x]\n", i, s.source_ext, s.source, (int)s.converged, s.last_epoch,
            (long long)s.st.iterations, (long long)s.st.persist_launches, (long long)s.st.persist_aborts, (long long)s.st.binned_sweeps,
            ok ? "" : "(unreadable) ", h[0], h[1], h[2], h[3], h[5], h[6], (unsigned)h[7]);
    }
    for (size_t i = 0; i < e->groups.size(); ++i) {
        const Group &g = e->groups[i];
        int h[3 * GS_MAX] = {0}, st = 0;
        const bool ok = fetch(h, g.cnt, sizeof(h)) && fetch(&st, g.mlog, sizeof(int));
        long long f[3] = {0, 0, 0};
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < GS_MAX; ++k) f[r] += h[r * GS_MAX + k];
        add("group %zu: %d sources, rows of %d doubles, converged %d last_epoch %d iterations %lld multi-sweep launches %lld aborts %lld; device: %sfrontier "
            "pairs in the three rotating rows %lld %lld %lld, multi-sweep status 0x%x\n", i, g.n, g.gw, (int)g.converged, g.last_epoch,
            (long long)g.st.iterations, (long long)g.st.persist_launches, (long long)g.st.persist_aborts, ok ? "" : "(unreadable) ", f[0], f[1], f[2],
            (unsigned)st);
    }
    if (side_stuck) add("(a device read did not complete within 2 s: the remaining ones were skipped, the side stream is abandoned)\n");
    if (have_side && !side_stuck) (void)hipStreamDestroy(side);
#undef add
    const size_t n = std::min(o.size(), (size_t)cap - 1);
    memcpy(buf, o.data(), n);
    buf[n] = 0;
    return (int)n;
}

int dppr_init_solve(dppr_engine *e, int32_t slot, double eps, float *out_ms) { return dppr_init_solve_at(e, slot, -1, eps, out_ms); }

int dppr_init_solve_at(dppr_engine *e, int32_t slot, int32_t epoch, double eps, float *out_ms) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(k_init, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, s.p, s.r, e->V, s.source);
    HIP_TRY(hipGetLastError());
    s.converged = false;
    s.park_eps = 0.0; // (parked rows are zero again)
    int rc = main_loop_inspect(e, s, ep, 0, eps);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    s.converged = true;
    s.conv_eps = eps;
    s.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_incremental_batch_update(dppr_engine *e, int32_t slot, int32_t epoch) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if (!epoch_in_sequence(s.last_epoch, ep.id)) return fail(e, DPPR_ERR_INVALID, "epoch out of sequence for this source");
    HIP_TRY(hipSetDevice(e->device));
    // after a converged solve the update also lists the tails that left [-eps, eps] (dppr_seed_lists)
    const bool seeded = s.converged;
    int rc = prepare_epoch(e, ep);
    if (rc) return rc;
    rc = stream_update(e, s, ep, s.conv_eps, seeded);
    if (rc) return rc;
    s.seed_lists_valid = seeded;
    s.last_epoch = ep.id;
    s.converged = false;
    s.phase0_done = false;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_execute_main_loop(dppr_engine *e, int32_t slot, int32_t epoch, int phase, double eps) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if ((phase != 0 && phase != 1) || !(eps > 0)) return fail(e, DPPR_ERR_INVALID, "phase must be 0/1, eps > 0");
    HIP_TRY(hipSetDevice(e->device));
    int rc = settle_parked(e, s.p, s.r, 1, eps, &s.park_eps, &s.st);
    if (rc) return rc;
    rc = main_loop_inspect(e, s, ep, phase, eps);
    if (rc) return rc;
    if (phase == 0) {
        s.phase0_done = true;
        s.phase0_eps = eps;
    } else if (s.phase0_done && s.phase0_eps == eps) { // both phases done: |r| <= eps everywhere
        s.converged = true;
        s.conv_eps = eps;
    }
    return DPPR_OK;
}

int dppr_update(dppr_engine *e, int32_t slot, int32_t epoch, double eps, float *out_ms) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    if (!epoch_in_sequence(s.last_epoch, ep.id)) return fail(e, DPPR_ERR_INVALID, "epoch out of sequence for this source");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->persist_ok && e->persist_mode && e->persist_retry > 0 && --e->persist_retry == 0)
        e->persist_ok = true; // a resident launch gave up a while ago (the CUs were shared): try them again
    s.seed_lists_valid = false;
    // Seeding from the batch tails is exact only if every |r| <= eps beforehand
    // (the state a completed solve leaves). Otherwise fall back to full Inspect passes.
    // Merged loop (dppr_set_phase_merge, eager schedule): residuals of both signs are pushed in ONE loop, to eps / merge_div.
    const bool merged = e->merge_phases && e->schedule == DPPR_SCHEDULE_EAGER;
    if (merged) eps = eps / e->merge_div;
    const bool seeded = s.converged && s.conv_eps <= eps;
    const bool ahead = seeded && can_batch_ahead(e, s, ep) && resident_arena(e, ep);
    int rc = settle_parked(e, s.p, s.r, 1, eps, &s.park_eps, &s.st);
    if (rc) return rc;
    rc = prepare_epoch(e, ep);
    if (rc) return rc;
    if (e->raw_backoff > 0) --e->raw_backoff;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    // A whole-batch resident launch applies the records itself (PLAN_UPDATE) -- grouped at slide time and cut into the sweep groups'
    // ranges, or (default accounting) RAW: the launch finds, orders and applies every group's records itself. Only the counters
    // and the GridBar are cleared here. Should the launch call itself off, nothing was changed and the update runs as its own
    // kernels after all.
    const bool raw_ok = !ep.grouped && ep.L > 0 && ep.L <= RES_RAW_STEPS * sweep_block(e) && e->raw_backoff == 0;
    bool inline_su = ahead && e->res_update && ((ep.su_inline && ep.grouped) || raw_ok);
    if (inline_su) {
        hipLaunchKernelGGL(k_su_keys, dim3(1), dim3(BLOCK), 0, e->stream, ep.b1, 0, e->su_k[0], e->su_v[0],
                           reinterpret_cast<unsigned long long *>(e->bar), (int)(sizeof(GridBar) / sizeof(unsigned long long)), s.cnt, 5);
        HIP_TRY(hipGetLastError());
    } else {
        rc = stream_update(e, s, ep, eps, seeded, ahead);
    }
    if (rc) return rc;
    s.converged = false;
    auto update_after_abort = [&]() -> int { // (inline_su only) the launch called itself off
        if (!inline_su) return DPPR_OK;
        if (e->launch_called_off) {
            inline_su = false;
            return stream_update(e, s, ep, eps, seeded, false);
        }
        s.st.records += ep.L;
        return DPPR_OK;
    };
    if (merged && ahead) { // a window that runs resident: the whole merged loop as ONE launch that seeds itself
        int stage = 0;
        bool p1 = false;
        LoopEntry en0, en1;
        rc = batch_ahead(e, s, ep, eps, &stage, &en0, &en1, &p1, true, inline_su);
        if (rc) return rc;
        rc = update_after_abort();
        if (rc) return rc;
        if (stage != 2) { // out of sweeps, or the roll-call failed (then the update's lists stand: add the negative tails)
            if (en0.it == 0 && !en0.dense) {
                hipLaunchKernelGGL(k_filter, dim3(grid_for(std::max(ep.L, 1))), dim3(BLOCK), 0, e->stream, s.neg, s.cnt + 3, s.r, 1, eps,
                                   s.ft[0], s.cnt + 0);
                HIP_TRY(hipGetLastError());
            }
            rc = run_frontier_loop(e, s, ep, PHASE_BOTH, eps, 0, 0, en0);
            if (rc) return rc;
        }
    } else if (merged) {
        if (seeded) { // the frontier: the tails the update left above eps (ft[0]) and those it left below -eps (the candidates)
            hipLaunchKernelGGL(k_filter, dim3(grid_for(std::max(ep.L, 1))), dim3(BLOCK), 0, e->stream, s.neg, s.cnt + 3, s.r, 1, eps,
                               s.ft[0], s.cnt + 0);
            HIP_TRY(hipGetLastError());
            rc = run_frontier_loop(e, s, ep, PHASE_BOTH, eps, 0, 0);
        } else {
            rc = main_loop_inspect(e, s, ep, PHASE_BOTH, eps);
        }
        if (rc) return rc;
    } else if (seeded) {
        int stage = 0;
        bool p1_seeded = false;
        LoopEntry en0, en1;
        if (ahead) {
            rc = batch_ahead(e, s, ep, eps, &stage, &en0, &en1, &p1_seeded, false, inline_su);
            if (rc) return rc;
            rc = update_after_abort();
            if (rc) return rc;
        }
        if (stage == 0) {
            rc = run_frontier_loop(e, s, ep, 0, eps, 0, 0, en0);
            if (rc) return rc;
        }
        if (stage <= 1 && !p1_seeded && inline_su) {
            // the update ran inside the launch and recorded no candidates: phase 1 starts from a full Inspect
            rc = main_loop_inspect(e, s, ep, 1, eps);
            if (rc) return rc;
        } else if (stage <= 1) {
            if (!p1_seeded) { // phase 1: candidates recorded by the update, re-checked now
                HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * 3, e->stream));
                hipLaunchKernelGGL(k_filter, dim3(grid_for(std::max(ep.L, 1))), dim3(BLOCK), 0, e->stream, s.neg,
                                   s.cnt + 3, s.r, 1, eps, s.ft[0], s.cnt + 0);
                HIP_TRY(hipGetLastError());
            }
            rc = run_frontier_loop(e, s, ep, 1, eps, 0, 0, en1);
            if (rc) return rc;
        }
    } else {
        rc = main_loop_inspect(e, s, ep, 0, eps);
        if (rc) return rc;
        rc = main_loop_inspect(e, s, ep, 1, eps);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    s.st.gpu_ms += ms;
    s.st.batches++;
    s.converged = true;
    s.conv_eps = eps; // (the merged loop's eps / merge_div)
    s.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_read(dppr_engine *e, int32_t slot, double *p, double *r) {
    GET_SLOT(e, slot);
    HIP_TRY(hipSetDevice(e->device));
    int rc = sync_map(e);
    if (rc) return rc;
    const double *src[2] = {s.p, s.r};
    double *dst[2] = {p, r};
    for (int k = 0; k < 2; ++k) {
        if (!dst[k]) continue;
        hipLaunchKernelGGL(k_int_to_ext, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, src[k], e->d_ext2int, e->V,
                           e->d_xfer);
        HIP_TRY(hipMemcpyAsync(dst[k], e->d_xfer, sizeof(double) * (size_t)e->V, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return DPPR_OK;
}

int dppr_write(dppr_engine *e, int32_t slot, const double *p, const double *r) {
    GET_SLOT(e, slot);
    HIP_TRY(hipSetDevice(e->device));
    // vertices that carry a value get an internal id first
    for (int v = 0; v < e->V; ++v)
        if ((p && p[v] != 0.0) || (r && r[v] != 0.0)) (void)to_int(e, v);
    int rc = flush_moves(e);
    if (rc) return rc;
    rc = sync_map(e);
    if (rc) return rc;
    rc = recut_stale_groups(e);
    if (rc) return rc;
    const double *src[2] = {p, r};
    double *dst[2] = {s.p, s.r};
    for (int k = 0; k < 2; ++k) {
        if (!src[k]) continue;
        HIP_TRY(hipMemcpyAsync(e->d_xfer, src[k], sizeof(double) * (size_t)e->V, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemsetAsync(dst[k], 0, sizeof(double) * (size_t)e->V, e->stream));
        hipLaunchKernelGGL(k_ext_to_int, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, e->d_xfer, e->d_ext2int, e->V,
                           dst[k]);
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    s.converged = false;
    s.phase0_done = false;
    if (r) s.park_eps = 0.0; // (every vertex that carries a value has a live id now; parked rows were written as zeros)
    s.last_epoch = -2; // the caller supplied the state: which batches it contains is the caller's business
    s.seed_lists_valid = false;
    return DPPR_OK;
}

int dppr_seed_lists(dppr_engine *e, int32_t slot, int phase, int32_t *out_ids, int32_t *out_count) {
    GET_SLOT(e, slot);
    if (!out_ids || !out_count || (phase != 0 && phase != 1)) return DPPR_ERR_INVALID;
    if (!s.seed_lists_valid)
        return fail(e, DPPR_ERR_INVALID, "seed_lists: only right after dppr_incremental_batch_update on a converged state");
    HIP_TRY(hipSetDevice(e->device));
    int n = 0;
    int rc = read_count(e, s.cnt + (phase == 0 ? 0 : 3), &n);
    if (rc) return rc;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(out_ids, phase == 0 ? s.ft[0] : s.neg, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    for (int i = 0; i < n; ++i) out_ids[i] = e->int2ext[(size_t)out_ids[i]];
    *out_count = n;
    return DPPR_OK;
}

int dppr_stats(dppr_engine *e, int32_t slot, dppr_stats_t *out) {
    GET_SLOT(e, slot);
    if (!out) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    int rc = pull_device_stats(e, s);
    if (rc) return rc;
    // every enqueued vertex is a frontier member of a later iteration, except the seeds
    s.st.sum_N = s.st.sum_F;
    // SURVEY.md 8(d); its Inspect term (8 bytes per vertex and pass) is counted for the passes that RAN:
    // after a converged solve the frontier is seeded from the batch tails and no vertex is scanned
    s.st.algorithmic_bytes = 8ll * s.st.inspected + 45ll * s.st.records + 72ll * s.st.sum_F + 24ll * s.st.sum_E +
                             4ll * s.st.sum_N;
    *out = s.st;
    return DPPR_OK;
}

int dppr_reset_stats(dppr_engine *e, int32_t slot) {
    GET_SLOT(e, slot);
    HIP_TRY(hipSetDevice(e->device));
    s.st = dppr_stats_t{};
    HIP_TRY(hipMemsetAsync(s.dstats, 0, 2 * sizeof(IterStats), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_inspect(dppr_engine *e, int32_t slot, int phase, double eps, int32_t *out_ids, int32_t *out_count) {
    GET_SLOT(e, slot);
    if (!out_ids || !out_count || (phase != 0 && phase != 1)) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemsetAsync(s.cnt + 4, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL(k_inspect, dim3(grid_for(e->n_int, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream, s.r,
                       e->n_int, phase, eps, s.ft[1], s.cnt + 4);
    HIP_TRY(hipGetLastError());
    int n = 0;
    int rc = read_count(e, s.cnt + 4, &n);
    if (rc) return rc;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(out_ids, s.ft[1], sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    if (e->n_parked > 0 && eps < s.park_eps) { // parked rows are inert down to park_eps only
        const int base = e->V - e->n_parked;
        int m = 0;
        HIP_TRY(hipMemsetAsync(s.cnt + 4, 0, sizeof(int), e->stream));
        hipLaunchKernelGGL(k_inspect, dim3(grid_for(e->n_parked, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream, s.r + base,
                           e->n_parked, phase, eps, s.ft[1], s.cnt + 4);
        HIP_TRY(hipGetLastError());
        rc = read_count(e, s.cnt + 4, &m);
        if (rc) return rc;
        if (m > 0) {
            HIP_TRY(hipMemcpyAsync(out_ids + n, s.ft[1], sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            for (int i = n; i < n + m; ++i) out_ids[i] += base;
        }
        n += m;
    }
    for (int i = 0; i < n; ++i) out_ids[i] = e->int2ext[(size_t)out_ids[i]];
    *out_count = n;
    return DPPR_OK;
}

int dppr_graph_edges(dppr_engine *e, int32_t epoch, int32_t *out) {
    if (!e || !out) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    *out = ep.Ed;
    return DPPR_OK;
}

// internal CSR (rows by internal id, columns internal) -> external CSR with ascending rows
static void csr_to_external(const dppr_engine *e, const std::vector<int> &irow, const std::vector<int> &icol,
                            int32_t *row_ptr, int32_t *col) {
    int off = 0;
    std::vector<int> tmp;
    for (int v = 0; v < e->V; ++v) {
        if (row_ptr) row_ptr[v] = off;
        const int m = e->ext2int[(size_t)v];
        if (m >= 0) {
            tmp.clear();
            for (int j = irow[(size_t)m]; j < irow[(size_t)m + 1]; ++j) tmp.push_back(e->int2ext[(size_t)icol[(size_t)j]]);
            std::sort(tmp.begin(), tmp.end());
            if (col) std::copy(tmp.begin(), tmp.end(), col + off);
            off += (int)tmp.size();
        }
    }
    if (row_ptr) row_ptr[e->V] = off;
}

int dppr_read_graph(dppr_engine *e, int32_t epoch, int32_t *row_ptr, int32_t *col, int32_t *out_degree) {
    if (!e) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    HIP_TRY(hipSetDevice(e->device));
    std::vector<int> irow((size_t)e->V + 1), icol((size_t)std::max(ep.Ed, 1)), ideg((size_t)e->V);
    HIP_TRY(hipMemcpyAsync(irow.data(), ep.row_ptr, sizeof(int) * ((size_t)e->V + 1), hipMemcpyDeviceToHost, e->stream));
    if (ep.Ed > 0) {
        int *tmp = reinterpret_cast<int *>(e->keys_a); // scratch
        hipLaunchKernelGGL(k_split_adj, dim3(grid_for(ep.Ed)), dim3(BLOCK), 0, e->stream, ep.adj, ep.Ed, tmp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(icol.data(), tmp, sizeof(int) * (size_t)ep.Ed, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipMemcpyAsync(ideg.data(), e->outdeg, sizeof(int) * (size_t)e->V, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    csr_to_external(e, irow, icol, row_ptr, col);
    if (out_degree)
        for (int v = 0; v < e->V; ++v) {
            const int m = e->ext2int[(size_t)v];
            out_degree[v] = m >= 0 ? ideg[(size_t)m] : 0;
        }
    return DPPR_OK;
}

int dppr_read_out_graph(dppr_engine *e, int32_t epoch, int32_t *row_ptr, int32_t *col) {
    if (!e) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    HIP_TRY(hipSetDevice(e->device));
    std::vector<int> irow((size_t)e->V + 1), icol((size_t)std::max(ep.Ed, 1));
    HIP_TRY(hipMemcpyAsync(irow.data(), ep.out_row_ptr, sizeof(int) * ((size_t)e->V + 1), hipMemcpyDeviceToHost, e->stream));
    if (ep.Ed > 0)
        HIP_TRY(hipMemcpyAsync(icol.data(), ep.out_col, sizeof(int) * (size_t)ep.Ed, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    csr_to_external(e, irow, icol, row_ptr, col);
    return DPPR_OK;
}

int dppr_trace_enable(dppr_engine *e, int32_t slot, int on) {
    GET_SLOT(e, slot);
    s.trace = on != 0;
    s.trace_ids.clear();
    s.trace_off.assign(1, 0);
    return DPPR_OK;
}

int dppr_trace_get(dppr_engine *e, int32_t slot, int64_t *n_iters, int64_t *n_ids, int64_t *offsets, int32_t *ids) {
    GET_SLOT(e, slot);
    const int64_t ni = s.trace_off.empty() ? 0 : (int64_t)s.trace_off.size() - 1;
    if (n_iters) *n_iters = ni;
    if (n_ids) *n_ids = (int64_t)s.trace_ids.size();
    if (offsets && !s.trace_off.empty()) memcpy(offsets, s.trace_off.data(), sizeof(int64_t) * s.trace_off.size());
    if (ids && !s.trace_ids.empty()) memcpy(ids, s.trace_ids.data(), sizeof(int32_t) * s.trace_ids.size());
    return DPPR_OK;
}

#define GET_GROUP(e, gid)                                                                             \
    if (!(e) || (gid) < 0 || (gid) >= (int)(e)->groups.size()) return fail((e), DPPR_ERR_INVALID, "bad group"); \
    if ((e)->broken) return fail((e), DPPR_ERR_INVALID, "engine unusable after a failed renumbering");           \
    Group &g = (e)->groups[(size_t)(gid)]

int dppr_add_source_group(dppr_engine *e, const int32_t *sources, int32_t n, int32_t *out_group) {
    if (!e || e->broken || !sources || n < 1 || n > GS_MAX) return fail(e, DPPR_ERR_INVALID, "add_source_group: 1..16 sources");
    if (!ids_in_range(e, sources, n)) return fail(e, DPPR_ERR_INVALID, "add_source_group: vertex out of range"); // before anything is touched
    HIP_TRY(hipSetDevice(e->device));
    Group g;
    g.n = n;
    g.gw = e->group_full_rows ? (n > OCT ? 2 * OCT : OCT) : row_width(n); // doubles per vertex: the sources + at most one of padding
    g.spl = row_spl(g.gw);
    for (int s = 0; s < GS_MAX; ++s) g.src.s[s] = -1;
    for (int s = 0; s < n; ++s) g.src_ext[s] = sources[s];
    const size_t V = (size_t)e->V, row = sizeof(double) * (size_t)g.gw;
    g.act_bytes = (V / 32 + 1024 / 32 + 4) * sizeof(uint32_t);
    auto release = [&]() { // (an allocation failed: nothing of this group stays behind)
        (void)hipFree(g.p); (void)hipFree(g.r); (void)hipFree(g.x); (void)hipFree(g.x2); (void)hipFree(g.act[0]); (void)hipFree(g.act[1]);
        (void)hipFree(g.cnt); (void)hipFree(g.mlog); (void)hipFree(g.dstats); (void)hipFree(g.gq);
    };
#define GRP_TRY(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            release();                                                                                  \
            e->err = std::string("add_source_group: ") + hipGetErrorString(_e);                         \
            return _e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP;                           \
        }                                                                                               \
    } while (0)
    GRP_TRY(hipMalloc((void **)&g.p, row * V));
    GRP_TRY(hipMalloc((void **)&g.r, row * V));
    const size_t xrow = sizeof(double) * (size_t)x_stride(g.gw); // snapshot rows never straddle a 128-byte line (dppr_multi.hpp)
    GRP_TRY(hipMalloc((void **)&g.x, xrow * V));
    GRP_TRY(hipMalloc((void **)&g.x2, xrow * V));
    GRP_TRY(hipMalloc((void **)&g.act[0], g.act_bytes));
    GRP_TRY(hipMalloc((void **)&g.act[1], g.act_bytes));
    GRP_TRY(hipMalloc((void **)&g.cnt, sizeof(int) * (5 * GS_MAX + MAX_CHUNK * GS_MAX))); // rows 3, 4: scratch of multi-sweep launches
    GRP_TRY(hipMalloc((void **)&g.mlog, sizeof(int) * (size_t)(GMULTI_MAX + 2) * GS_MAX));
    GRP_TRY(hipMalloc((void **)&g.dstats, 2 * sizeof(IterStats)));
    GRP_TRY(hipMalloc((void **)&g.gq, sizeof(int) * 3 * GQ_PAD));
    GRP_TRY(hipMemsetAsync(g.gq, 0, sizeof(int) * 3 * GQ_PAD, e->stream));
    GRP_TRY(hipMemsetAsync(g.act[0], 0, g.act_bytes, e->stream));
    GRP_TRY(hipMemsetAsync(g.act[1], 0, g.act_bytes, e->stream));
    GRP_TRY(hipMemsetAsync(g.cnt, 0, sizeof(int) * (5 * GS_MAX + MAX_CHUNK * GS_MAX), e->stream));
    GRP_TRY(hipMemsetAsync(g.dstats, 0, 2 * sizeof(IterStats), e->stream));
    // the memory is there: now the ids (a source outside the window receives one; a parked one is revived)
    for (int s = 0; s < n; ++s) (void)to_int(e, sources[s]);
    for (int s = 0; s < n; ++s) g.src.s[s] = e->ext2int[(size_t)sources[s]]; // (after ALL revivals: one may move another)
    if (int rc = flush_moves(e)) {
        release();
        return rc;
    }
    hipLaunchKernelGGL(k_ginit, dim3(grid_for((int64_t)e->V * g.gw)), dim3(BLOCK), 0, e->stream, g.p, g.r, e->V, g.gw, g.src);
    GRP_TRY(hipGetLastError());
    GRP_TRY(hipStreamSynchronize(e->stream));
#undef GRP_TRY
    e->groups.push_back(g);
    e->any_groups = true; // (recut_stale_groups below adds the second group table to resident epochs)
    if (g.spl == 2) e->wide_groups = true;
    if (int rc = recut_stale_groups(e)) return rc;
    if (out_group) *out_group = (int)e->groups.size() - 1;
    return DPPR_OK;
}

int dppr_group_init_solve(dppr_engine *e, int32_t group, double eps, float *out_ms) {
    return dppr_group_init_solve_at(e, group, -1, eps, out_ms);
}

int dppr_group_init_solve_at(dppr_engine *e, int32_t group, int32_t epoch, double eps, float *out_ms) {
    GET_GROUP(e, group);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(k_ginit, dim3(grid_for((int64_t)e->V * g.gw)), dim3(BLOCK), 0, e->stream, g.p, g.r, e->V, g.gw, g.src);
    HIP_TRY(hipGetLastError());
    g.converged = false;
    g.park_eps = 0.0;
    int rc = group_loop(e, g, ep, 0, eps, false);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    g.converged = true; // r = e_s >= 0 and phase 0 left every r <= eps: nothing is below -eps
    g.conv_eps = eps;
    g.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_group_update(dppr_engine *e, int32_t group, int32_t epoch, double eps, float *out_ms) {
    GET_GROUP(e, group);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    if (!epoch_in_sequence(g.last_epoch, ep.id)) return fail(e, DPPR_ERR_INVALID, "epoch out of sequence for this group");
    HIP_TRY(hipSetDevice(e->device));
    // seeding from the batch tails is exact only if every |r| <= eps beforehand (dppr_update has the same rule)
    const bool merged = e->merge_phases && e->schedule == DPPR_SCHEDULE_EAGER; // (dppr_set_phase_merge)
    if (merged) eps = eps / e->merge_div;
    const bool tails = g.converged && g.conv_eps <= eps && e->group_tail_seeding;
    int rc = settle_parked(e, g.p, g.r, g.gw, eps, &g.park_eps, &g.st);
    if (rc) return rc;
    rc = prepare_epoch(e, ep);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    rc = group_stream_update(e, g, ep);
    if (rc) return rc;
    g.converged = false;
    if (merged) {
        rc = group_loop(e, g, ep, PHASE_BOTH, eps, tails);
        if (rc) return rc;
    } else {
        rc = group_loop(e, g, ep, 0, eps, tails);
        if (rc) return rc;
        rc = group_loop(e, g, ep, 1, eps, tails);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    g.st.gpu_ms += ms;
    g.st.batches++;
    g.converged = true;
    g.conv_eps = eps;
    g.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_group_read(dppr_engine *e, int32_t group, int32_t index, double *p, double *r) {
    GET_GROUP(e, group);
    if (index < 0 || index >= g.n) return fail(e, DPPR_ERR_INVALID, "group_read: bad source index");
    HIP_TRY(hipSetDevice(e->device));
    int rc = sync_map(e);
    if (rc) return rc;
    const double *src[2] = {g.p, g.r};
    double *dst[2] = {p, r};
    for (int k = 0; k < 2; ++k) {
        if (!dst[k]) continue;
        hipLaunchKernelGGL(k_gint_to_ext, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, src[k], g.gw, index, e->d_ext2int,
                           e->V, e->d_xfer);
        HIP_TRY(hipMemcpyAsync(dst[k], e->d_xfer, sizeof(double) * (size_t)e->V, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return DPPR_OK;
}

int dppr_group_reset_stats(dppr_engine *e, int32_t group) {
    GET_GROUP(e, group);
    HIP_TRY(hipSetDevice(e->device));
    g.st = dppr_stats_t{};
    HIP_TRY(hipMemsetAsync(g.dstats, 0, 2 * sizeof(IterStats), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_group_stats(dppr_engine *e, int32_t group, dppr_stats_t *out) {
    GET_GROUP(e, group);
    if (!out) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    static thread_local IterStats h[2];
    HIP_TRY(hipMemcpyAsync(h, g.dstats, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long t = 0, ts = 0;
    for (int i = 0; i < STAT_SLOTS; ++i) {
        t += h[0].blk_E[i];
        ts += h[1].blk_E[i];
    }
    g.st.sum_E = (int64_t)(t + ts);
    g.st.sweep_E = (int64_t)ts;
    g.st.sum_N = g.st.sum_F;
    g.st.algorithmic_bytes = 8ll * g.st.inspected + 45ll * g.st.records + 72ll * g.st.sum_F + 24ll * g.st.sum_E +
                             4ll * g.st.sum_N;
    *out = g.st;
    return DPPR_OK;
}

#ifdef DPPR_STAMPS
// diagnostic build only: copy the stage stamps of the last sweep (rows x 8 clock values)
extern "C" int dppr_debug_bin_stamps(unsigned long long *out, int which, int rows) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dppr::g_bin_stamps), sizeof(unsigned long long) * 6 * (size_t)rows,
                               sizeof(unsigned long long) * 6 * 16384 * (size_t)which) == hipSuccess ? 0 : -2;
}
extern "C" int dppr_debug_stamps(unsigned long long *out, int rows) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dppr::g_stamps), sizeof(unsigned long long) * 8 * (size_t)rows) == hipSuccess
               ? 0 : -2;
}
#endif

int dppr_bench_atomics(int device, int64_t table_elems, int64_t n, int scope, int reps, float *out_ms) {
    if (table_elems <= 0 || (table_elems & (table_elems - 1)) || n <= 0 || reps <= 0 || !out_ms) return DPPR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DPPR_ERR_HIP;
    double *table = nullptr, *sink = nullptr;
    hipEvent_t a, b;
    if (hipMalloc((void **)&table, sizeof(double) * (size_t)table_elems) != hipSuccess) return DPPR_ERR_NOMEM;
    (void)hipMalloc((void **)&sink, sizeof(double));
    (void)hipMemset(table, 0, sizeof(double) * (size_t)table_elems);
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 2048;
    auto launch = [&]() {
        if (scope == 2) // (calibration only) adds whose result is not used
            hipLaunchKernelGGL(k_bench_scatter<0>, dim3(grid), dim3(BLOCK), 0, 0, table, (uint64_t)table_elems - 1, n);
        else if (scope == 3) // ... and plain scattered 8-byte stores
            hipLaunchKernelGGL(k_bench_scatter<1>, dim3(grid), dim3(BLOCK), 0, 0, table, (uint64_t)table_elems - 1, n);
        else if (scope == 0)
            hipLaunchKernelGGL(k_bench_atomics<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(BLOCK), 0, 0, table,
                               (uint64_t)table_elems - 1, n, sink);
        else
            hipLaunchKernelGGL(k_bench_atomics<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(BLOCK), 0, 0, table,
                               (uint64_t)table_elems - 1, n, sink);
    };
    launch(); // warm-up
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    hipError_t err = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    *out_ms = ms / reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(table);
    (void)hipFree(sink);
    return err == hipSuccess ? DPPR_OK : DPPR_ERR_HIP;
}

int dppr_bench_line_fills(int device, int64_t table_bytes, int64_t lines, int reps, float *out_ms) {
    if (table_bytes < 128 || (table_bytes & (table_bytes - 1)) || lines <= 0 || reps <= 0 || !out_ms) return DPPR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DPPR_ERR_HIP;
    double2 *table = nullptr;
    double *sink = nullptr;
    if (hipMalloc((void **)&table, (size_t)table_bytes) != hipSuccess) return DPPR_ERR_NOMEM;
    (void)hipMalloc((void **)&sink, sizeof(double));
    (void)hipMemset(table, 0, (size_t)table_bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 2048; // 2048 x 128 octets
    const int per = (int)std::max<int64_t>(8, (lines + (int64_t)grid * 128 - 1) / ((int64_t)grid * 128) / 8 * 8);
    auto launch = [&]() {
        hipLaunchKernelGGL(k_bench_lines<8>, dim3(grid), dim3(1024), 0, 0, table, (uint64_t)(table_bytes / 128) - 1, per, sink);
    };
    launch(); // warm-up
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    hipError_t err = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    // per launch, scaled to the number of lines asked for (the launch fetches grid * 128 * per of them)
    *out_ms = ms / reps * (float)((double)lines / ((double)grid * 128.0 * per));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(table);
    (void)hipFree(sink);
    return err == hipSuccess ? DPPR_OK : DPPR_ERR_HIP;
}

int dppr_bench_stream_copy(int device, int64_t bytes, int reps, float *out_ms) {
    if (bytes < 16 || reps <= 0 || !out_ms) return DPPR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DPPR_ERR_HIP;
    double2 *src = nullptr, *dst = nullptr;
    if (hipMalloc((void **)&src, (size_t)bytes) != hipSuccess) return DPPR_ERR_NOMEM;
    if (hipMalloc((void **)&dst, (size_t)bytes) != hipSuccess) {
        (void)hipFree(src);
        return DPPR_ERR_NOMEM;
    }
    (void)hipMemset(src, 0, (size_t)bytes);
    (void)hipMemset(dst, 0, (size_t)bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    auto launch = [&]() { hipLaunchKernelGGL(k_bench_copy, dim3(4096), dim3(1024), 0, 0, src, dst, bytes / 16); };
    launch();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    hipError_t err = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    *out_ms = ms / reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(src);
    (void)hipFree(dst);
    return err == hipSuccess ? DPPR_OK : DPPR_ERR_HIP;
}

unsigned long long dppr_heartbeat(const dppr_engine *e) { return e ? e->heartbeat.load(std::memory_order_relaxed) : 0ull; }

#ifndef DPPR_BUILD_ID
#define DPPR_BUILD_ID "unstamped"
#endif
const char *dppr_build_id(void) { return DPPR_BUILD_ID; }

} // extern "C"
