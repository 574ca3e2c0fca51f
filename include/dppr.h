/*
 * dppr.h -- C ABI of libdppr_hip.so: the MI355X-native dynamic reverse-push PPR
 * engine (hot path of guowentian/dynamicppr, gpu/ tree).
 *
 * Drop-in boundary. The reference has no FFI layer; its seam is the C++ virtual
 * interface PPRGPU::{GPUBuildSlidingGraph, IncrementalBatchUpdate,
 * ExecuteMainLoop(phase), ValidateResult} (gpu/PPRGPU.cuh:179-182) plus the data
 * handed across it (DeviceMemory, GPUEdgeBatch, SlidingGraphBuilder). Each entry
 * point below names the reference interface it replaces. The host program
 * (./pagerank, dynamicppr_amd/host/) and any other binding call only this.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types. Every function
 * returns 0 (DPPR_OK) or a negative dppr_status; nothing exits the process (the
 * reference prints and exit(-1)s, gpu/GPUUtil.cuh:7-19). An engine is confined to
 * one host thread -- except for the builder / solver pair of dppr_slide_concurrent and the
 * helper of dppr_hint_next_batch --; engines on different devices are independent. Host buffers are
 * borrowed for the duration of the call only (as EdgeBatch arrays are,
 * SlidingGraphVec.h:17-26). All device memory is owned by the engine
 * (DeviceMemory / SlidingGraphBuilder ownership, gpu/DeviceMemory.cuh:31-50).
 *
 * Index type int32, value type double, ALPHA = 0.15 (Meta.h:25-31).
 */
#ifndef DPPR_H
#define DPPR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPPR_ABI_VERSION 5

typedef struct dppr_engine dppr_engine; /* opaque */

typedef enum dppr_status {
    DPPR_OK = 0,
    DPPR_ERR_INVALID = -1,   /* bad argument / call order */
    DPPR_ERR_HIP = -2,       /* a HIP runtime call failed (see dppr_last_error) */
    DPPR_ERR_NOMEM = -3,
    DPPR_ERR_NO_DEVICE = -4, /* no usable gfx950 device: the HIP path is mandatory, no CPU fallback */
    DPPR_ERR_NOT_CONVERGED = -5 /* iteration cap hit */
} dppr_status;

/* Push schedule. Both are legal interleavings of the reference kernels
 * (gpu/ExpandRev.cuh:8-183 + :708-743); see DESIGN.md "Schedules".
 *   EAGER : sparse (push) iterations take a frontier vertex's residual with an atomic
 *           exchange at the moment it is pushed (reads whatever has arrived, like the
 *           racy `ru = residual[u]` of ExpandUnifiedRev); dense iterations run as pull
 *           sweeps, which are synchronous by construction. Production mode.
 *   SYNC  : every frontier residual is snapshotted before any push of the iteration
 *           lands, in sparse iterations too; results are independent of thread
 *           scheduling up to the rounding of the sums, and the per-iteration frontier
 *           SETS equal the oracle's synchronous schedule exactly. Validation mode. */
#define DPPR_SCHEDULE_EAGER 0
#define DPPR_SCHEDULE_SYNC 1

typedef struct dppr_stats_t {
    int64_t iterations;   /* frontier-loop iterations (both phases) since last reset */
    int64_t sum_F;        /* sum of frontier sizes */
    int64_t sum_E;        /* traversed in-edges */
    int64_t sum_N;        /* vertices enqueued into next frontiers */
    int64_t records;      /* batch records applied by IncrementalBatchUpdate */
    int64_t inspected;    /* vertices scanned by full Inspect passes */
    int64_t batches;      /* dppr_update calls */
    int64_t pull_iterations; /* iterations evaluated as a dense pull sweep (subset of iterations) */
    int64_t algorithmic_bytes; /* 8 * inspected + 45 * records + sum(72F + 24E + 4N) */
    double gpu_ms;        /* sum of event-timed regions */
    double push_ms;       /* sum of per-launch event times of the push kernel (profiling on only) */
    int64_t push_launches; /* push-kernel launches timed into push_ms */
    int64_t persist_launches; /* launches of the resident multi-iteration sweep (k_pull_resident) */
    int64_t persist_aborts;   /* of those, launches that stopped at a grid-barrier time-out */
    int64_t binned_sweeps;    /* pull iterations evaluated as a binned sweep (k_bin_scatter + k_bin_reduce; subset of pull_iterations) */
    /* ABI 3: the dense sweeps on their own (one launch per sweep: k_pull_iter, k_bin_scatter + k_bin_reduce, k_gsweep), so that
     * the roofline of the sweep kernel is its own bytes over its own time, reproducible from a rocprofv3 kernel-stats file */
    int64_t sweep_F;          /* sum of the frontier sizes the sweeps pushed (subset of sum_F) */
    int64_t sweep_E;          /* in-edges the sweeps traversed (subset of sum_E) */
    double sweep_ms;          /* sum of the sweeps' event times (profiling on only; subset of push_ms) */
    int64_t sweep_launches;   /* sweeps timed into sweep_ms */
} dppr_stats_t;

/* ---- lifetime ----------------------------------------------------------- */

/* HIP devices visible to the process (0 when there is none or the runtime fails): lets a host program that has no HIP of
 * its own map `-g N` device threads onto the devices that exist. No reference counterpart (single implicit device 0). */
int dppr_device_count(void);

/* IncrementalBatchUpdate replays a batch's records tail by tail, each tail's records in batch order (lock-free; the
 * reference serialises them with a spin lock per tail, gpu/StreamUpdate.cuh:50-72), after CopyOutDegree (gpu/StreamUpdate.cuh:7-17).
 * at_slide = 0 (DEFAULT since ABI 4): both run inside dppr_update / dppr_group_update / dppr_incremental_batch_update -- the
 * reference's bracket (gpu/PPRGPU.cuh:138-164 times all of IncrementalBatchUpdate). Up to 4 Ki records (SU_RANK_MAX) the grouping is one
 * ranking launch (k_su_group_rank), beyond that a stable LDS radix placement on the tail id; a whole-batch resident launch takes the records raw and
 * groups them itself (dppr_resident.hpp). at_slide = 1 (rounds 3-4): the grouping and the degree gather, functions of the batch
 * alone, are done when the batch is uploaded (dppr_slide; for epochs that already exist: on entry to the next call, BEFORE its
 * event bracket opens) and the timed region holds only the replay. Same results bit for bit. */
int dppr_set_batch_grouping(dppr_engine *e, int at_slide);

/* The reference's four variants (-o, Meta.h; gpu/PPRRevPushGPUVariants.cuh:6-150) as MECHANISMS of the push iterations:
 *   0 OPTIMIZED     eager residual read + threshold-crossing duplicate filter (ExpandUnifiedRev + RepairFrontierRev)      [default]
 *   1 FAST_FRONTIER residuals pre-extracted and zeroed at the snapshot (InspectExtra, gpu/Inspect.cuh:51-65), crossing filter, no repair
 *   2 EAGER         eager read + STATUS-ARRAY filter: legal(curr) && atomicExch(status[v], level) < level (gpu/ExpandRev.cuh:255,298,340)
 *   3 VANILLA       pre-extracted + status-array filter (gpu/ExpandRev.cuh:603,646,688)
 * 1 and 3 imply the synchronous schedule, 0 and 2 the eager one (dppr_set_schedule may be called afterwards). Dense iterations
 * run as sweeps in every variant (a sweep computes the next frontier directly and needs neither filter): to time the four
 * mechanisms against each other, as the paper's ablation does, pin the push form (dppr_set_tuning pull_min_frontier = -1,
 * ./pagerank --push-only). Results: the variants' frontier SETS per iteration equal the oracle's restatements of the reference's
 * variants (tests); p / r as for the two schedules. */
int dppr_set_variant(dppr_engine *e, int variant);

/* MERGED LOOP (off by default; not the reference's schedule). The reference pushes the positive residuals of a batch to
 * convergence (ExecuteMainLoop(0)) and then the negative ones (ExecuteMainLoop(1), gpu/PPRGPU.cuh:138-164). Both loops
 * spread the same batch's disturbance over the same part of the graph, and most of what they move cancels: with
 * on != 0, dppr_update / dppr_group_update (eager schedule) run ONE loop that pushes every residual with |r| > eps',
 * eps' = eps / eps_divisor (0 keeps the divisor; default 4). Every push is the reference's push (gpu/ExpandRev.cuh:70-77,
 * :708-743), the loop invariant of SURVEY.md section 0 holds after every iteration, and the state it ends in satisfies
 * |r| <= eps' < eps -- a state the reference's own Validate() accepts (cpu/PPRCPUMTCilkRev.h:291-309), closer to the
 * fixed point than the reference's result (LiveJournal stand-in, eps 1e-9: max |p - p_cilk| 4.8e-10 with divisor 4;
 * 7.7e-10 with divisor 1) in a third to a half of the sweeps. The split interface (dppr_incremental_batch_update +
 * dppr_execute_main_loop(0 / 1)) and the synchronous schedule always run the reference's two loops. */
int dppr_set_phase_merge(dppr_engine *e, int on, int eps_divisor);

/* Replaces: DeviceMemory ctor + CudaAllocAppData + InitForDynamicGraph
 * (gpu/DeviceMemory.cuh:9-74) and SlidingGraphBuilder ctor
 * (gpu/SlidingGraphBuilder.cuh:64-76), as called from PPRGPU ctor
 * (gpu/PPRGPU.cuh:23-34).
 *   window_edges : W, the sliding window size in STREAM edges (undirected streams
 *                  are mirrored inside, as InitWindowStream does)
 *   max_batch    : c, stream edges per batch (edge batches hold up to 4c records)
 *   n_epochs     : how many graph epochs (CSR + batch records) stay resident in
 *                  HBM; 1 = rebuild in place like the reference, K+1 lets a caller
 *                  pre-stage K batches and then run the timed path back to back. */
int dppr_create(dppr_engine **out, int device, int32_t vertex_count, int32_t window_edges,
                int directed, int32_t max_batch, int32_t n_epochs);
void dppr_destroy(dppr_engine *e);
const char *dppr_strerror(int status);
const char *dppr_last_error(const dppr_engine *e); /* detail of the last failure */
int dppr_abi_version(void);
int dppr_set_schedule(dppr_engine *e, int schedule);
/* When on, every launch of the push kernel (the dominant kernel) is bracketed by its own
 * hipEvent pair on the engine's stream and summed into dppr_stats_t.push_ms. Off by default
 * (the extra events perturb the whole-batch time slightly). */
int dppr_set_profiling(dppr_engine *e, int on);
/* Execution-path knobs (defaults 256 / 512 / 0); only valid right after dppr_create.
 * hub_min_degree: out-degree from which a vertex's incoming pushes are aggregated in LDS (at
 * most 2048 hubs per epoch); big_row_edges: in-degree from which a frontier vertex's row is
 * expanded by the whole grid; pull_min_frontier: frontier size from which an iteration is
 * evaluated as a dense pull sweep instead of push atomics (0 = auto: max(1024, edges/192),
 * negative = never); chunk_iters: iterations enqueued per host read-back of the frontier size
 * (default 24, 1 = read back every iteration like the reference, <= 0 keeps the default; an
 * explicit value also caps the sweeps a resident launch may run before the host looks);
 * pull_block: workgroup size of the sweep, 256 / 512 / 1024 (0 = chosen from the graph size).
 * Results never depend on them beyond floating-point summation order; tests set them so
 * small graphs exercise every path. */
int dppr_set_tuning(dppr_engine *e, int hub_min_degree, int big_row_edges, int pull_min_frontier,
                    int chunk_iters, int pull_block);
/* Resident sweeps: when a window's sweep groups all fit on the chip at once, a run of dense
 * iterations is ONE launch that keeps the per-vertex state on chip; workgroups exchange the
 * per-iteration snapshot through memory and wait for each other's values, not for the host (no
 * counterpart in the reference, whose loop reads the frontier size back after every iteration,
 * gpu/PPRRevPushGPU.cuh:107). mode 1 = automatic (default; in the steady
 * state of a stream the launches of BOTH phases of a batch are enqueued ahead, without a
 * read-back in between), 2 = resident launches but one host read-back per launch, 0 = never
 * (every iteration its own launch). Such a launch first checks that all its workgroups are
 * running (waiting on one another needs that); timeout_us is how long that roll-call may take before the
 * launch gives up WITHOUT having changed anything and the engine continues with per-iteration
 * launches, trying resident ones again 64 batches later (0 keeps the default, 50 ms; negative = a roll-call that cannot succeed, which tests
 * use to exercise that path; the results are the same either way).
 * Only valid right after dppr_create. */
int dppr_set_persistent(dppr_engine *e, int mode, int64_t timeout_us);

/* Single-source per-iteration sweeps (windows too large for a resident launch) can test an activity
 * bitmap of the snapshot (1 bit per vertex, L2-resident) before each 8-byte gather of x[u], so that
 * heads outside the frontier cost no random sector (k_pull_iter<.., true>). Off by default: on the
 * LiveJournal and twitter stand-ins it fetches 30 % / 0 % fewer bytes and is 0 % / 23 % SLOWER (the
 * extra dependent L2 round trip per edge costs more than the sectors it saves; profiles/r02_pmc_*_bits*).
 * Source groups always use their bitmap (dppr_multi.hpp). Results are identical either way. */
int dppr_set_sweep_bitmap(dppr_engine *e, int on);

/* Single-source dense iterations on windows whose snapshot vector is far beyond the L2s run as a BINNED
 * (propagation-blocked) sweep: two streaming passes over a per-epoch layout of the window's edges instead of one
 * random 64-byte sector per edge (dppr_binned.hpp: k_bin_scatter + k_bin_reduce replace k_pull_iter; the per-edge
 * term is still gpu/ExpandRev.cuh:72's expression, results equal k_pull_iter's up to the order of each row's sum).
 *   mode         : 0 never, 1 (default) when a source slot exists and the window has at least min_ids vertices with
 *                  an id, 2 always (tests: tiny windows)
 *   ha_tiles     : an A-block holds at most 64 x ha_tiles heads (8 bytes of LDS per head; 0 keeps the default, 128)
 *   hb_tiles     : a B-block holds at most 64 x hb_tiles rows (20 bytes of LDS per row; 0 keeps 60 = 77 KB, two workgroups per CU)
 *   target_edges : edges a B-block is cut for (0 keeps the automatic choice, clamp(window edges / 256, 16 Ki, 384 Ki): twitter stand-in,
 *                  single source, 192 Ki / 384 Ki / 768 Ki: 74.6 / 70.2 / 77.7 ms per batch); a row of a quarter of that is a block of its own
 *   min_ids      : mode 1 threshold (0 keeps 1 Mi vertices with an id: smaller windows run resident or cannot fill the chip with blocks)
 *   chunk_edges  : runs (one head into one B-block: the unit x[u] travels in since ABI 5) per workgroup of k_bin_scatter (0 keeps 32768)
 *   target_a_edges : edges an A-block is cut for (0 keeps 4 Mi: large, the longer the runs a tile's values are written in)
 * The layout costs at most 4 bytes per window edge and epoch (2 per edge + 2 per run + 4 per tile) plus 32 bytes per window edge of engine
 * state (the values of pass 1, and both orders of the window's edges as sorted words), and is built in dppr_load_window and PATCHED by
 * dppr_slide (untimed, like the CSRs):
 * a slide merges the words of its retired and inserted edges into the two orders instead of sorting the window twice; the block
 * cuts are renewed every 32 slides (DPPR_BIN_RECUT_EVERY; DPPR_BIN_INCREMENTAL=0: the sorts every epoch). Only valid right after dppr_create. */
int dppr_set_binned_sweep(dppr_engine *e, int mode, int ha_tiles, int hb_tiles, int64_t target_edges, int64_t min_ids,
                          int64_t chunk_edges, int64_t target_a_edges);

/* ---- graph side (UNTIMED in the reference's metric) --------------------- */

/* Replaces: SlidingGraphBuilder::InitWindowStream (gpu/SlidingGraphBuilder.cuh:182-192)
 * + BuildInGraph (:137-153) / DeviceMemory::CudaMemcpyGraph (gpu/DeviceMemory.cuh:76-112).
 * e1/e2: the n == W window edges in stream order, NOT mirrored. Builds epoch 0. */
int dppr_load_window(dppr_engine *e, const int32_t *e1, const int32_t *e2, int32_t n);

/* Replaces: GPUEdgeBatch::CudaMemcpy(EdgeBatch*, HostToDevice) (gpu/GPUEdgeBatch.cuh:20-26).
 * The L = 2c (directed) or 4c (undirected) records exactly as
 * SlidingGraphVec::StreamUpdates lays them out (SlidingGraphVec.h:241-272):
 * [c deletes][c inserts] then the mirrored copy. Stages them for the NEXT epoch. */
int dppr_set_batch(dppr_engine *e, const int32_t *edge1, const int32_t *edge2,
                   const uint8_t *is_insert, int32_t length);

/* Replaces: SlidingGraphBuilder::IncBuildInGraph(new_stream, ...)
 * (gpu/SlidingGraphBuilder.cuh:117-133, :163-181, :193-221) = PPRGPU::GPUBuildSlidingGraph.
 * Drops the c oldest window edges, appends the c new ones, rebuilds the in-CSR
 * (rows sorted ascending like the reference's pair sort) and the out-degrees, and
 * makes the result the newest epoch. Call after dppr_set_batch. *out_epoch (may be
 * NULL) receives the new epoch id. */
int dppr_slide(dppr_engine *e, const int32_t *new_e1, const int32_t *new_e2, int32_t c,
               int32_t *out_epoch);

/* The same graph update, allowed to run CONCURRENTLY with one solver call on an OLDER epoch (ABI 4; the overlap of the reference's
 * untimed region, gpu/PPRGPU.cuh:114-135, with its timed one). Contract: ONE builder thread (dppr_set_batch +
 * dppr_slide_concurrent for batch k + 1) beside ONE solver thread (dppr_update / dppr_group_update / dppr_incremental_batch_update /
 * dppr_execute_main_loop / dppr_read / dppr_stats with an EXPLICIT epoch id <= k; never -1 = "newest", which the builder is
 * changing); the solver must not be given epoch k + 1 before this call has returned; n_epochs >= 2 (epoch k + 1 takes the ring
 * entry of epoch k + 1 - n_epochs, which nothing may still be solving on). The builder works on its own HIP stream and scratch;
 * what it shares with the solver are the state rows of vertices that are NEW or REVIVED in batch k + 1, which no older epoch
 * touches. Such a slide never renumbers the id space and never grows the resident-launch arena: when dppr_renumbering_due(e)
 * says so, make the next graph update an exclusive dppr_slide (after the solver call has returned). Same results.
 * dppr_read / dppr_group_read beside the builder (ABI 5): the engine holds its id-map lock from the first id a batch assigns or revives
 * to the end of the row moves that go with it, and a read holds it from its copy of the map to the end of its gathers -- a read sees
 * the id space of before a revival or of after it, never a mixture (it may wait for the builder's id-assigning head, never for the
 * graph build itself). */
int dppr_slide_concurrent(dppr_engine *e, const int32_t *new_e1, const int32_t *new_e2, int32_t c, int32_t *out_epoch);
int dppr_renumbering_due(const dppr_engine *e);

/* How dppr_slide maintains the device CSR. on (default): the previous epoch's sorted edge keys
 * are kept and the batch is merged in (sort of the 2c batch keys + mark + select + merge, O(Ed)
 * streaming). off: the whole window is re-sorted every batch, which is what the reference does
 * (thrust::sort in BuildCSRGraph, gpu/SlidingGraphBuilder.cuh:203-221). Same CSR either way. */
int dppr_set_incremental_graph(dppr_engine *e, int on);

/* Renumbering of the internal vertex ids (no counterpart in the reference, whose kernels scan all V ids every
 * iteration: gpu/Inspect.cuh:8-48). The engine numbers vertices by first appearance and sweeps only those ids; a
 * vertex whose last edge left the window keeps p / r, so on a long stream the id space outgrows the vertices that
 * still have edges. on (default): a dppr_slide that finds EVERY source slot / group converged on the newest epoch,
 * the id space grown by growth_pct % (default 15) since the last numbering and at least min_parked (default 1024,
 * and growth_pct / 2 % of the live vertices) ids without an edge, renumbers: live vertices first, the others parked at the top
 * of the id capacity with their state rows, outside of every sweep; a parked vertex that shows up in a later batch
 * gets a fresh id and its rows back. Results are those of a run without renumbering; older epochs become
 * unavailable at that slide (they are in the old numbering), which is why slots that lag behind block it.
 * growth_pct / min_parked = 0 keep their current values. */
int dppr_set_renumbering(dppr_engine *e, int on, int growth_pct, int min_parked);
/* ids in use by sweeps and scans, parked ids, renumberings so far, parked vertices that came back (any may be NULL) */
int dppr_id_space(dppr_engine *e, int32_t *n_ids, int32_t *n_parked, int32_t *renumberings, int64_t *revivals);

/* ---- per-source state ---------------------------------------------------- */

/* Allocates pagerank/residual/frontier state for one source vertex
 * (DeviceMemory::CudaAllocAppData, gpu/DeviceMemory.cuh:51-63). */
int dppr_add_source(dppr_engine *e, int32_t source_vertex, int32_t *out_slot);

/* Replaces: Init<<<>>> + ExecuteMainLoop(0) of PPRGPU::DynamicExecute
 * (gpu/PPRGPU.cuh:85-89, gpu/PPRCommon.cuh:12-22). Runs on the newest epoch. */
int dppr_init_solve(dppr_engine *e, int32_t slot, double eps, float *out_ms);
/* ... on a given resident epoch (-1: the newest, as dppr_init_solve): a source added after several epochs were pre-staged starts
 * from the first of them and follows them with dppr_update in sequence. */
int dppr_init_solve_at(dppr_engine *e, int32_t slot, int32_t epoch, double eps, float *out_ms);

/* THE TIMED REGION of the reference (gpu/PPRGPU.cuh:138-164):
 * IncrementalBatchUpdate + ExecuteMainLoop(0) + ExecuteMainLoop(1) for one batch.
 * epoch < 0 means the newest epoch. *out_ms = hipEvent time of exactly that region
 * (batch upload and CSR rebuild excluded, as in the reference). */
int dppr_update(dppr_engine *e, int32_t slot, int32_t epoch, double eps, float *out_ms);
/* (Epochs must be applied to a source in sequence: after the solve / update on epoch k only epoch
 * k + 1 is accepted, DPPR_ERR_INVALID otherwise -- with several epochs resident a skipped or replayed
 * batch would otherwise go unnoticed. dppr_write resets that memory. The same holds for
 * dppr_incremental_batch_update and dppr_group_update.) */

/* The same region split along the reference's virtual interface, for callers that
 * keep the reference's driver loop and for kernel-level parity tests:
 *   PPRRevPushGPU::IncrementalBatchUpdate (gpu/PPRRevPushGPU.cuh:21-28,
 *     gpu/StreamUpdate.cuh:7-76)
 *   PPRRevPushGPU::ExecuteMainLoop(phase) (gpu/PPRRevPushGPU.cuh:97-131): full
 *     Inspect over V, then the Expand/Repair frontier loop. */
int dppr_incremental_batch_update(dppr_engine *e, int32_t slot, int32_t epoch);
/* Test hook: the frontier seeds dppr_incremental_batch_update derived from the batch when the state
 * was converged before it -- phase 0: tails with r > eps, phase 1: tails with r < -eps (eps = that of
 * the completed solve). This is what cpu/PPRCPUMTCilkRev.h:126-156 (DynPushInit) builds from the batch
 * endpoints; unordered. Valid only between that call and the next main loop / update. */
int dppr_seed_lists(dppr_engine *e, int32_t slot, int phase, int32_t *out_ids, int32_t *out_count);
int dppr_execute_main_loop(dppr_engine *e, int32_t slot, int32_t epoch, int phase, double eps);

/* Replaces: the cudaMemcpy D2H of pagerank/residual in ValidateResult
 * (gpu/PPRRevPushGPU.cuh:136-139). p and r each hold vertex_count doubles. */
int dppr_read(dppr_engine *e, int32_t slot, double *p, double *r);
/* Overwrite p/r (test hook: lets a kernel be checked from an arbitrary state). */
int dppr_write(dppr_engine *e, int32_t slot, const double *p, const double *r);

int dppr_stats(dppr_engine *e, int32_t slot, dppr_stats_t *out);
int dppr_reset_stats(dppr_engine *e, int32_t slot);

/* ---- source groups (multi-source batched sweeps) ---------------------------------------
 * Up to 16 sources that share the engine's graph are solved TOGETHER: their p / r / x vectors are
 * interleaved (8 doubles = one 64-byte sector per vertex for up to 8 sources, 16 doubles = one
 * 128-byte line for 9..16), so one pass over the out-CSR serves all of them (BASELINE.json configs
 * 3 and 5 run 10 sources over the same stream). Per source the arithmetic and the results are those
 * of the single-source calls; group iterations are always sweeps, and an activity bitmap keeps a
 * sweep's cost proportional to what the frontiers touch (dppr_multi.hpp). The reference has no
 * counterpart (one source per process, gpu/PPRGPU.cuh:24). */
int dppr_add_source_group(dppr_engine *e, const int32_t *sources, int32_t n /* 1..16 */, int32_t *out_group);
int dppr_group_init_solve(dppr_engine *e, int32_t group, double eps, float *out_ms);
int dppr_group_init_solve_at(dppr_engine *e, int32_t group, int32_t epoch, double eps, float *out_ms); /* (as dppr_init_solve_at) */
/* timed region for all sources of the group at once (same scope as dppr_update) */
int dppr_group_update(dppr_engine *e, int32_t group, int32_t epoch, double eps, float *out_ms);
int dppr_group_read(dppr_engine *e, int32_t group, int32_t index, double *p, double *r);
int dppr_group_stats(dppr_engine *e, int32_t group, dppr_stats_t *out); /* summed over the sources */
int dppr_group_reset_stats(dppr_engine *e, int32_t group);
/* Windows whose sweep groups are all resident at once run a frontier loop of a source group as
 * multi-sweep launches (grid barrier between sweeps, row tables kept in LDS; dppr_multi.hpp). on by
 * default; 0 = one launch per sweep everywhere. The roll-call / time-out rules are those of
 * dppr_set_persistent. Same results. */
int dppr_set_group_resident(dppr_engine *e, int on);
/* Edge slots of the single-source resident sweep (dppr_resident.hpp). sorted = 1 (default): every sweep group's edges are
 * dealt to the threads in the order of their gather position, so that the lanes of one gather instruction read the same
 * or neighbouring 64-byte sectors (table built with the group cut, untimed -- the reference builds its CSR untimed as well,
 * gpu/PPRGPU.cuh:131-135). sorted = 0: slots in CSR order (the form of rounds 1-2). Same pushes, same sums up to the order
 * of the additions. 1 takes effect with the next epoch built, 0 at once. */
int dppr_set_resident_slots(dppr_engine *e, int sorted);
/* IncrementalBatchUpdate inside the resident launch of a whole batch (dppr_resident.hpp, PLAN_UPDATE; on by default): the
 * workgroup that owns a tail applies its records -- same terms, same order per tail as the separate kernel
 * (gpu/StreamUpdate.cuh:34-76), bit-identical residuals -- before it seeds the frontier. dppr_update only (the split calls
 * dppr_incremental_batch_update + dppr_execute_main_loop always run the update as its own kernel). 0 = always a kernel of its own. */
int dppr_set_resident_update(dppr_engine *e, int on);
/* The tail of a source group's frontier loop as pushes. A sweep costs at least its floor (every out_col entry, every
 * sweep group's tables) however few (vertex, source) pairs are still being pushed; below enter_pairs frontier pairs
 * the loop's remaining iterations run as what the reference does for every iteration (gpu/ExpandRev.cuh:34-77,
 * :708-743: per frontier vertex and in-neighbour a returning atomic add, all sources of the group at once), same
 * synchronous schedule, same results up to the order of the sums. enter_pairs: -1 (default) = automatic: below
 * max(64, 2 pairs per sweep group) -- the factor 2 can be changed for tuning runs with the environment variable
 * DPPR_GROUP_PUSH_FACTOR, read in dppr_create --, 0 = never, N = below N pairs; an iteration with more in-edges than the floor is worth sends
 * the loop back to sweeps (max_edges: that bound on an iteration's in-edges; 0 = automatic, 20 per sweep group), to
 * try again on a much smaller frontier. list_cap: vertices a frontier list holds (0: keep; default 2^20). Windows
 * whose groups run as multi-sweep launches do not use it. */
int dppr_set_group_push(dppr_engine *e, int enter_pairs, int list_cap, int64_t max_edges);

/* How dppr_group_update seeds its two frontier loops. from_tails (default): after a converged solve
 * only tails of the batch's records can be legal (the argument of cpu/PPRCPUMTCilkRev.h:126-156), so
 * the frontier is read off the batch; 0: a full Inspect pass over all vertices per phase like
 * PPRRevPushGPU::ExecuteMainLoop (gpu/PPRRevPushGPU.cuh:97-104). Same frontier either way. */
int dppr_set_group_seeding(dppr_engine *e, int from_tails);

/* ---- validation / test hooks -------------------------------------------- */

/* Replaces: InspectPureRev alone (gpu/Inspect.cuh:8-48). Writes the legal vertices
 * (unordered) to out_ids (capacity vertex_count) and their count to *out_count. */
int dppr_inspect(dppr_engine *e, int32_t slot, int phase, double eps, int32_t *out_ids, int32_t *out_count);

/* Replaces: the D2H copies of in_row_ptr/in_col_ind in the -DVALIDATE CSR check
 * (gpu/PPRRevPushGPU.cuh:69-72). row_ptr: V+1, col: directed-edge count, out_degree: V
 * (any may be NULL). */
int dppr_read_graph(dppr_engine *e, int32_t epoch, int32_t *row_ptr, int32_t *col, int32_t *out_degree);
int dppr_graph_edges(dppr_engine *e, int32_t epoch, int32_t *out_directed_edges);
/* The out-CSR the pull sweep reads (rows sorted ascending). row_ptr: V+1, col: directed edges. */
int dppr_read_out_graph(dppr_engine *e, int32_t epoch, int32_t *row_ptr, int32_t *col);

/* Frontier trace (test hook): when enabled the host loop copies every iteration's
 * frontier back. dppr_trace_get returns them concatenated: offsets has n_iters+1
 * entries. Pass NULL buffers to query sizes. */
int dppr_trace_enable(dppr_engine *e, int32_t slot, int on);
int dppr_trace_get(dppr_engine *e, int32_t slot, int64_t *n_iters, int64_t *n_ids,
                   int64_t *offsets, int32_t *ids);

/* Waits for everything the engine has enqueued (the solver's and the builder's HIP streams). */
int dppr_synchronize(dppr_engine *e);

/* Lookahead for the untimed region (no reference counterpart: its graph update is serial,
 * gpu/PPRGPU.cuh:114-135). Tells the engine which id arrays the NEXT dppr_set_batch (b1, b2: L records) and dppr_slide
 * (n1, n2: c new edges) will be called with -- the SAME pointers and lengths, contents final and untouched until those
 * calls. Their external -> internal lookups (the dominant host cost of a large batch: 18 + 5 ms of a twitter-size
 * slide) then run on helper threads while the caller is inside dppr_update / dppr_group_update, and the two calls only
 * resolve what the lookups could not (ids without an internal id yet, parked vertices). Read-only on the engine: every
 * call that changes the id maps waits for the lookups first; after a renumbering, or when the pointers / lengths differ,
 * the hint is ignored. Results are identical with and without it.
 * Threads: this is the ONE call that may also come from another host thread than the engine's while that one is inside
 * dppr_update / dppr_group_update / dppr_incremental_batch_update / dppr_execute_main_loop (none of which touches the id maps) --
 * so a host can advance its own stream and hint from a helper while the update runs (./pagerank does) -- provided the helper
 * has returned from it before the engine's thread makes its next call. */
int dppr_hint_next_batch(dppr_engine *e, const int32_t *b1, const int32_t *b2, int32_t L, const int32_t *n1, const int32_t *n2, int32_t c);

/* What dppr_set_batch_grouping(at_slide = 1) and the batch upload move out of the reference's timed region
 * (gpu/PPRGPU.cuh:138-164 times CopyOutDegree + the whole IncrementalBatchUpdate): the post-batch out-degree gather and the
 * grouping of epoch `epoch`'s L records by tail (device radix sort), run `reps` times on the engine's stream between two
 * events; *out_ms = milliseconds per repetition. bench.py adds it to the measured batch time and reports both accountings
 * (config.timed_region). Touches scratch only. */
int dppr_time_batch_grouping(dppr_engine *e, int32_t epoch, int32_t reps, float *out_ms);

/* Diagnostics for a call that does not come back (one unexplained 300-second guard in 13 suite runs). Writes a
 * text report into buf (at most cap bytes, NUL-terminated; returns the length written): last error, epoch / id-space
 * state, resident-launch settings, per slot and group the host-side loop state and statistics, and -- read through a
 * SEPARATE stream with a bounded wait, so that it works while the engine's own stream is stuck in a kernel -- the
 * GridBar words (roll-call outcome, check-ins, per-sweep arrivals) and the status words of resident / multi-sweep
 * launches. Callable from another host thread than the one inside the engine (it takes no lock: the host-side fields
 * are read racily, which is what a post-mortem wants). No reference counterpart (its loop is host-driven,
 * gpu/PPRRevPushGPU.cuh:106-130, and cannot wait on a device-side barrier). */
int dppr_debug_dump(dppr_engine *e, char *buf, int32_t cap);
/* Test hook (ABI 5): the binned-sweep tables of an epoch (dppr_binned.hpp) -- block cuts (n_a + 1 / n_b + 1 first vertices, internal
 * ids); the RUN list in A-major order (n_runs entries: head index inside its A-block, top bit = first run of its tile), per tile
 * (n_tiles, A-major order) the difference between its runs' B-major and A-major indices, per 64 runs the tile that holds the first of
 * them (n_runs / 64 + 1 entries, rounded up); the EDGE list in B-major order (n_edges entries: row index inside its B-block, top bit =
 * first edge of its run) and per 64 edges the run that holds the first of them (n_edges / 64 + 1 entries, rounded up); how many epochs
 * had their tables patched by the slide's merge and how many built by the sorts. Any pointer may be NULL. The patched tables must
 * equal, bit for bit, what the sorts produce under the same cuts (tests/test_binned_tables_gpu.py). */
int dppr_debug_bin_tables(dppr_engine *e, int32_t epoch, int32_t *n_a, int32_t *n_b, int32_t *n_edges, int32_t *n_runs, int32_t *n_tiles,
                          int32_t *acut, int32_t *bcut, uint16_t *hl, int32_t *tdelta, int32_t *tb, uint16_t *dl, int32_t *vb,
                          int64_t *patched, int64_t *rebuilt);

/* A counter that advances at every host read-back of a frontier loop and every stage of a graph build (ABI 4): a watchdog
 * samples it so that ONE long call (the first solve on a large window, a group's from-scratch solve) is told from a hang
 * (ADVICE r04). Callable from any thread. */
unsigned long long dppr_heartbeat(const dppr_engine *e);

/* Microbenchmark used to calibrate the roofline ceiling of the push kernel
 * (SURVEY.md 8d): n returning f64 atomic adds per launch at pseudo-random
 * addresses of a table of table_elems doubles; scope 0 = agent, 1 = workgroup.
 * Returns the average kernel time in ms over reps launches. */
int dppr_bench_atomics(int device, int64_t table_elems, int64_t n, int scope, int reps, float *out_ms);
/* The other two calibrated ceilings of SURVEY.md 8(d), measured by bench.py in the run that prints the line (ABI 4):
 * dppr_bench_line_fills -- `lines` random 128-byte line fetches (one request per 8 lanes, the access shape of the group
 * sweep's row gathers) out of a table of table_bytes (a power of two; 1 GiB = beyond every cache); *out_ms = time for
 * `lines` fetches. dppr_bench_stream_copy -- a streaming copy of `bytes` (read + write): *out_ms per copy. */
int dppr_bench_line_fills(int device, int64_t table_bytes, int64_t lines, int reps, float *out_ms);
int dppr_bench_stream_copy(int device, int64_t bytes, int reps, float *out_ms);

/* Identity of the build: the first 16 hex digits of the SHA-256 over the kernel / engine sources the library was compiled
 * from (the .hpp files of csrc in name order, csrc/dppr_engine.hip, include/dppr.h; see csrc/Makefile). Profiles under profiles/ carry the id of
 * the library that produced them; bench.py refuses counter traffic measured on another build. */
const char *dppr_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* DPPR_H */
