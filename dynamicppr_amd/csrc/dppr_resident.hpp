// dppr_resident.hpp -- a run of dense frontier iterations as ONE resident launch.
#pragma once

#include "dppr_common.hpp"
#include "dppr_pull.hpp" // STAMP (diagnostic builds)

namespace dppr {

#define PSTAMP(i)                \
    do {                         \
        if (it == 10) STAMP(i);  \
    } while (0)

// ---------------------------------------------------------------------------
// a4+a5, DENSE iterations on graphs whose sweep fits the chip in one wave of workgroups.
//
// On a small window (configs[1]: ~0.6 M edges) one sweep of k_pull_iter moves ~18 MB and is not
// bandwidth bound at all: its time is the dependent chain  grp_tile -> out_row_ptr -> out_col ->
// x[col] -> stores, plus the launch ramp, the kernel-end cache write-back and the gap to the next
// dispatch -- and a batch needs ~80 such iterations (gpu/PPRRevPushGPU.cuh:106-130 pays a blocking
// D2H on top of every one of them).
//
// k_pull_resident runs up to n_iter consecutive sweeps in one launch. Every workgroup owns ONE
// sweep group (<= PB consecutive vertices, cut by the graph builder) for the whole launch, so
// everything that does not change between iterations is computed once and kept on chip:
//   * row starts / lengths and their workgroup-wide prefix (LDS),
//   * the owner row, the out_col entry and the divisor (outdeg+1) of every edge slot (registers;
//     the group's concatenated edge list is dealt to the PB threads with stride PB, so long rows
//     are spread over all waves and out_col is read coalesced -- once; pre-reducing the stretches of
//     a long row inside a wave before the LDS atomics was measured and is slower than the atomics),
//   * residual / pagerank / snapshot value of the thread's own vertex (registers; r and p go back
//     to memory when the launch ends).
// One iteration is then: gather x[col] (the only dependent memory hop), LDS-atomic the terms
// into the owners' sums, repair + threshold + next snapshot exactly as k_pull_iter::finish does,
// and store x_new[v]. The arithmetic per vertex is k_pull_iter's: the same terms
// (1-ALPHA)*x[u]/(outdeg(v)+1), summed into residual[v] (gpu/ExpandRev.cuh:70-73), the same repair
// (:708-743) and legal-push test.
//
// SYNCHRONISATION BETWEEN ITERATIONS IS BY DATA FLOW. (A first version separated iterations with a
// grid barrier: stores complete ~1.7 K clocks -> arrive -> everybody has arrived ~5.4 K -> only
// then the next gathers ~8 K; three memory-side round trips in a row.) Here an iteration's
// gathers wait for exactly what they need -- the values themselves:
//   * FRESH VECTORS: x_j of a launch lives in its own vector A[j] of an arena (132 vectors; a launch runs at most 128
//     sweeps). Sweep g reads x_g from A[g] and ends with two stores per thread: x_{g+1} into A[g+1], and the EMPTY mark of
//     its entry of A[g+3] for the sweep after next (A[1..3] are marked on entry);
//   * a marked entry holds X_EMPTY, a NaN bit pattern no computed value has; a gather that reads
//     X_EMPTY is simply repeated until the owner has stored this round's value;
//   * because every address is written ONCE per launch (after its mark), a gather's first attempt is an ORDINARY load: a copy
//     of the line in the L1 or in the XCD's L2 -- fetched a moment ago by another workgroup of the same XCD -- holds either
//     the value or the mark, never an older value. Only a repeat goes to the memory side (agent scope, sc1). The 32
//     workgroups of an XCD share one fetch of a hub's line instead of pulling it through the fabric 32 times. (Rounds 1-2
//     rotated three vectors; an address then carried a new value every third sweep, every gather had to be an agent-scope
//     access that misses the L2 by construction, and the gather phase was bound by the fabric: 0.5 M lines per sweep on
//     the configs[1] stand-in, 8 K of an iteration's 10.6 K clocks.) The arena's addresses are reused by the next launch;
//     what an earlier launch left in the L1s / L2s is dropped by the acquire every kernel dispatch begins with -- the
//     mechanism every per-iteration kernel of this engine relies on when it reads, with ordinary loads, the snapshot vector
//     the previous kernel's workgroups wrote on other XCDs (an explicit agent-scope acquire by every wave on top of it was
//     measured: + 35 us per launch);
//   * arrival counters (one memory-side atomic per workgroup and sweep, fire and forget) tell
//     "every workgroup has finished sweep h" and carry the number of legal vertices it produced.
//     Nobody waits on them in the common case: during sweep g the first wave of a workgroup looks
//     at the arrivals of sweep g-1 -- issued a whole gather phase earlier, so normally complete --
//     to learn the size of the frontier sweep g consumes, i.e. whether the loop is over. Every workgroup
//     evaluates the same sums at the same iteration number, so all of them stop together.
// Why a gather never sees anything but X_EMPTY or the right value: a thread marks its entry of A[g+3] at the end of sweep g;
// in sweep g+1 the workgroup waits for all its outstanding stores (s_waitcnt vmcnt(0)) and only then passes the
// __syncthreads after which any of its waves stores x_{g+2}. A consumer reads A[g+3] no earlier than its sweep g+3, which
// it starts after it consumed x_{g+2} of every vertex it depends on -- stored after the mark completed; and nobody has read
// (or cached) a line of A[g+3] before that in this launch. A 128-byte line belongs to ONE wave's 512-byte store, so the
// entries a consumer does not depend on were marked by the same completed store. Progress: sweep g of any workgroup needs
// only values and arrivals of sweep g-1.
// Accesses to ONE address serialise at the memory side (~10 ns each; 242 workgroups polling the
// same 16 words took 4 us), so every counter exists BAR_REPS times: a workgroup arrives on all
// replicas (one 16-lane atomic instruction) and reads only the replica of its own sixteen. Two
// counter sets, for odd and even sweeps, keep a fast workgroup's next arrival out of the sums a
// slow one is still reading.
// The snapshot vectors are exchanged between XCDs (each has its own L2) with agent-scope
// accesses (sc1) and explicit waits for store completion; a workgroup barrier alone does NOT wait
// for outstanding stores on gfx942/950 (workgroup-scope release omits vmcnt), and an agent-scope
// release per wave (buffer_wbl2) costs ~75 us per iteration for the 4096 waves.
//
// CO-RESIDENCY. Waiting on other workgroups only terminates if every workgroup of the launch is
// resident. The engine sizes the grid to the occupancy the runtime reports, and the kernel verifies
// it with a ROLL-CALL before it changes anything that matters: every workgroup checks in on entry,
// workgroup 0 watches the check-ins and publishes READY with one compare-and-swap on the word
// everybody reads after their set-up; a workgroup that waits longer than the time limit swaps that
// word to BAR_ABORT instead. One word decides, so all workgroups agree; an aborted launch leaves
// the state as it found it and the host goes on with per-iteration launches (dppr_engine.hip).
// After a successful roll-call all workgroups are running and stay resident; a time-out later on
// cannot be a residency problem and is reported as a device fault (DPPR_ERR_HIP).
// ---------------------------------------------------------------------------
constexpr int BAR_SUBS = 16;
constexpr int BAR_REPS = 16;       // replicas of every arrival counter
constexpr int BAR_POLL_SLEEP = 4;  // s_sleep units (64 clocks) between two polls of the arrivals
constexpr unsigned long long BAR_ABORT = ~0ull;
constexpr unsigned long long BAR_READY = 1ull;
constexpr int PERSIST_SLOTS = 4; // edge slots per thread kept in registers (PB * 4 edges per group)
constexpr unsigned long long X_EMPTY = 0x7FF8DEADBEEFCAFEull;
constexpr int RES_MAX_SWEEPS = 128;               // sweeps of one launch at most
constexpr int RES_VECTORS = RES_MAX_SWEEPS + 4;   // vectors of the arena: x_0 .. x_128 and the marks three ahead
// SLOT TABLES (round 3). The gathers are the iteration's cost, and what the fabric moves for them is one 64-byte sector per
// distinct sector a wave instruction touches -- 0.99 sectors per edge with the slots in CSR order on hashed ids (configs[1]
// stand-in: the lanes of an instruction walk a few short rows whose columns are spread over the whole id range). Nothing ties
// a slot to CSR order: every slot carries its owner row. The graph build therefore sorts every group's edge list by gather
// position (k_res_slots: untimed, with the group cut), so that the lanes of one instruction read neighbouring positions -- the
// many edges of a group that lead to the same hub become ONE request, neighbours share sectors (0.88 sectors per edge by
// simulation, tools/r03/resident_slots_sim.py) -- and the lanes of one LDS-atomic instruction belong to different owner rows.
// Table entry: (gather position << 10) | owner row inside the group. No table (res_pk == nullptr): CSR order, as in round 2.
// Measured on the configs[1] stand-in: 0.566 -> 0.515 ms per batch. (Giving the 64 .. 16 K vertices of largest in-degree a
// second, PACKED home behind the vectors -- eight hubs per sector, 0.75 sectors per edge by the same simulation -- was built
// and measured twice, hub copies stored by their owners and staged through LDS into full-sector stores: 0.56 - 0.59 ms, worse
// with every hub added. 39 % of all gathers then go to a few hundred consecutive sectors, i.e. to a handful of memory
// channels; hashed ids spread the hot values over all of them.)
constexpr int RES_SORT_MAX = 16384;    // edges of one group the table build sorts in LDS (larger groups: no table for the epoch)
constexpr int RES_ID_LIMIT = 1 << 22;  // ids a table entry can name (22 + 10 bits)

struct alignas(128) BarWord {
    unsigned long long w;
    unsigned long long pad[15];
};
struct GridBar {                          // zeroed by the host before every launch
    BarWord gen;                          // roll-call outcome: 0 pending, BAR_READY, BAR_ABORT
    BarWord roll[BAR_SUBS];               // roll-call check-ins of the workgroups with blockIdx % BAR_SUBS == s
    BarWord sub[2][BAR_REPS][BAR_SUBS];   // per sweep parity and replica: arrivals << 32 | cumulative legal count
};
// status word of a launch
constexpr int PERSIST_ABORTED = 1 << 30;   // the roll-call failed, nothing was changed
constexpr int PERSIST_FAULT = 1 << 29;     // a wait timed out after a successful roll-call
constexpr int PERSIST_CONVERGED = 1 << 28; // the frontier emptied; all snapshot vectors are all zero again
constexpr int PERSIST_PHASE1 = 1 << 26;    // a launch that runs both phases had started phase 1
constexpr int PERSIST_SWEEPS = (1 << 16) - 1; // low bits: loop position g
// plan of a launch
constexpr int PLAN_SEED = 1;  // take the first snapshot from the registers: {v : legal(residual[v])} (valid after a converged solve)
constexpr int PLAN_BOTH = 2;  // when phase 0 is over, seed phase 1 the same way and go on
constexpr int PLAN_UPDATE = 4; // apply the batch's records to the residuals first (IncrementalBatchUpdate inside the launch)

// IncrementalBatchUpdate inside the launch (PLAN_UPDATE). A record changes the residual of its TAIL only, and a tail belongs to
// exactly one sweep group: the workgroup that owns it applies the group's records -- the batch's records sorted by tail at
// slide time (dppr_engine.hip: epoch_group_records), cut into the sweep groups' ranges by k_res_rec_ranges -- to the residuals it
// has just loaded, before it seeds the frontier from them. Same terms, same order per tail, same expressions as k_su_apply_fused
// (dppr_update.hpp; gpu/StreamUpdate.cuh:34-76, cpu/PPRCPUMTCilkRev.h:108-124): bit-identical residuals. What it saves is a
// kernel of ~20 us and the gap in front of it -- 6 % of a configs[1] batch, 9 % of a configs[0] one.
//
// Round 5 -- the records taken RAW (b1 != nullptr; the default accounting, in which nothing of IncrementalBatchUpdate is
// prepared at slide time): every workgroup reads the batch's L tails once (coalesced, 16 independent loads per lane, served
// by the L2s after the first workgroup), keeps the records whose tail is one of ITS rows -- found in batch order by ballots,
// ranked by (row, batch index) in LDS -- and proceeds as above. CopyOutDegree (gpu/StreamUpdate.cuh:7-17) is the length of
// the tail's row in the epoch's out-CSR, which the row's thread holds anyway. No radix sort, no range table, no degree
// array: the whole of gpu/StreamUpdate.cuh:7-76 runs inside the launch. A group that owns more than PB records (a hub's
// tail in a large batch) or a tail beyond the last group calls the launch off before anything is changed (cnt[4] = 1;
// the host applies the update with its own kernels and launches again).
constexpr int RES_RAW_STEPS = 16; // 64-record steps per wave: L <= RES_RAW_STEPS * PB
struct ResUpdate {
    const int *rng;          // per sweep group: first record (n_groups + 1 entries); nullptr: no update, or raw records
    const uint32_t *tails;   // the records' tails, ascending
    const uint32_t *order;   // ... and their indices in the batch (stable)
    const int *b2;           // heads
    const uint8_t *ins;      // 1 = insertion
    const int *deg_after;    // out-degree of the tail after the batch
    int source;              // the slot's source vertex
    const int *b1;           // raw form: the tails in batch order (rng / tails / order / deg_after unused)
    int L;                   // ... and their number
};
// first record of every sweep group's range (tails[] is sorted); rng[n_groups] = first record whose tail lies beyond the last
// group; stat[0] = the largest range (atomicMax), stat[1] = rng[n_groups]
__global__ __launch_bounds__(256) void k_res_rec_ranges(const uint32_t *__restrict__ tails, int L, const int *__restrict__ grp_tile,
                                                        int n_groups, int *__restrict__ rng, int *__restrict__ stat) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g > n_groups) return;
    auto lower = [&](int gi) { // first record with tail >= first vertex of group gi
        const long long key = (long long)grp_tile[gi] * WAVE;
        int lo = 0, hi = L;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((long long)tails[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int a = lower(g);
    rng[g] = a;
    if (g < n_groups) atomicMax(&stat[0], lower(g + 1) - a);
    else stat[1] = a;
}

__device__ __forceinline__ unsigned long long bar_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool bar_cas(unsigned long long *p, unsigned long long expect, unsigned long long desired) {
    return __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long xb_load(const double *p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a gather's FIRST attempt: an ordinary load, which may be served by the L1 / the XCD's L2 (see "FRESH VECTORS" below)
__device__ __forceinline__ unsigned long long xc_load(const double *p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void xb_store(double *p, unsigned long long bits) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int PB>
__global__ __launch_bounds__(PB) void k_pull_resident(int V, const int *__restrict__ grp_tile,
                                                      const int *__restrict__ out_row_ptr,
                                                      const int *__restrict__ out_col, double *b0, double *arena,
                                                      long long stride, double *r, double *p, int *cnt, int cur0, int phase,
                                                      double eps, IterStats *stats, int *log, int n_iter, GridBar *bar,
                                                      int *status, unsigned long long limit_ticks, int rollcall_extra,
                                                      int plan, const uint32_t *__restrict__ res_pk, ResUpdate upd) {
    constexpr int NW = PB / WAVE;
    constexpr int S = PERSIST_SLOTS;
    __shared__ int s_scan[PB + 1];
    __shared__ int s_rs[PB];
    __shared__ double s_acc[2][PB];
    __shared__ int s_wtot[NW];
    __shared__ int s_cnt[NW];
    __shared__ unsigned long long s_edges[NW];
    __shared__ unsigned s_next[2]; // {roll-call outcome / no fault, size of the frontier of the current sweep}
    __shared__ int s_fault;
    __shared__ int s_rtail[PB];    // PLAN_UPDATE: the group's records (tail as a row of the group, insertion flag)
    __shared__ uint8_t s_rins[PB];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned G = gridDim.x;
    const unsigned subs_used = G < (unsigned)BAR_SUBS ? G : (unsigned)BAR_SUBS;
    const unsigned long long n_sub = lane < (int)subs_used ? (G - lane + BAR_SUBS - 1) / BAR_SUBS : 0;
    const int my_rep = (blockIdx.x / BAR_SUBS) % BAR_REPS;
    const unsigned long long t_entry = wall_clock64();

    // ---- static part: the group's vertices and edge slots; the two vectors that are not the input
    // start EMPTY
    const int t0 = grp_tile[blockIdx.x], t1 = grp_tile[blockIdx.x + 1];
    const int v = t0 * WAVE + tid;
    const bool valid = tid < (t1 - t0) * WAVE && v < V;
    auto Av = [&](int j) { return arena + (long long)j * stride; }; // the vector of x_j
    if (valid) {
        xb_store(Av(1) + v, X_EMPTY);
        xb_store(Av(2) + v, X_EMPTY);
        xb_store(Av(3) + v, X_EMPTY);
    }
    if (tid == 0) s_fault = 0;
    int rs = 0, d = 0;
    double rv = 0.0, xv = 0.0, pv = 0.0;
    if (valid) {
        rs = out_row_ptr[v];
        d = out_row_ptr[v + 1] - rs;
        rv = r[v];
        pv = p[v];
        if (!(plan & PLAN_SEED)) xv = b0[v];
    }
    bool overflow = false; // (PLAN_UPDATE, raw records) this group cannot apply its records inside the launch
    if (plan & PLAN_UPDATE) {
        // the group's records: terms in parallel, then one lane per tail applies its records in batch order
        int nrec, row = -1, rdeg = 0, u = 0;
        double term = 0.0;
        bool rin = false;
        if (upd.b1) {
            // ---- raw records: this group's, in batch order (wave w looks at records [w * span, (w + 1) * span))
            const int L = upd.L;
            const int span = ((L + NW - 1) / NW + WAVE - 1) / WAVE * WAVE;
            const int vlo = t0 * WAVE, vhi = blockIdx.x + 1 == gridDim.x ? 0x7fffffff : t1 * WAVE; // (the last group also sees tails beyond it)
            uint64_t mine[RES_RAW_STEPS];
            int tl[RES_RAW_STEPS];
#pragma unroll
            for (int k = 0; k < RES_RAW_STEPS; ++k) {
                const int i = w * span + k * WAVE + lane;
                tl[k] = (k * WAVE < span && i < L) ? upd.b1[i] : -1;
            }
            int wn = 0;
#pragma unroll
            for (int k = 0; k < RES_RAW_STEPS; ++k) {
                mine[k] = __ballot(tl[k] >= vlo && tl[k] < vhi);
                wn += __popcll(mine[k]);
            }
            if (lane == 0) s_cnt[w] = wn;
            __syncthreads();
            int base = 0;
            nrec = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int c = s_cnt[k];
                base += k < w ? c : 0;
                nrec += c;
            }
            overflow = nrec > PB;
#pragma unroll
            for (int k = 0; k < RES_RAW_STEPS; ++k) {
                if ((mine[k] >> lane) & 1ull) {
                    const int pos = base + mbcnt(mine[k]);
                    if (pos < PB) {
                        s_scan[pos] = w * span + k * WAVE + lane; // record index
                        s_rs[pos] = tl[k] - vlo;                  // its tail as a row of the group
                    }
                }
                base += __popcll(mine[k]);
            }
            __syncthreads();
            if (overflow) nrec = 0;
            int rec = 0, key_row = 0x7fffffff;
            if (tid < nrec) {
                rec = s_scan[tid];
                key_row = s_rs[tid];
                overflow = key_row >= PB || vlo + key_row >= V; // (a tail beyond the groups: an id the cut does not cover)
            }
            overflow = __syncthreads_or(overflow ? 1 : 0) != 0;
            if (overflow) nrec = 0;
            // stable order by row: position = #{j : row_j < row or (row_j == row and j < tid)}
            int rank = 0;
            if (tid < nrec)
                for (int j = 0; j < nrec; ++j) {
                    const int rj = s_rs[j];
                    rank += (rj < key_row || (rj == key_row && j < tid)) ? 1 : 0;
                }
            __syncthreads(); // (s_scan / s_rs are read; s_scan now carries the rows' lengths = post-batch out-degrees)
            s_scan[tid] = d;
            s_rtail[tid] = -1;
            s_rins[tid] = 0;
            __syncthreads();
            if (tid < nrec) {
                const int uu = vlo + key_row;
                s_rtail[rank] = key_row;
                s_rins[rank] = upd.ins[rec] != 0 ? 1 : 0;
                s_acc[1][rank] = ONE_MINUS_ALPHA * p[upd.b2[rec]] - p[uu];
            }
            s_acc[0][tid] = rv;
            __syncthreads();
            if (tid < nrec) {
                row = s_rtail[tid];
                u = vlo + row;
                rdeg = s_scan[row];
            }
        } else {
        const int lo = upd.rng[blockIdx.x];
        nrec = upd.rng[blockIdx.x + 1] - lo; // <= PB (the host checked)
        if (tid < nrec) {
            u = (int)upd.tails[lo + tid];
            const int rec = (int)upd.order[lo + tid];
            rin = upd.ins[rec] != 0;
            rdeg = upd.deg_after[rec];
            term = ONE_MINUS_ALPHA * p[upd.b2[rec]] - p[u];
            row = u - t0 * WAVE;
        }
        s_rtail[tid] = row;
        s_rins[tid] = rin ? 1 : 0;
        s_acc[1][tid] = term;
        s_acc[0][tid] = rv;
        __syncthreads();
        }
        if (tid < nrec && (tid == 0 || s_rtail[tid - 1] != row)) {
            int end = tid, delta = 0; // extent of the tail's records and its net degree change (post-batch minus pre-batch)
            while (end < nrec && s_rtail[end] == row) {
                delta += s_rins[end] ? 1 : -1;
                ++end;
            }
            int dg = rdeg - delta; // RevertOutDegree (gpu/StreamUpdate.cuh:18-33)
            double ru = s_acc[0][row];
            const double src_term = ALPHA * (upd.source == u ? 1.0 : 0.0);
            for (int k = tid; k < end; ++k) {
                const double add = s_acc[1][k] - ALPHA * ru + src_term;
                if (s_rins[k]) {
                    dg++;
                    ru += add / (double)(dg + 1) / ALPHA;
                } else {
                    dg--;
                    ru -= add / (double)(dg + 1) / ALPHA;
                }
            }
            s_acc[0][row] = ru;
        }
        __syncthreads();
        rv = s_acc[0][tid];
        __syncthreads(); // (s_acc is written again below)
    }
    unsigned F = (unsigned)__hip_atomic_load(cnt + cur0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (plan & PLAN_SEED) {
        // Inspect + the head of ExpandUnifiedRev (gpu/PPRRevPushGPU.cuh:97-104, gpu/ExpandRev.cuh:34-42) on
        // the registers: the frontier is {v : legal(residual[v])}, x_0[v] = residual[v] for those,
        // pagerank[v] += ALPHA * residual[v]. After a converged solve plus a stream update these are the
        // vertices k_su_apply listed.
        const bool lg0 = valid && legal(rv, phase, eps);
        xv = lg0 ? rv : 0.0;
        if (lg0) pv = pv + ALPHA * rv;
        const int wl = __popcll(__ballot(lg0));
        if (lane == 0) s_cnt[w] = wl;
    }
    if (valid) xb_store(Av(0) + v, (unsigned long long)__double_as_longlong(xv)); // x_0, given or seeded
    const int incl = wave_inclusive_scan(d);
    if (lane == WAVE - 1) s_wtot[w] = incl;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the EMPTY marks (and seeds) are in place before this workgroup checks in
    __syncthreads();
    if (tid == 0 && overflow) { // the launch is called off before anybody is told to go: workgroup 0 publishes READY only after EVERY check-in
        cnt[4] = 1;
        (void)bar_cas(&bar->gen.w, 0ull, BAR_ABORT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the swap has been performed before this workgroup checks in)
    }
    if (tid == 0) { // roll-call: this workgroup is running and has initialised its entries
        unsigned long long seeds = 0;
        if (plan & PLAN_SEED)
            for (int k = 0; k < NW; ++k) seeds += (unsigned)s_cnt[k];
        __hip_atomic_fetch_add(&bar->roll[blockIdx.x % BAR_SUBS].w, (1ull << 32) | seeds, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    int woff = 0, Eg = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int t = s_wtot[k];
        woff += k < w ? t : 0;
        Eg += t;
    }
    s_scan[tid] = woff + incl - d;
    s_rs[tid] = rs;
    if (tid == 0) s_scan[PB] = Eg;
    s_acc[0][tid] = rv;
    __syncthreads();
    auto owner_of = [&](int e) {
        int lo = 0, hi = PB;
#pragma unroll
        for (int st = PB; st > 1; st >>= 1) {
            const int mid = (lo + hi) >> 1;
            if (s_scan[mid] <= e) lo = mid; else hi = mid;
        }
        return lo;
    };
    const int gbase = s_rs[0]; // the group's rows are consecutive: its edges are out_col / res_pk [gbase, gbase + Eg)
    auto slot_of = [&](int e, int *o, int *c) {
        if (res_pk) {
            const uint32_t pk = res_pk[gbase + e];
            *o = (int)(pk & 1023u);
            *c = (int)(pk >> 10);
        } else {
            *o = owner_of(e);
            *c = out_col[s_rs[*o] + (e - s_scan[*o])];
        }
    };
    int own[S], col[S];
    double den[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int e = tid + k * PB;
        own[k] = -1;
        col[k] = 0;
        den[k] = 1.0;
        if (e < Eg) {
            int o, c;
            slot_of(e, &o, &c);
            own[k] = o;
            col[k] = c;
            den[k] = (double)(s_scan[o + 1] - s_scan[o] + 1);
        }
    }

    // ---- roll-call outcome
    if (w == 0) {
        if (blockIdx.x == 0) {
            unsigned polls = 0;
            bool all_here = false;
            unsigned long long word = 0;
            for (;;) {
                word = lane < (int)subs_used ? bar_load(&bar->roll[lane].w) : 0;
                // rollcall_extra > 0 (tests only) makes the roll-call wait for a workgroup that does not exist
                if (__ballot((word >> 32) >= n_sub + (lane == 0 ? (unsigned long long)rollcall_extra : 0ull)) == ~0ull) {
                    all_here = true;
                    break;
                }
                if ((polls++ & 31u) == 0 && (bar_load(&bar->gen.w) != 0 || wall_clock64() - t_entry > limit_ticks)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // READY carries the seed count of all workgroups (PLAN_SEED) in its high half
            const unsigned seeds = (unsigned)__builtin_amdgcn_readlane(wave_inclusive_scan((int)(unsigned)word), WAVE - 1);
            if (lane == 0)
                (void)bar_cas(&bar->gen.w, 0ull, all_here ? (BAR_READY | ((unsigned long long)seeds << 32)) : BAR_ABORT);
        }
        if (lane == 0) {
            unsigned long long word;
            unsigned polls = 0;
            while ((word = bar_load(&bar->gen.w)) == 0) {
                if ((polls++ & 31u) == 0 && wall_clock64() - t_entry > limit_ticks)
                    (void)bar_cas(&bar->gen.w, 0ull, BAR_ABORT);
                __builtin_amdgcn_s_sleep(1);
            }
            s_next[0] = word != BAR_ABORT;
            s_next[1] = (plan & PLAN_SEED) ? (unsigned)(word >> 32) : F;
        }
    }
    __syncthreads();
    if (!s_next[0]) { // not co-resident: nothing the engine looks at was changed (the arena is scratch)
        if (blockIdx.x == 0 && tid == 0) *status = PERSIST_ABORTED;
        return;
    }

    // arrivals of sweep h: counter set h & 1, (h >> 1) + 1 rounds of that set so far. Returns false on a
    // time-out; on success *count = legal vertices summed over all workgroups, cumulative for the set.
    auto wait_arrivals = [&](int h, unsigned long long first_word, unsigned *cum) {
        const unsigned long long rounds = (unsigned long long)((h >> 1) + 1);
        unsigned long long word = first_word;
        unsigned polls = 0;
        const unsigned long long t_start = wall_clock64();
        for (;;) {
            if (__ballot((word >> 32) >= n_sub * rounds) == ~0ull) break;
            if ((polls++ & 63u) == 63u && wall_clock64() - t_start > limit_ticks + 100000000ull) return false;
            __builtin_amdgcn_s_sleep(BAR_POLL_SLEEP);
            if (lane < (int)subs_used) word = bar_load(&bar->sub[h & 1][my_rep][lane].w);
        }
        *cum = (unsigned)__builtin_amdgcn_readlane(wave_inclusive_scan((int)(unsigned)word), WAVE - 1);
        return true;
    };

    unsigned long long edges = 0;
    unsigned Cpar[2] = {0u, 0u};
    int logged = 0;
    bool fault = false, converged = false;
    int cur_phase = phase;
    const int last_phase = (plan & PLAN_BOTH) ? 1 : phase;
    int g = 0;
    for (; g < n_iter; ++g) {
        const double *xin = Av(g);
        double *xout = Av(g + 1);
        const int it = g; // (PSTAMP)
        PSTAMP(0);

        // first wave: the arrivals of sweep g-1, read BEFORE the gathers are issued (loads return in
        // order, so this is back long before they are)
        unsigned long long fw = 0;
        if (w == 0 && g >= 1 && lane < (int)subs_used) fw = bar_load(&bar->sub[(g - 1) & 1][my_rep][lane].w);
        unsigned long long gb[S];
#pragma unroll
        for (int k = 0; k < S; ++k) gb[k] = own[k] >= 0 ? xc_load(xin + col[k]) : 0ull;
        if (w == 0 && g >= 1) {
            unsigned cum = 0;
            const bool ok = wait_arrivals(g - 1, fw, &cum);
            const unsigned fsz = cum - Cpar[(g - 1) & 1];
            Cpar[(g - 1) & 1] = cum;
            if (lane == 0) {
                s_next[1] = fsz;
                if (!ok) s_fault = 1;
            }
            // (ok: every workgroup is done with sweep g-1, nobody reads x_{g-1} any more -- the licence for the reset at the
            // head of the NEXT iteration)
        }
        // gathers whose owner has not stored this round's value yet are repeated
#ifdef DPPR_STAMPS
        if (it == 10 && tid == PB - 1) { // (last wave: when its first attempt was back)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g_stamps[blockIdx.x * 8 + 6] = clock64();
        }
#endif
        {
            unsigned polls = 0;
            const unsigned long long t_start = wall_clock64();
            for (;;) {
                bool pend = false;
#pragma unroll
                for (int k = 0; k < S; ++k) pend |= gb[k] == X_EMPTY;
                if (__ballot(pend) == 0) break;
                if ((polls++ & 63u) == 63u && (s_fault || wall_clock64() - t_start > limit_ticks + 100000000ull)) {
                    s_fault = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int k = 0; k < S; ++k)
                    if (gb[k] == X_EMPTY) gb[k] = xb_load(xin + col[k]);
            }
#ifdef DPPR_STAMPS
            if (it == 10 && tid == PB - 1) g_stamps[blockIdx.x * 8 + 7] = polls;
#endif
        }
        PSTAMP(1);
        double *acc = s_acc[g & 1];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const double xg = __longlong_as_double((long long)gb[k]);
            const bool nz = xg != 0.0;
            const double term = ONE_MINUS_ALPHA * xg / den[k];
            if (nz) lds_add(&acc[own[k]], term);
            edges += (unsigned long long)__popcll(__ballot(nz));
        }
        for (int e0 = S * PB; e0 < Eg; e0 += PB) { // edges beyond the register slots
            const int e = e0 + tid;
            double xe = 0.0;
            int o = 0;
            if (e < Eg) {
                int c;
                slot_of(e, &o, &c);
                const double *src = xin + c;
                unsigned long long bits = xc_load(src);
                unsigned polls = 0;
                while (bits == X_EMPTY && !s_fault) {
                    // 2^20 polls of >= 2 us each: seconds, like the wall-clock limits of the other waits (reading the
                    // clock in THIS loop costs the whole kernel 12 %: the compiler schedules the gather loop around it)
                    if ((polls++ & 0xFFFFFu) == 0xFFFFFu) s_fault = 1; // a device fault, reported below
                    __builtin_amdgcn_s_sleep(8);
                    bits = xb_load(src);
                }
                xe = __longlong_as_double((long long)bits);
            }
            const bool nz = xe != 0.0 && !(xe != xe);
            if (nz) lds_add(&acc[o], ONE_MINUS_ALPHA * xe / (double)(s_scan[o + 1] - s_scan[o] + 1));
            edges += (unsigned long long)__popcll(__ballot(nz));
        }
        // (everybody's reset store is complete before the workgroup goes on to store x_{g+1})
        PSTAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PSTAMP(3);
        if (s_fault) {
            fault = true;
            break;
        }
        F = s_next[1]; // size of the frontier this sweep consumed
        if (blockIdx.x == 0 && tid == 0) log[g] = (int)F;
        logged = g + 1;
        if (F == 0 && cur_phase == last_phase) { // the loop is over; this sweep had nothing to do (every workgroup sees the same F)
            converged = true;
            break;
        }
        double rn = acc[tid];
        if (F == 0) {
            // phase 0 is over (x_g is all zero, the gathers above found nothing) and the launch goes on
            // with phase 1: this step seeds it from the registers the way PLAN_SEED does, in place of a
            // sweep (the vectors rotate as in any other iteration).
            cur_phase = 1;
            rn = rv;
        } else if (xv != 0.0) {
            rn -= xv;
        }
        const bool lg = valid && legal(rn, cur_phase, eps);
        if (valid) {
            rv = rn;
            xv = lg ? rn : 0.0;
            xb_store(xout + v, (unsigned long long)__double_as_longlong(xv));
            // The vector of x_{g+3} gets its EMPTY marks now, fire and forget next to the snapshot store: the NEXT iteration's
            // wait for outstanding stores covers both, one iteration before the vector is written.
            xb_store(Av(g + 3) + v, X_EMPTY);
            if (lg) pv = pv + ALPHA * rn;
        }
        s_acc[(g + 1) & 1][tid] = rv;
        const int wl = __popcll(__ballot(lg));
        if (lane == 0) s_cnt[w] = wl;
        PSTAMP(4);
        __syncthreads();
        PSTAMP(5);
        if (w == 0) { // arrive for sweep g (nobody waits for this now)
            const int part = lane < NW ? s_cnt[lane] : 0;
            const int tot = __builtin_amdgcn_readlane(wave_inclusive_scan(part), WAVE - 1);
            if (lane < BAR_REPS)
                __hip_atomic_fetch_add(&bar->sub[g & 1][lane][blockIdx.x % BAR_SUBS].w, (1ull << 32) | (unsigned)tot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- the launch ends. Converged: x_g is all zero, the other two vectors are cleaned. Out of
    // sweeps: one real barrier (everybody finished the last sweep), its count is the live frontier
    // size; the vector with x_{n_iter} stays, the other two are cleaned.
    if (!fault && !converged) {
        if (w == 0) {
            unsigned cum = 0;
            unsigned long long fw = lane < (int)subs_used ? bar_load(&bar->sub[(g - 1) & 1][my_rep][lane].w) : 0;
            const bool ok = wait_arrivals(g - 1, fw, &cum);
            if (lane == 0) {
                s_next[1] = cum - Cpar[(g - 1) & 1];
                if (!ok) s_fault = 1;
            }
        }
        __syncthreads();
        fault = s_fault != 0;
        F = s_next[1];
    }
    if (valid) {
        if (!fault) b0[v] = xv; // x_g, the live snapshot (all zero if converged)
        r[v] = rv;
        p[v] = pv;
    }
    if (blockIdx.x == 0 && tid == 0) {
        for (int k = logged; k < n_iter; ++k) log[k] = 0;
        cnt[0] = (int)F;
        cnt[1] = 0;
        cnt[2] = 0;
        *status = g | (fault ? PERSIST_FAULT : 0) | (cur_phase != phase ? PERSIST_PHASE1 : 0) |
                  ((converged || (!fault && F == 0 && cur_phase == last_phase)) ? PERSIST_CONVERGED : 0);
    }
    stat_add_edges<NW>(stats, edges, s_edges);
}

// ---- slot table of a resident-size epoch (part of the untimed graph build, next to the group cut) ----
// k_res_slots (one workgroup per sweep group): the group's edges as (gather position << 10 | owner row), sorted (bitonic, in LDS).
__global__ __launch_bounds__(1024) void k_res_slots(int NV, const int *__restrict__ grp_tile, const int *__restrict__ out_row_ptr,
                                                    const int *__restrict__ out_col, uint32_t *__restrict__ res_pk) {
    __shared__ uint32_t s_key[RES_SORT_MAX];
    const int tid = threadIdx.x;
    const int t0 = grp_tile[blockIdx.x], t1 = grp_tile[blockIdx.x + 1];
    const int va = min(t0 * WAVE, NV), vb = min(t1 * WAVE, NV);
    const int e0 = out_row_ptr[va], Eg = out_row_ptr[vb] - e0;
    if (Eg <= 0 || Eg > RES_SORT_MAX) return; // (the host does not use the table of an epoch with such a group)
    int n2 = 64;
    while (n2 < Eg) n2 <<= 1;
    for (int i = tid; i < n2; i += 1024) {
        uint32_t key = 0xFFFFFFFFu;
        if (i < Eg) {
            int lo = va, hi = vb; // owner row: the last v in [va, vb) with out_row_ptr[v] <= e0 + i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (out_row_ptr[mid] <= e0 + i) lo = mid; else hi = mid;
            }
            key = ((uint32_t)out_col[e0 + i] << 10) | (uint32_t)(lo - va);
        }
        s_key[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = s_key[i], b = s_key[l];
                    if (((i & k) == 0) == (a > b)) {
                        s_key[i] = b;
                        s_key[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < Eg; i += 1024) res_pk[e0 + i] = s_key[i];
}

} // namespace dppr
