// dppr_multi.hpp -- f2: multi-source batched sweeps.
//
// Several source vertices that share one device graph (BASELINE.json configs 3 and 5 run 10 of
// them) are solved TOGETHER. Their state is interleaved GW-wide, p/r/x[v] = GW doubles, where
// GW = 2 * ceil(sources / 2): a row holds the sources and at most one padding double (10 sources:
// 80-byte rows; until round 3 rows were 64 or 128 bytes whatever the count, so 10 sources moved
// 37.5 % padding). Eight lanes serve a row: SPL = 1 double per lane up to 8 sources, SPL = 2 beyond;
// lanes whose doubles lie beyond the row (j * SPL >= GW) issue no memory operation at all. A sweep
// reads every out_col entry once and one row per ACTIVE edge for all sources -- a single-source
// sweep pulls a whole sector for 8 useful bytes.
//
// Per source the arithmetic is that of k_pull_iter (dppr_pull.hpp), i.e. what the pushes u -> v of
// gpu/ExpandRev.cuh:70-73 plus the repair of :708-743 leave in residual[v]:
//     rn = residual[v] + sum_{u in out(v), u pushed} (1.0-ALPHA) * residual[u] / (outdeg(v)+1);  rn -= (v's own push)
// then the legal-push test and the next snapshot (x_new[v] = rn, pagerank[v] += ALPHA*rn).
//
// Kernels: k_ginit, k_gseed_dense / k_gseed_tails (the frontier of a loop), k_gtables (a sweep group's
// row tables, once per epoch), k_gsweep (one sweep per launch, or -- on windows whose groups are all
// resident -- a run of sweeps with a grid barrier in between).
//
// How a sweep is laid out on the machine (k_gsweep):
//  * ACTIVITY BITMAP. bit v of `act` says "some source has x[v] != 0". It is one bit per vertex
//    (LiveJournal stand-in: 150 KB; friendster: 16 MB), so it lives in L2 where x (75 MB .. 17 GB)
//    does not. An edge whose head is inactive costs its 4-byte out_col entry and one bit test, no
//    gather; a vertex that received nothing and is not in the frontier is not touched at all (no
//    residual read, no snapshot write): x[v] is only meaningful where the bit is set, and nothing
//    is ever zero-filled. Late iterations of a batch (small frontiers) therefore cost the stream of
//    out_col, not a full sweep -- the reference's gpu/ExpandRev.cuh:70-77 has no counterpart because
//    it pushes, at the price of one atomic per edge.
//  * OCTETS. Eight consecutive lanes serve one edge at a time: lane j holds sources j*SPL.., so
//    the gather of x[u] is ONE coalesced 64-byte (128-byte) request, not 64 lanes x 64 bytes.
//  * EDGE-BALANCED SLICES. A workgroup owns a sweep group (<= 1024 consecutive vertices, cut by the
//    graph builder for equal weight); their out-rows are one contiguous range of out_col, which is
//    split evenly over the workgroup's 128 octets whatever the degree distribution (a hub's row
//    is simply shared by many octets). An octet walks its slice in order with a cursor over the
//    group's non-empty rows (compacted in LDS), keeps the running sum of the current row in
//    registers and adds it to the row's LDS accumulator once, when the row (or the slice) ends:
//    one LDS atomic per row piece instead of one per edge and source.
//  * The vertex side (repair, threshold, next snapshot) is octet-cooperative too: 64/128-byte
//    coalesced accesses, only for vertices that were touched.
//  * ONE COPY OF THE RESIDUAL. While a vertex is active its snapshot row IS its residual row:
//    x[v][s] holds the current residual of EVERY source s (a source is in the frontier iff that value
//    is legal -- the gather applies the legal-push test instead of "non-zero"), and residual[v] is
//    not read or written until the vertex leaves the frontier, when the row is written back. A
//    frontier vertex costs x read + x_new write + pagerank read/write per sweep, not those plus a
//    residual read and write. For a legal source the new residual is exactly the sum of the sweep's
//    adds (what PPRRevPushGPUFF's `residual[u] = 0` at the snapshot gives, gpu/Inspect.cuh:51-65);
//    between loops every bit is clear and residual[] is complete.
//  * PAGERANK EVERY OTHER SWEEP (round 4). pagerank[v] += ALPHA * x is owed for every snapshot value that is pushed
//    (gpu/ExpandRev.cuh:42). Crediting it when the snapshot is taken costs a read and a write of the pagerank row per
//    active vertex and sweep -- 22 % of a dense LiveJournal sweep once the kernel kept its loads in flight. A vertex
//    phase has TWO consecutive snapshot values of a vertex in registers: the one being pushed in this sweep (cur) and
//    the one it creates (rn). So sweeps alternate: a DEFERRING sweep does not touch pagerank at all, the next,
//    CREDITING one adds ALPHA * cur (where cur was pushed) and then ALPHA * rn (where rn will be) -- the same two
//    additions in the same order as before, bit for bit, with one read-modify-write instead of two. The host tracks
//    whether the live snapshot is still owed (`credit`); a loop that ends has an empty frontier and owes nothing, a
//    loop that goes on as pushes (dppr_gpush.hpp) hands the debt to the first push iteration.
#pragma once

#include "dppr_kernels.hpp"

namespace dppr {

constexpr int GS_MAX = 16;      // sources per group (SPL = 2); 8 with SPL = 1
constexpr int OCT = 8;          // lanes that serve one edge / one vertex together

struct SrcN {
    int s[GS_MAX]; // internal source vertex per state lane, -1 = unused lane
};

__device__ __forceinline__ unsigned oct_mask(uint64_t ballot) { // the 8 ballot bits of this lane's octet
    return (unsigned)(ballot >> (lane_id() & ~(OCT - 1))) & 0xffu;
}

// Row geometry: GW doubles per vertex (even, 2 .. 16), SPL doubles per lane of the octet. Lane j of an octet owns
// doubles [j * SPL, j * SPL + SPL) of a row; it is LIVE iff they exist. (A full row -- GW = 8 * SPL -- has no dead lane
// and the test folds away.)
__host__ __device__ constexpr int row_width(int n_sources) { return n_sources <= 2 ? 2 : (n_sources + 1) / 2 * 2; }
__host__ __device__ constexpr int row_spl(int gw) { return gw > OCT ? 2 : 1; }
// The SNAPSHOT rows (x: the only rows that are gathered at random) keep a power-of-two stride so that a row never
// straddles a 128-byte line: the L2 fills whole lines from the fabric (TCC_EA0_RDREQ_128B is every read request of
// a sweep, profiles/r04_pmc_*), and an 80-byte row at an 80-byte stride costs two fills on every other gather.
// pagerank / residual rows are only ever streamed (consecutive vertices) and stay compact.
#ifndef DPPR_X_COMPACT
#define DPPR_X_COMPACT 0 // (A/B: 1 = snapshot rows at the compact stride too)
#endif
__host__ __device__ constexpr int x_stride(int gw) { return DPPR_X_COMPACT ? gw : gw > 8 ? 16 : gw > 4 ? 8 : gw; }
template <int SPL, int GW>
__device__ __forceinline__ bool oct_live(int j) {
    static_assert(GW >= 2 && GW <= OCT * SPL && GW % 2 == 0 && (SPL == 1 || GW > OCT), "row width / lane split");
    if constexpr (GW == OCT * SPL) return true;
    else return j * SPL < GW;
}

// r = e_s per source, p = 0. One thread per (vertex, state lane).
__global__ __launch_bounds__(BLOCK) void k_ginit(double *__restrict__ p, double *__restrict__ r, int V, int gw, SrcN src) {
    const int64_t n = (int64_t)V * gw;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int v = (int)(i / gw), s = (int)(i % gw);
        p[i] = 0.0;
        r[i] = (src.s[s] == v) ? 1.0 : 0.0;
    }
}

// Dense seeding of a phase: Inspect (gpu/Inspect.cuh:8-48) + the snapshot head of ExpandUnifiedRev
// (gpu/ExpandRev.cuh:34-42) for every source at once: where some source is legal, x[v][s] =
// legal(r) ? r : 0 for all s, p += ALPHA*x, and the vertex's activity bit is set. Writes the
// complete bitmap (every word up to V). Octet per vertex.
template <int SPL, int GW>
__global__ __launch_bounds__(BLOCK) void k_gseed_dense(int V, const double *__restrict__ r, double *__restrict__ x,
                                                       double *__restrict__ p, uint32_t *__restrict__ act, int phase,
                                                       double eps, int *__restrict__ cnt_out) {
    __shared__ int s_cnt[GS_MAX];
    if (threadIdx.x < GS_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int j = threadIdx.x & (OCT - 1);
    const bool live = oct_live<SPL, GW>(j);
    int nleg[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) nleg[q] = 0;
    uint8_t *act8 = reinterpret_cast<uint8_t *>(act);
    const int vper = BLOCK / OCT; // vertices per workgroup pass
    const int Vpad = (V + 7) & ~7;
    for (int vb = blockIdx.x * vper; vb < Vpad; vb += gridDim.x * vper) { // workgroup-uniform trip count
        const int v = vb + (int)threadIdx.x / OCT;
        bool any = false;
        double rv[SPL];
        bool lg[SPL];
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            rv[q] = v < V && live ? r[(size_t)v * GW + j * SPL + q] : 0.0;
            lg[q] = v < V && live && legal(rv[q], phase, eps);
            any |= lg[q];
            nleg[q] += lg[q] ? 1 : 0;
        }
        const uint64_t bal = __ballot(any);
        const unsigned m = oct_mask(bal);
        if (m && live) {
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                const size_t i = (size_t)v * GW + j * SPL + q;
                x[(size_t)v * x_stride(GW) + j * SPL + q] = rv[q]; // the whole residual row moves to the snapshot (dppr_multi.hpp header)
                if (lg[q]) p[i] = p[i] + ALPHA * rv[q];
            }
        }
        // one activity byte per wave pass: bit k = octet k's vertex
        if (lane_id() == 0) {
            unsigned byte = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) byte |= ((bal >> (k * OCT)) & 0xffull) ? (1u << k) : 0u;
            const int v0 = vb + wave_id() * 8;
            if (v0 < Vpad) act8[v0 >> 3] = (uint8_t)byte;
        }
    }
#pragma unroll
    for (int q = 0; q < SPL; ++q)
        if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
    __syncthreads();
    if (threadIdx.x < GW && s_cnt[threadIdx.x]) atomicAdd(&cnt_out[threadIdx.x], s_cnt[threadIdx.x]);
}

// Seeding from the batch tails (valid after a converged solve plus a stream update: only tails of
// batch records can have left [-eps, eps]; cpu/PPRCPUMTCilkRev.h:126-156 seeds from the batch
// endpoints for the same reason). skeys = the batch's tails, sorted; an octet takes the first
// record of each tail. `act` must be all zero on entry.
template <int SPL, int GW>
__global__ __launch_bounds__(BLOCK) void k_gseed_tails(const uint32_t *__restrict__ skeys, int L, const double *__restrict__ r,
                                                       double *__restrict__ x, double *__restrict__ p,
                                                       uint32_t *__restrict__ act, int phase, double eps,
                                                       int *__restrict__ cnt_out) {
    __shared__ int s_cnt[GS_MAX];
    if (threadIdx.x < GS_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int j = threadIdx.x & (OCT - 1);
    const bool live = oct_live<SPL, GW>(j);
    int nleg[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) nleg[q] = 0;
    const int per = BLOCK / OCT;
    const int Lpad = (L + per - 1) / per * per;
    for (int i0 = blockIdx.x * per; i0 < Lpad; i0 += gridDim.x * per) {
        const int i = i0 + (int)threadIdx.x / OCT;
        int u = -1;
        if (i < L) {
            u = (int)skeys[i];
            if (i > 0 && (int)skeys[i - 1] == u) u = -1; // not the group leader
        }
        bool any = false;
        double rv[SPL];
        bool lg[SPL];
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            rv[q] = u >= 0 && live ? r[(size_t)u * GW + j * SPL + q] : 0.0;
            lg[q] = u >= 0 && live && legal(rv[q], phase, eps);
            any |= lg[q];
            nleg[q] += lg[q] ? 1 : 0;
        }
        if (oct_mask(__ballot(any))) {
            if (live) {
#pragma unroll
                for (int q = 0; q < SPL; ++q) {
                    const size_t k = (size_t)u * GW + j * SPL + q;
                    x[(size_t)u * x_stride(GW) + j * SPL + q] = rv[q];
                    if (lg[q]) p[k] = p[k] + ALPHA * rv[q];
                }
            }
            if (j == 0) atomicOr(&act[u >> 5], 1u << (u & 31));
        }
    }
#pragma unroll
    for (int q = 0; q < SPL; ++q)
        if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
    __syncthreads();
    if (threadIdx.x < GW && s_cnt[threadIdx.x]) atomicAdd(&cnt_out[threadIdx.x], s_cnt[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// Row tables of the sweep groups, built ONCE per epoch (untimed graph build) and loaded by every sweep.
// What a workgroup of k_gsweep needs before it can walk its group's edges -- the group's extents, its
// non-empty rows compacted (first edge + local vertex index) and, per octet, the row its slice begins in --
// is a pure function of the epoch's graph and of the group cut. Per group, GT_STRIDE(NVX) ints:
//   [0..7]  v0, nv, E0, Eg, per (edges per octet slice), ncomp (non-empty rows), -, -
//   [8 ..]  cstart[NVX + 1]  (entries beyond ncomp hold Eg),  ostart[GNT / 8],  cvid[NVX] as 16-bit values
// (k_gsweep used to rebuild this per group and per sweep: two dependent scalar loads, the row extents, a
// ballot compaction across the workgroup, two integer divisions per row and three barriers; on the
// LiveJournal stand-in that set-up was half of the 55-60 us a near-empty sweep costs.)
// ---------------------------------------------------------------------------
constexpr int GNT = 1024; // threads per workgroup of k_gsweep
constexpr int GT_HDR = 8;
__host__ __device__ constexpr int GT_STRIDE(int nvx) { return (GT_HDR + nvx + 1 + GNT / OCT + nvx / 2 + 3) / 4 * 4; }

template <int NVX>
__global__ __launch_bounds__(GNT) void k_gtables(int V, const int *__restrict__ grp_tile, int n_groups,
                                                 const int *__restrict__ out_row_ptr, int *__restrict__ tables) {
    constexpr int NW = GNT / WAVE, NOCT = GNT / OCT, EB = 8;
    __shared__ int s_wcnt[NW];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        __syncthreads();
        int *T = tables + (size_t)g * GT_STRIDE(NVX);
        const int t0 = grp_tile[g], t1 = grp_tile[g + 1];
        const int v0 = t0 * WAVE;
        const int nv = min((t1 - t0) * WAVE, V - v0);
        const int E0 = out_row_ptr[v0];
        const int Eg = out_row_ptr[v0 + nv] - E0;
        int rs = 0, d = 0;
        if (tid < nv) {
            rs = out_row_ptr[v0 + tid] - E0;
            d = out_row_ptr[v0 + tid + 1] - E0 - rs;
        }
        const int per = ((Eg + NOCT - 1) / NOCT + EB - 1) / EB * EB;
        const uint64_t ne = __ballot(d > 0);
        if (lane == 0) s_wcnt[w] = __popcll(ne);
        __syncthreads();
        int woff = 0, ncomp = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int c = s_wcnt[k];
            woff += k < w ? c : 0;
            ncomp += c;
        }
        int *cstart = T + GT_HDR, *ostart = T + GT_HDR + NVX + 1;
        unsigned short *cvid = reinterpret_cast<unsigned short *>(T + GT_HDR + NVX + 1 + NOCT);
        // defaults first (entries beyond ncomp, octets without a slice), the real entries after the barrier
        if (tid <= NVX && tid >= ncomp) cstart[tid] = Eg;
        if (tid == 0 && NVX >= GNT) cstart[NVX] = Eg; // (NVX == GNT: entry NVX has no thread of its own)
        if (tid < NOCT) ostart[tid] = 0;
        if (tid < NVX && tid >= ncomp) cvid[tid] = 0;
        if (tid == 0) {
            T[0] = v0; T[1] = nv; T[2] = E0; T[3] = Eg; T[4] = per; T[5] = ncomp; T[6] = 0; T[7] = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the defaults have landed before another thread overwrites one
        __syncthreads();
        if (d > 0) {
            const int idx = woff + mbcnt(ne);
            cstart[idx] = rs;
            cvid[idx] = (unsigned short)tid;
            const int o_first = (rs + per - 1) / per, o_last = min((rs + d + per - 1) / per - 1, NOCT - 1);
            for (int o = o_first; o <= o_last; ++o) ostart[o] = idx; // octets whose slice begins inside this row
        }
    }
}

// Memory accesses of the sweep: plain in a one-sweep launch; in a multi-sweep launch everything another
// workgroup wrote during the launch is read past the L1 (agent scope, `sc1`) and every store another
// workgroup will read is agent-scope too (see k_pull_resident for the measured visibility rules).
template <bool COH>
__device__ __forceinline__ double gs_ld(const double *p) {
    if constexpr (COH)
        return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else
        return *p;
}
template <bool COH>
__device__ __forceinline__ uint32_t gs_ldu(const uint32_t *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void gs_st(double *p, double v) {
    if constexpr (COH)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <bool COH>
__device__ __forceinline__ void gs_stu(uint32_t *p, uint32_t v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Cache policy of a one-sweep launch (DPPR_GS_NT, a bit mask; the multi-sweep form keeps its agent-scope accesses).
// A dense sweep is bound by 128-byte line fills of the XCDs' L2s (tools/r04/gather_probe.hip: ~50 G random lines/s chip-wide
// whether the table sits in the Infinity Cache or in HBM; 190 G rows/s out of the L2), and an XCD's 4 MiB L2 sees ~25x its
// size per sweep, so under plain LRU only rows gathered every few microseconds survive. The ids are numbered in blocks of
// falling in-degree (dppr_idspace.hpp): rows below `hot_rows` take a third to a half of all gathers. They are loaded with
// the default policy, everything that is touched once per sweep is marked non-temporal so that it does not push them out:
//   bit 0: gathers of rows >= hot_rows;  bit 1: the vertex phase's row loads;  bit 2: the vertex phase's row stores
// Measured twice (before and after the spill fix) and off by default: bits 1 + 2 cost 12.5-12.7 against 11.6-11.8 ms per batch; bit 0
// gains 7-11 % in a pure gather loop (tools/r04/policy_probe.hip: only `nt` does anything, sc0 / sc1 do not) and nothing in the sweep
// (11.65 / 11.83 against 11.64 / 11.85): with every gather redirected to 4 096 rows (all L2 hits, -DDPPR_WHATIF=16) a launch still
// costs 137 us against 157 -- the sweep is bound by its chain of dependent round trips per group, not by the fills.
#ifndef DPPR_GS_NT
#define DPPR_GS_NT 0
#endif
#ifndef DPPR_WHATIF
#define DPPR_WHATIF 0 // (timing experiments that compute WRONG results: 1 = no pagerank traffic, 2 = no own-row read of active vertices)
#endif
template <bool COH, int BIT>
__device__ __forceinline__ double gs_ld_once(const double *p) {
    if constexpr (!COH && (DPPR_GS_NT & BIT)) return __builtin_nontemporal_load(p);
    else return gs_ld<COH>(p);
}
template <bool COH, int BIT>
__device__ __forceinline__ void gs_st_once(double *p, double v) {
    if constexpr (!COH && (DPPR_GS_NT & BIT)) __builtin_nontemporal_store(v, p);
    else gs_st<COH>(p, v);
}

// status word of a multi-sweep launch
constexpr int GSM_ABORTED = 1 << 30;   // the roll-call failed: nothing was changed
constexpr int GSM_FAULT = 1 << 29;     // a wait timed out after a successful roll-call
constexpr int GSM_CONVERGED = 1 << 28; // every frontier emptied
constexpr int GSM_SWEEPS = (1 << 16) - 1;

// One frontier iteration (ExpandUnifiedRev + RepairFrontierRev) for all sources of a group.
// NVX = vertices per sweep group (the LDS accumulators are NVX x GW doubles: at most 64 KB, two
// 1024-thread workgroups per CU).
//
// MULTI = a RUN of iterations as one launch, for windows whose groups are all resident at once (one
// workgroup per group): the group's row tables are built once and stay in LDS, iterations are
// separated by a grid barrier (arrival counters of dppr_resident.hpp, replicated, polled by one wave;
// ~5 us, which is what a dependent kernel boundary costs too -- what disappears is the launch ramp,
// the per-launch table set-up chain and the no-op launches at the end of a chunk). Per iteration g the
// frontier sizes of all sources are row g of `mlog` (row 0 = cnt_in): the loop ends when a row is
// all zero. Co-residency is verified by the same roll-call as k_pull_resident's before anything is
// changed; a failed roll-call leaves everything untouched and the host goes on with one-sweep launches.
// CM: how pagerank is credited -- 0 a deferring sweep, 1 a crediting one (one-sweep launches: the host alternates the two
// instantiations; a deferring sweep loads one row per vertex instead of two, so it finishes ALL of an octet's vertices in one
// step -- one memory round trip in its vertex phase instead of two), 2 decided per sweep inside the launch (multi-sweep form).
template <int SPL, int GW, int NVX, bool MULTI, int CM>
__global__ __launch_bounds__(GNT, MULTI ? 4 : 8) void k_gsweep(int V, const int *__restrict__ gtab, int n_groups,
                                                   const int *cnt_in, int hot_rows,
                                                   const int *__restrict__ out_col, double *x_a, double *x_b,
                                                   uint32_t *act_a, uint32_t *act_b, double *r, double *p, int *cnt_out,
                                                   int *cnt_zero, int phase, double eps, IterStats *__restrict__ stats,
                                                   int *log_slot, int n_iter, GridBar *bar, int *status,
                                                   unsigned long long limit_ticks, int rollcall_extra, int credit0,
                                                   int *q_take, int *q_zero) {
    constexpr int NOCT = GNT / OCT, WORDS = NVX / 32, XS = x_stride(GW);
    static_assert(sizeof(double) * NVX * GW <= 64 * 1024, "two workgroups per CU");
    constexpr int EB = 8;        // edges an octet tests per step (one per lane)
    constexpr int GB = EB / SPL; // ... and gathers per sub-step (registers: GB x SPL doubles)
    // vertices an octet finishes per step. A wait for loaded values also waits for every older store, so a second
    // step costs the store latency again: the multi-sweep form (one workgroup per CU is enough there: 128 VGPRs)
    // requests ALL its rows before any is stored; the one-sweep form has 64 VGPRs and takes steps of 4 / 2.
    // (the one-sweep 16-wide form with 128 VGPRs and all rows at once, one workgroup per CU: 19.05 vs 18.65 ms per
    // LiveJournal batch -- the halved occupancy costs more than the vertex phase gains)
#ifndef DPPR_GS_FU
#define DPPR_GS_FU (4 / SPL)
#endif
    constexpr int FU_WANT = CM == 0 && SPL == 2 ? 2 * (DPPR_GS_FU) : (DPPR_GS_FU); // (8-wide rows finish 4 vertices per step in either mode: 8 at once spill)
    constexpr int FU = MULTI ? NVX / NOCT : FU_WANT <= NVX / NOCT ? FU_WANT : NVX / NOCT;
    static_assert(MULTI == (CM == 2), "the multi-sweep form decides per sweep, a one-sweep launch is compiled for its mode");
    static_assert(NVX % NOCT == 0 && (NVX / NOCT) % FU == 0, "vertex phase covers the group in whole steps");
    __shared__ double s_acc[NVX * GW];   // per vertex and source: sum of this sweep's adds (zero between groups)
    __shared__ int s_cstart[NVX + 1];    // non-empty rows of the group, compacted: first edge (relative)
    __shared__ unsigned short s_cvid[NVX]; // ... and the row's local vertex index
    __shared__ int s_ostart[NOCT];       // compacted row in which each octet's slice begins
    __shared__ uint32_t s_actin[WORDS], s_actout[WORDS], s_touched[WORDS];
    __shared__ int s_cnt[GS_MAX];
    __shared__ unsigned long long s_edges;
    __shared__ int s_flag[2]; // MULTI: {go on (roll-call ok / frontier not empty), fault}
    __shared__ int s_ticket;  // one-sweep launches: the group this workgroup takes next (groups beyond the grid are dealt by a counter)
    const int tid0 = threadIdx.x, lane = lane_id(), w = wave_id();
    const int j = tid0 & (OCT - 1);
    // Everything a thread derives from its index (LDS addresses of its bitmap words, row bases, masks) is invariant over the
    // group loop; hoisted out of it, those values were what the 64-register budget spilled (17 VGPRs in round 3), and
    // every reload from scratch is a vector-memory load behind an `s_waitcnt vmcnt(0)` -- i.e. behind every row load in
    // flight: the vertex phase's two vertices per step were loaded one after the other. opaque() hides the index from
    // the optimiser inside the loop: the few integer operations are redone per phase, nothing is kept live.
    auto opaque = [](int v) {
#ifndef DPPR_NO_OPAQUE // (A/B switch: the round-3 code generation)
        asm volatile("" : "+v"(v));
#endif
        return v;
    };

    // frontier sizes of the sources; the group iterates while ANY of them is non-empty
    bool stamp_dense = false; // (diagnostic builds: a one-sweep launch is stamped when its frontier is dense)
    if constexpr (!MULTI) {
        const int my_cnt = lane < GW ? cnt_in[lane] : 0;
        if (blockIdx.x == 0 && tid0 < GW) {
            cnt_zero[tid0] = 0;
            log_slot[tid0] = my_cnt;
        }
        if (blockIdx.x == 0 && tid0 == 0) *q_zero = 0; // the NEXT launch's group counter (before the early return: a no-op launch keeps the rotation intact)
        if (__ballot(my_cnt != 0) == 0) return;
#ifdef DPPR_STAMPS
#if defined(DPPR_STAMP_MID) // (a "hovering" sweep: between V / 16 and V / 2 frontier pairs -- few pairs on hub heads that still reach most rows)
        {
            const long long tot = __builtin_amdgcn_readlane(wave_inclusive_scan(my_cnt), WAVE - 1);
            stamp_dense = tot * 16 > (long long)V && tot * 2 < (long long)V;
        }
#elif defined(DPPR_STAMP_SPARSE)
        stamp_dense = (long long)__builtin_amdgcn_readlane(wave_inclusive_scan(my_cnt), WAVE - 1) * 64 < (long long)V;
#else
        stamp_dense = (long long)__builtin_amdgcn_readlane(wave_inclusive_scan(my_cnt), WAVE - 1) > 4ll * V;
#endif
#endif
    }
    (void)stamp_dense;
#ifdef DPPR_STAMPS
    if (stamp_dense) STAMP(7);
    if (!MULTI && stamp_dense && tid0 == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + 5] = wall_clock64(); // (100 MHz, same on every CU)
#endif
    for (int k = tid0; k < NVX * GW; k += GNT) s_acc[k] = 0.0;
    if (tid0 < WORDS) {
        s_actout[tid0] = 0u;
        s_touched[tid0] = 0u;
    }
    if (tid0 < GS_MAX) s_cnt[tid0] = 0;
    if (tid0 == 0) {
        s_edges = 0ull;
        s_flag[0] = 1;
        s_flag[1] = 0;
    }
    int nleg[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) nleg[q] = 0;
    unsigned ecount = 0;

    // ---- MULTI: roll-call (every workgroup of the launch is running) before anything is changed
    const unsigned G = gridDim.x;
    const unsigned subs_used = G < (unsigned)BAR_SUBS ? G : (unsigned)BAR_SUBS;
    const unsigned long long n_sub = lane < (int)subs_used ? (G - lane + BAR_SUBS - 1) / BAR_SUBS : 0;
    const int my_rep = (blockIdx.x / BAR_SUBS) % BAR_REPS;
    if constexpr (MULTI) {
        const unsigned long long t_entry = wall_clock64();
        __syncthreads();
        if (tid0 == 0)
            __hip_atomic_fetch_add(&bar->roll[blockIdx.x % BAR_SUBS].w, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            if (blockIdx.x == 0) {
                unsigned polls = 0;
                bool all_here = false;
                for (;;) {
                    const unsigned long long word = lane < (int)subs_used ? bar_load(&bar->roll[lane].w) : 0;
                    if (__ballot((word >> 32) >= n_sub + (lane == 0 ? (unsigned long long)rollcall_extra : 0ull)) == ~0ull) {
                        all_here = true;
                        break;
                    }
                    if ((polls++ & 31u) == 0 && (bar_load(&bar->gen.w) != 0 || wall_clock64() - t_entry > limit_ticks)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (lane == 0) (void)bar_cas(&bar->gen.w, 0ull, all_here ? BAR_READY : BAR_ABORT);
            }
            if (lane == 0) {
                unsigned long long word;
                unsigned polls = 0;
                while ((word = bar_load(&bar->gen.w)) == 0) {
                    if ((polls++ & 31u) == 0 && wall_clock64() - t_entry > limit_ticks) (void)bar_cas(&bar->gen.w, 0ull, BAR_ABORT);
                    __builtin_amdgcn_s_sleep(1);
                }
                s_flag[0] = word != BAR_ABORT;
            }
        }
        __syncthreads();
        if (!s_flag[0]) {
            if (blockIdx.x == 0 && tid0 == 0) *status = GSM_ABORTED;
            return;
        }
    }

    // A group's work is a chain of dependent memory round trips (group table -> row extents -> out_col ->
    // activity bits -> x -> residuals), which is what a sweep of a SMALL window costs. The chain is kept
    // short by asking early: the next group's extents while this one is processed, the next step's bits
    // and the step after's out_col entries while this step's gathers are in flight, pagerank together
    // with the residual.
    constexpr int STRIDE = GT_STRIDE(NVX);
    // the next group's header (extents, slice length) is requested while this group is processed
    int hv0 = 0, hnv = 0, hE0 = 0, hEg = 0, hper = 0;
    if ((int)blockIdx.x < n_groups) {
        const int *H = gtab + (size_t)blockIdx.x * STRIDE;
        hv0 = H[0]; hnv = H[1]; hE0 = H[2]; hEg = H[3]; hper = H[4];
    }
    int sweeps_done = 0;
    bool converged = false, fault = false;
    // One-sweep launches: a workgroup's first group is its block index, every further one a ticket from a device counter
    // (the groups are cut for equal edge weight, not equal time: with a fixed stride the slots of the last round idle for
    // a fifth of a sweep; the counter's round trip hides behind the edge phase). ONE counter on purpose: returning atomics on
    // one word serialise at ~11 ns, which is 34 us for 3 075 tickets and most of a near-empty sweep's 47-52 us -- but eight
    // counters, one per XCD (with and without taking from the others' once dry), made those sweeps 42-44 us and every larger
    // one 3-7 % slower (groups handed out in global order keep the chip on neighbouring rows): 12.07 -> 12.21-12.37 ms per batch.
    // Nor does it pay to deal the groups of a SPARSE sweep (fewer than ids / 16 frontier pairs; the count is known at launch) by a
    // fixed stride without any ticket: 11.90 against 11.68 ms, and worse the higher the threshold (12.04 / 12.26 / 12.60).
    for (int g = blockIdx.x; g < n_groups;) { // workgroup-uniform loop (MULTI: one group per workgroup)
        __syncthreads(); // the previous group's tables are no longer read; the initial fills are in place
        int ticket = 0, g_next = n_groups;
        if (!MULTI && tid0 == 0) ticket = (int)gridDim.x + atomicAdd(q_take, 1);
        const int *T = gtab + (size_t)g * STRIDE;
        const int tid = opaque(tid0);
        const int j = tid & (OCT - 1), oid = tid / OCT;
        const bool live = oct_live<SPL, GW>(j);
        const int v0 = hv0, nv = hnv, E0 = hE0, Eg = hEg, per = hper; // nv <= NVX: the builder cuts these groups for this kernel
        // the group's row tables (built once per epoch, k_gtables) -> LDS; the first sweep's activity words and
        // out_col entries are requested in the same round trip
        const int e_begin = oid * per, e_end = min(Eg, e_begin + per);
        const int *cols = out_col + E0;
        {
            const int c0 = tid < NVX ? T[GT_HDR + tid] : 0;
            const int os = tid < NOCT ? T[GT_HDR + NVX + 1 + tid] : 0;
            const unsigned short cv = tid < NVX ? reinterpret_cast<const unsigned short *>(T + GT_HDR + NVX + 1 + NOCT)[tid] : 0;
            if (tid < WORDS) s_actin[tid] = tid * 32 < nv ? gs_ldu<MULTI>(act_a + (v0 >> 5) + tid) : 0u;
            if (tid < NVX) {
                s_cstart[tid] = c0;
                s_cvid[tid] = cv;
            }
            if (tid < NOCT) s_ostart[tid] = os;
            if (tid == 0) s_cstart[NVX] = Eg; // (entry NVX is Eg whatever ncomp is: rows are never more than NVX)
        }
        int mycol = e_begin + j < e_end ? ld_stream(&cols[e_begin + j]) : -1;
        int ncol = e_begin + EB + j < e_end ? ld_stream(&cols[e_begin + EB + j]) : -1;

        // ---- the sweeps over this group's tables: one (a one-sweep launch) or a run of them (MULTI)
        for (int it = 0; it < (MULTI ? n_iter : 1); ++it) {
            const double *x = (it & 1) ? x_b : x_a;
            double *x_new = (it & 1) ? x_a : x_b;
            const uint32_t *act_in = (it & 1) ? act_b : act_a;
            uint32_t *act_out = (it & 1) ? act_a : act_b;
            const bool credit = CM == 2 ? ((credit0 ^ it) & 1) != 0 : CM == 1; // this sweep settles pagerank for the value it pushes and the one it creates
            if constexpr (MULTI) {
                // frontier sizes of iteration `it`: row `it` of the log (row 0 = what the seeding left in cnt_in)
                const int *row = it == 0 ? cnt_in : log_slot + (size_t)it * GS_MAX;
                const int f = lane < GW ? (int)gs_ldu<true>(reinterpret_cast<const uint32_t *>(row + lane)) : 0;
                if (it == 0 && blockIdx.x == 0 && tid < GW) log_slot[tid] = f;
                if (__ballot(f != 0) == 0) { // every workgroup reads the same row: all of them stop here
                    converged = true;
                    break;
                }
            }
#define GSTAMP(i) do { if (MULTI ? it == 12 : stamp_dense) STAMP(i); } while (0)
            GSTAMP(0);
            if (it > 0) { // (MULTI only)
                if (tid < WORDS) s_actin[tid] = tid * 32 < nv ? gs_ldu<MULTI>(act_in + (v0 >> 5) + tid) : 0u;
                mycol = e_begin + j < e_end ? ld_stream(&cols[e_begin + j]) : -1;
                ncol = e_begin + EB + j < e_end ? ld_stream(&cols[e_begin + EB + j]) : -1;
            }
            __syncthreads();

            // ---- edge phase
            if (e_begin < e_end) {
                int crow = s_ostart[oid];
                int row_beg = s_cstart[crow], row_end = s_cstart[crow + 1];
                double den = (double)(row_end - row_beg + 1);
                double rcp = 1.0 / den;
                double acc[SPL];
#pragma unroll
                for (int q = 0; q < SPL; ++q) acc[q] = 0.0;
                auto flush = [&]() { // the running sums of row `crow` go to its LDS accumulator
                    const int vl = s_cvid[crow];
                    bool nz = false;
#pragma unroll
                    for (int q = 0; q < SPL; ++q) {
                        if (acc[q] != 0.0) lds_add(&s_acc[vl * GW + j * SPL + q], acc[q]);
                        nz |= acc[q] != 0.0;
                        acc[q] = 0.0;
                    }
                    if (nz) atomicOr(&s_touched[vl >> 5], 1u << (vl & 31));
                };
                uint32_t aw = mycol >= 0 ? gs_ldu<MULTI>(act_in + (mycol >> 5)) : 0u; // activity word of the first step's head
                for (int e = e_begin; e < e_end; e += EB) {
                    // which of the step's edges have an active head (one bit per lane); the NEXT step's word and
                    // the out_col entry of the step after are requested before this step's gathers
                    const bool a = mycol >= 0 && ((aw >> (mycol & 31)) & 1u);
                    const uint32_t naw = ncol >= 0 ? gs_ldu<MULTI>(act_in + (ncol >> 5)) : 0u;
                    const int nncol = e + 2 * EB + j < e_end ? ld_stream(&cols[e + 2 * EB + j]) : -1;
                    const unsigned m = oct_mask(__ballot(a));
#pragma unroll
                    for (int h = 0; h < EB; h += GB) {
                        const unsigned mh = (m >> h) & ((1u << GB) - 1u);
                        if (mh) {
                            // the sub-step's gathers are all issued before any is used
                            double xv[GB][SPL];
#pragma unroll
                            for (int k = 0; k < GB; ++k) {
                                const int ck = __shfl(mycol, (lane & ~(OCT - 1)) + h + k, WAVE);
#pragma unroll
                                for (int q = 0; q < SPL; ++q) xv[k][q] = 0.0;
#if DPPR_WHATIF & 4
                                if (((mh >> k) & 1u) && live) { xv[k][0] = 1e-7 * (ck & 7); } // (timing experiment: no gather)
                                else
#endif
#if DPPR_WHATIF & 16
                                if (((mh >> k) & 1u) && live) { // (timing experiment: every gather goes to one of 4096 rows: always an L2 hit)
#pragma unroll
                                    for (int q = 0; q < SPL; ++q) xv[k][q] = gs_ld<MULTI>(x + (size_t)(ck & 4095) * XS + j * SPL + q);
                                } else
#endif
                                if (((mh >> k) & 1u) && live) {
                                    if (MULTI || !(DPPR_GS_NT & 1) || ck < hot_rows) {
#pragma unroll
                                        for (int q = 0; q < SPL; ++q) xv[k][q] = gs_ld<MULTI>(x + (size_t)ck * XS + j * SPL + q);
                                    } else {
#pragma unroll
                                        for (int q = 0; q < SPL; ++q) xv[k][q] = gs_ld_once<MULTI, 1>(x + (size_t)ck * XS + j * SPL + q);
                                    }
                                }
                            }
#pragma unroll
                            for (int k = 0; k < GB; ++k) {
                                if ((mh >> k) & 1u) {
                                    const int ek = e + h + k;
                                    if (ek >= row_end) { // the cursor moves on: non-empty rows are contiguous in edge space
                                        flush();
                                        do {
                                            ++crow;
                                            row_beg = row_end;
                                            row_end = s_cstart[crow + 1];
                                        } while (ek >= row_end);
                                        den = (double)(row_end - row_beg + 1);
                                        rcp = 1.0 / den;
                                    }
#pragma unroll
                                    for (int q = 0; q < SPL; ++q) {
                                        if (legal(xv[k][q], phase, eps)) { // the head's residual for this source is being pushed
                                            acc[q] += push_term(xv[k][q], den, rcp);
                                            ++ecount;
                                        }
                                    }
                                }
                            }
                        }
                    }
                    mycol = ncol;
                    aw = naw;
                    ncol = nncol;
                }
                flush();
            }
            if (!MULTI && tid0 == 0) s_ticket = ticket;
            GSTAMP(1);
            __syncthreads();
            GSTAMP(2);
            if constexpr (!MULTI) { // the next group's header (extents, slice length) is requested while this one's vertices are finished
                g_next = __builtin_amdgcn_readfirstlane(s_ticket);
                if (g_next < n_groups) {
                    const int *H = gtab + (size_t)g_next * STRIDE;
                    hv0 = H[0]; hnv = H[1]; hE0 = H[2]; hEg = H[3]; hper = H[4];
                }
            }

            // ---- vertex phase: repair, threshold, next snapshot for the vertices that were touched
            for (int i0 = 0; i0 < NVX / NOCT; i0 += FU) {
                const int tid_v = opaque(tid0);
                const int j = tid_v & (OCT - 1), oid = tid_v / OCT;
                const bool live = oct_live<SPL, GW>(j);
                int vl[FU];
                bool tch[FU], wasact[FU];
                double cur[FU][SPL], pv[FU][SPL]; // cur: the residual row -- from x if the vertex was active, else from r
#pragma unroll
                for (int i = 0; i < FU; ++i) {
                    vl[i] = oid + (i0 + i) * NOCT;
                    const unsigned bit = 1u << (vl[i] & 31);
                    wasact[i] = (s_actin[vl[i] >> 5] & bit) != 0;
                    tch[i] = vl[i] < nv && (wasact[i] || (s_touched[vl[i] >> 5] & bit) != 0);
#pragma unroll
                    for (int q = 0; q < SPL; ++q) {
                        cur[i][q] = 0.0;
                        pv[i][q] = 0.0;
                    }
                    if (tch[i] && live) {
                        const size_t base = (size_t)(v0 + vl[i]) * GW + j * SPL;
                        const double *row = wasact[i] ? x + (size_t)(v0 + vl[i]) * XS + j * SPL : r + base;
#pragma unroll
                        for (int q = 0; q < SPL; ++q) {
#if DPPR_WHATIF & 2
                            if (!wasact[i])
#endif
                            cur[i][q] = gs_ld_once<MULTI, 2>(row + q);
#if DPPR_WHATIF & 2
                            else cur[i][q] = 1.0; // (timing experiment only: "every source of an active row was pushed")
#endif
#if !(DPPR_WHATIF & 1)
                            if (credit) pv[i][q] = gs_ld_once<MULTI, 2>(p + base + q); // (asked for with the residual, not after the tests)
#endif // needed only if the vertex ends up legal: asked for now, not after the test
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < FU; ++i) {
                    if (tch[i]) {
                        const size_t base = (size_t)(v0 + vl[i]) * GW + j * SPL;
                        double rn[SPL];
                        bool lg[SPL], pushed[SPL], any = false, changed = false;
#pragma unroll
                        for (int q = 0; q < SPL; ++q) {
                            double a = 0.0;
                            if (live) {
                                double *ap = &s_acc[vl[i] * GW + j * SPL + q];
                                a = *ap;
                                *ap = 0.0;
                            }
                            // RepairFrontierRev: a source that was pushed keeps only what arrived during the sweep
                            pushed[q] = wasact[i] && legal(cur[i][q], phase, eps);
                            rn[q] = pushed[q] ? a : cur[i][q] + a;
                            lg[q] = legal(rn[q], phase, eps);
                            any |= lg[q];
                            changed |= rn[q] != cur[i][q];
                            nleg[q] += lg[q] ? 1 : 0;
                        }
#if !(DPPR_WHATIF & 1)
                        if (credit && live) { // pagerank: what this sweep pushed, then what the next one will (header: every other sweep)
#pragma unroll
                            for (int q = 0; q < SPL; ++q) {
                                if (pushed[q] || lg[q]) {
                                    double pn = pv[i][q];
                                    if (pushed[q]) pn = pn + ALPHA * cur[i][q];
                                    if (lg[q]) pn = pn + ALPHA * rn[q];
                                    gs_st_once<false, 4>(p + base + q, pn);
                                }
                            }
                        }
#endif
                        if (oct_mask(__ballot(any))) { // stays / becomes active: the row lives in the next snapshot
                            if (live && !(DPPR_WHATIF & 8)) { // (8: timing experiment, no row stores)
#pragma unroll
                                for (int q = 0; q < SPL; ++q) gs_st_once<MULTI, 4>(x_new + (size_t)(v0 + vl[i]) * XS + j * SPL + q, rn[q]);
                            }
                            if (j == 0) atomicOr(&s_actout[vl[i] >> 5], 1u << (vl[i] & 31));
                        } else if ((wasact[i] || changed) && live) { // inactive now: the row goes (back) to residual[]
#pragma unroll
                            for (int q = 0; q < SPL; ++q) gs_st_once<false, 4>(r + base + q, rn[q]);
                        }
                    }
                }
            }
            GSTAMP(3);
            __syncthreads();
            GSTAMP(4);
            if (tid < WORDS) { // the group's words of the next bitmap (complete), tables back to zero
                if (tid * 32 < nv) gs_stu<MULTI>(act_out + (v0 >> 5) + tid, s_actout[tid]);
                s_actout[tid] = 0u;
                s_touched[tid] = 0u;
            }
            if constexpr (MULTI) {
                // ---- end of iteration `it`: this workgroup's frontier counts go to row it + 1 of the log, its stores are
                // drained, it arrives, and it waits until everybody has
#pragma unroll
                for (int q = 0; q < SPL; ++q) {
                    if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
                    nleg[q] = 0;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                GSTAMP(5);
                if (tid < GW) {
                    if (s_cnt[tid])
                        __hip_atomic_fetch_add(log_slot + (size_t)(it + 1) * GS_MAX + tid, s_cnt[tid], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    s_cnt[tid] = 0;
                }
                if (w == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the count adds of this wave's lanes)
                    if (lane < BAR_REPS)
                        __hip_atomic_fetch_add(&bar->sub[it & 1][lane][blockIdx.x % BAR_SUBS].w, 1ull << 32, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long rounds = (unsigned long long)((it >> 1) + 1);
                    unsigned polls = 0;
                    const unsigned long long t_start = wall_clock64();
                    bool ok = true;
                    for (;;) {
                        const unsigned long long word = lane < (int)subs_used ? bar_load(&bar->sub[it & 1][my_rep][lane].w) : 0;
                        if (__ballot((word >> 32) >= n_sub * rounds) == ~0ull) break;
                        if ((polls++ & 63u) == 63u && wall_clock64() - t_start > limit_ticks + 100000000ull) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(BAR_POLL_SLEEP);
                    }
                    if (lane == 0 && !ok) s_flag[1] = 1;
                }
                __syncthreads();
                GSTAMP(6);
                sweeps_done = it + 1;
                if (s_flag[1]) {
                    fault = true;
                    break;
                }
            }
        }
        if constexpr (MULTI) break; // one group per workgroup
        g = g_next;
    }
    const int tid = tid0;
    if constexpr (MULTI) {
        if (blockIdx.x == 0 && tid < GS_MAX) { // the live frontier sizes for whoever continues, counters as a one-sweep launch leaves them
            const int f = (tid < GW && !converged && !fault)
                              ? (int)gs_ldu<true>(reinterpret_cast<const uint32_t *>(log_slot + (size_t)sweeps_done * GS_MAX + tid)) : 0;
            cnt_out[tid] = f;
            cnt_zero[tid] = 0;
        }
        if (blockIdx.x == 0 && tid == 0)
            *status = sweeps_done | (fault ? GSM_FAULT : 0) | (converged ? GSM_CONVERGED : 0);
        if (ecount) atomicAdd(&s_edges, (unsigned long long)ecount);
        __syncthreads();
        if (tid == 0 && s_edges) stats->blk_E[blockIdx.x] += s_edges;
        return;
    }
#ifdef DPPR_STAMPS
    if (!MULTI && stamp_dense && tid == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + 6] = wall_clock64();
#endif
    // next frontier sizes: one fire-and-forget atomic per source and workgroup
#pragma unroll
    for (int q = 0; q < SPL; ++q)
        if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
    if (ecount) atomicAdd(&s_edges, (unsigned long long)ecount);
    __syncthreads();
    if (tid < GW && s_cnt[tid]) atomicAdd(&cnt_out[tid], s_cnt[tid]);
    if (tid == 0 && s_edges) stats->blk_E[blockIdx.x] += s_edges;
}

// strided copy from the interleaved group state to an external-id vector of one source
__global__ __launch_bounds__(BLOCK) void k_gint_to_ext(const double *__restrict__ a_int, int gw, int lane_s,
                                                       const int *__restrict__ ext2int, int V,
                                                       double *__restrict__ a_ext) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        const int m = ext2int[v];
        a_ext[v] = m >= 0 ? a_int[(size_t)m * gw + lane_s] : 0.0;
    }
}

} // namespace dppr
