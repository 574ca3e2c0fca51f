// dppr_push.hpp -- Init, Inspect, dense snapshot and the sparse (push) frontier iteration.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

// ---------------------------------------------------------------------------
// a2  Init (gpu/PPRCommon.cuh:12-22): r = e_s, p = 0. 16 B per lane stores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_init(double *__restrict__ p, double *__restrict__ r, int V, int s) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t u = (int64_t)blockIdx.x * BLOCK + threadIdx.x; u < V; u += stride) {
        p[u] = 0.0;
        r[u] = (u == s) ? 1.0 : 0.0;
    }
}

// ---------------------------------------------------------------------------
// a3  Inspect (gpu/Inspect.cuh:8-48): compact {u : legal(r[u])} into ft.
// One pass over r (the reference reads it twice), ballot/mbcnt ranks inside each
// wave, LDS staging per workgroup, ONE global counter atomic per chunk of
// BLOCK*INSPECT_ITEMS vertices.
// ---------------------------------------------------------------------------
constexpr int INSPECT_ITEMS = 8;
__global__ __launch_bounds__(BLOCK) void k_inspect(const double *__restrict__ r, int V, int phase, double eps,
                                                   int *__restrict__ ft, int *__restrict__ cnt) {
    __shared__ int s_buf[BLOCK * INSPECT_ITEMS];
    __shared__ int s_n;
    __shared__ int s_base;
    const int64_t chunk = (int64_t)BLOCK * INSPECT_ITEMS;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < V; base += (int64_t)gridDim.x * chunk) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < INSPECT_ITEMS; ++k) {
            const int64_t u = base + (int64_t)k * BLOCK + threadIdx.x;
            const bool hit = (u < V) && legal(r[u], phase, eps);
            const uint64_t m = __ballot(hit);
            if (m) {
                int wbase = 0;
                if (lane_id() == 0) wbase = atomicAdd(&s_n, __popcll(m)); // LDS atomic
                wbase = __shfl(wbase, 0, WAVE);
                if (hit) s_buf[wbase + mbcnt(m)] = (int)u;
            }
        }
        __syncthreads();
        const int n = s_n;
        if (n) {
            if (threadIdx.x == 0) s_base = atomicAdd(cnt, n);
            __syncthreads();
            const int gb = s_base;
            for (int i = threadIdx.x; i < n; i += BLOCK) ft[gb + i] = s_buf[i];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Snapshot of the frontier residuals into the DENSE vector x (x[u] = amount u pushes
// this iteration, 0 for every vertex outside the frontier). The head of
// ExpandUnifiedRev (gpu/ExpandRev.cuh:34-42) for ALL frontier vertices before any push
// lands: ru = residual[u]; (vertex_ft_r =) x[u] = ru; pagerank[u] += ALPHA * ru.
// Used by the synchronous schedule and when a sparse (push) iteration is followed by a
// dense (pull) one.
// ---------------------------------------------------------------------------
// `dedup`: the list may name a vertex twice (merged loop: adds of both signs can take a residual across the threshold,
// back, and across again within one push iteration, and every crossing appends): the entry that claims x[u] -- all zero
// before a snapshot -- with a compare-and-swap credits pagerank, the others do nothing.
// `extract`: InspectExtra's pre-extraction (gpu/Inspect.cuh:51-65, variants FAST_FRONTIER / VANILLA): residual[u] = 0 at the
// snapshot, so that the push iteration that follows needs no repair (k_push_iter<true> with extract).
__global__ __launch_bounds__(BLOCK) void k_snapshot_dense(const int *__restrict__ ft, const int *__restrict__ cnt_in,
                                                          double *__restrict__ r, double *__restrict__ p,
                                                          double *__restrict__ x, uint32_t *__restrict__ act, int dedup, int extract) {
    const int F = *cnt_in;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < F; i += gridDim.x * BLOCK) {
        const int u = ft[i];
        const double ru = r[u];
        if (extract) r[u] = 0.0;
        if (dedup) {
            if (ru == 0.0 || atomicCAS(reinterpret_cast<unsigned long long *>(&x[u]), 0ull, (unsigned long long)__double_as_longlong(ru)) != 0ull)
                continue;
        } else {
            x[u] = ru;
        }
        p[u] += ALPHA * ru;
        if (act && ru != 0.0) atomicOr(&act[u >> 5], 1u << (u & 31)); // activity bitmap of the snapshot (cleared by the host)
    }
}

// ---------------------------------------------------------------------------
// a4+a5  one frontier iteration: ExpandUnifiedRev (gpu/ExpandRev.cuh:8-183) with
// RepairFrontierRev (:708-743) fused in.
//
// k_push_iter -- per wave: a tile of 64 frontier vertices. Lane i owns vertex i:
//   EAGER: ru = atomic_exchange(r[u], 0)  (= "ru = residual[u]" ... "residual[u] -= ru"
//          collapsed to one instant; everything that arrives later stays and may
//          cross the threshold again), p[u] += ALPHA*ru.
//   DENSE: ru = x[u] (snapshot taken by k_snapshot_dense or by the preceding pull sweep;
//          p already updated), x[u] = 0, repair = returning atomic add of -ru; if the
//          result is still legal the vertex re-enters the next frontier.
// Row extents are scanned across the wave and staged in LDS; the wave then walks
// the concatenated edge list 4 x 64 edges per round (four independent returning
// atomics in flight per lane): edge e belongs to the vertex found by a binary
// search of the scan, so consecutive lanes read consecutive Adj entries. Per edge
// (gpu/ExpandRev.cuh:70-77):
//   add  = (1.0-ALPHA) * ru / (degv + 1)
//   prer = atomicAdd(&residual[v], add); curr = prer + add
//   enqueue v iff !legal(prer) && legal(curr)
//
// Load balance (replaces the CTA / warp / scan tiers of gpu/ExpandRev.cuh:44-176):
//   * rows with >= BIG_ROW edges are not expanded by their wave; (row, ru) goes to a
//     small device list and k_push_big spreads 1024-edge chunks of those rows over
//     the whole grid;
//   * HUB TARGETS: a vertex with a very large out-degree receives one add per
//     frontier neighbour; thousands of returning atomics on ONE address serialise
//     at the memory-side atomic unit (~88 per us). The CSR builder tags the top
//     out-degree vertices (Adj.degp1 < 0 -> hub slot), pushes to them accumulate in a
//     per-workgroup LDS table (ds_add_f64) and each workgroup issues ONE global atomic
//     per touched hub. Residual adds within a phase all have the same sign, so the
//     crossing test on the aggregated add still fires exactly once.
// Crossing vertices are ranked with ballot+mbcnt into a per-wave LDS tile and
// flushed with one global counter atomic per workgroup (per wave on overflow).
//
// Counter rotation: the kernel reads cnt_in, appends to cnt_out and zeroes cnt_zero
// (the counter the NEXT iteration appends to); likewise big_cnt / big_zero. Iteration
// kernels can therefore be chained without host round trips or memsets.
// ---------------------------------------------------------------------------
struct BigItem { // one BIG_CHUNK-edge piece of a deferred big row (the list holds pieces, so that k_push_big finds its work by index)
    int row_start; // first Adj entry of the piece
    int len;       // edges in it (<= BIG_CHUNK)
    double ru;
};

constexpr int BIG_ROW_DEFAULT = 512; // rows at least this long go to k_push_big (runtime tunable)
constexpr int BIG_CHUNK = 1024; // edges per workgroup chunk there
constexpr int HUB_CAP = 2048;   // hub slots (16 KiB of LDS accumulators)
constexpr int UNROLL = 4;

struct HubTable { // per-epoch hub directory (device pointers)
    const int *v;      // hub slot -> vertex
    const int *degp1;  // hub slot -> outdeg + 1
    int n;
};

// per-wave staging of next-frontier entries
struct OutStage {
    int *s_out;   // this wave's LDS tile (OUT_CAP ints)
    int n;        // wave-uniform fill
    int *ft_out;
    int *cnt_out;
    __device__ __forceinline__ void flush_wave() {
        int gb = 0;
        if (lane_id() == 0) gb = atomicAdd(cnt_out, n);
        gb = __shfl(gb, 0, WAVE);
        for (int i = lane_id(); i < n; i += WAVE) ft_out[gb + i] = s_out[i];
        n = 0;
    }
    __device__ __forceinline__ void stage(bool hit, int v) {
        const uint64_t m = __ballot(hit);
        if (m) {
            if (hit) s_out[n + mbcnt(m)] = v;
            n += __popcll(m);
            if (n > OUT_CAP - WAVE) flush_wave();
        }
    }
};


// Does the add that took residual[v] from prer to prer + add queue v for the next frontier? Two duplicate filters, as in the
// reference's variants: the threshold CROSSING (ExpandUnifiedRev / ExpandFastFrontierRev, gpu/ExpandRev.cuh:75-77,430-432) or the
// STATUS array (ExpandEagerRev / ExpandVanillaRev, :255,298,340 and :603,646,688): legal(curr) && atomicExch(status[v], level) < level.
struct Dedup {
    int *status; // nullptr: crossing test
    int level;   // number of this push launch (monotonic over the engine's life: stale entries are always smaller)
};
__device__ __forceinline__ bool queues(double prer, double add, int v, int phase, double eps, const Dedup &dd) {
    if (dd.status) return legal(prer + add, phase, eps) && atomicExch(&dd.status[v], dd.level) < dd.level;
    return !legal(prer, phase, eps) && legal(prer + add, phase, eps);
}

// issue the push of one edge; returns the pre-add residual (or NaN-free dummy for hubs)
struct EdgePush {
    int v;
    double add;
    double prer;
    bool direct; // a global atomic was issued and prer is meaningful
};
__device__ __forceinline__ EdgePush push_edge(bool valid, Adj a, double ru, double *__restrict__ r,
                                              double *s_hub, const HubTable &hubs) {
    EdgePush o;
    o.v = a.v;
    o.add = 0.0;
    o.prer = 0.0;
    o.direct = false;
    if (valid) {
        if (a.degp1 < 0) {
            const int slot = ~a.degp1;
            lds_add(&s_hub[slot], ONE_MINUS_ALPHA * ru / (double)hubs.degp1[slot]);
        } else {
            o.add = ONE_MINUS_ALPHA * ru / (double)a.degp1;
            o.prer = atomic_add_ret(&r[a.v], o.add);
            o.direct = true;
        }
    }
    return o;
}

// workgroup epilogue shared by both push kernels: flush hub accumulators, then the staged frontier
__device__ __forceinline__ void push_epilogue(OutStage &out, double *s_hub, const HubTable &hubs,
                                              double *__restrict__ r, int phase, double eps, int *s_cnt, int *s_base,
                                              unsigned long long edges, IterStats *stats,
                                              unsigned long long *s_edges, const Dedup &dd) {
    __syncthreads(); // all LDS hub adds of the workgroup done
    for (int s0 = 0; s0 < hubs.n; s0 += BLOCK) {
        const int slot = s0 + threadIdx.x;
        bool hit = false;
        int v = 0;
        if (slot < hubs.n) {
            const double acc = s_hub[slot];
            if (acc != 0.0) {
                v = hubs.v[slot];
                const double prer = atomic_add_ret(&r[v], acc);
                hit = queues(prer, acc, v, phase, eps, dd);
            }
        }
        out.stage(hit, v);
    }
    const int lane = lane_id(), w = wave_id();
    if (lane == 0) s_cnt[w] = out.n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        *s_base = tot ? atomicAdd(out.cnt_out, tot) : 0;
    }
    __syncthreads();
    int gb = *s_base;
    for (int k = 0; k < w; ++k) gb += s_cnt[k];
    for (int i = lane; i < out.n; i += WAVE) out.ft_out[gb + i] = out.s_out[i];
    stat_add_edges<WAVES_PER_BLOCK>(stats, edges, s_edges);
}

template <bool DENSE>
__global__ __launch_bounds__(BLOCK) void k_push_iter(const int *__restrict__ ft, const int *__restrict__ cnt_in,
                                                     int *__restrict__ ft_out, int *__restrict__ cnt_out,
                                                     int *__restrict__ cnt_zero, double *__restrict__ x,
                                                     const int *__restrict__ row_ptr, const Adj *__restrict__ adj,
                                                     HubTable hubs, BigItem *__restrict__ big, int *__restrict__ big_cnt,
                                                     int *__restrict__ big_zero, int big_row, double *__restrict__ r,
                                                     double *__restrict__ p, int phase, double eps,
                                                     IterStats *__restrict__ stats, int *__restrict__ log_slot, Dedup dd, int extract) {
    __shared__ int s_scan[WAVES_PER_BLOCK][WAVE + 1];
    __shared__ int s_start[WAVES_PER_BLOCK][WAVE];
    __shared__ double s_ru[WAVES_PER_BLOCK][WAVE];
    __shared__ int s_out[WAVES_PER_BLOCK][OUT_CAP];
    __shared__ double s_hub[HUB_CAP];
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ unsigned long long s_edges[WAVES_PER_BLOCK];
    __shared__ int s_base;

    const int lane = lane_id();
    const int w = wave_id();
    const int F = *cnt_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *cnt_zero = 0;
        *big_zero = 0;
        *log_slot = F; // frontier size of this iteration (0: the loop is over), read by the host per chunk
    }
    const int n_tiles = (F + WAVE - 1) / WAVE;
    if ((int)blockIdx.x >= n_tiles) return; // the grid is sized for the largest sparse frontier
    for (int i = threadIdx.x; i < hubs.n; i += BLOCK) s_hub[i] = 0.0;
    __syncthreads();

    OutStage out{s_out[w], 0, ft_out, cnt_out};
    unsigned long long edges = 0; // wave-uniform

    // tile t -> (block t % gridDim, wave (t / gridDim) % 4): small frontiers spread over CUs
    for (int t = blockIdx.x + gridDim.x * w; t < n_tiles; t += gridDim.x * WAVES_PER_BLOCK) {
        const int i = t * WAVE + lane;
        const bool valid = i < F;
        int u = -1, rs = 0, d = 0;
        double ru = 0.0;
        bool requeue = false;
        if (valid) {
            u = ft[i];
            rs = row_ptr[u];
            d = row_ptr[u + 1] - rs;
            if (DENSE) {
                ru = x[u];
                x[u] = 0.0; // x is all-zero again once the sparse iteration is over
                if (!extract) { // (extract: residual[u] was zeroed at the snapshot, InspectExtra -- nothing to repair)
                    // RepairFrontierRev: residual[u] -= vertex_ft_r[i]; still legal -> next frontier
                    const double prer = atomic_add_ret(&r[u], -ru);
                    requeue = legal(prer - ru, phase, eps);
                    // (status filter: an add may have queued u for the next frontier already -- one entry per vertex and launch)
                    if (requeue && dd.status) requeue = atomicExch(&dd.status[u], dd.level) < dd.level;
                }
            } else {
                ru = atomic_exch(&r[u], 0.0);
                p[u] += ALPHA * ru;
            }
        }
        if (DENSE) out.stage(requeue, u);

        // big rows: hand their BIG_CHUNK-edge pieces (row piece, ru) to k_push_big
        const bool is_big = d >= big_row;
        const uint64_t mb = __ballot(is_big);
        if (mb) {
            const int nch = is_big ? (d + BIG_CHUNK - 1) / BIG_CHUNK : 0;
            const int incl_ch = wave_inclusive_scan(nch);
            int gb = 0;
            if (lane == WAVE - 1) gb = atomicAdd(big_cnt, incl_ch);
            gb = __shfl(gb, WAVE - 1, WAVE) + incl_ch - nch;
            for (int c = 0; c < nch; ++c) {
                BigItem it;
                it.row_start = rs + c * BIG_CHUNK;
                it.len = min(BIG_CHUNK, d - c * BIG_CHUNK);
                it.ru = ru;
                big[gb + c] = it;
            }
            if (is_big) d = 0;
        }

        const int incl = wave_inclusive_scan(d);
        const int total = __shfl(incl, WAVE - 1, WAVE);
        s_scan[w][lane] = incl - d;
        s_start[w][lane] = rs;
        s_ru[w][lane] = ru;
        if (lane == 0) s_scan[w][WAVE] = total;
        __builtin_amdgcn_wave_barrier(); // LDS ops of one wave execute in order
        edges += (unsigned long long)total;

        for (int e0 = 0; e0 < total; e0 += WAVE * UNROLL) {
            EdgePush q[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const int e = e0 + k * WAVE + lane;
                const bool ok = e < total;
                Adj a{0, 1};
                double ruk = 0.0;
                if (ok) {
                    int lo = 0, hi = WAVE; // owner: last index with scan[idx] <= e
#pragma unroll
                    for (int s = 0; s < 6; ++s) {
                        const int mid = (lo + hi) >> 1;
                        if (s_scan[w][mid] <= e) lo = mid; else hi = mid;
                    }
                    a = adj[s_start[w][lo] + (e - s_scan[w][lo])];
                    ruk = s_ru[w][lo];
                }
                q[k] = push_edge(ok, a, ruk, r, s_hub, hubs);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const bool hit = q[k].direct && queues(q[k].prer, q[k].add, q[k].v, phase, eps, dd);
                out.stage(hit, q[k].v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    push_epilogue(out, s_hub, hubs, r, phase, eps, s_cnt, &s_base, edges, stats, s_edges, dd);
}

// Deferred big rows: piece c of the list goes to workgroup c % gridDim; 256 lanes x 4 edges.
__global__ __launch_bounds__(BLOCK) void k_push_big(const BigItem *__restrict__ big, const int *__restrict__ big_cnt,
                                                    int *__restrict__ ft_out, int *__restrict__ cnt_out,
                                                    const Adj *__restrict__ adj, HubTable hubs, double *__restrict__ r,
                                                    int phase, double eps, IterStats *__restrict__ stats, Dedup dd) {
    __shared__ int s_out[WAVES_PER_BLOCK][OUT_CAP];
    __shared__ double s_hub[HUB_CAP];
    __shared__ int s_cnt[WAVES_PER_BLOCK];
    __shared__ unsigned long long s_edges[WAVES_PER_BLOCK];
    __shared__ int s_base;
    const int nbig = *big_cnt;
    if (nbig == 0) return; // uniform for the whole grid
    for (int i = threadIdx.x; i < hubs.n; i += BLOCK) s_hub[i] = 0.0;
    __syncthreads();
    OutStage out{s_out[wave_id()], 0, ft_out, cnt_out};
    unsigned long long edges = 0;

    for (int c = blockIdx.x; c < nbig; c += gridDim.x) { // piece c of the list: 256 lanes x 4 edges
        const BigItem item = big[c];
        EdgePush q[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const int e = k * BLOCK + threadIdx.x;
            const bool ok = e < item.len;
            Adj a{0, 1};
            if (ok) a = adj[item.row_start + e];
            q[k] = push_edge(ok, a, item.ru, r, s_hub, hubs);
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const bool hit = q[k].direct && queues(q[k].prer, q[k].add, q[k].v, phase, eps, dd);
            out.stage(hit, q[k].v);
        }
        if (wave_id() == 0) edges += (unsigned long long)item.len;
    }
    push_epilogue(out, s_hub, hubs, r, phase, eps, s_cnt, &s_base, edges, stats, s_edges, dd);
}

} // namespace dppr
