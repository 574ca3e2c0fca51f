// dppr_pull.hpp -- the dense (pull sweep) frontier iteration.
#pragma once

#include "dppr_common.hpp"
#include "dppr_push.hpp"

namespace dppr {

// ---------------------------------------------------------------------------
// a4+a5, DENSE iterations: the same frontier iteration evaluated as a PULL sweep.
//
// When the frontier covers a large part of the graph (on the benchmark streams it is
// the whole active component for most iterations) one random returning atomic per
// traversed edge is bounded by the memory-side atomic units (~23 G/s).
// The identical arithmetic can be gathered instead: for every vertex v
//     rv = residual[v]
//     for u in out(v), ascending:  if x[u] != 0:  rv += (1.0-ALPHA) * x[u] / (outdeg(v)+1)
//     rv -= x[v]                                   (RepairFrontierRev for frontier members)
// which is exactly what the pushes u -> v of gpu/ExpandRev.cuh:70-73 followed by the
// repair of :708-743 leave in residual[v] when the atomics happen to arrive in CSR
// order. No atomics on global memory: out_col is streamed (the rows of 64 consecutive
// vertices are one contiguous range), x[u] is an 8-byte gather, the per-vertex sums are
// LDS atomics inside the owning wave. The next frontier is {v : legal(rv)} (residual
// adds of a phase all have one sign, so this equals the reference's crossing test plus
// repaired members); for those the kernel immediately takes the next snapshot
// (x_new[v] = rv, pagerank[v] += ALPHA*rv), so consecutive dense iterations are ONE
// kernel each. The sparse list / counter are produced as well, so a push iteration
// can follow.
// ---------------------------------------------------------------------------
// workgroup size of the sweep = consecutive vertices per pass: 1024 for large graphs (few
// workgroups -> few counter atomics), 512 / 256 when that would leave CUs idle
constexpr int PULL_BIG_ROW_DEFAULT = 128; // rows at least this long are gathered by the whole workgroup
constexpr int PULL_BIG_CAP = 64;  // such rows per workgroup pass (more: the owning wave does them itself)
constexpr int PU = 4;             // gathers in flight per lane (short rows)

struct PullBig {
    int v, rs, d;
    double rv, xv, pv;
};


// The sweep does not build the sparse frontier list (a following dense iteration does not
// need it): it only COUNTS the next frontier, with one fire-and-forget atomic per workgroup.
// k_list_from_dense materialises the list when a sparse iteration follows (or for tracing).
// Diagnostic build only (-DDPPR_STAMPS, tools/stamps.sh): shader-clock stamps of the sweep's
// stages, one row per workgroup, written to a buffer nothing else reads.
#ifdef DPPR_STAMPS
__device__ unsigned long long g_stamps[4096 * 8];
#define STAMP(i)                                                                         \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + (i)] = clock64(); \
    } while (0)
#else
#define STAMP(i) ((void)0)
#endif

// 8 waves per SIMD (<= 64 VGPRs, a handful of spilled registers): two 1024-thread workgroups per CU
// instead of one. On windows that run this kernel per iteration the sweep is latency-bound per group
// (DESIGN.md section 6), and the second resident workgroup is worth 12 % (LiveJournal stand-in:
// 88 -> 77 us per sweep). The block sizes that are not powers of two exist for tests only.
// BITS: the sweep is given the ACTIVITY BITMAP of the snapshot it reads (bit u = x[u] != 0) and writes the one
// of the snapshot it produces. A gather is then preceded by a bit test: one bit per vertex stays in L2
// (LiveJournal stand-in 150 KB, twitter 2.5 MB, friendster 8 MB) where x (9 MB .. 500 MB) does not, and
// the share of a sweep's edges whose head is NOT in the frontier -- 27 % on the configs[1] stand-in, 45 %
// on the friendster one, nearly all of them in the tail of a loop -- costs an L2 hit instead of a random
// sector from Infinity Cache / HBM. x stays a complete snapshot (zeros included) either way, so the other
// kernels do not care which form ran. Used on windows that cannot run resident (dppr_engine.hip).
template <int PULL_BLOCK, bool BITS>
__global__ __launch_bounds__(PULL_BLOCK, (PULL_BLOCK & (PULL_BLOCK - 1)) == 0 ? 8 : 1) void k_pull_iter(int V, const int *__restrict__ grp_tile, int n_groups,
                                                          const int *__restrict__ cnt_in,
                                                          const int *__restrict__ out_row_ptr,
                                                          const int *__restrict__ out_col,
                                                          const double *__restrict__ x, double *__restrict__ x_new,
                                                          double *__restrict__ r, double *__restrict__ p,
                                                          int *__restrict__ cnt_out, int *__restrict__ cnt_zero,
                                                          int phase, double eps, IterStats *__restrict__ stats,
                                                          int *__restrict__ log_slot, int pull_big_row,
                                                          const uint32_t *__restrict__ act_in,
                                                          uint32_t *__restrict__ act_out) {
    constexpr int PULL_WAVES = PULL_BLOCK / WAVE;
    __shared__ uint32_t s_bits[BITS ? 2 * PULL_WAVES : 2]; // next snapshot's activity words of the group's tiles
    __shared__ int s_own[PULL_WAVES][WAVE * PU];   // per round: owner marks of the wave's edge window
    __shared__ int s_scan[PULL_WAVES][WAVE + 1];
    __shared__ int s_start[PULL_WAVES][WAVE];
    __shared__ double s_acc[PULL_WAVES][WAVE];
    __shared__ double s_rcp[PULL_WAVES][WAVE];     // 1 / (outdeg + 1) per row: one division per vertex, none per edge (push_term)
    __shared__ int s_cnt[PULL_WAVES];
    __shared__ unsigned long long s_edges[PULL_WAVES];
    __shared__ PullBig s_big[PULL_BIG_CAP];
    __shared__ double s_bigacc[PULL_BIG_CAP];
    __shared__ int s_chunk0[PULL_WAVES][WAVE + 1]; // per wave copy: first chunk id of each long row
    __shared__ int s_nbig;
    const int lane = lane_id(), w = wave_id();
    // Work is dealt in GROUPS of consecutive 64-vertex tiles: group g = tiles [grp_tile[g],
    // grp_tile[g+1]), at most PULL_WAVES of them, cut by the graph builder so that every group
    // carries about the same number of edges (a hub's group has few tiles, and all the
    // workgroup's waves share its long rows). Wave w takes the group's w-th tile.
    const int F = *cnt_in;
    // the first group's tile loads are issued BEFORE F is consumed: the (cold) read of the
    // frontier size overlaps them instead of heading the dependent chain
    int rs = 0, d = 0;
    double rv = 0.0, xv = 0.0, pv = 0.0;
    int t0 = 0, t1 = 0;
    if ((int)blockIdx.x < n_groups) {
        t0 = grp_tile[blockIdx.x];
        t1 = grp_tile[blockIdx.x + 1];
    }
    {
        const int v0 = (t0 + w) * WAVE + lane;
        if (t0 + w < t1 && v0 < V) {
            rs = out_row_ptr[v0];
            d = out_row_ptr[v0 + 1] - rs;
            rv = r[v0];
            xv = x[v0];
            pv = p[v0];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *cnt_zero = 0;
        *log_slot = F;
    }
    if (F == 0) return; // empty frontier: x / x_new are not touched
    STAMP(0);
    int n_legal = 0;    // per-lane count of next-frontier vertices
    unsigned long long edges = 0;

    // repair, threshold, next snapshot. pv = pagerank[v], loaded up front with r and x so the
    // "pagerank[v] += ALPHA*rn" of the next snapshot costs no extra round trip
    auto finish = [&](bool valid, int v, double rv, double xv, double pv, double rn) -> bool {
        if (xv != 0.0) rn -= xv;
        const bool lg = valid && legal(rn, phase, eps);
        if (valid) {
            if (rn != rv) r[v] = rn;
            x_new[v] = lg ? rn : 0.0; // every entry is rewritten: x_new is a complete snapshot
            if (lg) p[v] = pv + ALPHA * rn;
        }
        n_legal += lg ? 1 : 0;
        return lg;
    };
    // x[c] for a head c: behind its activity bit when the sweep has the bitmap
    auto head_active = [&](int c) -> bool { return ((act_in[c >> 5] >> (c & 31)) & 1u) != 0u; };

    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) { // workgroup-uniform loop
        if (threadIdx.x == 0) s_nbig = 0;
        __syncthreads();
        if (g != (int)blockIdx.x) {
            t0 = grp_tile[g];
            t1 = grp_tile[g + 1];
        }
        const int v = (t0 + w) * WAVE + lane;
        const bool valid = t0 + w < t1 && v < V; // waves beyond the group's tiles only help with long rows
        if (g != (int)blockIdx.x) { // later groups (the first one was loaded above)
            rs = 0; d = 0; rv = 0.0; xv = 0.0; pv = 0.0;
            if (valid) {
                rs = out_row_ptr[v];
                d = out_row_ptr[v + 1] - rs;
                rv = r[v];
                xv = x[v];
                pv = p[v];
            }
        }
        // long rows go to the workgroup list; the owning lane keeps them only if the list is full
        bool deferred = false;
        if (d >= pull_big_row) {
            const int slot = atomicAdd(&s_nbig, 1);
            if (slot < PULL_BIG_CAP) {
                s_big[slot] = PullBig{v, rs, d, rv, xv, pv};
                s_bigacc[slot] = 0.0;
                deferred = true;
            }
        }
        const int dd = deferred ? 0 : d;
        const int incl = wave_inclusive_scan(dd);
        const int scan_ex = incl - dd;
        const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
        s_scan[w][lane] = scan_ex; // deferred rows have length 0 here and are never visited
        s_start[w][lane] = rs;
        if (lane == 0) s_scan[w][WAVE] = total;
        s_acc[w][lane] = rv;
        s_rcp[w][lane] = 1.0 / (double)(dd + 1);
        STAMP(1);

        // ---- the wave's own (short) rows: 64*PU consecutive edges of the concatenated list per
        // round. Owner of edge e = last non-empty row whose start is <= e: rows starting inside the
        // round's window mark their lane id at their start position, a max-scan propagates it.
        for (int e0 = 0; e0 < total; e0 += WAVE * PU) {
#pragma unroll
            for (int k = 0; k < PU; ++k) s_own[w][k * WAVE + lane] = -1;
            __builtin_amdgcn_wave_barrier();
            const int pos = scan_ex - e0;
            if (dd > 0 && pos >= 0 && pos < WAVE * PU) s_own[w][pos] = lane;
            __builtin_amdgcn_wave_barrier();
            const uint64_t before = __ballot(dd > 0 && scan_ex <= e0);
            int carry = before ? 63 - __clzll(before) : -1; // row that owns edge e0
            int own[PU], col[PU];
#pragma unroll
            for (int k = 0; k < PU; ++k) {
                const int e = e0 + k * WAVE + lane;
                int o = wave_inclusive_max(s_own[w][k * WAVE + lane]);
                o = max(o, carry);
                carry = __builtin_amdgcn_readlane(o, WAVE - 1);
                own[k] = e < total ? o : -1;
                col[k] = 0;
                if (own[k] >= 0) col[k] = ld_stream(&out_col[s_start[w][o] + (e - s_scan[w][o])]);
            }
            double xa[PU];
            if constexpr (BITS) {
                bool on[PU];
#pragma unroll
                for (int k = 0; k < PU; ++k) on[k] = own[k] >= 0 && head_active(col[k]);
#pragma unroll
                for (int k = 0; k < PU; ++k) xa[k] = on[k] ? x[col[k]] : 0.0;
            } else {
#pragma unroll
                for (int k = 0; k < PU; ++k) xa[k] = own[k] >= 0 ? x[col[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < PU; ++k) {
                const bool nz = xa[k] != 0.0;
                if (nz) {
                    const int dk = s_scan[w][own[k] + 1] - s_scan[w][own[k]];
                    lds_add(&s_acc[w][own[k]], push_term(xa[k], (double)(dk + 1), s_rcp[w][own[k]]));
                }
                edges += (unsigned long long)__popcll(__ballot(nz));
            }
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(2);
        const bool lgw = finish(valid && !deferred, v, rv, xv, pv, s_acc[w][lane]); // deferred vertices are finished below
        if constexpr (BITS) {
            const uint64_t lb = __ballot(lgw);
            if (lane == 0) {
                s_bits[2 * w] = (uint32_t)lb;
                s_bits[2 * w + 1] = (uint32_t)(lb >> 32);
            }
        }
        STAMP(3);

        // ---- the workgroup's long rows, cut into chunks of PULL_CHUNK edges dealt round-robin to
        // the waves: no per-edge search, per-lane partial sums, one wave reduction per chunk.
        __syncthreads(); // long-row list complete
        const int nbig = min(s_nbig, PULL_BIG_CAP);
        if (nbig) { // workgroup-uniform
            constexpr int PULL_CHUNK = WAVE * 8;
            const int nch = lane < nbig ? (s_big[lane].d + PULL_CHUNK - 1) / PULL_CHUNK : 0;
            const int inc = wave_inclusive_scan(nch);
            s_chunk0[w][lane] = inc - nch;
            const int n_chunks = __builtin_amdgcn_readlane(inc, WAVE - 1);
            if (lane == 0) s_chunk0[w][WAVE] = n_chunks;
            __builtin_amdgcn_wave_barrier();
            for (int ch = w; ch < n_chunks; ch += PULL_WAVES) {
                int lo = 0, hi = WAVE; // row of this chunk: wave-uniform search, once per 512 edges
#pragma unroll
                for (int s2 = 0; s2 < 6; ++s2) {
                    const int mid = (lo + hi) >> 1;
                    if (s_chunk0[w][mid] <= ch) lo = mid; else hi = mid;
                }
                const int row_rs = s_big[lo].rs, row_d = s_big[lo].d;
                const int c0 = (ch - s_chunk0[w][lo]) * PULL_CHUNK;
                const int c1 = min(c0 + PULL_CHUNK, row_d);
                const double denom = (double)(row_d + 1), rdenom = 1.0 / denom;
                double part = 0.0;
                constexpr int CH_SLOTS = PULL_CHUNK / WAVE; // all of a chunk's loads are issued before any use
                int colb[CH_SLOTS];
                double xb[CH_SLOTS];
#pragma unroll
                for (int k = 0; k < CH_SLOTS; ++k) {
                    const int e = c0 + k * WAVE + lane;
                    colb[k] = e < c1 ? ld_stream(&out_col[row_rs + e]) : -1;
                }
                if constexpr (BITS) {
                    bool on[CH_SLOTS];
#pragma unroll
                    for (int k = 0; k < CH_SLOTS; ++k) on[k] = colb[k] >= 0 && head_active(colb[k]);
#pragma unroll
                    for (int k = 0; k < CH_SLOTS; ++k) xb[k] = on[k] ? x[colb[k]] : 0.0;
                } else {
#pragma unroll
                    for (int k = 0; k < CH_SLOTS; ++k) xb[k] = colb[k] >= 0 ? x[colb[k]] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < CH_SLOTS; ++k) {
                    const bool nz = xb[k] != 0.0;
                    if (nz) part += push_term(xb[k], denom, rdenom);
                    edges += (unsigned long long)__popcll(__ballot(nz));
                }
                part = wave_sum(part);
                if (lane == 0 && part != 0.0) lds_add(&s_bigacc[lo], part);
            }
            __syncthreads(); // all long-row partial sums are in
            if (w == 0) {
                const bool has = lane < nbig;
                PullBig big{0, 0, 0, 0.0, 0.0, 0.0};
                double acc = 0.0;
                if (has) {
                    big = s_big[lane];
                    acc = s_bigacc[lane];
                }
                const bool lgb = finish(has, big.v, big.rv, big.xv, big.pv, big.rv + acc);
                if constexpr (BITS) {
                    if (lgb) atomicOr(&s_bits[(big.v >> 5) - 2 * t0], 1u << (big.v & 31));
                }
            }
        }
        STAMP(4);
        __syncthreads();
        if constexpr (BITS) { // the group's tiles are consecutive: 2 words per tile, complete
            if ((int)threadIdx.x < 2 * (t1 - t0) && t0 * WAVE + (int)threadIdx.x * 32 < V)
                act_out[2 * t0 + threadIdx.x] = s_bits[threadIdx.x];
        }
    }
    STAMP(5);
    // count of the next frontier: wave reduce, then ONE fire-and-forget atomic per workgroup
    int cw = wave_inclusive_scan(n_legal);
    if (lane == WAVE - 1) s_cnt[w] = cw;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < PULL_WAVES; ++k) tot += s_cnt[k];
        if (tot) atomicAdd(cnt_out, tot);
    }
    stat_add_edges<PULL_WAVES>(stats, edges, s_edges);
    STAMP(6);
}

// dense -> sparse: the frontier list {v : x[v] != 0} (k_inspect's compaction on the snapshot).
// Used when a sparse iteration follows a sweep, and by the frontier trace.
__global__ __launch_bounds__(BLOCK) void k_list_from_dense(const double *__restrict__ x, int V, const int *__restrict__ cnt_f,
                                                           int *__restrict__ ft, int *__restrict__ cnt) {
    __shared__ int s_buf[BLOCK * INSPECT_ITEMS];
    __shared__ int s_n;
    __shared__ int s_base;
    if (*cnt_f == 0) return;
    const int64_t chunk = (int64_t)BLOCK * INSPECT_ITEMS;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < V; base += (int64_t)gridDim.x * chunk) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < INSPECT_ITEMS; ++k) {
            const int64_t u = base + (int64_t)k * BLOCK + threadIdx.x;
            const bool hit = (u < V) && x[u] != 0.0;
            const uint64_t m = __ballot(hit);
            if (m) {
                int wbase = 0;
                if (lane_id() == 0) wbase = atomicAdd(&s_n, __popcll(m));
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

} // namespace dppr
