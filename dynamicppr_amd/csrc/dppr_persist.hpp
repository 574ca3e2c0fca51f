// dppr_persist.hpp -- a run of dense frontier iterations as ONE resident launch.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

// ---------------------------------------------------------------------------
// a4+a5, DENSE iterations on graphs whose sweep fits the chip in one wave of workgroups.
//
// On a small window (configs[1]: ~0.6 M edges) one sweep of k_pull_iter moves ~18 MB and is not
// bandwidth bound at all: its time is the dependent chain  grp_tile -> out_row_ptr -> out_col ->
// x[col] -> stores, plus the launch ramp, the kernel-end cache write-back and the gap to the next
// dispatch -- and a batch needs ~80 such iterations (gpu/PPRRevPushGPU.cuh:106-130 pays a blocking
// D2H on top of every one of them).
//
// k_pull_persist runs up to n_iter consecutive sweeps in one launch. Every workgroup owns ONE
// sweep group (<= PB consecutive vertices, cut by the graph builder) for the whole launch, so
// everything that does not change between iterations is computed once and kept on chip:
//   * row starts / lengths and their workgroup-wide prefix (LDS),
//   * the owner row, the out_col entry and the divisor (outdeg+1) of every edge slot (registers;
//     the group's concatenated edge list is dealt to the PB threads with stride PB, so long rows
//     are spread over all waves and out_col is read coalesced -- once),
//   * residual / pagerank / snapshot value of the thread's own vertex (registers; r and p go back
//     to memory when the launch ends).
// One iteration is then: gather x[col] (the only dependent memory hop), LDS-atomic the terms
// into the owners' sums, repair + threshold + next snapshot exactly as k_pull_iter::finish does,
// store x_new[v], and a grid barrier that also carries the size of the next frontier.
// The arithmetic per vertex is k_pull_iter's: the same terms (1-ALPHA)*x[u]/(outdeg(v)+1), summed
// into residual[v] (gpu/ExpandRev.cuh:70-73), the same repair (:708-743) and legal-push test.
//
// Grid barrier: workgroups arrive on one of BAR_SUBS counters (64-bit: arrivals << 32 | running
// sum of next-frontier counts); workgroup 0 watches the counters and publishes
// {generation, cumulative count} with ONE compare-and-swap on the word everybody polls. A
// workgroup that waits longer than the time limit swaps the same word to BAR_ABORT instead, so
// "barrier g completed" and "launch aborted at barrier g" are decided at a single point: either
// way every workgroup has finished exactly g sweeps and the state in memory is that of g complete
// iterations. The host then continues with per-iteration launches (dppr_engine.hip). The limit
// only matters if the grid is not co-resident (another context holding CUs): the engine sizes the
// grid to the occupancy the runtime reports, so it is a safety net, not a code path that is
// expected to run.
// ---------------------------------------------------------------------------
constexpr int BAR_SUBS = 16;
constexpr unsigned long long BAR_ABORT = ~0ull;
constexpr int PERSIST_SLOTS = 4; // edge slots per thread kept in registers (PB * 4 edges per group)

struct alignas(128) BarWord {
    unsigned long long w;
    unsigned long long pad[15];
};
struct GridBar {            // zeroed by the host before every launch
    BarWord gen;            // generation << 32 | cumulative next-frontier count; BAR_ABORT after a time-out
    BarWord sub[BAR_SUBS];  // arrivals << 32 | cumulative count of the workgroups with blockIdx % BAR_SUBS == s
};
constexpr int PERSIST_ABORTED = 1 << 30; // flag in the launch's status word (low bits: complete sweeps)

__device__ __forceinline__ unsigned long long bar_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The snapshot vectors are exchanged between workgroups on different XCDs (each XCD has its own
// L2) once per iteration. They are read and written with agent-scope accesses (sc1: through the
// L2 to the memory side), so an iteration needs no L2 write-back / invalidate at all; a full
// "release" per wave (buffer_wbl2) was measured at ~75 us per iteration for the 4096 waves.
__device__ __forceinline__ double x_load(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void x_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool bar_cas(unsigned long long *p, unsigned long long expect, unsigned long long desired) {
    return __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
}

template <int PB>
__global__ __launch_bounds__(PB) void k_pull_persist(int V, const int *__restrict__ grp_tile,
                                                     const int *__restrict__ out_row_ptr,
                                                     const int *__restrict__ out_col, double *xa, double *xb, double *r,
                                                     double *p, int *cnt, int cur0, int phase, double eps,
                                                     IterStats *stats, int *log, int n_iter, GridBar *bar,
                                                     int *status, unsigned long long limit_ticks) {
    constexpr int NW = PB / WAVE;
    constexpr int S = PERSIST_SLOTS;
    __shared__ int s_scan[PB + 1];
    __shared__ int s_rs[PB];
    __shared__ double s_acc[2][PB];
    __shared__ int s_wtot[NW];
    __shared__ int s_cnt[NW];
    __shared__ unsigned long long s_edges[NW];
    __shared__ unsigned s_next[2]; // barrier outcome: {completed, size of the next frontier}
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();

    // ---- static part: the group's vertices and edge slots
    const int t0 = grp_tile[blockIdx.x], t1 = grp_tile[blockIdx.x + 1];
    const int v = t0 * WAVE + tid;
    const bool valid = tid < (t1 - t0) * WAVE && v < V;
    int rs = 0, d = 0;
    double rv = 0.0, xv = 0.0, pv = 0.0;
    if (valid) {
        rs = out_row_ptr[v];
        d = out_row_ptr[v + 1] - rs;
        rv = r[v];
        xv = xa[v];
        pv = p[v];
    }
    unsigned F = (unsigned)__hip_atomic_load(cnt + cur0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int incl = wave_inclusive_scan(d);
    if (lane == WAVE - 1) s_wtot[w] = incl;
    __syncthreads();
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
    // owner of concatenated edge e = last row whose exclusive start is <= e (empty rows share the
    // start of their successor and are never the last such row)
    auto owner_of = [&](int e) {
        int lo = 0, hi = PB;
#pragma unroll
        for (int st = PB; st > 1; st >>= 1) {
            const int mid = (lo + hi) >> 1;
            if (s_scan[mid] <= e) lo = mid; else hi = mid;
        }
        return lo;
    };
    int own[S], col[S];
    double den[S];
    bool uni[S]; // wave-uniform: all 64 slots of this wave row belong to one (long) row
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int e = tid + k * PB;
        own[k] = -1;
        col[k] = 0;
        den[k] = 1.0;
        if (e < Eg) {
            const int o = owner_of(e);
            own[k] = o;
            col[k] = out_col[s_rs[o] + (e - s_scan[o])];
            den[k] = (double)(s_scan[o + 1] - s_scan[o] + 1);
        }
        const int first = __builtin_amdgcn_readfirstlane(own[k]);
        uni[k] = __ballot(own[k] >= 0 && own[k] == first) == ~0ull;
    }

    unsigned long long edges = 0;
    unsigned Cprev = 0; // cumulative next-frontier count published by the barriers so far
    int sweeps = 0, logged = 0;
    bool aborted = false;
    for (int it = 0; it < n_iter; ++it) {
        const double *xin = (it & 1) ? xb : xa;
        double *xout = (it & 1) ? xa : xb;
        // the gathers do not depend on F: issue them first
        double xg[S];
#pragma unroll
        for (int k = 0; k < S; ++k) xg[k] = own[k] >= 0 ? x_load(xin + col[k]) : 0.0;
        if (blockIdx.x == 0 && tid == 0) log[it] = (int)F;
        logged = it + 1;
        if (F == 0) break; // every workgroup sees the same F
        double *acc = s_acc[it & 1];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const bool nz = xg[k] != 0.0;
            const double term = ONE_MINUS_ALPHA * xg[k] / den[k];
            if (uni[k]) { // one long row: reduce in the wave, one LDS atomic
                const double sum = wave_sum(nz ? term : 0.0);
                if (lane == 0 && sum != 0.0) lds_add(&acc[own[k]], sum);
            } else if (nz) {
                lds_add(&acc[own[k]], term);
            }
            edges += (unsigned long long)__popcll(__ballot(nz));
        }
        // edges beyond the register slots (a group heavier than PB * PERSIST_SLOTS edges)
        for (int e = tid + S * PB; e < Eg; e += PB) {
            const int o = owner_of(e);
            const double xe = x_load(xin + out_col[s_rs[o] + (e - s_scan[o])]);
            const bool nz = xe != 0.0;
            if (nz) lds_add(&acc[o], ONE_MINUS_ALPHA * xe / (double)(s_scan[o + 1] - s_scan[o] + 1));
            edges += (unsigned long long)__popcll(__ballot(nz));
        }
        __syncthreads();
        // repair, threshold, next snapshot (k_pull_iter::finish), on registers
        double rn = acc[tid];
        if (xv != 0.0) rn -= xv;
        const bool lg = valid && legal(rn, phase, eps);
        if (valid) {
            rv = rn;
            xv = lg ? rn : 0.0;
            x_store(xout + v, xv); // every entry is rewritten: xout is a complete snapshot
            if (lg) pv = pv + ALPHA * rn;
        }
        s_acc[(it + 1) & 1][tid] = rv;
        const int wl = __popcll(__ballot(lg));
        if (lane == 0) s_cnt[w] = wl;
        ++sweeps;

        // ---- grid barrier #sweeps, carrying the next frontier's size
        __syncthreads(); // (workgroup release: every wave's xout stores are complete before wave 0 arrives)
        if (w == 0) {
            const unsigned gen = (unsigned)sweeps;
            const unsigned G = gridDim.x;
            const unsigned subs_used = G < (unsigned)BAR_SUBS ? G : (unsigned)BAR_SUBS;
            const unsigned long long old_word = ((unsigned long long)(gen - 1) << 32) | Cprev;
            const unsigned long long t_start = wall_clock64();
            const int part = lane < NW ? s_cnt[lane] : 0;
            const int tot = __builtin_amdgcn_readlane(wave_inclusive_scan(part), WAVE - 1);
            if (lane == 0)
                __hip_atomic_fetch_add(&bar->sub[blockIdx.x % BAR_SUBS].w, (1ull << 32) | (unsigned)tot, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) { // the deciding workgroup: lane s watches counter s
                const unsigned long long n_sub = lane < (int)subs_used ? (G - lane + BAR_SUBS - 1) / BAR_SUBS : 0;
                unsigned long long word = 0;
                unsigned polls = 0;
                bool gave_up = false;
                for (;;) {
                    if (lane < (int)subs_used) word = bar_load(&bar->sub[lane].w);
                    const bool here = lane >= (int)subs_used || (word >> 32) >= n_sub * gen;
                    if (__ballot(here) == ~0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((polls++ & 31u) == 0 && (bar_load(&bar->gen.w) == BAR_ABORT || wall_clock64() - t_start > limit_ticks)) {
                        gave_up = true;
                        break;
                    }
                }
                if (!gave_up) {
                    const unsigned mine = lane < (int)subs_used ? (unsigned)word : 0u;
                    const unsigned C = (unsigned)__builtin_amdgcn_readlane(wave_inclusive_scan((int)mine), WAVE - 1);
                    if (lane == 0) (void)bar_cas(&bar->gen.w, old_word, ((unsigned long long)gen << 32) | C);
                } else if (lane == 0) {
                    (void)bar_cas(&bar->gen.w, old_word, BAR_ABORT);
                }
            }
            if (lane == 0) {
                unsigned long long word;
                unsigned polls = 0;
                while ((word = bar_load(&bar->gen.w)) == old_word) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((polls++ & 31u) == 0 && wall_clock64() - t_start > limit_ticks)
                        (void)bar_cas(&bar->gen.w, old_word, BAR_ABORT); // decided here or by the publisher, never both
                }
                s_next[0] = word != BAR_ABORT;
                s_next[1] = (unsigned)word - Cprev;
            }
        }
        __syncthreads();
        if (!s_next[0]) {
            aborted = true;
            break;
        }
        F = s_next[1];
        Cprev += F;
    }

    // ---- the launch ends: registers back to memory, counters in the state the per-iteration
    // kernels expect (cnt[0] = live frontier size, the other two zero)
    if (valid) {
        r[v] = rv;
        p[v] = pv;
    }
    if (blockIdx.x == 0 && tid == 0) {
        for (int k = logged; k < n_iter; ++k) log[k] = 0; // iterations this launch did not get to
        cnt[0] = aborted ? 0 : (int)F;
        cnt[1] = 0;
        cnt[2] = 0;
        *status = sweeps | (aborted ? PERSIST_ABORTED : 0);
    }
    stat_add_edges<NW>(stats, edges, s_edges);
}

} // namespace dppr
