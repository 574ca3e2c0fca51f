// dppr_persist.hpp -- a run of dense frontier iterations as ONE resident launch.
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
// Co-residency and the grid barrier. A grid barrier only terminates if every workgroup of the
// launch is resident. The engine sizes the grid to the occupancy the runtime reports, and the
// kernel verifies it with a ROLL-CALL before it changes anything: every workgroup checks in on
// entry, workgroup 0 watches the check-ins and publishes READY with one compare-and-swap on the
// word everybody reads after their set-up; a workgroup that waits longer than the time limit swaps
// that word to BAR_ABORT instead. One word decides, so all workgroups agree, and an aborted launch
// has not touched the state (the host goes on with per-iteration launches, dppr_engine.hip).
// After a successful roll-call all workgroups are running and stay resident, so the per-iteration
// barriers always complete and can be the cheap all-to-all kind: a workgroup adds
// (1 << 32 | its next-frontier count) to one of BAR_SUBS counters -- there are two sets, used by
// odd and even iterations, so a fast workgroup's next arrival never mixes into the sums a slow one
// is still reading -- and then 16 lanes of its first wave poll the 16 counters of that parity until
// all arrivals are in: one memory-side atomic and one read, and the sum of the low halves is the
// size of the next frontier. A time-out there cannot be a residency problem; it is reported as a
// device fault (DPPR_ERR_HIP), never silently survived.
// ---------------------------------------------------------------------------
constexpr int BAR_SUBS = 16;
#ifndef DPPR_BAR_REPS
#define DPPR_BAR_REPS 16
#endif
#ifndef DPPR_BAR_SLEEP
#define DPPR_BAR_SLEEP 4
#endif
constexpr int BAR_REPS = DPPR_BAR_REPS;       // replicas of every arrival counter
constexpr int BAR_POLL_SLEEP = DPPR_BAR_SLEEP; // s_sleep units (64 clocks) between two polls
constexpr unsigned long long BAR_ABORT = ~0ull;
constexpr int PERSIST_SLOTS = 4; // edge slots per thread kept in registers (PB * 4 edges per group)

struct alignas(128) BarWord {
    unsigned long long w;
    unsigned long long pad[15];
};
struct GridBar {                // zeroed by the host before every launch
    BarWord gen;                // roll-call outcome: 0 pending, BAR_READY, BAR_ABORT
    BarWord roll[BAR_SUBS];     // roll-call check-ins of the workgroups with blockIdx % BAR_SUBS == s
    // per iteration parity and replica: arrivals << 32 | cumulative next-frontier count. Accesses to
    // ONE address serialise at the memory side (~10 ns each, measured: 242 workgroups polling the
    // same 16 words made the barrier take 4 us), so every workgroup arrives on all BAR_REPS replicas
    // of its counter (one 16-lane atomic instruction) and polls only the replica of its own
    // sixteen: no word sees more than ~16 arrivals or ~16 pollers.
    BarWord sub[2][BAR_REPS][BAR_SUBS];
};
constexpr unsigned long long BAR_READY = 1ull;
constexpr int PERSIST_ABORTED = 1 << 30; // status word: the roll-call failed, nothing was changed (low bits: complete sweeps)
constexpr int PERSIST_FAULT = 1 << 29;   // status word: a grid barrier timed out after a successful roll-call
constexpr int PERSIST_CONVERGED = 1 << 28; // status word: the frontier emptied; both snapshot vectors are all zero again
constexpr int PERSIST_SKIPPED = 1 << 27;   // status word: the launch was enqueued ahead and its guard said no
constexpr int PERSIST_SWEEPS = (1 << 16) - 1;

__device__ __forceinline__ unsigned long long bar_load(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The snapshot vectors are exchanged between workgroups on different XCDs (each XCD has its own
// L2) once per iteration. They are read and written with agent-scope accesses (sc1: through the
// L2 to the memory side) and every wave waits for its stores to complete (s_waitcnt vmcnt(0))
// before its workgroup arrives at the grid barrier, so an iteration needs no L2 write-back /
// invalidate: a full agent-scope release per wave (buffer_wbl2) was measured at ~75 us per
// iteration for the 4096 waves, one per workgroup at ~2.7 us.
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
                                                     int *status, unsigned long long limit_ticks, int rollcall_extra,
                                                     const int *guard) {
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
    // a launch enqueued ahead of time runs only if the one before it converged (dppr_engine.hip, batch_ahead)
    if (guard && !(__hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PERSIST_CONVERGED)) {
        if (blockIdx.x == 0 && tid == 0) *status = PERSIST_SKIPPED;
        return;
    }
    const unsigned G = gridDim.x;
    const unsigned subs_used = G < (unsigned)BAR_SUBS ? G : (unsigned)BAR_SUBS;
    // workgroups that share counter `lane` (lanes >= subs_used watch nothing)
    const unsigned long long n_sub = lane < (int)subs_used ? (G - lane + BAR_SUBS - 1) / BAR_SUBS : 0;
    const unsigned long long t_entry = wall_clock64();
    if (tid == 0) // roll-call: this workgroup is running
        __hip_atomic_fetch_add(&bar->roll[blockIdx.x % BAR_SUBS].w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

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

    // ---- roll-call outcome (the set-up above ran while the other workgroups were starting)
    if (w == 0) {
        if (blockIdx.x == 0) { // lane s watches check-in counter s
            unsigned polls = 0;
            bool all_here = false;
            for (;;) {
                const unsigned long long word = lane < (int)subs_used ? bar_load(&bar->roll[lane].w) : 0;
                // rollcall_extra > 0 (tests only) makes the roll-call wait for a workgroup that does not exist
                if (__ballot(word >= n_sub + (lane == 0 ? (unsigned long long)rollcall_extra : 0ull)) == ~0ull) {
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
                if ((polls++ & 31u) == 0 && wall_clock64() - t_entry > limit_ticks)
                    (void)bar_cas(&bar->gen.w, 0ull, BAR_ABORT); // decided here or by workgroup 0, never both
                __builtin_amdgcn_s_sleep(1);
            }
            s_next[0] = word == BAR_READY;
        }
    }
    __syncthreads();
    if (!s_next[0]) { // not co-resident: leave everything as it was
        if (blockIdx.x == 0 && tid == 0) *status = PERSIST_ABORTED;
        return;
    }

    unsigned long long edges = 0;
    unsigned Cpar[2] = {0u, 0u}; // (first wave) cumulative counts read from the two counter sets so far
    int sweeps = 0, logged = 0;
    bool fault = false, converged = false;
    for (int it = 0; it < n_iter; ++it) {
        const double *xin = (it & 1) ? xb : xa;
        double *xout = (it & 1) ? xa : xb;
        // the gathers do not depend on F: issue them first
        double xg[S];
#pragma unroll
        for (int k = 0; k < S; ++k) xg[k] = own[k] >= 0 ? x_load(xin + col[k]) : 0.0;
        if (blockIdx.x == 0 && tid == 0) log[it] = (int)F;
        logged = it + 1;
        if (F == 0) { // every workgroup sees the same F
            // the frontier is empty: the last sweep wrote an all-zero snapshot (xin); zero the one
            // before it too, so both vectors are clean for the next loop
            if (valid) x_store(xout + v, 0.0);
            converged = true;
            break;
        }
        PSTAMP(0);
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
        PSTAMP(1);
        __syncthreads();
        PSTAMP(2);
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
        PSTAMP(3);
        // every wave's xout stores must be COMPLETE before wave 0 arrives for the workgroup. The
        // barrier alone does not wait for them (a workgroup-scope release on gfx942/950 omits
        // vmcnt: the waves of a workgroup share their CU's L1), and a remote workgroup would read
        // last iteration's value.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PSTAMP(4);
        if (w == 0) {
            const int par = sweeps & 1;
            const unsigned long long rounds = (unsigned long long)((sweeps + 1) >> 1); // barriers of this parity so far
            const unsigned long long t_start = wall_clock64();
            const int part = lane < NW ? s_cnt[lane] : 0;
            const int tot = __builtin_amdgcn_readlane(wave_inclusive_scan(part), WAVE - 1);
            if (lane < BAR_REPS)
                __hip_atomic_fetch_add(&bar->sub[par][lane][blockIdx.x % BAR_SUBS].w, (1ull << 32) | (unsigned)tot,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int my_rep = (blockIdx.x / BAR_SUBS) % BAR_REPS;
            unsigned long long word = 0;
            unsigned polls = 0;
            bool ok = true;
            // (more polls in flight make the barrier SLOWER: four per workgroup took it from 5.4 K to
            // 8.3 K cycles -- reads of one word serialise at the memory side like atomics do)
            for (;;) {
                if (lane < (int)subs_used) word = bar_load(&bar->sub[par][my_rep][lane].w);
                if (__ballot((word >> 32) >= n_sub * rounds) == ~0ull) break;
                if ((polls++ & 63u) == 63u && wall_clock64() - t_start > limit_ticks + 100000000ull) { // + 1 s
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(BAR_POLL_SLEEP);
            }
            const unsigned C = (unsigned)__builtin_amdgcn_readlane(wave_inclusive_scan((int)(unsigned)word), WAVE - 1);
            if (lane == 0) {
                s_next[0] = ok;
                s_next[1] = C - Cpar[par];
            }
            Cpar[par] = C;
        }
        __syncthreads();
        PSTAMP(5);
        if (!s_next[0]) {
            fault = true;
            break;
        }
        F = s_next[1];
    }

    if (!fault && !converged && F == 0) { // emptied exactly at the last sweep this launch was given
        if (valid) x_store(((sweeps & 1) ? xa : xb) + v, 0.0);
        converged = true;
    }

    // ---- the launch ends: registers back to memory, counters in the state the per-iteration
    // kernels expect (cnt[0] = live frontier size, the other two zero)
    if (valid) {
        r[v] = rv;
        p[v] = pv;
    }
    if (blockIdx.x == 0 && tid == 0) {
        for (int k = logged; k < n_iter; ++k) log[k] = 0; // iterations this launch did not get to
        cnt[0] = (int)F;
        cnt[1] = 0;
        cnt[2] = 0;
        *status = sweeps | (fault ? PERSIST_FAULT : 0) | (converged ? PERSIST_CONVERGED : 0);
    }
    stat_add_edges<NW>(stats, edges, s_edges);
}

} // namespace dppr
