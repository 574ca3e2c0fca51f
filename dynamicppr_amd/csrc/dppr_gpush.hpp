// dppr_gpush.hpp -- the TAIL of a source group's frontier loop as pushes.
//
// A sweep of k_gsweep (dppr_multi.hpp) costs at least its floor -- every out_col entry, every group's tables: 61 us
// on the LiveJournal stand-in, 2.5 ms on the friendster one -- however few (vertex, source) pairs are still being
// pushed, and a loop ends with a dozen or more iterations of a few hundred pairs or fewer (the last mass trickling
// through the hubs). Those iterations run here as what the reference does for every iteration (gpu/ExpandRev.cuh:34-77
// + :708-743): per frontier vertex u, per in-neighbour v, residual[v] += (1.0-ALPHA)*residual[u]/(outdeg(v)+1) with a
// returning atomic, v enters the next frontier when that add takes it across the threshold -- for all sources of the
// group at once (an octet per edge, a lane per source, the 64 / 128-byte state row of v as one coalesced atomic
// instruction). The schedule stays the synchronous one of the sweeps: every frontier vertex's residual row is
// snapshotted (and zeroed where it is pushed) BEFORE any add of the iteration lands.
//
// State while in this mode is the plain one (residual[] complete, every activity bit clear except the ones that queue
// a vertex for the next iteration), i.e. exactly what a finished loop leaves: a loop that ends here needs no conversion
// back. Entering: the frontier's rows move from the snapshot back to residual[] (k_gpush_enter). Leaving early (the
// frontier grew again): the queued vertices' rows move to the snapshot, their bits are the sweep's bitmap
// (k_gpush_leave). One iteration = three launches (scan of the in-degrees, snapshot, expand); the sizes live in
// device memory (GPushCtl), so a chunk of iterations is enqueued without the host in between.
#pragma once

#include "dppr_multi.hpp"

namespace dppr {

constexpr int GPUSH_LOG = 16; // iterations one chunk may hold (rows of the log are indexed by it % GPUSH_LOG; a power of two)

struct GPushCtl {
    int n[2];          // vertices in list 0 / 1 (iteration it reads list it & 1, fills the other)
    int etot;          // in-edges of the current list (scan)
    int overflow;      // the next list did not hold every queued vertex: the bits do (the host leaves the mode)
    int it;            // iterations run in this mode so far
    int stop;          // set by the scan when an iteration would be too large for this mode: nothing further is changed
    unsigned int done; // workgroups of the running expand kernel that have finished
    int pad;
    int F[GPUSH_LOG][GS_MAX]; // legal (vertex, source) pairs each iteration of the chunk pushed, per source
    long long atomics[GPUSH_LOG]; // edge x source adds of each iteration
};

// activity bitmap -> vertex list (the frontier a sweep left), one thread per word
__global__ __launch_bounds__(BLOCK) void k_gpush_list(const uint32_t *__restrict__ act, int n_words, int *__restrict__ list, int cap,
                                                      GPushCtl *__restrict__ ctl) {
    for (int w = blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += gridDim.x * BLOCK) {
        uint32_t m = act[w];
        if (!m) continue;
        const int n = __popc(m);
        const int pos = atomicAdd(&ctl->n[0], n);
        for (int k = 0; m; ++k, m &= m - 1) {
            if (pos + k < cap) list[pos + k] = w * 32 + (__ffs((int)m) - 1);
            else ctl->overflow = 1;
        }
    }
}

// rows of the listed vertices between snapshot and residual[] (to_x = false: entering, x -> r; true: leaving, r -> x).
// An octet per vertex; the snapshot row is GW wide (lanes beyond the sources are zero), the residual row too.
template <int SPL, int GW>
__global__ __launch_bounds__(BLOCK) void k_gpush_rows(const int *__restrict__ list, const GPushCtl *__restrict__ ctl, int which,
                                                      double *__restrict__ x, double *__restrict__ r, bool to_x) {
    if (ctl->overflow) return; // (the list is incomplete: nothing moves, the scan calls the mode off)
    const int n = ctl->n[which];
    const int j = threadIdx.x & (OCT - 1);
    if (!oct_live<SPL, GW>(j)) return;
    for (int i = (blockIdx.x * BLOCK + threadIdx.x) / OCT; i < n; i += gridDim.x * BLOCK / OCT) {
        const size_t base = (size_t)list[i] * GW + j * SPL, xbase = (size_t)list[i] * x_stride(GW) + j * SPL;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            if (to_x) x[xbase + q] = r[base + q];
            else r[base + q] = x[xbase + q];
        }
    }
}

// Snapshot of an iteration (the head of ExpandUnifiedRev, gpu/ExpandRev.cuh:34-42, with the zeroing of
// PPRRevPushGPUFF's snapshot, gpu/Inspect.cuh:51-65): per listed vertex, the legal sources' residuals go to the
// snapshot row (0 elsewhere), pagerank += ALPHA * residual (iteration 0 excepted), residual = 0; its bit (which queued it) is cleared; its
// in-degree was scanned by k_gpush_scan, which runs FIRST and may call the iteration off.
template <int SPL, int GW>
__global__ __launch_bounds__(BLOCK) void k_gpush_snap(const int *__restrict__ list0, const int *__restrict__ list1, GPushCtl *ctl,
                                                      double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                      uint32_t *__restrict__ bits, int phase, double eps, int credit_first) {
    __shared__ int s_cnt[GS_MAX];
    if (ctl->stop) return;
    const int it = ctl->it, which = it & 1;
    const int n = ctl->n[which];
    const int *list = which ? list1 : list0;
    if (threadIdx.x < GS_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int j = threadIdx.x & (OCT - 1);
    int nleg[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) nleg[q] = 0;
    const bool live = oct_live<SPL, GW>(j);
    for (int i = (blockIdx.x * BLOCK + threadIdx.x) / OCT; i < n; i += gridDim.x * BLOCK / OCT) {
        const int u = list[i];
        const size_t base = (size_t)u * GW + j * SPL;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            if (!live) continue;
            const double rv = r[base + q];
            const bool lg = legal(rv, phase, eps);
            x[(size_t)u * x_stride(GW) + j * SPL + q] = lg ? rv : 0.0;
            if (lg) {
                // (the frontier a sweep hands over was credited when the sweep took its snapshot in place -- unless that was a
                // deferring sweep, dppr_multi.hpp "pagerank every other sweep": credit_first)
                if (it > 0 || credit_first) p[base + q] = p[base + q] + ALPHA * rv;
                r[base + q] = 0.0;
                nleg[q]++;
            }
        }
        if (j == 0) atomicAnd(&bits[u >> 5], ~(1u << (u & 31)));
    }
#pragma unroll
    for (int q = 0; q < SPL; ++q)
        if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
    __syncthreads();
    if (threadIdx.x < GW && s_cnt[threadIdx.x]) atomicAdd(&ctl->F[it & (GPUSH_LOG - 1)][threadIdx.x], s_cnt[threadIdx.x]);
}

// First kernel of an iteration: exclusive scan of the listed vertices' in-degrees (one workgroup; the list of a tail
// iteration is short) and the decision whether the iteration belongs here at all: more than max_n vertices or more
// than max_edges in-edges (the late frontier is made of hubs) and `stop` is raised -- snapshot and expand of this
// and of every later enqueued iteration then return at once, the state is untouched and the host goes back to sweeps.
__global__ __launch_bounds__(1024) void k_gpush_scan(GPushCtl *ctl, const int *__restrict__ list0, const int *__restrict__ list1,
                                                     const int *__restrict__ in_row_ptr, int *__restrict__ pre, int max_n,
                                                     long long max_edges) {
    __shared__ long long s_w[16];
    __shared__ long long s_carry;
    if (ctl->stop) return;
    const int which = ctl->it & 1;
    const int n = ctl->n[which];
    const int *list = which ? list1 : list0;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    if (n > max_n || ctl->overflow) {
        if (tid == 0) ctl->stop = 1;
        return;
    }
    if (tid < GS_MAX) ctl->F[ctl->it & (GPUSH_LOG - 1)][tid] = 0; // (this iteration's row of the log: the snapshot adds to it)
    if (tid == 0) ctl->atomics[ctl->it & (GPUSH_LOG - 1)] = 0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        int d = 0;
        if (i < n) {
            const int u = list[i];
            d = in_row_ptr[u + 1] - in_row_ptr[u];
        }
        const int inc = wave_inclusive_scan(d); // (a wave's 64 rows: < 2^31 as long as the epoch's edges are)
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < w; ++k) woff += s_w[k];
        const long long carry = s_carry;
        const long long mine = carry + woff + inc - d;
        if (i < n) pre[i] = (int)(mine < 0x7fffffffll ? mine : 0x7fffffffll);
        __syncthreads();
        if (tid == 1023) s_carry = carry + woff + inc;
        __syncthreads();
    }
    if (tid == 0) {
        const long long tot = s_carry;
        if (tot > max_edges || tot > 0x7fffffffll) ctl->stop = 1;
        else {
            pre[n] = (int)tot;
            ctl->etot = (int)tot;
            ctl->n[which ^ 1] = 0;
        }
    }
}

// Expand of an iteration: the in-edges of the listed vertices, dealt evenly over all octets (an octet finds the vertex
// of its first edge by bisection of the scan, then walks). Per edge u <- v and source lane with a snapshot value:
// residual[v] += (1.0-ALPHA)*x/(outdeg(v)+1) with a returning atomic; the add that takes residual[v] across the
// threshold queues v (its activity bit; the first to set it appends v to the next list) -- adds of a phase have one
// sign, so exactly one add per (v, source) sees the crossing. The last workgroup to finish closes the iteration.
template <int SPL, int GW>
__global__ __launch_bounds__(BLOCK) void k_gpush_expand(const int *__restrict__ list0, const int *__restrict__ list1, GPushCtl *ctl,
                                                        const int *__restrict__ pre, const int *__restrict__ in_row_ptr,
                                                        const Adj *__restrict__ adj, const int *__restrict__ hub_degp1,
                                                        const double *__restrict__ x, double *__restrict__ r,
                                                        uint32_t *__restrict__ bits, int *__restrict__ nlist0, int *__restrict__ nlist1,
                                                        int cap, int phase, double eps, IterStats *__restrict__ stats) {
    __shared__ unsigned long long s_edges;
    if (ctl->stop) return;
    const int it = ctl->it, which = it & 1;
    const int n = ctl->n[which], etot = ctl->etot;
    const int *list = which ? list1 : list0;
    int *nlist = which ? nlist0 : nlist1; // (the other list)
    if (threadIdx.x == 0) s_edges = 0ull;
    __syncthreads();
    const int j = threadIdx.x & (OCT - 1);
    const bool live = oct_live<SPL, GW>(j);
    const long long n_oct = (long long)gridDim.x * (BLOCK / OCT);
    const long long oct = (long long)blockIdx.x * (BLOCK / OCT) + threadIdx.x / OCT;
    const long long per = (etot + n_oct - 1) / n_oct;
    long long e0 = oct * per, e1 = e0 + per < (long long)etot ? e0 + per : (long long)etot;
    unsigned ecount = 0;
    if (e0 < e1 && n > 0) {
        int lo = 0, hi = n; // largest i with pre[i] <= e0
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= (int)e0) lo = mid; else hi = mid;
        }
        int i = lo;
        while (e0 < e1) {
            while (pre[i + 1] <= (int)e0) ++i; // (vertices without in-edges are stepped over)
            const int u = list[i];
            double xv[SPL];
#pragma unroll
            for (int q = 0; q < SPL; ++q) xv[q] = live ? x[(size_t)u * x_stride(GW) + j * SPL + q] : 0.0;
            const long long row_end = pre[i + 1] < (int)e1 ? pre[i + 1] : e1;
            const Adj *row = adj + in_row_ptr[u] + ((int)e0 - pre[i]);
            for (int k = 0; k < (int)(row_end - e0); ++k) {
                const Adj a = row[k];
                const int dp1 = a.degp1 < 0 ? hub_degp1[~a.degp1] : a.degp1;
                const double den = (double)dp1;
                bool crossed = false;
#pragma unroll
                for (int q = 0; q < SPL; ++q) {
                    if (xv[q] != 0.0) {
                        const double add = ONE_MINUS_ALPHA * xv[q] / den;
                        const double prer = atomic_add_ret(&r[(size_t)a.v * GW + j * SPL + q], add);
                        crossed |= !legal(prer, phase, eps) && legal(prer + add, phase, eps);
                        ++ecount;
                    }
                }
                if (oct_mask(__ballot(crossed)) && j == 0) {
                    const uint32_t bit = 1u << (a.v & 31);
                    const uint32_t old = atomicOr(&bits[a.v >> 5], bit);
                    if (!(old & bit)) {
                        const int pos = atomicAdd(&ctl->n[which ^ 1], 1);
                        if (pos < cap) nlist[pos] = a.v;
                        else ctl->overflow = 1;
                    }
                }
            }
            e0 = row_end;
        }
    }
    if (ecount) atomicAdd(&s_edges, (unsigned long long)ecount);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_edges) {
            stats->blk_E[blockIdx.x] += s_edges;
            atomicAdd(reinterpret_cast<unsigned long long *>(&ctl->atomics[it & (GPUSH_LOG - 1)]), s_edges);
        }
        __threadfence();
        if (atomicAdd(&ctl->done, 1u) == gridDim.x - 1) { // everybody's appends are in: the iteration is over
            ctl->done = 0u;
            if (ctl->n[which ^ 1] > cap) ctl->n[which ^ 1] = cap; // (overflow is flagged; the bits hold the whole set)
            ctl->it = it + 1;
        }
    }
}

// A RUN of iterations on a frontier of a few hundred vertices as ONE single-workgroup launch: the scan, the snapshot
// and the expand of every iteration with workgroup barriers in between, lists, scan and snapshot rows in LDS. What the
// waves exchange through global memory goes through the L2 only: residual[] is read with agent-scope loads, zeroed with
// a returning exchange and added to with returning atomics (a wave that waits for the value knows the operation is
// done), pagerank is credited with an atomic add, the activity bits with returning atomics. Stops -- at an iteration
// boundary, lists and counters written back for whoever goes on -- when the frontier is empty, when an iteration is too
// large for it (more than TINY_N vertices or TINY_E in-edges: the three-kernel form, with an octet per edge, takes it) or after max_it
// iterations; a next list that outgrows TINY_N raises `stop` (the bits hold the set: the host leaves the mode).
constexpr int TINY_N = 512;
constexpr int TINY_E = 1024; // (an octet walks its share of the edges one returning atomic after the other: 8 per octet at most)
// gfx950 only: the 16-wide form keeps TINY_N snapshot rows of 128 bytes in LDS (64 KB) next to the lists and the scan,
// about 70 KB of static LDS -- beyond the 64 KB of earlier targets, within the 160 KB of a CDNA4 CU
static_assert(sizeof(double) * TINY_N * 16 + sizeof(int) * (3 * TINY_N + 1) + 1024 <= 160 * 1024,
              "k_gpush_tiny: TINY_N snapshot rows + lists + scan must fit the 160 KB LDS of a gfx950 CU");

template <int SPL, int GW>
__global__ __launch_bounds__(1024) void k_gpush_tiny(GPushCtl *ctl, int *__restrict__ list0, int *__restrict__ list1,
                                                     const int *__restrict__ in_row_ptr, const Adj *__restrict__ adj,
                                                     const int *__restrict__ hub_degp1, double *r, double *p, uint32_t *bits, int phase,
                                                     double eps, IterStats *__restrict__ stats, int max_it, int credit_first) {
    constexpr int NT = 1024, NOCT = NT / OCT;
    __shared__ int s_list[2][TINY_N];
    __shared__ int s_pre[TINY_N + 1];
    __shared__ double s_x[TINY_N * GW];
    __shared__ int s_w[NT / WAVE];
    __shared__ int s_n[2];
    __shared__ int s_cnt[GS_MAX];
    __shared__ int s_flag; // 1: the next list overflowed
    __shared__ unsigned long long s_edges;
    if (ctl->stop || ctl->overflow) return;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int j = tid & (OCT - 1), oid = tid / OCT;
    const bool live = oct_live<SPL, GW>(j);
    int it = ctl->it, which = it & 1;
    const int n0 = ctl->n[which];
    if (n0 > TINY_N) return; // (not a frontier for this kernel)
    for (int i = tid; i < n0; i += NT) s_list[which][i] = (which ? list1 : list0)[i];
    if (tid == 0) {
        s_n[which] = n0;
        s_n[which ^ 1] = 0;
        s_flag = 0;
        s_edges = 0ull;
    }
    if (tid < GS_MAX) s_cnt[tid] = 0;
    unsigned long long edges_before = 0; // (thread 0)
    for (int step = 0; step < max_it; ++step) {
        __syncthreads();
        const int n = s_n[which];
        if (n == 0 || s_flag) break;
        // ---- scan of the in-degrees (one vertex per thread: n <= TINY_N <= NT)
        int d = 0;
        if (tid < n) {
            const int u = s_list[which][tid];
            d = in_row_ptr[u + 1] - in_row_ptr[u];
        }
        const int inc = wave_inclusive_scan(d);
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < NT / WAVE; ++k) {
            woff += k < w ? s_w[k] : 0;
            tot += s_w[k];
        }
        if (tid < n) s_pre[tid] = woff + inc - d;
        if (tid == 0) s_pre[n] = tot;
        if (tot > TINY_E) break; // (uniform: every thread computed the same total)
        // ---- snapshot
        int nleg[SPL];
#pragma unroll
        for (int q = 0; q < SPL; ++q) nleg[q] = 0;
        for (int i = oid; i < n; i += NOCT) {
            const int u = s_list[which][i];
            const size_t base = (size_t)u * GW + j * SPL;
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                if (!live) continue;
                double rv = gs_ld<true>(r + base + q);
                const bool lg = legal(rv, phase, eps);
                if (lg) {
                    rv = atomic_exch(r + base + q, 0.0); // (nobody adds between the load and this: same value)
                    if (it > 0 || credit_first) (void)__hip_atomic_fetch_add(p + base + q, ALPHA * rv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nleg[q]++;
                }
                s_x[i * GW + j * SPL + q] = lg ? rv : 0.0;
            }
            if (j == 0) (void)atomicAnd(&bits[u >> 5], ~(1u << (u & 31)));
        }
#pragma unroll
        for (int q = 0; q < SPL; ++q)
            if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the credits and bit clears of this wave are at the L2)
        __syncthreads();
        if (tid < GS_MAX) {
            ctl->F[it & (GPUSH_LOG - 1)][tid] = s_cnt[tid];
            s_cnt[tid] = 0;
        }
        // ---- expand
        const int e_per = (tot + NOCT - 1) / NOCT;
        int e0 = oid * e_per;
        const int e1 = min(tot, e0 + e_per);
        unsigned long long adds = 0;
        if (e0 < e1) {
            int lo = 0, hi = n;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_pre[mid] <= e0) lo = mid; else hi = mid;
            }
            int i = lo;
            while (e0 < e1) {
                while (s_pre[i + 1] <= e0) ++i;
                const int u = s_list[which][i];
                double xv[SPL];
#pragma unroll
                for (int q = 0; q < SPL; ++q) xv[q] = live ? s_x[i * GW + j * SPL + q] : 0.0;
                const int row_end = min(s_pre[i + 1], e1);
                const Adj *row = adj + in_row_ptr[u] + (e0 - s_pre[i]);
                for (int k = 0; k < row_end - e0; ++k) {
                    const Adj a = row[k];
                    const double den = (double)(a.degp1 < 0 ? hub_degp1[~a.degp1] : a.degp1);
                    bool crossed = false;
#pragma unroll
                    for (int q = 0; q < SPL; ++q) {
                        if (xv[q] != 0.0) {
                            const double add = ONE_MINUS_ALPHA * xv[q] / den;
                            const double prer = atomic_add_ret(&r[(size_t)a.v * GW + j * SPL + q], add);
                            crossed |= !legal(prer, phase, eps) && legal(prer + add, phase, eps);
                            ++adds;
                        }
                    }
                    if (oct_mask(__ballot(crossed)) && j == 0) {
                        const uint32_t bit = 1u << (a.v & 31);
                        const uint32_t old = atomicOr(&bits[a.v >> 5], bit);
                        if (!(old & bit)) {
                            const int pos = atomicAdd(&s_n[which ^ 1], 1);
                            if (pos < TINY_N) s_list[which ^ 1][pos] = a.v;
                            else s_flag = 1;
                        }
                    }
                }
                e0 = row_end;
            }
        }
        if (adds) atomicAdd(&s_edges, adds);
        __syncthreads();
        if (tid == 0) {
            ctl->atomics[it & (GPUSH_LOG - 1)] = (long long)(s_edges - edges_before);
            edges_before = s_edges;
            s_n[which] = 0; // (becomes the next "next" list)
        }
        ++it;
        which ^= 1;
    }
    __syncthreads();
    // ---- hand over: the live list (which), its size, the iteration count
    const int n = min(s_n[which], TINY_N);
    for (int i = tid; i < n; i += NT) (which ? list1 : list0)[i] = s_list[which][i];
    if (tid == 0) {
        ctl->n[which] = n;
        ctl->n[which ^ 1] = 0;
        ctl->it = it;
        if (s_flag) ctl->stop = 1; // the list is incomplete, the bits are not: back to sweeps
        if (s_edges) stats->blk_E[0] += s_edges;
    }
}

// Leaving the mode before the loop is over (the frontier grew again): the queued vertices -- exactly the set bits --
// get their residual rows into the snapshot and their pagerank share (credit: not if no iteration ran here -- then
// they are the frontier the sweep handed over, credited already), which is what a sweep expects of an active vertex, and the per-source
// frontier sizes the sweep's launches look at are counted. An octet per vertex over all ids.
template <int SPL, int GW>
__global__ __launch_bounds__(BLOCK) void k_gpush_leave(int V, const uint32_t *__restrict__ bits, double *__restrict__ x,
                                                       const double *__restrict__ r, double *__restrict__ p, int credit, int phase,
                                                       double eps, int *__restrict__ cnt_out) {
    __shared__ int s_cnt[GS_MAX];
    if (threadIdx.x < GS_MAX) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int j = threadIdx.x & (OCT - 1);
    const bool live = oct_live<SPL, GW>(j);
    int nleg[SPL];
#pragma unroll
    for (int q = 0; q < SPL; ++q) nleg[q] = 0;
    for (int v = (blockIdx.x * BLOCK + threadIdx.x) / OCT; v < V; v += gridDim.x * BLOCK / OCT) {
        if (!((bits[v >> 5] >> (v & 31)) & 1u) || !live) continue;
        const size_t base = (size_t)v * GW + j * SPL;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const double rv = r[base + q];
            x[(size_t)v * x_stride(GW) + j * SPL + q] = rv;
            if (legal(rv, phase, eps)) { // a sweep's frontier has its pagerank share already (snapshot in place)
                if (credit) p[base + q] = p[base + q] + ALPHA * rv;
                nleg[q]++;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SPL; ++q)
        if (nleg[q]) atomicAdd(&s_cnt[j * SPL + q], nleg[q]);
    __syncthreads();
    if (threadIdx.x < GW && s_cnt[threadIdx.x]) atomicAdd(&cnt_out[threadIdx.x], s_cnt[threadIdx.x]);
}

} // namespace dppr
