// dppr_update.hpp -- IncrementalBatchUpdate: lock-free grouped stream update and phase-1 filter.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

// ---------------------------------------------------------------------------
// a7  IncrementalBatchUpdate (gpu/PPRRevPushGPU.cuh:21-28; kernels
// gpu/StreamUpdate.cuh:7-76), lock-free formulation.
//
// The reference serialises records that share a tail u with a per-vertex spin
// lock taken inside a WarpAny loop; under wave64 lock-step that is a deadlock
// hazard and its application order is arbitrary. Here the records are stably
// grouped by tail (radix sort of (u, index)), and one lane applies each group in
// batch-index order -- exactly the order cpu/PPRCPUMTCilkRev.h:108-124 applies them
// at -t 1, so the updated residuals are bit-identical to that CPU path. Records of
// different tails are independent (only r[u] and predeg[u] are written; p is
// read-only during the update).
//
//  k_su_keys : keys = tail u, vals = record index
//  k_su_terms: per record (parallel): t = (1-ALPHA)*p[v] - p[u]   (first two terms of
//              the reference's add expression, evaluated left to right)
//  k_su_apply: per group leader (sequential over the group):
//              add = t - ALPHA*r[u] + ALPHA*[u==s]
//              insert: d++; r[u] += add/(d+1)/ALPHA    delete: d--; r[u] -= add/(d+1)/ALPHA
//              where d starts at the PRE-batch out-degree = post-batch degree reverted
//              by the group's own records (CopyOutDegree + RevertOutDegree).
//              Afterwards the leader seeds the phase-0 frontier (r[u] > eps) and the
//              phase-1 candidate list (r[u] < -eps): only tails can leave [-eps, eps]
//              (cpu/PPRCPUMTCilkRev.h:126-156 seeds from batch endpoints for the same reason).
// ---------------------------------------------------------------------------
// (also zeroes `nz` 8-byte words at `zero` and `nzi` ints at `zero_ints` when given: the first kernel of a
// batch clears the counters of what follows instead of separate fills. Tried for the grouping of small
// batches and NOT faster than the device radix sort's five launches (26 us of kernels + ~20 us of gaps for
// the 12 K records of the configs[1] stand-in): one workgroup's bitonic sort in LDS (3x slower), rocPRIM's
// single-workgroup path (39 us), and a hand-written single-workgroup stable LSD radix sort, two 10-bit
// passes with ballot-ladder ranks (45 us: 160 ballot steps per wave and pass on ONE CU's four SIMDs))
__global__ __launch_bounds__(BLOCK) void k_su_keys(const int *__restrict__ e1, int L, uint32_t *__restrict__ keys,
                                                   uint32_t *__restrict__ vals, unsigned long long *__restrict__ zero,
                                                   int nz, int *__restrict__ zero_ints, int nzi) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nz; i += gridDim.x * BLOCK) zero[i] = 0ull;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nzi; i += gridDim.x * BLOCK) zero_ints[i] = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < L; i += gridDim.x * BLOCK) {
        keys[i] = (uint32_t)e1[i];
        vals[i] = (uint32_t)i;
    }
}

// The grouping INSIDE the timed region (round 5; the reference times all of IncrementalBatchUpdate, gpu/PPRGPU.cuh:138-164,
// gpu/StreamUpdate.cuh:7-33, so the default accounting keeps it there). For batches of up to SU_RANK_MAX records the stable
// order by tail needs no sort passes: record i goes to position  #{j : (tail_j, j) < (tail_i, i)}. One thread per record,
// 256 records per workgroup; the batch's tails pass through LDS in tiles of SU_RANK_TILE, every thread compares its own
// key with four broadcast keys per LDS read. L^2 / 256 comparisons per workgroup, L / 256 workgroups side by side: a few
// microseconds for a thousand records, where the device radix sort is five dispatches and 60-80 us with their gaps; beyond
// SU_RANK_MAX the quadratic work catches up with the sort (12 K records: ~40 us either way) and the sort is used. (First form
// of this kernel: the inner loop over b1 through the scalar cache -- one s_load_dwordx8 per 8 pairs, waited for every time:
// 400 us for 12 K records.) The same launch does CopyOutDegree
// (gpu/StreamUpdate.cuh:7-17) -- the post-batch out-degree of a tail is the length of its row in the epoch's out-CSR, built
// from the post-batch window -- and clears the loop's counters.
constexpr int SU_RANK_MAX = 4096;
constexpr int SU_RANK_TILE = 2048;
__global__ __launch_bounds__(BLOCK) void k_su_group_rank(const int *__restrict__ e1, int L, const int *__restrict__ out_row_ptr,
                                                        uint32_t *__restrict__ skeys, uint32_t *__restrict__ svals,
                                                        int *__restrict__ deg_after, unsigned long long *__restrict__ zero, int nz,
                                                        int *__restrict__ zero_ints, int nzi) {
    __shared__ int4 s_t[SU_RANK_TILE / 4];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < nz; i += BLOCK) zero[i] = 0ull;
        for (int i = threadIdx.x; i < nzi; i += BLOCK) zero_ints[i] = 0;
    }
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int t = i < L ? e1[i] : 0x7ffffffe; // (threads beyond the batch rank nothing)
    if (i < L && out_row_ptr) deg_after[i] = out_row_ptr[t + 1] - out_row_ptr[t];
    int *s_flat = reinterpret_cast<int *>(s_t);
    int rank = 0;
    for (int j0 = 0; j0 < L; j0 += SU_RANK_TILE) {
        __syncthreads();
        for (int k = threadIdx.x; k < SU_RANK_TILE; k += BLOCK) s_flat[k] = j0 + k < L ? e1[j0 + k] : 0x7fffffff; // (padding ranks behind every record)
        __syncthreads();
        const int n4 = (min(SU_RANK_TILE, L - j0) + 3) / 4;
        // records that precede i in batch order count on equality, the others only when smaller: tail_j < t + [j < i].
        // Quads entirely before i, the one that straddles it, quads behind it: one comparison and one add per pair.
        const int nb = min(max((i - j0) >> 2, 0), n4);
        const int t1 = t + 1;
        for (int k4 = 0; k4 < nb; ++k4) {
            const int4 q = s_t[k4];
            rank += (q.x < t1 ? 1 : 0) + (q.y < t1 ? 1 : 0) + (q.z < t1 ? 1 : 0) + (q.w < t1 ? 1 : 0);
        }
        if (nb < n4) {
            const int4 q = s_t[nb];
            const int j = j0 + 4 * nb;
            rank += (q.x < t + (j < i ? 1 : 0) ? 1 : 0) + (q.y < t + (j + 1 < i ? 1 : 0) ? 1 : 0) + (q.z < t + (j + 2 < i ? 1 : 0) ? 1 : 0) +
                    (q.w < t + (j + 3 < i ? 1 : 0) ? 1 : 0);
        }
        for (int k4 = nb + 1; k4 < n4; ++k4) {
            const int4 q = s_t[k4];
            rank += (q.x < t ? 1 : 0) + (q.y < t ? 1 : 0) + (q.z < t ? 1 : 0) + (q.w < t ? 1 : 0);
        }
    }
    if (i < L) {
        skeys[rank] = (uint32_t)t;
        svals[rank] = (uint32_t)i;
    }
}

// ---- The grouping of LARGER batches inside the timed region, without a library sort (round 6; rounds 1-5:
// rocprim::radix_sort_pairs -- a block sort and eight merge launches for the headline's 138 K records). Three launches:
//   k_su_grp_hist    : histogram of the tails' LOW bits (nb <= 4 096 buckets; LDS per workgroup, one global add
//                      per workgroup and occupied bucket) + CopyOutDegree + the loop's counters cleared;
//   k_su_grp_scatter : every workgroup scans the nb counts itself (no scan launch), claims room for its 2 048 records per bucket
//                      with one atomic each and writes (tail, index) pairs into the buckets -- in NO particular order inside one;
//   k_su_grp_rank    : one workgroup per 256 records of a bucket: record i goes to  bucket start + #{j in the bucket :
//                      (tail_j, j) < (tail_i, i)} -- k_su_group_rank's rule inside a bucket, the pairs passing through LDS in tiles.
// The result is a stable GROUPING by tail: every tail's records side by side in batch order, tails ascending inside a bucket (that is all
// IncrementalBatchUpdate and the seeding need: leaders are found by comparing neighbours). The LOW bits pick the bucket on purpose: ids are
// numbered by falling in-degree (dppr_idspace.hpp), and the id range of the first high-bit bucket owned a third of a twitter-size batch's
// records -- 39 ms of ranking. Work per thread of the last launch = the size of its bucket (a few hundred records; a hub tail's thousands
// at most; a batch whose records all share ONE tail degenerates to L comparisons per thread -- bounded, never seen).
constexpr int SU_GRP_MAX_BUCKETS = 4096, SU_GRP_PER_WG = 2048, SU_GRP_TILE = 2048, SU_GRP_MAX_RECORDS = 1 << 22;
__global__ __launch_bounds__(BLOCK) void k_su_grp_hist(const int *__restrict__ e1, int L, int nb, const int *__restrict__ out_row_ptr,
                                                      int *__restrict__ deg_after, int *__restrict__ hist, unsigned long long *__restrict__ zero, int nz,
                                                      int *__restrict__ zero_ints, int nzi) {
    __shared__ int s_h[SU_GRP_MAX_BUCKETS];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < nz; i += BLOCK) zero[i] = 0ull;
        for (int i = threadIdx.x; i < nzi; i += BLOCK) zero_ints[i] = 0;
    }
    for (int k = threadIdx.x; k < nb; k += BLOCK) s_h[k] = 0;
    __syncthreads();
    const int i0 = blockIdx.x * SU_GRP_PER_WG, i1 = min(i0 + SU_GRP_PER_WG, L);
    for (int i = i0 + threadIdx.x; i < i1; i += BLOCK) {
        const int t = e1[i];
        deg_after[i] = out_row_ptr[t + 1] - out_row_ptr[t]; // CopyOutDegree (gpu/StreamUpdate.cuh:7-17)
        atomicAdd(&s_h[t & (nb - 1)], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += BLOCK)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}
// ctl: [0, nb] bucket starts, [nb + 1, 2 nb + 1] first slice (256 records) of every bucket, written by workgroup 0 for the ranking launch
__global__ __launch_bounds__(BLOCK) void k_su_grp_scatter(const int *__restrict__ e1, int L, int nb, const int *__restrict__ hist,
                                                         int *__restrict__ cursor, int *__restrict__ ctl, uint32_t *__restrict__ tk,
                                                         uint32_t *__restrict__ tv) {
    __shared__ int s_start[SU_GRP_MAX_BUCKETS], s_cnt[SU_GRP_MAX_BUCKETS];
    __shared__ int s_wsum[WAVES_PER_BLOCK], s_wsl[WAVES_PER_BLOCK];
    const int lane = lane_id(), w = wave_id();
    // exclusive scan of the nb counts (and of their slice counts): each thread owns nb / BLOCK consecutive buckets
    const int per = (nb + BLOCK - 1) / BLOCK;
    int mine = 0, mine_sl = 0;
    for (int k = 0; k < per; ++k) {
        const int b = threadIdx.x * per + k;
        const int c = b < nb ? hist[b] : 0;
        mine += c;
        mine_sl += (c + BLOCK - 1) / BLOCK;
    }
    const int inc = wave_inclusive_scan(mine), inc_sl = wave_inclusive_scan(mine_sl);
    if (lane == WAVE - 1) {
        s_wsum[w] = inc;
        s_wsl[w] = inc_sl;
    }
    __syncthreads();
    int run = inc - mine, run_sl = inc_sl - mine_sl;
    for (int k = 0; k < w; ++k) {
        run += s_wsum[k];
        run_sl += s_wsl[k];
    }
    for (int k = 0; k < per; ++k) {
        const int b = threadIdx.x * per + k;
        if (b < nb) {
            const int c = hist[b];
            s_start[b] = run;
            s_cnt[b] = 0;
            if (blockIdx.x == 0) {
                ctl[b] = run;
                ctl[nb + 1 + b] = run_sl;
            }
            run += c;
            run_sl += (c + BLOCK - 1) / BLOCK;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == BLOCK - 1) {
        ctl[nb] = run; // (= L)
        ctl[2 * nb + 1] = run_sl;
    }
    __syncthreads();
    // this workgroup's records: count per bucket, one global claim per occupied bucket, then the placements
    const int i0 = blockIdx.x * SU_GRP_PER_WG, i1 = min(i0 + SU_GRP_PER_WG, L);
    for (int i = i0 + threadIdx.x; i < i1; i += BLOCK) atomicAdd(&s_cnt[e1[i] & (nb - 1)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += BLOCK) {
        const int c = s_cnt[b];
        if (c) s_start[b] += atomicAdd(&cursor[b], c);
        s_cnt[b] = 0;
    }
    __syncthreads();
    for (int i = i0 + threadIdx.x; i < i1; i += BLOCK) {
        const int t = e1[i], b = t & (nb - 1);
        const int pos = s_start[b] + atomicAdd(&s_cnt[b], 1);
        tk[pos] = (uint32_t)t;
        tv[pos] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(BLOCK) void k_su_grp_rank(const uint32_t *__restrict__ tk, const uint32_t *__restrict__ tv, const int *__restrict__ ctl, int nb,
                                                      uint32_t *__restrict__ skeys, uint32_t *__restrict__ svals, int *__restrict__ hist,
                                                      int *__restrict__ cursor) {
    __shared__ unsigned long long s_key[SU_GRP_TILE];
    const int *bstart = ctl, *sstart = ctl + nb + 1;
    const int wg = blockIdx.x;
    if (wg >= sstart[nb]) return; // (the grid is the upper bound L / 256 + nb)
    int lo = 0, hi = nb; // the bucket of this slice: the LAST bucket whose first slice is <= wg (empty buckets share their successor's)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sstart[mid] <= wg) lo = mid; else hi = mid;
    }
    const int b = lo, base = bstart[b], n = bstart[b + 1] - base;
    const int i = (wg - sstart[b]) * BLOCK + (int)threadIdx.x;
    const unsigned long long me = i < n ? ((unsigned long long)tk[base + i] << 32) | tv[base + i] : ~0ull;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += SU_GRP_TILE) {
        __syncthreads();
        for (int k = threadIdx.x; k < SU_GRP_TILE; k += BLOCK)
            s_key[k] = j0 + k < n ? ((unsigned long long)tk[base + j0 + k] << 32) | tv[base + j0 + k] : ~0ull; // (padding ranks behind every record)
        __syncthreads();
        const int m = min(SU_GRP_TILE, n - j0);
        int k = 0;
        for (; k + 3 < m; k += 4)
            rank += (s_key[k] < me ? 1 : 0) + (s_key[k + 1] < me ? 1 : 0) + (s_key[k + 2] < me ? 1 : 0) + (s_key[k + 3] < me ? 1 : 0);
        for (; k < m; ++k) rank += s_key[k] < me ? 1 : 0;
    }
    if (i < n) {
        skeys[base + rank] = (uint32_t)(me >> 32);
        svals[base + rank] = (uint32_t)me;
    }
    // the counters of the NEXT batch's first two launches (this launch reads neither): cleared by the slices of bucket 0 .. nb / 256
    if (wg * BLOCK + (int)threadIdx.x < nb) {
        hist[wg * BLOCK + threadIdx.x] = 0;
        cursor[wg * BLOCK + threadIdx.x] = 0;
    }
}

// CopyOutDegree for the larger batches (the radix-sort path): post-batch out-degree of every record's tail from the epoch's out-CSR
__global__ __launch_bounds__(BLOCK) void k_copy_out_degree(const int *__restrict__ e1, int L, const int *__restrict__ out_row_ptr,
                                                          int *__restrict__ deg_after) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < L; i += gridDim.x * BLOCK) {
        const int t = e1[i];
        deg_after[i] = out_row_ptr[t + 1] - out_row_ptr[t];
    }
}

// blockIdx.y = source lane of a group (0 for a single source); state element (v, lane) sits at
// base[v * stride + lane]
__global__ __launch_bounds__(BLOCK) void k_su_terms(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ svals,
                                                    const int *__restrict__ e2, const uint8_t *__restrict__ ins, int L,
                                                    const double *__restrict__ p_base, int stride,
                                                    double *__restrict__ term_base, uint8_t *__restrict__ sins) {
    const double *p = p_base + blockIdx.y;
    double *term = term_base + (size_t)blockIdx.y * (size_t)L;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < L; j += gridDim.x * BLOCK) {
        const int u = (int)skeys[j];
        const int rec = (int)svals[j];
        const int v = e2[rec];
        term[j] = ONE_MINUS_ALPHA * p[(size_t)v * stride] - p[(size_t)u * stride];
        if (blockIdx.y == 0) sins[j] = ins[rec];
    }
}

struct SuSources {
    int s[16]; // source vertex per lane (blockIdx.y)
};
// A tail's records are applied one after the other (the order is the point). Short groups: by their leader's lane. A group of
// more than SU_LONG records (a hub's tail in a batch of millions: thousands) would leave that one lane walking global memory at
// one load latency per record: its leader only notes it, and the WAVE then walks it together -- 64 terms per coalesced load,
// the recurrence evaluated uniformly by all lanes (same operations in the same order: bit-identical).
constexpr int SU_LONG = 96;
__global__ __launch_bounds__(BLOCK) void k_su_apply(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ svals,
                                                    const double *__restrict__ term_base, const uint8_t *__restrict__ sins,
                                                    const int *__restrict__ deg_after, int L, double *__restrict__ r_base,
                                                    int stride, SuSources srcs, double eps, int *__restrict__ ft_pos,
                                                    int *__restrict__ cnt_pos, int *__restrict__ ft_neg,
                                                    int *__restrict__ cnt_neg) {
    __shared__ int s_long[WAVES_PER_BLOCK][WAVE][2]; // (first record, end) of the long groups a wave's lanes lead
    double *r = r_base + blockIdx.y;
    const double *term = term_base + (size_t)blockIdx.y * (size_t)L;
    const int source = srcs.s[blockIdx.y];
    const bool seed = ft_pos != nullptr; // groups seed densely instead (k_gseed)
    const int nthreads = gridDim.x * BLOCK;
    const int lane = lane_id(), w = wave_id();
    for (int j0 = blockIdx.x * BLOCK; j0 < L; j0 += nthreads) {
        const int j = j0 + threadIdx.x;
        bool pos = false, neg = false, is_long = false;
        int u = 0, end = 0;
        if (j < L) {
            u = (int)skeys[j];
            const bool leader = (j == 0) || ((int)skeys[j - 1] != u);
            if (leader) {
                int lo = j, hi = L; // end of the group: first record behind j whose tail is not u (the keys are sorted)
                while (hi - lo > 1) {
                    const int mid = lo + ((hi - lo) >> 1);
                    if ((int)skeys[mid] == u) lo = mid; else hi = mid;
                }
                end = hi;
                is_long = end - j > SU_LONG;
                if (!is_long) {
                    int delta = 0; // post-batch degree minus pre-batch degree
                    for (int k = j; k < end; ++k) delta += sins[k] ? 1 : -1;
                    int d = deg_after[svals[j]] - delta; // RevertOutDegree (gpu/StreamUpdate.cuh:18-33)
                    double ru = r[(size_t)u * stride];
                    const double src_term = ALPHA * (source == u ? 1.0 : 0.0);
                    for (int k = j; k < end; ++k) {
                        const double add = term[k] - ALPHA * ru + src_term;
                        if (sins[k]) {
                            d++;
                            ru += add / (double)(d + 1) / ALPHA;
                        } else {
                            d--;
                            ru -= add / (double)(d + 1) / ALPHA;
                        }
                    }
                    r[(size_t)u * stride] = ru;
                    pos = seed && ru > eps;
                    neg = seed && ru < -eps;
                }
            }
        }
        // ---- the wave's long groups, one after the other, all lanes together
        const uint64_t ml = __ballot(is_long);
        if (ml) { // wave-uniform
            if (is_long) {
                s_long[w][mbcnt(ml)][0] = j;
                s_long[w][mbcnt(ml)][1] = end;
            }
            __builtin_amdgcn_wave_barrier();
            const int n_long = __popcll(ml);
            for (int g = 0; g < n_long; ++g) {
                const int gj = s_long[w][g][0], gend = s_long[w][g][1];
                const int gu = (int)skeys[gj];
                int part = 0;
                for (int k = gj + lane; k < gend; k += WAVE) part += sins[k] ? 1 : -1;
                const int delta = __builtin_amdgcn_readlane(wave_inclusive_scan(part), WAVE - 1);
                int d = deg_after[svals[gj]] - delta;
                double ru = r[(size_t)gu * stride];
                const double src_term = ALPHA * (source == gu ? 1.0 : 0.0);
                for (int base = gj; base < gend; base += WAVE) {
                    const int k = base + lane;
                    const double tk = k < gend ? term[k] : 0.0;
                    const int ik = k < gend ? (int)sins[k] : 0;
                    const int n = min(WAVE, gend - base);
                    for (int i = 0; i < n; ++i) { // wave-uniform: every lane evaluates the same recurrence
                        const double t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tk), i), __builtin_amdgcn_readlane(__double2loint(tk), i));
                        const int is_ins = __builtin_amdgcn_readlane(ik, i);
                        const double add = t - ALPHA * ru + src_term;
                        if (is_ins) {
                            d++;
                            ru += add / (double)(d + 1) / ALPHA;
                        } else {
                            d--;
                            ru -= add / (double)(d + 1) / ALPHA;
                        }
                    }
                }
                if (lane == 0) {
                    r[(size_t)gu * stride] = ru;
                    if (seed && ru > eps) ft_pos[atomicAdd(cnt_pos, 1)] = gu;
                    if (seed && ru < -eps) ft_neg[atomicAdd(cnt_neg, 1)] = gu;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // wave-aggregated appends
        uint64_t m = __ballot(pos);
        if (m) {
            int gb = 0;
            if (lane_id() == 0) gb = atomicAdd(cnt_pos, __popcll(m));
            gb = __shfl(gb, 0, WAVE);
            if (pos) ft_pos[gb + mbcnt(m)] = u;
        }
        m = __ballot(neg);
        if (m) {
            int gb = 0;
            if (lane_id() == 0) gb = atomicAdd(cnt_neg, __popcll(m));
            gb = __shfl(gb, 0, WAVE);
            if (neg) ft_neg[gb + mbcnt(m)] = u;
        }
    }
}

// Single-source form of k_su_terms + k_su_apply in one launch. The group leaders walk their records
// one after the other (the order is the point), so what they walk must not be a chain of global
// loads: every workgroup stages a window of SU_WIN sorted records -- tail, insert flag and the
// term t, computed here in parallel -- in LDS, its own BLOCK records plus a halo for groups that
// run past them; a group that outgrows even the window finishes from global memory.
constexpr int SU_WIN = 1024;
__global__ __launch_bounds__(BLOCK) void k_su_apply_fused(const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ svals,
                                                          const int *__restrict__ e2, const uint8_t *__restrict__ ins,
                                                          const int *__restrict__ deg_after, int L,
                                                          const double *__restrict__ p, double *__restrict__ r, int source,
                                                          double eps, int *__restrict__ ft_pos, int *__restrict__ cnt_pos,
                                                          int *__restrict__ ft_neg, int *__restrict__ cnt_neg) {
    __shared__ uint32_t s_key[SU_WIN];
    __shared__ double s_term[SU_WIN];
    __shared__ uint8_t s_ins[SU_WIN];
    auto term_of = [&](int u, int rec) { return ONE_MINUS_ALPHA * p[e2[rec]] - p[u]; };
    for (int j0 = blockIdx.x * BLOCK; j0 < L; j0 += gridDim.x * BLOCK) {
        __syncthreads(); // the window of the previous pass is no longer read
        const int wn = min(SU_WIN, L - j0);
        for (int i = threadIdx.x; i < wn; i += BLOCK) {
            const int u = (int)skeys[j0 + i], rec = (int)svals[j0 + i];
            s_key[i] = (uint32_t)u;
            s_ins[i] = ins[rec];
            s_term[i] = term_of(u, rec);
        }
        __syncthreads();
        const int j = j0 + threadIdx.x;
        bool pos = false, neg = false;
        int u = 0;
        if (j < L) {
            u = (int)s_key[threadIdx.x];
            const bool leader = (j == 0) || ((int)skeys[j - 1] != u);
            if (leader) {
                // extent of the group and its net degree change (post-batch minus pre-batch)
                int end = threadIdx.x, delta = 0;
                while (end < wn && (int)s_key[end] == u) {
                    delta += s_ins[end] ? 1 : -1;
                    ++end;
                }
                int gend = j0 + end; // the part beyond the window, if any
                if (end == wn)
                    while (gend < L && (int)skeys[gend] == u) {
                        delta += ins[svals[gend]] ? 1 : -1;
                        ++gend;
                    }
                int d = deg_after[svals[j]] - delta; // RevertOutDegree (gpu/StreamUpdate.cuh:18-33)
                double ru = r[u];
                const double src_term = ALPHA * (source == u ? 1.0 : 0.0);
                auto apply = [&](double t, bool is_ins) {
                    const double add = t - ALPHA * ru + src_term;
                    if (is_ins) {
                        d++;
                        ru += add / (double)(d + 1) / ALPHA;
                    } else {
                        d--;
                        ru -= add / (double)(d + 1) / ALPHA;
                    }
                };
                for (int k = threadIdx.x; k < end; ++k) apply(s_term[k], s_ins[k] != 0);
                for (int k = j0 + end; k < gend; ++k) {
                    const int rec = (int)svals[k];
                    apply(term_of(u, rec), ins[rec] != 0);
                }
                r[u] = ru;
                pos = ru > eps;
                neg = ru < -eps;
            }
        }
        uint64_t m = __ballot(pos);
        if (m) {
            int gb = 0;
            if (lane_id() == 0) gb = atomicAdd(cnt_pos, __popcll(m));
            gb = __shfl(gb, 0, WAVE);
            if (pos) ft_pos[gb + mbcnt(m)] = u;
        }
        m = __ballot(neg);
        if (m) {
            int gb = 0;
            if (lane_id() == 0) gb = atomicAdd(cnt_neg, __popcll(m));
            gb = __shfl(gb, 0, WAVE);
            if (neg) ft_neg[gb + mbcnt(m)] = u;
        }
    }
}

// phase-1 seed: keep the candidates that are still below -eps after phase 0
// (phase 0 only adds positive amounts, so no new vertex can have dropped below).
__global__ __launch_bounds__(BLOCK) void k_filter(const int *__restrict__ cand, const int *__restrict__ cnt_cand,
                                                  const double *__restrict__ r, int phase, double eps,
                                                  int *__restrict__ ft, int *__restrict__ cnt) {
    const int n = *cnt_cand;
    for (int i0 = blockIdx.x * BLOCK; i0 < n; i0 += gridDim.x * BLOCK) {
        const int i = i0 + threadIdx.x;
        int u = 0;
        bool hit = false;
        if (i < n) {
            u = cand[i];
            hit = legal(r[u], phase, eps);
        }
        const uint64_t m = __ballot(hit);
        if (m) {
            int gb = 0;
            if (lane_id() == 0) gb = atomicAdd(cnt, __popcll(m));
            gb = __shfl(gb, 0, WAVE);
            if (hit) ft[gb + mbcnt(m)] = u;
        }
    }
}

} // namespace dppr
