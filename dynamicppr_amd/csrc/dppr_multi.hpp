// dppr_multi.hpp -- f2: multi-source batched sweeps.
//
// Several source vertices that share one device graph (BASELINE.json configs 3 and 5 run 10 of
// them) are solved TOGETHER: their state is interleaved 8-wide, p/r/x[v] = 8 doubles = one
// 64-byte sector. One sweep then reads every out_col entry once, computes every edge's owner
// once and gathers ONE sector per edge for all 8 sources -- a single-source sweep pulls the same
// sector for 8 useful bytes. The per-source arithmetic is exactly that of k_pull_iter
// (rv += (1.0-ALPHA)*x[u]/(outdeg(v)+1) in CSR order, rv -= x[v], threshold, next snapshot).
// Group iterations are always dense (the sweep's cost is shared by 8 sources, so sparse push
// iterations are not worth a second code path): seeding is a dense pass too.
#pragma once

#include "dppr_kernels.hpp"

namespace dppr {

constexpr int GS = 8; // sources per group = doubles per 64-byte sector

struct alignas(64) D8 {
    double v[GS];
};
struct Src8 {
    int s[GS]; // internal source vertex per lane of the group, -1 = unused lane
};

// r = e_s per source, p = 0
__global__ __launch_bounds__(BLOCK) void k_ginit(D8 *__restrict__ p, D8 *__restrict__ r, int V, Src8 src) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        D8 z, e;
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            z.v[s] = 0.0;
            e.v[s] = (src.s[s] == v) ? 1.0 : 0.0;
        }
        p[v] = z;
        r[v] = e;
    }
}

// Dense seeding of a phase: Inspect (gpu/Inspect.cuh:8-48) + the snapshot head of ExpandUnifiedRev
// (gpu/ExpandRev.cuh:34-42) for every source at once: x[v][s] = legal(r) ? r : 0, p += ALPHA*r.
__global__ __launch_bounds__(BLOCK) void k_gseed(int V, const D8 *__restrict__ r, D8 *__restrict__ x, D8 *__restrict__ p,
                                                 int phase, double eps, int *__restrict__ cnt_out) {
    __shared__ int s_cnt[WAVES_PER_BLOCK][GS];
    int n_legal[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) n_legal[s] = 0;
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        const D8 rv = r[v];
        D8 xn;
        bool any = false;
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            const bool lg = legal(rv.v[s], phase, eps);
            xn.v[s] = lg ? rv.v[s] : 0.0;
            n_legal[s] += lg ? 1 : 0;
            any |= lg;
        }
        x[v] = xn;
        if (any) {
            D8 pv = p[v];
#pragma unroll
            for (int s = 0; s < GS; ++s) pv.v[s] += ALPHA * xn.v[s];
            p[v] = pv;
        }
    }
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        const int t = wave_inclusive_scan(n_legal[s]);
        if (lane_id() == WAVE - 1) s_cnt[wave_id()][s] = t;
    }
    __syncthreads();
    if (threadIdx.x < GS) {
        int tot = 0;
        for (int k = 0; k < WAVES_PER_BLOCK; ++k) tot += s_cnt[k][threadIdx.x];
        if (tot) atomicAdd(&cnt_out[threadIdx.x], tot);
    }
}

constexpr int GPB = 512; // workgroup = 512 consecutive vertices (LDS: 32 KiB of per-vertex accumulators)
constexpr int GPU_SLOTS = 2; // sector gathers in flight per lane

__global__ __launch_bounds__(GPB) void k_pull_multi(int V, const int *__restrict__ cnt_in,
                                                    const int *__restrict__ out_row_ptr,
                                                    const int *__restrict__ out_col, const D8 *__restrict__ x,
                                                    D8 *__restrict__ x_new, D8 *__restrict__ r, D8 *__restrict__ p,
                                                    int *__restrict__ cnt_out, int *__restrict__ cnt_zero, int phase,
                                                    double eps, IterStats *__restrict__ stats,
                                                    int *__restrict__ log_slot, int pull_big_row) {
    constexpr int NW = GPB / WAVE;
    constexpr int PUM = GPU_SLOTS;
    __shared__ int s_own[NW][WAVE * PUM];
    __shared__ int s_scan[NW][WAVE + 1];
    __shared__ int s_start[NW][WAVE];
    __shared__ double s_acc[NW][GS * WAVE]; // [source][lane]: conflict-free for lane-contiguous access
    __shared__ int s_cnt[NW][GS];
    __shared__ unsigned long long s_edges[NW];
    __shared__ int s_bigv[PULL_BIG_CAP], s_bigrs[PULL_BIG_CAP], s_bigd[PULL_BIG_CAP];
    __shared__ double s_bigacc[PULL_BIG_CAP][GS];
    __shared__ int s_chunk0[NW][WAVE + 1];
    __shared__ int s_nbig;
    const int lane = lane_id(), w = wave_id();
    // frontier sizes of the 8 sources; the group iterates while ANY of them is non-empty
    const int my_cnt = lane < GS ? cnt_in[lane] : 0;
    if (blockIdx.x == 0 && threadIdx.x < GS) {
        cnt_zero[threadIdx.x] = 0;
        log_slot[threadIdx.x] = my_cnt;
    }
    if (__ballot(my_cnt != 0) == 0) return;
    int n_legal[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) n_legal[s] = 0;
    unsigned long long edges = 0;

    // repair, threshold, next snapshot for one vertex, all sources (acc = residual + gathered adds)
    auto finish = [&](bool valid, int v, const D8 &acc) {
        if (!valid) return;
        const D8 xv = x[v];
        D8 rn, xn;
        bool any = false;
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            double t = acc.v[s];
            if (xv.v[s] != 0.0) t -= xv.v[s];
            const bool lg = legal(t, phase, eps);
            rn.v[s] = t;
            xn.v[s] = lg ? t : 0.0;
            n_legal[s] += lg ? 1 : 0;
            any |= lg;
        }
        r[v] = rn;
        x_new[v] = xn;
        if (any) {
            D8 pv = p[v];
#pragma unroll
            for (int s = 0; s < GS; ++s) pv.v[s] += ALPHA * xn.v[s];
            p[v] = pv;
        }
    };

    const int n_groups = (V + GPB - 1) / GPB;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) { // workgroup-uniform loop
        if (threadIdx.x == 0) s_nbig = 0;
        __syncthreads();
        const int v = (g * NW + w) * WAVE + lane;
        const bool valid = v < V;
        int rs = 0, d = 0;
        D8 rv;
#pragma unroll
        for (int s = 0; s < GS; ++s) rv.v[s] = 0.0;
        if (valid) {
            rs = out_row_ptr[v];
            d = out_row_ptr[v + 1] - rs;
            rv = r[v];
        }
        bool deferred = false;
        if (d >= pull_big_row) {
            const int slot = atomicAdd(&s_nbig, 1);
            if (slot < PULL_BIG_CAP) {
                s_bigv[slot] = v;
                s_bigrs[slot] = rs;
                s_bigd[slot] = d;
#pragma unroll
                for (int s = 0; s < GS; ++s) s_bigacc[slot][s] = 0.0;
                deferred = true;
            }
        }
        const int dd = deferred ? 0 : d;
        const int incl = wave_inclusive_scan(dd);
        const int scan_ex = incl - dd;
        const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
        s_scan[w][lane] = scan_ex;
        s_start[w][lane] = rs;
        if (lane == 0) s_scan[w][WAVE] = total;
#pragma unroll
        for (int s = 0; s < GS; ++s) s_acc[w][s * WAVE + lane] = rv.v[s];

        // ---- short rows: same owner scheme as k_pull_iter (marks + max-scan), one sector per edge
        for (int e0 = 0; e0 < total; e0 += WAVE * PUM) {
#pragma unroll
            for (int k = 0; k < PUM; ++k) s_own[w][k * WAVE + lane] = -1;
            __builtin_amdgcn_wave_barrier();
            const int pos = scan_ex - e0;
            if (dd > 0 && pos >= 0 && pos < WAVE * PUM) s_own[w][pos] = lane;
            __builtin_amdgcn_wave_barrier();
            const uint64_t before = __ballot(dd > 0 && scan_ex <= e0);
            int carry = before ? 63 - __clzll(before) : -1;
            int own[PUM], col[PUM];
#pragma unroll
            for (int k = 0; k < PUM; ++k) {
                const int e = e0 + k * WAVE + lane;
                int o = wave_inclusive_max(s_own[w][k * WAVE + lane]);
                o = max(o, carry);
                carry = __builtin_amdgcn_readlane(o, WAVE - 1);
                own[k] = e < total ? o : -1;
                col[k] = 0;
                if (own[k] >= 0) col[k] = out_col[s_start[w][o] + (e - s_scan[w][o])];
            }
            D8 xa[PUM];
#pragma unroll
            for (int k = 0; k < PUM; ++k) {
                if (own[k] >= 0) xa[k] = x[col[k]];
                else {
#pragma unroll
                    for (int s = 0; s < GS; ++s) xa[k].v[s] = 0.0;
                }
            }
#pragma unroll
            for (int k = 0; k < PUM; ++k) {
                const int o = own[k] >= 0 ? own[k] : 0;
                const double denom = (double)(s_scan[w][o + 1] - s_scan[w][o] + 1);
#pragma unroll
                for (int s = 0; s < GS; ++s) {
                    const bool nz = xa[k].v[s] != 0.0;
                    if (nz) lds_add(&s_acc[w][s * WAVE + o], ONE_MINUS_ALPHA * xa[k].v[s] / denom);
                    edges += (unsigned long long)__popcll(__ballot(nz));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
        {
            D8 acc;
#pragma unroll
            for (int s = 0; s < GS; ++s) acc.v[s] = s_acc[w][s * WAVE + lane];
            finish(valid && !deferred, v, acc);
        }

        // ---- long rows: 512-edge chunks dealt to the waves, 8 partial sums per lane
        __syncthreads();
        const int nbig = min(s_nbig, PULL_BIG_CAP);
        if (nbig) { // workgroup-uniform
            constexpr int CHUNK = WAVE * 8;
            const int nch = lane < nbig ? (s_bigd[lane] + CHUNK - 1) / CHUNK : 0;
            const int inc = wave_inclusive_scan(nch);
            s_chunk0[w][lane] = inc - nch;
            const int n_chunks = __builtin_amdgcn_readlane(inc, WAVE - 1);
            if (lane == 0) s_chunk0[w][WAVE] = n_chunks;
            __builtin_amdgcn_wave_barrier();
            for (int ch = w; ch < n_chunks; ch += NW) {
                int lo = 0, hi = WAVE;
#pragma unroll
                for (int s2 = 0; s2 < 6; ++s2) {
                    const int mid = (lo + hi) >> 1;
                    if (s_chunk0[w][mid] <= ch) lo = mid; else hi = mid;
                }
                const int row_rs = s_bigrs[lo], row_d = s_bigd[lo];
                const int c0 = (ch - s_chunk0[w][lo]) * CHUNK;
                const int c1 = min(c0 + CHUNK, row_d);
                const double denom = (double)(row_d + 1);
                double part[GS];
#pragma unroll
                for (int s = 0; s < GS; ++s) part[s] = 0.0;
                for (int e0 = c0; e0 < c1; e0 += WAVE) { // wave-uniform trip count (ballots inside)
                    const int e = e0 + lane;
                    D8 xb;
                    if (e < c1) xb = x[out_col[row_rs + e]];
                    else {
#pragma unroll
                        for (int s = 0; s < GS; ++s) xb.v[s] = 0.0;
                    }
#pragma unroll
                    for (int s = 0; s < GS; ++s) {
                        const bool nz = xb.v[s] != 0.0;
                        if (nz) part[s] += ONE_MINUS_ALPHA * xb.v[s] / denom;
                        edges += (unsigned long long)__popcll(__ballot(nz));
                    }
                }
#pragma unroll
                for (int s = 0; s < GS; ++s) {
                    const double t = wave_sum(part[s]);
                    if (lane == 0 && t != 0.0) lds_add(&s_bigacc[lo][s], t);
                }
            }
            __syncthreads();
            if (w == 0 && lane < nbig) {
                const int bv = s_bigv[lane];
                const D8 rb = r[bv];
                D8 acc;
#pragma unroll
                for (int s = 0; s < GS; ++s) acc.v[s] = rb.v[s] + s_bigacc[lane][s];
                finish(true, bv, acc);
            }
        }
        __syncthreads();
    }
    // next frontier sizes: one fire-and-forget atomic per source and workgroup
#pragma unroll
    for (int s = 0; s < GS; ++s) {
        const int t = wave_inclusive_scan(n_legal[s]);
        if (lane == WAVE - 1) s_cnt[w][s] = t;
    }
    __syncthreads();
    if (threadIdx.x < GS) {
        int tot = 0;
        for (int k = 0; k < NW; ++k) tot += s_cnt[k][threadIdx.x];
        if (tot) atomicAdd(&cnt_out[threadIdx.x], tot);
    }
    stat_add_edges<NW>(stats, edges, s_edges);
}

// strided copies between the interleaved group state and an external-id vector of one source
__global__ __launch_bounds__(BLOCK) void k_gint_to_ext(const D8 *__restrict__ a_int, int lane_s,
                                                       const int *__restrict__ ext2int, int V,
                                                       double *__restrict__ a_ext) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        const int m = ext2int[v];
        a_ext[v] = m >= 0 ? a_int[m].v[lane_s] : 0.0;
    }
}

} // namespace dppr
