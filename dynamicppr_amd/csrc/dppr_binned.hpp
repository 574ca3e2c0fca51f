// dppr_binned.hpp -- the dense frontier iteration of ONE source on a window whose snapshot vector is far larger
// than the L2s: a BINNED (propagation-blocked) sweep.
//
// k_pull_iter (dppr_pull.hpp) gathers x[u] for every out-edge (v, u): an 8-byte value out of a 64-byte sector, and once
// x no longer fits the caches every gather is a random sector from Infinity Cache / HBM (twitter stand-in: 4.4 GB
// fetched per sweep for 1.2 GB of useful values). The arithmetic does not care in which order the
// pushes u -> v of gpu/ExpandRev.cuh:70-73 reach residual[v], so the same sums are formed in two STREAMING passes over
// a per-epoch layout of the window's edges:
//
//   * the heads u are cut into A-BLOCKS (consecutive vertices, at most 64 ha_tiles of them), the rows v into B-BLOCKS
//     (at most 64 hb_tiles rows, about `target` out-edges, a hub row alone). The edges of A-block a, ordered by
//     (B-block of v, v, u), are its A-major run [in_row_ptr[first head of a], in_row_ptr[first head of a + 1]); the edges
//     of B-block b, ordered by (A-block of u, v, u), are its B-major run [out_row_ptr[first row of b], ...). Both runs
//     hold the same edges tile by tile (a tile = the edges from A-block a into B-block b) in the same inner order.
//   * k_bin_scatter (workgroups take chunks of an A-block's run): the block's slice of x is staged in LDS; the run is streamed
//     (2-byte head index + 4-byte B-major position per edge) and x[u] is written to vals[position]: inside a tile the
//     positions are consecutive, so a wave's 64 stores are runs of whole lines, not 64 sectors.
//   * k_bin_reduce (one workgroup per B-block): row accumulators (starting at residual[v]), 1 / (outdeg + 1) and
//     outdeg + 1 of the block's rows sit in LDS; the B-major run is streamed (8-byte value + 2-byte row index per
//     edge), add = (1.0-ALPHA) * x[u] / (outdeg(v) + 1) exactly as gpu/ExpandRev.cuh:72 (push_term), runs of one row
//     inside a wave are summed in registers first (segmented DPP scan) and cost ONE LDS atomic; then repair
//     (gpu/ExpandRev.cuh:708-743), threshold and the next snapshot per row, as k_pull_iter does them.
//
// 24 streamed bytes per edge in lines instead of a 64-byte sector per edge; no global atomics. Results equal
// k_pull_iter's up to the order of each row's sum.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

constexpr int BIN_NT = 1024;   // threads of both passes
#ifndef DPPR_BIN_U
#define DPPR_BIN_U 4           // (8, and a software-pipelined loop, were measured: 3-5 % slower on the twitter stand-in)
#endif
constexpr int BIN_U = DPPR_BIN_U; // entries in flight per lane

// ---- block cuts (graph build, untimed). A cut is a list of first VERTICES, cut[0] = 0 < ... < cut[n_blocks] = NV; it holds
// every multiple of the vertex cap, the first vertex behind every `target` edges of the CSR, and both sides of every row
// of at least target / 4 edges: blocks of at most cap vertices and about target edges, a hub alone in its block.
// quant[k - 1] = first vertex with row_ptr[v] >= k * target, k = 1 .. K - 1
__global__ __launch_bounds__(BLOCK) void k_bin_quantiles(const int *__restrict__ row_ptr, int NV, long long target, int K,
                                                         int *__restrict__ quant) {
    for (int k = 1 + blockIdx.x * BLOCK + threadIdx.x; k < K; k += gridDim.x * BLOCK) {
        const long long want = (long long)k * target;
        int lo = 0, hi = NV; // first v in [0, NV] with row_ptr[v] >= want
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if ((long long)row_ptr[mid] < want) lo = mid + 1; else hi = mid;
        }
        quant[k - 1] = lo;
    }
}
// rows of at least min_deg edges (first `cap` takers)
__global__ __launch_bounds__(BLOCK) void k_bin_big_rows(const int *__restrict__ row_ptr, int NV, int min_deg, int cap,
                                                        int *__restrict__ list, int *__restrict__ count) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < NV; v += gridDim.x * BLOCK)
        if (row_ptr[v + 1] - row_ptr[v] >= min_deg) {
            const int slot = atomicAdd(count, 1);
            if (slot < cap) list[slot] = v;
        }
}
// vertex -> block of a cut; also start[k] = row_ptr[cut[k]] (the block's first edge), k = 0 .. n_blocks
__global__ __launch_bounds__(BLOCK) void k_bin_vertex_block(const int *__restrict__ cut, int n_blocks, int NV,
                                                            const int *__restrict__ row_ptr, int *__restrict__ vblk,
                                                            int *__restrict__ start) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < NV && vblk != nullptr; v += gridDim.x * BLOCK) { // (no table wanted: only the blocks' first entries)
        int lo = 0, hi = n_blocks; // last block whose first vertex is <= v
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cut[mid] <= v) lo = mid; else hi = mid;
        }
        vblk[v] = lo;
    }
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k <= n_blocks; k += gridDim.x * BLOCK) start[k] = row_ptr[cut[k]];
}

// The two orders of the tables come out of two KEY-ONLY radix sorts of 64-bit words that carry everything the tables need (a
// pair sort moves a 32-bit key and a 64-bit value: 12 bytes per entry and pass instead of 8, and the fill kernels looked the
// blocks' first vertices up again):
//   word 1 = B-block | A-block | row inside its B-block (BIN_RL bits) | head inside its A-block (BIN_HL bits)
//            sorted by the two block fields, stable: the (row, head) order of the input survives inside a (B, A) cell -> B-major
//   word 2 = A-block | B-major position (31 bits) | head inside its A-block
//            sorted by the A-block field, stable -> A-major
constexpr int BIN_RL = 13, BIN_HL = 15; // rows of a B-block <= 64 x BIN_MAX_HB_TILES = 7 680, heads of an A-block <= 64 x BIN_MAX_HA_TILES = 17 408

// The keys come in (row, head) order: consecutive entries share their row or its neighbours (vblk_b[v] is a cached, coalesced
// read), the heads are random -- their A-block is found by bisection of the block cuts (a few thousand entries, staged in LDS when
// they fit `cuts_in_lds` ints) instead of a random 4-byte gather per edge out of a V-sized table (361 M of them on the friendster
// stand-in: 4.0 ms of a 40-ms slide).
// amajor = 0: word 1 (B-block | A-block | row | head); amajor = 1: the same edge's A-MAJOR word (A-block | B-block | row | head) -- the form the
// persistent A-major array of the incremental tables is kept in (below): plain 64-bit order = (A, B, row, head).
__global__ __launch_bounds__(BLOCK) void k_bin_keys(const uint64_t *__restrict__ out_keys, int Ed, int bits,
                                                    const int *__restrict__ acut, int n_a, int cuts_in_lds,
                                                    const int *__restrict__ vblk_b, const int *__restrict__ bcut, int abits,
                                                    uint64_t *__restrict__ w1, int bbits = 0, int amajor = 0) {
    extern __shared__ int s_acut[];
    const int *cut = acut;
    if (cuts_in_lds) {
        for (int k = threadIdx.x; k <= n_a; k += BLOCK) s_acut[k] = acut[k];
        __syncthreads();
        cut = s_acut;
    }
    const uint64_t mask = (1ull << bits) - 1;
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < Ed; o += gridDim.x * BLOCK) {
        const uint64_t k = out_keys[o];
        const int v = (int)(k >> bits), u = (int)(k & mask);
        int lo = 0, hi = n_a; // last block whose first vertex is <= u (k_bin_vertex_block's rule)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cut[mid] <= u) lo = mid; else hi = mid;
        }
        const int b = vblk_b[v];
        const uint64_t blocks = amajor ? (((uint64_t)(uint32_t)lo << bbits) | (uint64_t)(uint32_t)b) : (((uint64_t)(uint32_t)b << abits) | (uint64_t)(uint32_t)lo);
        w1[o] = (blocks << (BIN_RL + BIN_HL)) | ((uint64_t)(uint32_t)(v - bcut[b]) << BIN_HL) | (uint64_t)(uint32_t)(u - cut[lo]);
    }
}

// B-major order without a radix sort. The words arrive in (row, head) order, i.e. B-block-major already: what the first sort does is
// group every B-block's segment by A-block, stably -- four radix passes over the whole window for a key of which the upper half is
// sorted. Here one workgroup owns a B-block: a histogram of its segment over the A-blocks (LDS, all waves), an exclusive scan, and
// then ONE wave walks the segment in order and places every word at (start of its A-block inside the segment + number of earlier
// words of that A-block): the rank inside a 64-word step from ballots (one per bit of the A-block number: the lanes that agree in
// every bit hold the same block; no memory access), the running starts in LDS (gathered once per step, advanced by the last lane of each A-block). Deterministic and equal to
// the stable sort. A block that is ONE row (a hub alone in its block) is sorted already -- heads ascend -- and is copied by all waves.
// Needs n_a <= BIN_CS_MAX_A counters; beyond, the radix sort stays.
constexpr int BIN_CS_NT = 256, BIN_CS_MAX_A = 8192, BIN_CS_AHEAD = 8;
__global__ __launch_bounds__(BIN_CS_NT) void k_bin_bmajor(const uint64_t *__restrict__ w1, const int *__restrict__ bstart,
                                                          const int *__restrict__ bcut, int n_a, int n_pad, int abits,
                                                          uint64_t *__restrict__ out) {
    extern __shared__ int s_cnt[]; // n_pad (a power of two >= max(n_a, 64))
    const int b = blockIdx.x, tid = threadIdx.x, lane = lane_id();
    const int s0 = bstart[b], s1 = bstart[b + 1];
    if (s1 <= s0) return;
    if (bcut[b + 1] - bcut[b] == 1) { // one row: in order already
        int i = s0 + tid;
        for (; i + 7 * BIN_CS_NT < s1; i += 8 * BIN_CS_NT) { // (a hub's row can hold a million words: eight loads in flight per thread)
            uint64_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = w1[i + k * BIN_CS_NT];
#pragma unroll
            for (int k = 0; k < 8; ++k) out[i + k * BIN_CS_NT] = w[k];
        }
        for (; i < s1; i += BIN_CS_NT) out[i] = w1[i];
        return;
    }
    const uint32_t amask = (1u << abits) - 1u;
    for (int k = tid; k < n_pad; k += BIN_CS_NT) s_cnt[k] = 0;
    __syncthreads();
    {
        int i = s0 + tid;
        for (; i + 3 * BIN_CS_NT < s1; i += 4 * BIN_CS_NT) { // (four loads in flight per thread)
            uint64_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = w1[i + k * BIN_CS_NT];
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(&s_cnt[(uint32_t)(w[k] >> (BIN_RL + BIN_HL)) & amask], 1);
        }
        for (; i < s1; i += BIN_CS_NT) atomicAdd(&s_cnt[(uint32_t)(w1[i] >> (BIN_RL + BIN_HL)) & amask], 1);
    }
    __syncthreads();
    if (tid >= WAVE) return; // the walk is one wave's (nothing below needs a workgroup barrier)
    { // exclusive scan over the A-blocks: a lane sums its n_pad / 64 consecutive counters, the wave scans the lane sums
        const int per = n_pad / WAVE;
        int sum = 0;
        for (int k = 0; k < per; ++k) sum += s_cnt[lane * per + k];
        int run = s0 + wave_inclusive_scan(sum) - sum;
        for (int k = 0; k < per; ++k) {
            const int c = s_cnt[lane * per + k];
            s_cnt[lane * per + k] = run;
            run += c;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the starts are in LDS before the first step gathers them
    __builtin_amdgcn_wave_barrier();
    const uint64_t gt = lane == WAVE - 1 ? 0ull : (~0ull << (lane + 1)); // the lanes above this one
    uint64_t nxt[BIN_CS_AHEAD]; // the words of the NEXT run of steps are requested before this run's are placed
#pragma unroll
    for (int h = 0; h < BIN_CS_AHEAD; ++h) {
        const int i = s0 + h * WAVE + lane;
        nxt[h] = i < s1 ? w1[i] : 0ull;
    }
    for (int base = s0; base < s1; base += WAVE * BIN_CS_AHEAD) {
        uint64_t w[BIN_CS_AHEAD];
#pragma unroll
        for (int h = 0; h < BIN_CS_AHEAD; ++h) {
            w[h] = nxt[h];
            const int i = base + WAVE * BIN_CS_AHEAD + h * WAVE + lane;
            nxt[h] = i < s1 ? w1[i] : 0ull;
        }
#pragma unroll
        for (int h = 0; h < BIN_CS_AHEAD; ++h) {
            const int i = base + h * WAVE + lane;
            const bool valid = i < s1;
            const int a = (int)((uint32_t)(w[h] >> (BIN_RL + BIN_HL)) & amask);
            uint64_t m = __ballot(valid); // -> the valid lanes of this step that hold the same A-block as this lane: one ballot per key bit
            if (!m) break;
            for (int bit = 0; bit < abits; ++bit) {
                const bool set = (a >> bit) & 1;
                const uint64_t bm = __ballot(valid && set);
                m &= set ? bm : ~bm;
            }
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const bool last = (m & gt) == 0ull;
            int pos = 0;
            if (valid) pos = s_cnt[a] + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid && last) s_cnt[a] = pos + 1;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (valid) out[pos] = w[h];
        }
    }
}

// ---- Tables PATCHED per slide instead of re-sorted per epoch (round 5). Two radix sorts of the whole window were
// 6 of a twitter-size slide's 14 ms of device work and 13 of a friendster-size one's 27 (k_bin_bmajor included) -- and since the graph
// build runs beside the solve (dppr_slide_concurrent) every millisecond of it is a millisecond of wall time. The engine keeps both
// orders PERSISTENT as sorted 64-bit words (B-major: word 1; A-major: A | B | row | head) under block cuts that stay FROZEN between
// re-cuts (new ids extend the last blocks / append blocks; a re-cut every few dozen slides restores the balance). A slide then
//   * forms the words of its 2c retired and 2c inserted edges (k_bin_keys, both forms), sorts those, and merges them into the two
//     arrays with the key merge of the CSR build (k_del_positions + k_merge_tiles, dppr_builder.hpp): 16 bytes per edge and array;
//   * finds every tile's first entry in both orders (k_bin_tile_starts: where the block pair changes) -- a tile is a run in both, in
//     the same inner (row, head) order, so an edge's B-major position is  first_B(tile) + (its A-major index - first_A(tile));
//   * writes dl from the B-major words and hl, apos from the A-major ones (k_bin_rows / k_bin_heads_pos).
// The full build (first epoch, re-cut, renumbering, a merge that missed a key) produces the same two arrays by the sorts and runs the
// same tail, so every binned test exercises the tail; tests/test_binned_tables_gpu.py holds the patched tables to the sorted ones.
__global__ __launch_bounds__(BLOCK) void k_bin_to_amajor(const uint64_t *__restrict__ wb, int Ed, int abits, int bbits, uint64_t *__restrict__ wa) {
    const uint64_t lo_mask = (1ull << (BIN_RL + BIN_HL)) - 1ull, amask = (1ull << abits) - 1ull;
    for (int q = blockIdx.x * BLOCK + threadIdx.x; q < Ed; q += gridDim.x * BLOCK) {
        const uint64_t w = wb[q], hi = w >> (BIN_RL + BIN_HL);
        const uint64_t a = hi & amask, b = hi >> abits;
        wa[q] = (((a << bbits) | b) << (BIN_RL + BIN_HL)) | (w & lo_mask);
    }
}
// first[first_block * n_second + second_block] = index of the first word of that block pair (only pairs that occur are written -- and read)
__global__ __launch_bounds__(BLOCK) void k_bin_tile_starts(const uint64_t *__restrict__ w, int Ed, int second_bits, int n_second,
                                                           int *__restrict__ first) {
    const uint64_t smask = (1ull << second_bits) - 1ull;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < Ed; i += gridDim.x * BLOCK) {
        const uint64_t hi = w[i] >> (BIN_RL + BIN_HL);
        if (i == 0 || hi != (w[i - 1] >> (BIN_RL + BIN_HL))) first[(size_t)(hi >> second_bits) * (size_t)n_second + (size_t)(hi & smask)] = i;
    }
}
__global__ __launch_bounds__(BLOCK) void k_bin_rows(const uint64_t *__restrict__ wb, int Ed, uint16_t *__restrict__ dl) {
    for (int q = blockIdx.x * BLOCK + threadIdx.x; q < Ed; q += gridDim.x * BLOCK) dl[q] = (uint16_t)((wb[q] >> BIN_HL) & ((1u << BIN_RL) - 1u));
}
__global__ __launch_bounds__(BLOCK) void k_bin_heads_pos(const uint64_t *__restrict__ wa, int Ed, int bbits, int n_a, int n_b,
                                                         const int *__restrict__ first_a, const int *__restrict__ first_b,
                                                         uint16_t *__restrict__ hl, int *__restrict__ apos) {
    const uint64_t bmask = (1ull << bbits) - 1ull;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < Ed; j += gridDim.x * BLOCK) {
        const uint64_t w = wa[j], hi = w >> (BIN_RL + BIN_HL);
        const size_t a = (size_t)(hi >> bbits), b = (size_t)(hi & bmask);
        hl[j] = (uint16_t)(w & ((1ull << BIN_HL) - 1ull));
        apos[j] = first_b[b * (size_t)n_a + a] + (j - first_a[a * (size_t)n_b + b]);
    }
}

// Diagnostic build only (-DDPPR_STAMPS, tools/r03/stamps_bin.sh): wall-clock (100 MHz) stamps per workgroup of the last
// launch whose frontier held at least a third of the vertices
#ifdef DPPR_STAMPS
__device__ unsigned long long g_bin_stamps[2][16384 * 6];
#define BSTAMP(K, i, val)                                                                                                      \
    do {                                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 16384 && 3 * (long long)*cnt_in >= NV) g_bin_stamps[K][blockIdx.x * 6 + (i)] = (val); \
    } while (0)
#else
#define BSTAMP(K, i, val) ((void)0)
#endif

// Pass 1: vals[B-major position] = x[head] for a CHUNK of an A-block's edges (chunk = {block, first entry, end}: a block of
// many edges is dealt to several workgroups, each stages the block's slice of x). LDS: that slice.
struct BinChunk {
    int blk, j0, j1;
};
__global__ __launch_bounds__(BIN_NT) void k_bin_scatter(int NV, const int *__restrict__ cnt_in, const int *__restrict__ acut,
                                                        const BinChunk *__restrict__ chunks, const uint16_t *__restrict__ hl,
                                                        const int *__restrict__ apos, const double *__restrict__ x,
                                                        double *__restrict__ vals) {
    extern __shared__ double s_x[];
    if (*cnt_in == 0) return; // empty frontier: nothing is read or written (k_bin_reduce returns as well)
    const BinChunk ch = chunks[blockIdx.x];
    const int h0 = acut[ch.blk], h1 = acut[ch.blk + 1];
    const int j0 = ch.j0, j1 = ch.j1;
    BSTAMP(0, 0, wall_clock64());
    BSTAMP(0, 4, (unsigned long long)(j1 - j0));
    for (int i = threadIdx.x; i < h1 - h0; i += BIN_NT) s_x[i] = x[h0 + i];
    __syncthreads();
    BSTAMP(0, 1, wall_clock64());
    for (int j = j0 + (int)threadIdx.x; j < j1; j += BIN_NT * BIN_U) {
        int h[BIN_U], q[BIN_U];
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const int jj = j + k * BIN_NT;
            h[k] = jj < j1 ? (int)hl[jj] : -1;
            q[k] = jj < j1 ? apos[jj] : 0;
        }
#pragma unroll
        for (int k = 0; k < BIN_U; ++k)
            if (h[k] >= 0) vals[q[k]] = s_x[h[k]];
    }
#ifdef DPPR_STAMPS
    __syncthreads();
    BSTAMP(0, 2, wall_clock64());
#endif
}

// In-edges of the vertices of a frontier list (what pushing it costs in returning atomics): one thread per vertex.
__global__ __launch_bounds__(BLOCK) void k_front_degree(const int *__restrict__ list, const int *__restrict__ cnt, const int *__restrict__ in_row_ptr,
                                                  unsigned long long *__restrict__ out) {
    const int n = *cnt;
    unsigned long long d = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const int v = list[i];
        d += (unsigned long long)(in_row_ptr[v + 1] - in_row_ptr[v]);
    }
    // wave total (two 32-bit halves through the integer DPP ladder would overflow on hubs: go through LDS-free shuffles)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, WAVE);
    if (lane_id() == 0 && d) atomicAdd(out, d);
}

// wave64 segmented inclusive sum on the DPP path: `head` marks the first lane of a run; every lane receives the sum
// of its run up to itself. Same ladder as wave_inclusive_scan; a lane that has seen a head inside its window stops
// taking from below.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(double &x, int &f) {
    const double y = dpp_take_f64<CTRL, ROW_MASK>(x);
    const int g = __builtin_amdgcn_update_dpp(0, f, CTRL, ROW_MASK, 0xf, false);
    if (!f) x += y; // (lanes without a source received 0.0 and 0)
    f |= g;
}
__device__ __forceinline__ double wave_segmented_sum(double x, bool head) {
    int f = head ? 1 : 0;
    seg_step<0x111, 0xf>(x, f);
    seg_step<0x112, 0xf>(x, f);
    seg_step<0x114, 0xf>(x, f);
    seg_step<0x118, 0xf>(x, f);
    seg_step<0x142, 0xa>(x, f);
    seg_step<0x143, 0xc>(x, f);
    return x;
}

// Pass 2: the rows of B-block blockIdx.x. LDS (dynamic): per row accumulator (8) + reciprocal (8) + outdeg + 1 (4).
// Workgroups beyond the n_b blocks take the rows [NV_bin, NV) in pieces of rows_cap: vertices that received their id after
// the tables were built (they have no edge in this epoch, but may hold state).
__global__ __launch_bounds__(BIN_NT) void k_bin_reduce(int NV, int NV_bin, int n_b, const int *__restrict__ cnt_in, const int *__restrict__ bcut,
                                                       int rows_cap, const int *__restrict__ out_row_ptr,
                                                       const uint16_t *__restrict__ dl, const double *__restrict__ vals,
                                                       const double *__restrict__ x, double *__restrict__ x_new,
                                                       double *__restrict__ r, double *__restrict__ p,
                                                       int *__restrict__ cnt_out, int *__restrict__ cnt_zero, int phase, double eps,
                                                       IterStats *__restrict__ stats, int *__restrict__ log_slot,
                                                       const int *__restrict__ in_row_ptr, unsigned long long *__restrict__ deg_out) {
    extern __shared__ double s_bin[];
    __shared__ int s_cnt[BIN_NT / WAVE];
    __shared__ unsigned long long s_deg[BIN_NT / WAVE];
    __shared__ unsigned long long s_edges[BIN_NT / WAVE];
    double *s_acc = s_bin, *s_rcp = s_bin + rows_cap;
    int *s_den = reinterpret_cast<int *>(s_bin + 2 * rows_cap);
    const int F = *cnt_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *cnt_zero = 0;
        *log_slot = F;
    }
    if (F == 0) return;
    const int lane = lane_id(), w = wave_id();
    int v0, v1;
    if ((int)blockIdx.x < n_b) {
        v0 = bcut[blockIdx.x];
        v1 = bcut[blockIdx.x + 1];
    } else {
        v0 = min(NV_bin + ((int)blockIdx.x - n_b) * rows_cap, NV);
        v1 = min(v0 + rows_cap, NV);
    }
    const int nrows = v1 - v0;
    for (int i = threadIdx.x; i < nrows; i += BIN_NT) {
        const int d = out_row_ptr[v0 + i + 1] - out_row_ptr[v0 + i];
        s_acc[i] = r[v0 + i];
        s_rcp[i] = 1.0 / (double)(d + 1);
        s_den[i] = d + 1;
    }
    const int e0 = out_row_ptr[v0], e1 = out_row_ptr[v1];
    BSTAMP(1, 0, wall_clock64());
    BSTAMP(1, 4, (unsigned long long)(e1 - e0));
    __syncthreads();
    BSTAMP(1, 1, wall_clock64());
    unsigned long long edges = 0;
    for (int q0 = e0; q0 < e1; q0 += BIN_NT * BIN_U) { // workgroup-uniform trip count (the DPP steps below want whole waves);
        const int q = q0 + (int)threadIdx.x;           // a wave's lanes hold 64 consecutive entries per k
        double xv[BIN_U];
        int row[BIN_U];
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const int qq = q + k * BIN_NT;
            xv[k] = qq < e1 ? vals[qq] : 0.0;
            row[k] = qq < e1 ? (int)dl[qq] : -1;
        }
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const bool nz = xv[k] != 0.0;
            const uint64_t any = __ballot(nz);
            if (any == 0) continue; // wave-uniform
            edges += (unsigned long long)__popcll(any);
            const int i = nz ? row[k] : -1 - lane; // a lane without a term is a run of its own
            const double t = nz ? push_term(xv[k], (double)s_den[row[k]], s_rcp[row[k]]) : 0.0;
            // Lanes of one row are neighbours (a tile's entries are in (row, head) order). Many short runs: one LDS atomic
            // per lane (a run of k lanes is a k-way conflict, cheap for small k). Few long runs (a hub's row): the runs are
            // summed in registers first and cost ONE atomic each -- thousands of atomics on one LDS word would serialise.
            const int below = __builtin_amdgcn_update_dpp(-0x7fffffff, i, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool head = lane == 0 || below != i;
            if (__popcll(__ballot(head)) >= 16) { // wave-uniform
                if (nz) lds_add(&s_acc[i], t);
            } else {
                const int above = __builtin_amdgcn_update_dpp(-0x7fffffff, i, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                const double sum = wave_segmented_sum(t, head);
                if (nz && (lane == WAVE - 1 || above != i)) lds_add(&s_acc[i], sum); // the run's last lane
            }
        }
    }
    __syncthreads();
    BSTAMP(1, 2, wall_clock64());
    // repair, threshold, next snapshot (k_pull_iter's `finish`)
    // ... and what the NEXT iteration would cost as a push: the in-edges of the vertices that enter the frontier (one
    // returning atomic each, gpu/ExpandRev.cuh:70-73) -- the host weighs that against another sweep (dppr_engine.hip)
    int n_legal = 0, deg = 0;
    for (int i = threadIdx.x; i < nrows; i += BIN_NT) {
        const int v = v0 + i;
        const double rv = r[v], xvv = x[v];
        double rn = s_acc[i];
        if (xvv != 0.0) rn -= xvv;
        const bool lg = legal(rn, phase, eps);
        if (rn != rv) r[v] = rn;
        x_new[v] = lg ? rn : 0.0;
        if (lg) {
            p[v] += ALPHA * rn;
            n_legal++;
            if (deg_out) deg += in_row_ptr ? in_row_ptr[v + 1] - in_row_ptr[v] : s_den[i] - 1; // (undirected: the in-edges are the out-edges)
        }
    }
    const int cw = wave_inclusive_scan(n_legal);
    const int dw = wave_inclusive_scan(deg); // (a wave's 64 rows: below 2^31 as long as the epoch's edges are)
    if (lane == WAVE - 1) {
        s_cnt[w] = cw;
        s_deg[w] = (unsigned long long)(unsigned)dw;
    }
    if (lane == 0) s_edges[w] = edges;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        unsigned long long te = 0, td = 0;
#pragma unroll
        for (int k = 0; k < BIN_NT / WAVE; ++k) {
            tot += s_cnt[k];
            te += s_edges[k];
            td += s_deg[k];
        }
        if (tot) atomicAdd(cnt_out, tot);
        if (td && deg_out) atomicAdd(deg_out, td);
        if (te) atomicAdd(&stats->blk_E[blockIdx.x & (STAT_SLOTS - 1)], te); // (more workgroups than slots: slots are shared)
    }
    BSTAMP(1, 3, wall_clock64());
}

} // namespace dppr
