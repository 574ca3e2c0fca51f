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
//     (at most 64 hb_tiles rows, about `target` out-edges, a hub row alone). A TILE = the edges from A-block a into B-block b;
//     inside a tile the edges are ordered by (head, row), so the edges of one head into one B-block -- a RUN -- are neighbours.
//     On a skewed window many edges share their run (twitter stand-in: 0.83 runs per edge under these cuts, tools/r06/dedup_count.py),
//     and x[u] is handed from pass 1 to pass 2 once per RUN, not once per edge (round 6; rounds 3-5: once per edge, 24 B / edge).
//   * k_bin_scatter (workgroups take chunks of an A-block's runs, A-major order (A, B, head)): the block's slice of x is staged in
//     LDS; per run a 2-byte entry (head index inside the block + a "first run of its tile" bit) is streamed and x[u] is written to
//     vals[run's B-major index]: inside a tile both orders list the runs by ascending head, so the destination is
//     (A-major run index) + delta(tile) -- one 4-byte delta per TILE, found through the tile bits (ballot + per-64-runs ordinal).
//   * k_bin_reduce (one workgroup per B-block): row accumulators (starting at residual[v]), 1 / (outdeg + 1) and outdeg + 1 of the
//     block's rows sit in LDS; the B-major edge list is streamed (2 bytes per edge: row index inside the block + a "first edge of
//     its run" bit), a wave's 64 edges need the <= 64 CONSECUTIVE values vals[vb .. ] (vb: one int per 64 edges, loaded independently
//     of the edge entries; the lane's own value comes through a cross-lane read at the ordinal of its run),
//     add = (1.0-ALPHA) * x[u] / (outdeg(v) + 1) exactly as gpu/ExpandRev.cuh:72 (push_term) into the row's LDS accumulator
//     (a hub row alone in its block: the wave's terms are summed in registers first, segmented DPP scan, ONE LDS atomic); then repair
//     (gpu/ExpandRev.cuh:708-743), threshold and the next snapshot per row, as k_pull_iter does them.
//
// Streamed bytes per edge: 2 + rho (2 + 8 + 8) + 4 tiles / edge, rho = runs per edge (twitter stand-in 0.83: 17 B; rounds 3-5: 24);
// no global atomics. Results equal k_pull_iter's up to the order of each row's sum.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

constexpr int BIN_NT = 1024;   // threads of both passes
#ifndef DPPR_BIN_U
#define DPPR_BIN_U 4           // (8, and a software-pipelined loop, were measured: 3-5 % slower on the twitter stand-in)
#endif
constexpr int BIN_U = DPPR_BIN_U; // entries in flight per lane
#ifndef DPPR_BIN_WHATIF
#define DPPR_BIN_WHATIF 0 // (timing experiments that compute WRONG results: bit 0 = no LDS atomics in pass 2, bit 1 = no divisor reads, bit 2 = no cross-lane value read)
#endif

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

// The two orders of the tables are kept as sorted 64-bit words that carry everything the tables need:
//   B-major word = B-block | A-block | head inside its A-block (BIN_HL bits) | row inside its B-block (BIN_RL bits)
//   A-major word = A-block | B-block | head | row
// plain 64-bit order = (blocks, head, row): a run (one head into one B-block) is a stretch of words equal above the row field,
// a tile a stretch equal above the head field.
constexpr int BIN_RL = 13, BIN_HL = 15; // rows of a B-block <= 64 x BIN_MAX_HB_TILES = 7 680, heads of an A-block <= 64 x BIN_MAX_HA_TILES = 17 408
constexpr int BIN_LO = BIN_RL + BIN_HL; // bits below the block fields
constexpr uint16_t BIN_FLAG = 0x8000u;  // top bit of a 2-byte table entry: first edge of its run (dl) / first run of its tile (hl)

// The keys come in (head, row) order (the in-orientation keys of the CSR build): consecutive entries share their head or its
// neighbours (vblk_a[u] is a cached, coalesced read), the rows are random -- their B-block is found by bisection of the block cuts
// (a few thousand entries, staged in LDS when they fit `cuts_in_lds` ints) instead of a random 4-byte gather per edge out of a
// V-sized table. amajor = 1: A-major word, 0: B-major word.
__global__ __launch_bounds__(BLOCK) void k_bin_keys(const uint64_t *__restrict__ in_keys, int Ed, int bits,
                                                    const int *__restrict__ bcut, int n_b, int cuts_in_lds,
                                                    const int *__restrict__ vblk_a, const int *__restrict__ acut, int abits, int bbits,
                                                    uint64_t *__restrict__ w, int amajor) {
    extern __shared__ int s_bcut[];
    const int *cut = bcut;
    if (cuts_in_lds) {
        for (int k = threadIdx.x; k <= n_b; k += BLOCK) s_bcut[k] = bcut[k];
        __syncthreads();
        cut = s_bcut;
    }
    const uint64_t mask = (1ull << bits) - 1;
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < Ed; o += gridDim.x * BLOCK) {
        const uint64_t k = in_keys[o];
        const int u = (int)(k >> bits), v = (int)(k & mask); // head, row
        int lo = 0, hi = n_b; // last block whose first vertex is <= v (k_bin_vertex_block's rule)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cut[mid] <= v) lo = mid; else hi = mid;
        }
        const int a = vblk_a[u];
        const uint64_t blocks = amajor ? (((uint64_t)(uint32_t)a << bbits) | (uint64_t)(uint32_t)lo) : (((uint64_t)(uint32_t)lo << abits) | (uint64_t)(uint32_t)a);
        w[o] = (blocks << BIN_LO) | ((uint64_t)(uint32_t)(u - acut[a]) << BIN_RL) | (uint64_t)(uint32_t)(v - cut[lo]);
    }
}

// A-major order without a radix sort. The A-major words arrive in (head, row) order, i.e. A-block-major already: what the first sort
// does is group every A-block's segment by B-block, stably -- several radix passes over the whole window for a key of which the upper
// half is sorted. Here one workgroup owns an A-block: a histogram of its segment over the B-blocks (LDS, all waves), an exclusive scan,
// and then ONE wave walks the segment in order and places every word at (start of its B-block inside the segment + number of earlier
// words of that B-block): the rank inside a 64-word step from ballots (one per bit of the B-block number: the lanes that agree in
// every bit hold the same block; no memory access), the running starts in LDS (gathered once per step, advanced by the last lane of
// each B-block). Deterministic and equal to the stable sort. A block that is ONE head (a hub alone in its block) is sorted already --
// rows ascend, so do their blocks -- and is copied by all waves. Needs n_cnt <= BIN_CS_MAX counters; beyond, the radix sort stays.
constexpr int BIN_CS_NT = 256, BIN_CS_MAX = 8192, BIN_CS_AHEAD = 8;
__global__ __launch_bounds__(BIN_CS_NT) void k_bin_group(const uint64_t *__restrict__ w1, const int *__restrict__ segstart,
                                                         const int *__restrict__ segcut, int n_cnt, int n_pad, int cbits,
                                                         uint64_t *__restrict__ out) {
    extern __shared__ int s_cnt[]; // n_pad (a power of two >= max(n_cnt, 64))
    const int b = blockIdx.x, tid = threadIdx.x, lane = lane_id();
    const int s0 = segstart[b], s1 = segstart[b + 1];
    if (s1 <= s0) return;
    if (segcut[b + 1] - segcut[b] == 1) { // one vertex: in order already
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
    const uint32_t cmask = (1u << cbits) - 1u;
    for (int k = tid; k < n_pad; k += BIN_CS_NT) s_cnt[k] = 0;
    __syncthreads();
    {
        int i = s0 + tid;
        for (; i + 3 * BIN_CS_NT < s1; i += 4 * BIN_CS_NT) { // (four loads in flight per thread)
            uint64_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = w1[i + k * BIN_CS_NT];
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(&s_cnt[(uint32_t)(w[k] >> BIN_LO) & cmask], 1);
        }
        for (; i < s1; i += BIN_CS_NT) atomicAdd(&s_cnt[(uint32_t)(w1[i] >> BIN_LO) & cmask], 1);
    }
    __syncthreads();
    if (tid >= WAVE) return; // the walk is one wave's (nothing below needs a workgroup barrier)
    { // exclusive scan over the counted blocks: a lane sums its n_pad / 64 consecutive counters, the wave scans the lane sums
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
            const int a = (int)((uint32_t)(w[h] >> BIN_LO) & cmask);
            uint64_t m = __ballot(valid); // -> the valid lanes of this step that hold the same block as this lane: one ballot per key bit
            if (!m) break;
            for (int bit = 0; bit < cbits; ++bit) {
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

// ---- Tables PATCHED per slide instead of re-sorted per epoch (round 5). The engine keeps both orders PERSISTENT
// as sorted 64-bit words under block cuts that stay FROZEN between re-cuts (new ids extend the last blocks / append blocks; a re-cut
// every few dozen slides restores the balance). A slide
//   * forms the words of its 2c retired and 2c inserted edges (k_bin_keys, both forms), sorts those, and merges them into the two
//     arrays with the key merge of the CSR build (k_del_positions + k_merge_tiles, dppr_builder.hpp): 16 bytes per edge and array;
//   * and runs the TAIL below on the two arrays. The full build (first epoch, re-cut, renumbering, a merge that missed a key) produces
//     the same two arrays by sorts and runs the same tail, so every binned test exercises it; tests/test_binned_tables_gpu.py holds
//     the patched tables to the sorted ones bit for bit.
// The tail (round 6: runs and tiles). Per aligned block of 64 words: how many runs / tiles START in it (k_bin_count), an exclusive
// scan of those counts, and then
//   * from the B-major words (k_bin_btables): dl[q] = row | first-of-run bit; vb[k] = index of the run that holds word 64 k (the
//     values a wave's 64 edges need are vals[vb[k] ..], consecutive); vfirst[b, a] = run index of tile (a, b)'s first run;
//   * from the A-major words (k_bin_atables): the RUN list hl[j] = head | first-of-tile bit (j = A-major run index), per tile (in
//     A-major order) tdelta[t] = vfirst[b, a] - (A-major index of its first run), and arun[a] = first run of A-block a;
//   * from the run list (k_bin_count16 + scan + k_bin_tb): tb[k] = ordinal of the tile that holds run 64 k.
// swap the two block fields of a word: (H << lbits | L) -> (L << hbits | H)
__global__ __launch_bounds__(BLOCK) void k_bin_swap_blocks(const uint64_t *__restrict__ win, int Ed, int lbits, int hbits, uint64_t *__restrict__ wout) {
    const uint64_t lo_mask = (1ull << BIN_LO) - 1ull, lmask = (1ull << lbits) - 1ull;
    for (int q = blockIdx.x * BLOCK + threadIdx.x; q < Ed; q += gridDim.x * BLOCK) {
        const uint64_t w = win[q], blocks = w >> BIN_LO;
        const uint64_t l = blocks & lmask, h = blocks >> lbits;
        wout[q] = (((l << hbits) | h) << BIN_LO) | (w & lo_mask);
    }
}
// cnt[k] = (runs that start in words [64 k, 64 k + 64)) << 32 | tiles that start there; cnt[n_blk] = 0 (so that the scan's last entry is the total)
__global__ __launch_bounds__(BLOCK) void k_bin_count(const uint64_t *__restrict__ w, int Ed, unsigned long long *__restrict__ cnt) {
    const int n_blk = (Ed + WAVE - 1) / WAVE, lane = lane_id();
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wave_id(); k <= n_blk; k += gridDim.x * WAVES_PER_BLOCK) {
        const int i = k * WAVE + lane;
        bool rf = false, tf = false;
        if (i < Ed) {
            const uint64_t me = w[i], prev = i > 0 ? w[i - 1] : ~0ull;
            rf = (me >> BIN_RL) != (prev >> BIN_RL);
            tf = (me >> BIN_LO) != (prev >> BIN_LO);
        }
        const unsigned long long r = (unsigned long long)__popcll(__ballot(rf)), t = (unsigned long long)__popcll(__ballot(tf));
        if (lane == 0) cnt[k] = (r << 32) | t;
    }
}
// X = exclusive scan of k_bin_count's counts over the B-major words (n_blk + 1 entries, the last one the totals)
__global__ __launch_bounds__(BLOCK) void k_bin_btables(const uint64_t *__restrict__ wb, int Ed, const unsigned long long *__restrict__ X, int n_a, int abits,
                                                       uint16_t *__restrict__ dl, int *__restrict__ vb, int *__restrict__ vfirst) {
    const int n_blk = (Ed + WAVE - 1) / WAVE, lane = lane_id();
    const uint64_t amask = (1ull << abits) - 1ull;
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wave_id(); k < n_blk; k += gridDim.x * WAVES_PER_BLOCK) {
        const int q = k * WAVE + lane;
        bool rf = false, tf = false;
        uint64_t me = 0;
        if (q < Ed) {
            me = wb[q];
            const uint64_t prev = q > 0 ? wb[q - 1] : ~0ull;
            rf = (me >> BIN_RL) != (prev >> BIN_RL);
            tf = (me >> BIN_LO) != (prev >> BIN_LO);
        }
        const uint64_t rm = __ballot(rf);
        const int before = (int)(X[k] >> 32);
        if (q < Ed) dl[q] = (uint16_t)((me & ((1u << BIN_RL) - 1u)) | (rf ? BIN_FLAG : 0));
        if (tf) { // (a tile's first word starts a run: its index = the runs before it)
            const uint64_t blocks = me >> BIN_LO;
            vfirst[(size_t)(blocks >> abits) * (size_t)n_a + (size_t)(blocks & amask)] = before + mbcnt(rm);
        }
        if (lane == 0) {
            vb[k] = before - 1 + (int)(rm & 1ull);
            if (k == n_blk - 1) vb[n_blk] = (int)(X[n_blk] >> 32) - 1; // (the last run: what bounds the last block's value loads)
        }
    }
}
// X = the same scan over the A-major words
__global__ __launch_bounds__(BLOCK) void k_bin_atables(const uint64_t *__restrict__ wa, int Ed, const unsigned long long *__restrict__ X, int n_a, int bbits,
                                                       const int *__restrict__ vfirst, uint16_t *__restrict__ hl, int *__restrict__ tdelta,
                                                       int *__restrict__ arun) {
    const int n_blk = (Ed + WAVE - 1) / WAVE, lane = lane_id();
    const uint64_t bmask = (1ull << bbits) - 1ull;
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wave_id(); k < n_blk; k += gridDim.x * WAVES_PER_BLOCK) {
        const int i = k * WAVE + lane;
        bool rf = false, tf = false, af = false;
        uint64_t me = 0;
        if (i < Ed) {
            me = wa[i];
            const uint64_t prev = i > 0 ? wa[i - 1] : ~0ull;
            rf = (me >> BIN_RL) != (prev >> BIN_RL);
            tf = (me >> BIN_LO) != (prev >> BIN_LO);
            af = (me >> (BIN_LO + bbits)) != (prev >> (BIN_LO + bbits));
        }
        const uint64_t rm = __ballot(rf), tm = __ballot(tf);
        const unsigned long long x = X[k];
        if (rf) {
            const int j = (int)(x >> 32) + mbcnt(rm);
            hl[j] = (uint16_t)(((me >> BIN_RL) & ((1u << BIN_HL) - 1u)) | (tf ? BIN_FLAG : 0));
            if (tf) {
                const uint64_t blocks = me >> BIN_LO;
                const size_t a = (size_t)(blocks >> bbits), b = (size_t)(blocks & bmask);
                tdelta[(int)(x & 0xffffffffull) + mbcnt(tm)] = vfirst[b * (size_t)n_a + a] - j;
                if (af) arun[a] = j;
            }
        }
    }
}
// the same count over the run list's tile bits (one int per 64 runs; cnt[n_rb] = 0)
__global__ __launch_bounds__(BLOCK) void k_bin_count16(const uint16_t *__restrict__ hl, int R, int *__restrict__ cnt) {
    const int n_rb = (R + WAVE - 1) / WAVE, lane = lane_id();
    for (int k = blockIdx.x * WAVES_PER_BLOCK + wave_id(); k <= n_rb; k += gridDim.x * WAVES_PER_BLOCK) {
        const int j = k * WAVE + lane;
        const bool tf = j < R && (hl[j] & BIN_FLAG);
        const int t = __popcll(__ballot(tf));
        if (lane == 0) cnt[k] = t;
    }
}
// tb[k] = ordinal of the tile that holds run 64 k (X2 = exclusive scan of k_bin_count16's counts); tb[n_rb] = the last tile
__global__ __launch_bounds__(BLOCK) void k_bin_tb(const uint16_t *__restrict__ hl, int R, const int *__restrict__ X2, int *__restrict__ tb) {
    const int n_rb = (R + WAVE - 1) / WAVE;
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k <= n_rb; k += gridDim.x * BLOCK)
        tb[k] = k < n_rb ? X2[k] - 1 + ((hl[(size_t)k * WAVE] & BIN_FLAG) ? 1 : 0) : X2[n_rb] - 1;
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

// the lanes 1 .. lane of a wave (a table entry's ordinal inside its aligned block of 64 counts the bits set there)
__device__ __forceinline__ uint64_t lanes_1_to_me() { return (~0ull >> (WAVE - 1 - lane_id())) & ~1ull; }

// Pass 1: vals[run's B-major index] = x[head] for a CHUNK of an A-block's runs (chunk = {block, first run, end}: a block of
// many runs is dealt to several workgroups, each stages the block's slice of x). LDS: that slice. A wave takes ALIGNED blocks of 64
// runs (the tables tb / the tile bits count per aligned block) and masks the entries outside its chunk.
struct BinChunk {
    int blk, j0, j1;
};
__global__ __launch_bounds__(BIN_NT, 8) void k_bin_scatter(int NV, const int *__restrict__ cnt_in, const int *__restrict__ acut,
                                                        const BinChunk *__restrict__ chunks, const uint16_t *__restrict__ hl,
                                                        const int *__restrict__ tb, const int *__restrict__ tdelta, int R,
                                                        const double *__restrict__ x, double *__restrict__ vals) {
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
    const int lane = lane_id();
    const uint64_t below = lanes_1_to_me();
    const int kb1 = (j1 + WAVE - 1) / WAVE;
    // The tile ordinals of a step's blocks are requested one step AHEAD: the per-tile differences are addressed with them, and two
    // dependent round trips per step is what a streaming loop of this shape cannot hide (k_bin_reduce: 514 -> 674 us before this)
    int t0n[BIN_U], t1n[BIN_U];
#pragma unroll
    for (int k = 0; k < BIN_U; ++k) {
        const int kk = __builtin_amdgcn_readfirstlane(j0 / WAVE + wave_id() + k * (BIN_NT / WAVE)); // (wave-uniform: scalar loads, scalar registers)
        t0n[k] = kk < kb1 ? tb[kk] : 0;
        t1n[k] = kk < kb1 ? tb[kk + 1] : 0;
    }
    for (int kb = j0 / WAVE + wave_id(); kb < kb1; kb += (BIN_NT / WAVE) * BIN_U) {
        int hv[BIN_U], td[BIN_U];
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const int kk = kb + k * (BIN_NT / WAVE);
            hv[k] = -1;
            td[k] = 0;
            if (kk < kb1) {
                const int jj = kk * WAVE + lane;
                if (jj < R) hv[k] = (int)hl[jj];
                if (lane <= t1n[k] - t0n[k]) td[k] = tdelta[t0n[k] + lane];
            }
        }
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) { // the next step's ordinals
            const int kk = __builtin_amdgcn_readfirstlane(kb + (BIN_NT / WAVE) * BIN_U + k * (BIN_NT / WAVE));
            if (kk < kb1) {
                t0n[k] = tb[kk];
                t1n[k] = tb[kk + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const int kk = kb + k * (BIN_NT / WAVE);
            if (kk >= kb1) break; // wave-uniform
            const uint64_t tm = __ballot(hv[k] >= 0 && (hv[k] & BIN_FLAG));
            const int delta = __shfl(td[k], __popcll(tm & below), WAVE);
            const int jj = kk * WAVE + lane;
            if (jj >= j0 && jj < j1) vals[jj + delta] = s_x[hv[k] & (BIN_FLAG - 1)];
        }
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
__global__ __launch_bounds__(BIN_NT, 8) void k_bin_reduce(int NV, int NV_bin, int n_b, const int *__restrict__ cnt_in, const int *__restrict__ bcut,
                                                       int rows_cap, const int *__restrict__ out_row_ptr,
                                                       const uint16_t *__restrict__ dl, const int *__restrict__ vb, int Ed, const double *__restrict__ vals,
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
    const uint64_t runs_below = lanes_1_to_me();
    const int kb1 = (e1 + WAVE - 1) / WAVE;
    // ALIGNED blocks of 64 edges (vb and the run bits count per aligned block); a wave's lanes hold 64 consecutive entries per k.
    // The two loads of a block are independent: the edge entries, and the <= 64 consecutive values its runs need.
    int i0n[BIN_U], i1n[BIN_U]; // (requested one step ahead, as k_bin_scatter's tile ordinals)
#pragma unroll
    for (int k = 0; k < BIN_U; ++k) {
        const int kk = __builtin_amdgcn_readfirstlane(e0 / WAVE + w + k * (BIN_NT / WAVE)); // (wave-uniform: scalar loads, scalar registers)
        i0n[k] = kk < kb1 ? vb[kk] : 0;
        i1n[k] = kk < kb1 ? vb[kk + 1] : 0;
    }
    for (int kb = e0 / WAVE + w; kb < kb1; kb += (BIN_NT / WAVE) * BIN_U) {
        double xw[BIN_U];
        int dv[BIN_U];
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const int kk = kb + k * (BIN_NT / WAVE);
            xw[k] = 0.0;
            dv[k] = -1;
            if (kk < kb1) {
                const int qq = kk * WAVE + lane;
                if (qq < Ed) dv[k] = (int)dl[qq];
                if (lane <= i1n[k] - i0n[k]) xw[k] = vals[i0n[k] + lane];
            }
        }
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) { // the next step's run ordinals
            const int kk = __builtin_amdgcn_readfirstlane(kb + (BIN_NT / WAVE) * BIN_U + k * (BIN_NT / WAVE));
            if (kk < kb1) {
                i0n[k] = vb[kk];
                i1n[k] = vb[kk + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < BIN_U; ++k) {
            const int kk = kb + k * (BIN_NT / WAVE);
            if (kk >= kb1) break; // wave-uniform
            const uint64_t rm = __ballot(dv[k] >= 0 && (dv[k] & BIN_FLAG));
            const int src = __popcll(rm & runs_below); // the lane that loaded this edge's run value
#if DPPR_BIN_WHATIF & 4
            const double xv = xw[k] + (double)src * 1e-300;
#else
            const double xv = __hiloint2double(__shfl(__double2hiint(xw[k]), src, WAVE), __shfl(__double2loint(xw[k]), src, WAVE));
#endif
            const int qq = kk * WAVE + lane;
            const int rowk = dv[k] & ((1 << BIN_RL) - 1);
            const bool nz = qq >= e0 && qq < e1 && xv != 0.0;
            const uint64_t any = __ballot(nz);
            if (any == 0) continue; // wave-uniform
            edges += (unsigned long long)__popcll(any);
            const int i = nz ? rowk : -1 - lane; // a lane without a term is a run of its own
#if DPPR_BIN_WHATIF & 2
            const double t = nz ? push_term(xv, 1024.0, 1.0 / 1024.0) : 0.0;
#else
            const double t = nz ? push_term(xv, (double)s_den[rowk], s_rcp[rowk]) : 0.0;
#endif
            // A tile's entries are in (head, row) order: neighbouring lanes hold different rows as a rule (one LDS atomic per lane)
            // -- except in the block of a hub row, where all of them hold the same one: the runs are summed in registers first and
            // cost ONE atomic each (thousands of atomics on one LDS word would serialise).
            const int below = __builtin_amdgcn_update_dpp(-0x7fffffff, i, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool head = lane == 0 || below != i;
#if DPPR_BIN_WHATIF & 1
            if (t == 1.2345e-300) s_acc[0] = t;
            else
#endif
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
