// dppr_builder.hpp -- device graph builder (sliding window -> CSRs), id translation, calibration microbenchmark.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

// ---------------------------------------------------------------------------
// a9  SlidingGraphBuilder (gpu/SlidingGraphBuilder.cuh:62-242) kernels.
// The window lives in a ring in stream order (nothing is memmoved per slide, unlike
// IncCopyStreamFromCPU :163-181); out-degrees are a plain int array updated from
// the batch (replaces CollectOutDegree + exclusive_scan, :49-60,193-201).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_deg_update(const int *__restrict__ w1, const int *__restrict__ w2, int n,
                                                      int directed, int sign, int *__restrict__ outdeg) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        atomicAdd(&outdeg[w1[i]], sign);
        if (!directed) atomicAdd(&outdeg[w2[i]], sign);
    }
}

// key = dst << bits | src, one per directed edge (EdgePairScatter :11-24 + the (x,y) order of :41-47)
__global__ __launch_bounds__(BLOCK) void k_make_keys(const int *__restrict__ w1, const int *__restrict__ w2, int W,
                                                     int directed, int bits, uint64_t *__restrict__ keys) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < W; i += gridDim.x * BLOCK) {
        const uint64_t a = (uint32_t)w1[i], b = (uint32_t)w2[i];
        keys[i] = (b << bits) | a; // edge a -> b, row b of the in-CSR
        if (!directed) keys[(int64_t)W + i] = (a << bits) | b;
    }
}

// ---------------------------------------------------------------------------
// f1  Incremental CSR maintenance. The reference re-sorts the WHOLE window every batch
// (thrust::sort in BuildCSRGraph, gpu/SlidingGraphBuilder.cuh:203-221). Here the sorted key
// array of the previous epoch is kept; a slide sorts only the batch's own keys (c deleted +
// c inserted edges), marks the deleted instances in the old array by binary search, and
// produces the new sorted array with one select + one merge pass: O(Ed) streaming instead
// of O(Ed log Ed). Duplicate edges are handled by rank (the i-th deletion of a key removes
// its i-th instance).
// ---------------------------------------------------------------------------
// keys of a segment of the ring: in-orientation (dst << bits | src) and, for directed graphs,
// out-orientation (src << bits | dst). Undirected: both directions go to `in` (the two
// orientations are the same multiset, one array serves both CSRs).
__global__ __launch_bounds__(BLOCK) void k_make_keys_seg(const int *__restrict__ w1, const int *__restrict__ w2, int n,
                                                         int directed, int bits, uint64_t *__restrict__ in_keys,
                                                         uint64_t *__restrict__ out_keys) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint64_t a = (uint32_t)w1[i], b = (uint32_t)w2[i];
        if (directed) {
            in_keys[i] = (b << bits) | a;
            out_keys[i] = (a << bits) | b;
        } else {
            in_keys[2 * i] = (b << bits) | a;
            in_keys[2 * i + 1] = (a << bits) | b;
        }
    }
}

__device__ __forceinline__ int lower_bound_u64(const uint64_t *__restrict__ a, int n, uint64_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// position in the persistent sorted keys of every key a slide retires: the rank-th deletion of a key takes its rank-th
// instance. Ascending, because del_sorted is. A key that is not there -- never, on a consistent window -- gets n AND is counted
// in *miss: k_merge_tiles bisects delpos and would silently drop or duplicate keys behind a hole, so a slide that
// finds a miss discards the merge and re-sorts the whole window (dppr_engine.hip: dppr_slide).
__global__ __launch_bounds__(BLOCK) void k_del_positions(const uint64_t *__restrict__ sorted, int n,
                                                         const uint64_t *__restrict__ del_sorted, int nd, int *__restrict__ delpos,
                                                         int *__restrict__ miss) {
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < nd; j += gridDim.x * BLOCK) {
        const uint64_t key = del_sorted[j];
        const int rank = j - lower_bound_u64(del_sorted, nd, key);
        const int pos = lower_bound_u64(sorted, n, key) + rank;
        const bool found = pos < n && sorted[pos] == key;
        delpos[j] = found ? pos : n;
        if (!found) atomicAdd(miss, 1);
    }
}
__device__ __forceinline__ int lower_bound_i32(const int *__restrict__ a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The key merge of a slide (f1) in ONE pass over the persistent sorted keys: a workgroup takes a tile of CMP_TILE old keys, drops
// the retired ones (their positions: delpos), and merges what is left with the inserted keys that fall into the tile's key range
// -- every key goes straight to its final place: tile start - retired before it + inserted before it, then the rank inside the
// tile's merge (kept keys: binary search of the tile's inserted keys; inserted keys: of the kept ones in LDS; equal keys: old
// before new). Round 4 first replaced rocprim::select (4.45 ms per 147 M keys) by a count / scan / compact triple followed by
// rocprim::merge: 35 bytes per key in five kernels; this is 16.
constexpr int CMP_TILE = 2048;                 // keys per workgroup (16 KB of LDS; 4 096: 1.30 / 1.92 ms per twitter / friendster array, 2 048: 0.98 / 1.36, 1 024: 1.24 / 1.85)
constexpr int CMP_PER = CMP_TILE / BLOCK;      // ... per thread, strided (coalesced): element k * BLOCK + tid of the tile
constexpr int CMP_INS_LDS = 256;               // inserted keys of a tile searched in LDS when they are at most this many
__global__ __launch_bounds__(BLOCK) void k_merge_tiles(const uint64_t *__restrict__ keys, int n, const int *__restrict__ delpos, int nd,
                                                       const uint64_t *__restrict__ ins, int ni, uint64_t *__restrict__ out, size_t n_out) {
    __shared__ uint64_t s_key[CMP_TILE];
    __shared__ uint64_t s_ins[CMP_INS_LDS];
    __shared__ uint32_t s_del[CMP_TILE / 32];
    __shared__ int s_cnt[CMP_PER * WAVES_PER_BLOCK + 1];
    __shared__ int s_b[4];
    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const int t0 = blockIdx.x * CMP_TILE, t1 = min(n, t0 + CMP_TILE);
    if (tid < CMP_TILE / 32) s_del[tid] = 0u;
    if (tid == 0) s_b[0] = lower_bound_i32(delpos, nd, t0);
    if (tid == 1) s_b[1] = lower_bound_i32(delpos, nd, t1);
    if (tid == 2) s_b[2] = blockIdx.x == 0 ? 0 : lower_bound_u64(ins, ni, keys[t0]); // (keys below the first old key: the first tile's)
    if (tid == 3) s_b[3] = t1 < n ? lower_bound_u64(ins, ni, keys[t1]) : ni;
    __syncthreads();
    const int d0 = s_b[0], d1 = s_b[1], k0 = s_b[2], k1 = s_b[3], nk = k1 - k0;
    for (int d = d0 + tid; d < d1; d += BLOCK) {
        const int o = delpos[d] - t0;
        if ((unsigned)o < (unsigned)CMP_TILE) atomicOr(&s_del[o >> 5], 1u << (o & 31)); // (always, while delpos ascends)
    }
    if (nk <= CMP_INS_LDS)
        for (int k = tid; k < nk; k += BLOCK) s_ins[k] = ins[k0 + k];
    __syncthreads();
    uint64_t key[CMP_PER];
    bool kp[CMP_PER];
    int rank[CMP_PER];
#pragma unroll
    for (int k = 0; k < CMP_PER; ++k) { // coalesced: consecutive lanes, consecutive keys
        const int o = k * BLOCK + tid, i = t0 + o;
        kp[k] = i < t1 && !((s_del[o >> 5] >> (o & 31)) & 1u);
        key[k] = i < t1 ? keys[i] : 0ull;
        const uint64_t bal = __ballot(kp[k]);
        rank[k] = mbcnt(bal);
        if (lane == 0) s_cnt[k * WAVES_PER_BLOCK + w] = __popcll(bal);
    }
    __syncthreads();
    if (tid == 0) { // the (k, wave) pieces are consecutive runs of the tile: their kept counts -> starts
        int run = 0;
        for (int q = 0; q < CMP_PER * WAVES_PER_BLOCK; ++q) {
            const int c = s_cnt[q];
            s_cnt[q] = run;
            run += c;
        }
        s_cnt[CMP_PER * WAVES_PER_BLOCK] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CMP_PER; ++k)
        if (kp[k]) s_key[s_cnt[k * WAVES_PER_BLOCK + w] + rank[k]] = key[k];
    __syncthreads();
    const int kept = s_cnt[CMP_PER * WAVES_PER_BLOCK];
    const size_t base = (size_t)(t0 - d0) + (size_t)k0;
    const uint64_t *tin = nk <= CMP_INS_LDS ? s_ins : ins + k0; // (generic address space: LDS or global)
    for (int j = tid; j < kept; j += BLOCK) { // an old key: after the inserted keys below it
        const uint64_t kk = s_key[j];
        int lo = 0, hi = nk;
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if (tin[mid] < kk) lo = mid + 1; else hi = mid;
        }
        if (base + (size_t)j + (size_t)lo < n_out) out[base + (size_t)j + (size_t)lo] = kk; // (n_out: a retired key that was not there must not push the tail past the array)
    }
    for (int k = tid; k < nk; k += BLOCK) { // an inserted key: after the old keys up to and including it
        const uint64_t kk = tin[k];
        int lo = 0, hi = kept;
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if (s_key[mid] <= kk) lo = mid + 1; else hi = mid;
        }
        if (base + (size_t)k + (size_t)lo < n_out) out[base + (size_t)k + (size_t)lo] = kk;
    }
}

// hub selection: hist[b] = #vertices with min_deg * 2^b <= outdeg < min_deg * 2^(b+1)
constexpr int HUB_MIN_DEGREE_DEFAULT = 256;
__global__ __launch_bounds__(BLOCK) void k_deg_hist(const int *__restrict__ outdeg, int V, int min_deg,
                                                    int *__restrict__ hist) {
    __shared__ int s_hist[32]; // (per-workgroup counts first: a vertex above min_deg is no rarity on a large window)
    if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        const int d = outdeg[v];
        if (d >= min_deg) atomicAdd(&s_hist[31 - __clz(d / min_deg)], 1);
    }
    __syncthreads();
    if (threadIdx.x < 32 && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}
// degp1_enc[v] = ~slot for vertices with outdeg >= thresh (first HUB_CAP takers), else outdeg + 1: what an Adj entry carries for
// its tail, so that the CSR build gathers ONE word per edge (two -- slot, then degree -- were 361 M + 361 M random reads on the
// friendster stand-in)
__global__ __launch_bounds__(BLOCK) void k_assign_hubs(const int *__restrict__ outdeg, int V, int thresh,
                                                       int *__restrict__ degp1_enc, int *__restrict__ hub_v,
                                                       int *__restrict__ hub_degp1, int *__restrict__ n_hubs) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        int slot = -1;
        const int d = outdeg[v];
        if (d >= thresh) {
            slot = atomicAdd(n_hubs, 1);
            if (slot < HUB_CAP) {
                hub_v[slot] = v;
                hub_degp1[slot] = d + 1;
            } else {
                slot = -1;
            }
        }
        degp1_enc[v] = slot >= 0 ? ~slot : d + 1;
    }
}

// sorted keys -> row_ptr + Adj entries (cusparseXcoo2csr + EdgePairGather, :214-220)
__global__ __launch_bounds__(BLOCK) void k_build_csr(const uint64_t *__restrict__ skeys, int Ed, int V, int bits,
                                                     const int *__restrict__ degp1_enc, int *__restrict__ row_ptr,
                                                     Adj *__restrict__ adj) {
    const uint64_t mask = (1ull << bits) - 1;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < Ed; j += gridDim.x * BLOCK) {
        const uint64_t k = skeys[j];
        const int dst = (int)(k >> bits), src = (int)(k & mask);
        Adj a;
        a.v = src;
        a.degp1 = degp1_enc[src]; // negative: hub slot (see k_push_iter)
        adj[j] = a;
        const int prev = (j == 0) ? -1 : (int)(skeys[j - 1] >> bits);
        for (int x = prev + 1; x <= dst; ++x) row_ptr[x] = j;
    }
    // rows after the last non-empty one (with compacted ids: most of the id capacity), in parallel
    const int last = Ed ? (int)(skeys[Ed - 1] >> bits) : -1;
    for (int x = last + 1 + blockIdx.x * BLOCK + threadIdx.x; x <= V; x += gridDim.x * BLOCK) row_ptr[x] = Ed;
}

// out-CSR for the pull sweep: key = src << bits | dst, sorted -> out_row_ptr + out_col
__global__ __launch_bounds__(BLOCK) void k_make_out_keys(const int *__restrict__ w1, const int *__restrict__ w2, int W,
                                                         int directed, int bits, uint64_t *__restrict__ keys) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < W; i += gridDim.x * BLOCK) {
        const uint64_t a = (uint32_t)w1[i], b = (uint32_t)w2[i];
        keys[i] = (a << bits) | b;
        if (!directed) keys[(int64_t)W + i] = (b << bits) | a;
    }
}
__global__ __launch_bounds__(BLOCK) void k_build_out_csr(const uint64_t *__restrict__ skeys, int Ed, int V, int bits,
                                                         int *__restrict__ row_ptr, int *__restrict__ col) {
    const uint64_t mask = (1ull << bits) - 1;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < Ed; j += gridDim.x * BLOCK) {
        const uint64_t k = skeys[j];
        const int src = (int)(k >> bits);
        col[j] = (int)(k & mask);
        const int prev = (j == 0) ? -1 : (int)(skeys[j - 1] >> bits);
        for (int xx = prev + 1; xx <= src; ++xx) row_ptr[xx] = j;
    }
    const int last = Ed ? (int)(skeys[Ed - 1] >> bits) : -1;
    for (int xx = last + 1 + blockIdx.x * BLOCK + threadIdx.x; xx <= V; xx += gridDim.x * BLOCK) row_ptr[xx] = Ed;
}

// tile_prefix[t] = out_row_ptr[min(64 t, V)]: edges before tile t (input of the host's group cut)
__global__ __launch_bounds__(BLOCK) void k_tile_prefix(const int *__restrict__ out_row_ptr, int V, int n_tiles,
                                                       int *__restrict__ tile_prefix) {
    for (int t = blockIdx.x * BLOCK + threadIdx.x; t <= n_tiles; t += gridDim.x * BLOCK) {
        const long long v = (long long)t * WAVE;
        tile_prefix[t] = out_row_ptr[v < V ? (int)v : V];
    }
}

__global__ __launch_bounds__(BLOCK) void k_gather_deg(const int *__restrict__ e1, int L, const int *__restrict__ outdeg,
                                                      int *__restrict__ deg_after) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < L; i += gridDim.x * BLOCK) deg_after[i] = outdeg[e1[i]];
}

__global__ __launch_bounds__(BLOCK) void k_split_adj(const Adj *__restrict__ adj, int Ed, int *__restrict__ col) {
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < Ed; j += gridDim.x * BLOCK) col[j] = adj[j].v;
}

// ---------------------------------------------------------------------------
// Vertex compaction. The .bin header's V is an id RANGE (encoder/GraphEncoder.h:27-44) and a
// 10 % window touches only a fraction of it (14 % on the configs[1] stand-in). The engine
// numbers vertices by first appearance (internal ids 0..n_int) so every Theta(V) pass --
// Inspect, the pull sweep, hub selection, the CSR row fill -- covers only vertices that
// ever had an edge, and the hot state is contiguous. The C ABI speaks external ids; these
// two kernels translate p / r at the boundary.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_int_to_ext(const double *__restrict__ a_int, const int *__restrict__ ext2int,
                                                      int V, double *__restrict__ a_ext) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        const int m = ext2int[v];
        a_ext[v] = m >= 0 ? a_int[m] : 0.0;
    }
}
__global__ __launch_bounds__(BLOCK) void k_ext_to_int(const double *__restrict__ a_ext, const int *__restrict__ ext2int,
                                                      int V, double *__restrict__ a_int) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < V; v += gridDim.x * BLOCK) {
        const int m = ext2int[v];
        if (m >= 0) a_int[m] = a_ext[v];
    }
}

// ---------------------------------------------------------------------------
// calibration microbenchmark: returning f64 atomic adds at pseudo-random addresses
// ---------------------------------------------------------------------------
template <int SCOPE>
__global__ __launch_bounds__(BLOCK) void k_bench_atomics(double *__restrict__ table, uint64_t mask, int64_t n,
                                                         double *__restrict__ sink) {
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint64_t z = (uint64_t)i + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        acc += __hip_atomic_fetch_add(&table[z & mask], 1e-9, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 123.456) *sink = acc; // keep the returned values live
}
// ... and the same adds with the result unused (global_atomic_add_f64 without a return: fire and forget); KIND 1 = plain 8-byte
// stores at the same addresses, for the rate of scattered stores
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_bench_scatter(double *__restrict__ table, uint64_t mask, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        uint64_t z = (uint64_t)i + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (KIND == 0) (void)__hip_atomic_fetch_add(&table[z & mask], 1e-9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else table[z & mask] = 1e-9;
    }
}

// ---------------------------------------------------------------------------
// Renumbering of the internal ids (dppr_engine.hip: compact_ids / flush_moves). Internal ids are handed out on
// first sight and a vertex keeps its id when its last edge leaves the window -- its p / r must stay readable --
// so over a long stream the id space [0, n_int) outgrows the vertices that still have edges (stand-ins:
// + 29 % / + 42 % after the reference's default 100 batches). When every solver state has caught up with the
// newest epoch, a slide may renumber: vertices with an edge in the window (and the sources) keep their relative
// order at the front, the others are PARKED at the top of the id capacity [V - n_parked, V), outside of every
// sweep and scan, state rows included. A parked vertex that shows up in a later batch is given a fresh id and its
// rows are moved there (the parked zone stays dense: its lowest entry fills the hole).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_mark_live(const int *__restrict__ w1, const int *__restrict__ w2, int W,
                                                     uint8_t *__restrict__ live) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < W; i += gridDim.x * BLOCK) {
        live[w1[i]] = 1;
        live[w2[i]] = 1;
    }
}

__global__ __launch_bounds__(BLOCK) void k_in_degree(const int *__restrict__ w1, const int *__restrict__ w2, int W, int directed,
                                                     int *__restrict__ indeg) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < W; i += gridDim.x * BLOCK) {
        atomicAdd(&indeg[w2[i]], 1);
        if (!directed) atomicAdd(&indeg[w1[i]], 1);
    }
}

// Numbering order of a renumbering, on the device (the host version is dppr::numbering_order, dppr_idspace.hpp: same
// hash, same blocks of falling in-degree). key = block << 58 | hash >> 6 for a live vertex, top bit set for the others
// (they sort behind every live one, in id order); a radix sort of (key, id) pairs gives the order.
struct HotThresholds {
    int n;      // thresholds in use (0: hashed order only)
    int thr[8]; // in-degree of rank 512 K, 256 K, ..., 8 K among the live vertices (non-decreasing)
};
__device__ __forceinline__ uint64_t id_hash_dev(int v) { // = dppr::id_hash
    uint64_t z = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(BLOCK) void k_live_degree(const uint8_t *__restrict__ live, const int *__restrict__ indeg, int n,
                                                       int *__restrict__ out) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < n; v += gridDim.x * BLOCK) out[v] = live[v] ? indeg[v] : -1;
}
__global__ __launch_bounds__(BLOCK) void k_number_keys(const uint8_t *__restrict__ live, const int *__restrict__ int2ext,
                                                       const int *__restrict__ indeg, HotThresholds ht, int n,
                                                       uint64_t *__restrict__ keys, int *__restrict__ vals) {
    for (int v = blockIdx.x * BLOCK + threadIdx.x; v < n; v += gridDim.x * BLOCK) {
        uint64_t key = (1ull << 63) | (uint64_t)(uint32_t)v;
        if (live[v]) {
            const uint64_t h = id_hash_dev(int2ext[v]);
            if (ht.n > 0) {
                const int dg = indeg[v];
                uint64_t block = 0; // 0 = hottest
                for (int k = 0; k < ht.n; ++k) block += dg <= ht.thr[k] ? 1u : 0u;
                key = (h >> 6) | (block << 58);
            } else {
                key = h >> 1;
            }
        }
        keys[v] = key;
        vals[v] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void k_remap_ids(int *__restrict__ a, int n, const int *__restrict__ perm) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) a[i] = perm[a[i]];
}

// dst[perm[v]] = src[v] for every old position v that holds a vertex (perm >= 0); rows of w elements; dst is zero
// wherever no vertex lands (fresh ids find zero rows)
template <class T>
__global__ __launch_bounds__(BLOCK) void k_permute_rows(T *__restrict__ dst, const T *__restrict__ src,
                                                        const int *__restrict__ perm, int V, int w) {
    const int64_t n = (int64_t)V * w;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int v = (int)(i / w), k = (int)(i % w);
        const int m = perm[v];
        if (m >= 0) dst[(int64_t)m * w + k] = src[i];
    }
}

// row moves of a revival, in two steps so that a position may be source and destination at once:
// tmp[i] = a[src[i]];  then  a[dst[i]] = tmp[i]  (dst < 0: the row is only vacated);  then the vacated rows are zeroed
template <class T>
__global__ __launch_bounds__(BLOCK) void k_rows_gather(T *__restrict__ tmp, const T *__restrict__ a,
                                                       const int *__restrict__ src, int n, int w) {
    const int64_t tot = (int64_t)n * w;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < tot; i += (int64_t)gridDim.x * BLOCK)
        tmp[i] = a[(int64_t)src[i / w] * w + i % w];
}
template <class T>
__global__ __launch_bounds__(BLOCK) void k_rows_scatter(T *__restrict__ a, const T *__restrict__ tmp,
                                                        const int *__restrict__ dst, int n, int w) {
    const int64_t tot = (int64_t)n * w;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < tot; i += (int64_t)gridDim.x * BLOCK)
        a[(int64_t)dst[i / w] * w + i % w] = tmp[i];
}
template <class T>
__global__ __launch_bounds__(BLOCK) void k_rows_zero(T *__restrict__ a, const int *__restrict__ pos, int n, int w) {
    const int64_t tot = (int64_t)n * w;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < tot; i += (int64_t)gridDim.x * BLOCK)
        a[(int64_t)pos[i / w] * w + i % w] = T(0);
}

// A parked vertex has no edge: a push from it (gpu/ExpandRev.cuh:34-77 with an empty neighbour list, then
// RepairFrontierRev) is pagerank += ALPHA * residual, residual = 0. Parked rows satisfy |residual| <= the eps they
// were parked under; a solve with a SMALLER eps settles them here first. n = rows x lanes; counts what it pushed.
__global__ __launch_bounds__(BLOCK) void k_settle_parked(double *__restrict__ p, double *__restrict__ r, int64_t n,
                                                         double eps, int *__restrict__ cnt) {
    int hits = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const double rv = r[i];
        if (rv > eps || rv < -eps) {
            p[i] = p[i] + ALPHA * rv;
            r[i] = 0.0;
            ++hits;
        }
    }
    if (hits) atomicAdd(cnt, hits);
}

} // namespace dppr
