// dppr_idspace.hpp -- host side of the vertex numbering: external <-> internal ids, the parked zone, the row moves of
// revived vertices and the permutation of a renumbering. No HIP in here: the engine (dppr_engine.hip) inherits the
// maps and applies the moves / the permutation to its device arrays, and tests/native/idspace_test.cpp drives the
// same code against plain host arrays (CPU test suite).
//
// Internal ids are handed out on first sight. [0, n_int) is the LIVE zone -- what sweeps and scans cover --
// and [cap - n_parked, cap) the PARKED zone: vertices that had no edge in the window when the ids were last
// renumbered, kept there with their state rows. Every vertex with an id sits in exactly one of the two, the parked
// zone has no gaps, so n_int + n_parked never exceeds the number of vertices that have an id (<= cap) and the zones
// cannot overlap.
#pragma once

#include <algorithm>
#include <cstddef>
#include <functional>
#include <cstdint>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

namespace dppr {

// [0, n) in contiguous pieces on up to 8 host threads (small n: inline). The loops handed to it touch disjoint elements
// (a permutation's targets, one output per input): at twitter scale the id maps are 134 MB tables, and a serial pass
// of random reads / writes over them was most of a slide's host time (profiles/r03_slide_*).
#ifndef DPPR_PAR_MIN_PIECE
#define DPPR_PAR_MIN_PIECE 0 // (tests: a tiny piece size so that toy arrays take the threaded path too)
#endif
// Engines alive in this process (dppr_create / dppr_destroy). Every engine's host thread spins on its stream inside a timed
// region (loop_sync) and may have a lookahead running beside it: the helper threads of one call are capped at what the cores
// leave after one per engine (with -g N, N spinners plus 8 N helpers perturb the times they run beside).
inline std::atomic<int> g_live_engines{0};
template <class F>
inline void parallel_pieces(size_t n, size_t min_piece, F &&fn) {
    if (DPPR_PAR_MIN_PIECE) min_piece = DPPR_PAR_MIN_PIECE;
    static const size_t hw = std::thread::hardware_concurrency(); // (asked once: the call reads /sys)
    const size_t busy = (size_t)std::max(g_live_engines.load(std::memory_order_relaxed), 0);
    const size_t room = hw > busy ? hw - busy : 1;
    size_t nt = std::min<size_t>(std::min<size_t>(room, 8), n / std::max<size_t>(min_piece, 1));
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (size_t t = 1; t < nt; ++t) th.emplace_back([&fn, n, nt, t] { fn(n * t / nt, n * (t + 1) / nt); });
    fn((size_t)0, n / nt);
    for (auto &x : th) x.join();
}

// an atomic counter inside a struct that stays copyable (the native tests snapshot an IdSpace)
struct CopyableCounter {
    std::atomic<unsigned> v{1};
    CopyableCounter() = default;
    CopyableCounter(const CopyableCounter &o) : v(o.v.load(std::memory_order_relaxed)) {}
    CopyableCounter &operator=(const CopyableCounter &o) {
        v.store(o.v.load(std::memory_order_relaxed), std::memory_order_relaxed);
        return *this;
    }
    unsigned load(std::memory_order mo = std::memory_order_seq_cst) const { return v.load(mo); }
    unsigned fetch_add(unsigned d, std::memory_order mo = std::memory_order_seq_cst) { return v.fetch_add(d, mo); }
};

struct IdSpace {
    int cap = 0; // id capacity = the external id range V
    std::vector<int32_t> ext2int, int2ext; // both cap long; -1: no id / position holds no vertex
    int n_int = 0, n_parked = 0;
    long long revivals = 0;
    CopyableCounter map_gen; // bumped whenever ext2int changes: a device copy made at an older value is stale (the copy may be
                                      // taken by another thread than the one that assigns ids: dppr_slide_concurrent)
    unsigned long long renumber_epoch = 0; // bumped by renumber(): the only operation that changes the internal id of a LIVE vertex
    // pending row moves: position -> position whose rows it will receive (-1: zero rows). Composed on the host while
    // ids are assigned, applied to every state array in one gather + scatter + zero pass (take_moves).
    std::unordered_map<int32_t, int32_t> mv_origin;

    void init_ids(int V) {
        cap = V;
        ext2int.assign((size_t)V, -1);
        int2ext.assign((size_t)V, -1);
        n_int = n_parked = 0;
        revivals = 0;
        map_gen.fetch_add(1, std::memory_order_release);
        mv_origin.clear();
    }

    bool is_parked(int pos) const { return pos >= cap - n_parked; }

    // External ids -> internal ids for a whole array. False (and NOTHING changed) if an id lies outside [0, cap).
    // The lookups run in parallel and only read; ids that are new or parked -- rare -- then go through to_int one by one
    // in array order (a revival may move another parked vertex: parked entries are all resolved in that serial pass).
    // The read-only half of translate(): internal id of every live vertex, -1 for ids without one and for parked ones (whose id
    // changes when they are revived), and the positions of those -1 entries in array order. Touches nothing: may run on other
    // threads while no call that assigns, revives or renumbers is in progress (dppr_hint_next_batch). False if an id is outside
    // [0, cap).
    bool lookup_only(const int32_t *src, size_t n, int32_t *dst, std::vector<uint32_t> &miss) const {
        const int32_t lo_parked = cap - n_parked;
        std::atomic<int> bad{0};
        std::mutex mu;
        std::vector<std::pair<size_t, std::vector<uint32_t>>> parts; // (first index of a piece, its misses)
        parallel_pieces(n, 1 << 16, [&](size_t a, size_t b) {
            bool any_bad = false;
            std::vector<uint32_t> mine;
            for (size_t i = a; i < b; ++i) {
                const int32_t v = src[i];
                if (v < 0 || v >= cap) {
                    any_bad = true;
                    continue;
                }
                const int32_t m = ext2int[(size_t)v];
                dst[i] = (m >= 0 && m < lo_parked) ? m : -1;
                if (dst[i] < 0) mine.push_back((uint32_t)i);
            }
            if (any_bad) bad.store(1, std::memory_order_relaxed);
            if (!mine.empty()) {
                std::lock_guard<std::mutex> lk(mu);
                parts.emplace_back(a, std::move(mine));
            }
        });
        miss.clear();
        std::sort(parts.begin(), parts.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        for (const auto &pt : parts) miss.insert(miss.end(), pt.second.begin(), pt.second.end());
        return !bad.load();
    }

    // The lookups run in parallel and only read; ids that are new or parked -- rare -- then go through to_int one by one in
    // array order (a revival may move another parked vertex: parked entries are all resolved in that serial pass).
    bool translate(const int32_t *src, size_t n, int32_t *dst) {
        if (!lookup_only(src, n, dst, miss_scratch)) return false;
        resolve(src, dst, miss_scratch);
        return true;
    }
    // the entries a lookup left open, in array order (their map entries are cache misses on a large id range: asked for a few ahead)
    void resolve(const int32_t *src, int32_t *dst, const std::vector<uint32_t> &miss) {
        const size_t n = miss.size();
        for (size_t k = 0; k < n; ++k) {
            if (k + 8 < n) __builtin_prefetch(&ext2int[(size_t)src[miss[k + 8]]], 1);
            dst[miss[k]] = to_int(src[miss[k]]);
        }
    }
    std::vector<uint32_t> miss_scratch;

    // external -> internal id, assigning a new one on first sight, bringing a parked vertex back to the live zone
    int to_int(int ext) {
        int32_t &m = ext2int[(size_t)ext];
        if (m < 0) {
            m = n_int++;
            int2ext[(size_t)m] = ext;
            map_gen.fetch_add(1, std::memory_order_release);
        } else if (is_parked(m)) {
            revive(ext);
        }
        return ext2int[(size_t)ext];
    }

    // A parked vertex is needed again: fresh id at the end of the live zone, its rows follow, the lowest parked entry
    // fills the hole so that the parked zone stays dense. Only the maps change here; mv_origin composes the moves (a
    // position may receive rows and give its own away before the next take_moves).
    void revive(int ext) {
        const int q = ext2int[(size_t)ext];
        const int lo = cap - n_parked;
        auto origin = [&](int pos) {
            auto it = mv_origin.find(pos);
            return it == mv_origin.end() ? pos : it->second;
        };
        // fresh <= lo (== lo when every vertex has an id and the zones touch: then the revived vertex takes the slot
        // the parked zone gives up)
        const int fresh = n_int++;
        const int oq = origin(q), olo = origin(lo);
        if (q != lo) { // the lowest parked entry fills the hole
            const int y = int2ext[(size_t)lo];
            mv_origin[q] = olo;
            ext2int[(size_t)y] = q;
            int2ext[(size_t)q] = y;
        }
        mv_origin[lo] = -1; // vacated: zero rows (the live zone grows into it) -- unless it is `fresh` itself, below
        int2ext[(size_t)lo] = -1;
        mv_origin[fresh] = oq;
        ext2int[(size_t)ext] = fresh;
        int2ext[(size_t)fresh] = ext;
        n_parked--;
        revivals++;
        map_gen.fetch_add(1, std::memory_order_release);
    }

    // The pending moves as lists: rows[dst[i]] = OLD rows[src[i]] for all i at once (gather everything, then scatter),
    // then rows[zero[i]] = 0. Clears the pending set.
    void take_moves(std::vector<int32_t> &src, std::vector<int32_t> &dst, std::vector<int32_t> &zero) {
        src.clear();
        dst.clear();
        zero.clear();
        for (const auto &kv : mv_origin) {
            if (kv.second < 0) zero.push_back(kv.first);
            else if (kv.second != kv.first) {
                src.push_back(kv.second);
                dst.push_back(kv.first);
            }
        }
        mv_origin.clear();
    }

    // Renumbering: live[v] (v < n_int) says which live-zone ids keep a place in the live zone; `order` lists exactly
    // those ids in the order they are to be numbered (empty: keep their relative order). Everything else joins the
    // parked zone behind what is parked already. Fills perm (old position -> new position, -1 where no vertex sat)
    // and rewrites the maps. Pending moves must have been taken first.
    void renumber(const std::vector<uint8_t> &live, const std::vector<int32_t> &order, std::vector<int32_t> &perm) {
        const int n_old = n_int, R_old = n_parked;
        int n_live = 0;
        for (int v = 0; v < n_old; ++v) n_live += live[(size_t)v] ? 1 : 0;
        const int R_new = R_old + (n_old - n_live), base = cap - R_new;
        perm.resize((size_t)cap);
        parallel_pieces((size_t)cap, 1 << 18, [&](size_t a, size_t b) { std::fill(perm.begin() + (std::ptrdiff_t)a, perm.begin() + (std::ptrdiff_t)b, -1); });
        if (!order.empty()) {
            parallel_pieces(order.size(), 1 << 16, [&](size_t a, size_t b) {
                for (size_t k = a; k < b; ++k) perm[(size_t)order[k]] = (int32_t)k;
            });
            int np = 0;
            for (int v = 0; v < n_old; ++v)
                if (!live[(size_t)v]) perm[(size_t)v] = base + np++;
        } else {
            int nl = 0, np = 0;
            for (int v = 0; v < n_old; ++v) perm[(size_t)v] = live[(size_t)v] ? nl++ : base + np++;
        }
        const int np_live = n_old - n_live; // parked out of the live zone; the old parked zone follows in its order
        for (int i = 0; i < R_old; ++i) perm[(size_t)(cap - R_old + i)] = base + np_live + i;
        // the maps: only the two old zones hold vertices (the rest of int2ext is -1 already and stays so, except where
        // the parked zone grows into it)
        std::vector<int32_t> old_live(int2ext.begin(), int2ext.begin() + n_old);
        std::vector<int32_t> old_parked(int2ext.begin() + (cap - R_old), int2ext.end());
        std::fill(int2ext.begin(), int2ext.begin() + n_old, -1);
        std::fill(int2ext.begin() + (cap - R_old), int2ext.end(), -1);
        auto place = [&](int v, int ext) { // (perm is a bijection of the occupied positions: every thread writes its own targets)
            const int m = perm[(size_t)v];
            int2ext[(size_t)m] = ext;
            ext2int[(size_t)ext] = m;
        };
        parallel_pieces((size_t)n_old, 1 << 16, [&](size_t a, size_t b) {
            for (size_t v = a; v < b; ++v) place((int)v, old_live[v]);
        });
        for (int i = 0; i < R_old; ++i) place(cap - R_old + i, old_parked[(size_t)i]);
        n_int = n_live;
        n_parked = R_new;
        map_gen.fetch_add(1, std::memory_order_release);
        renumber_epoch++;
    }
};

// The order in which a set of vertices is numbered: `fresh` holds (hash of the external id, tag) pairs and comes back
// sorted. The hash alone gives a pseudo-random order (high-degree vertices show up early in a stream, and packing
// them into the first tiles would serialise the sweeps on a few workgroups).
// Large windows: the x[u] gathers of a sweep are random reads, and what an XCD's 4 MB L2 keeps of
// them saves sectors on the fabric. Gathers follow the in-degree, which is heavily skewed
// (LiveJournal stand-in: the 8 K / 32 K / 524 K vertices of highest in-degree, of 1.18 M, take 35 % /
// 55 % / 95 % of them), so vertices are numbered in BLOCKS of falling in-degree -- the top 8 K first,
// then ranks 8 K..16 K, 16 K..32 K, ... up to 512 K, everybody else last: whatever a vertex's state
// measures (8 bytes for one source, 64 / 128 for a source group), the ids that fit an L2 are the
// hottest ones. Inside a block the order stays hashed, so long rows are still spread over the
// tiles. Measured on that stand-in, single source, two blocks (524 K | rest): 73 -> 67 us per sweep.
// Only for windows beyond a resident launch (> 256 K vertices): below that everything is L2-resident
// anyway, and hot tiles next to each other would unbalance the <= 256 groups of a resident launch
// (configs[1] stand-in: 0.55 -> 0.91 ms per batch). indeg[tag] = in-degree in the window (nullptr: hash only).
constexpr size_t HOT_MIN = 8192, HOT_SET = 524288, HOT_WINDOW_MIN = 262144;
inline uint64_t id_hash(int v) {
    uint64_t z = (uint64_t)v + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline void numbering_order(std::vector<std::pair<uint64_t, int32_t>> &fresh, const int32_t *indeg, bool hot_blocks) {
    if (indeg && fresh.size() > HOT_WINDOW_MIN) {
        std::vector<int32_t> d;
        d.reserve(fresh.size());
        for (auto &kv : fresh) d.push_back(indeg[(size_t)kv.second]);
        std::vector<int32_t> thr; // in-degree of rank 512 K, 256 K, ..., 8 K (non-decreasing)
        size_t cur = d.size();
        for (size_t k = HOT_SET; k >= (hot_blocks ? HOT_MIN : HOT_SET); k >>= 1) {
            if (k >= cur) continue;
            std::nth_element(d.begin(), d.begin() + (std::ptrdiff_t)k, d.begin() + (std::ptrdiff_t)cur,
                             std::greater<int32_t>());
            thr.push_back(d[k]); // vertices with a larger in-degree belong to the first k (at most k of them)
            cur = k;
        }
        for (auto &kv : fresh) {
            const int32_t dg = indeg[(size_t)kv.second];
            uint64_t block = 0; // 0 = hottest
            for (int32_t t : thr) block += dg <= t ? 1u : 0u;
            kv.first = (kv.first >> 5) | (block << 59);
        }
    }
    // sorted by (key, tag): one counting pass on the top 16 / 21 key bits (block + hash bits: near-uniform), then the
    // buckets, a few dozen entries each, one by one -- a plain std::sort of a million pairs is most of what a
    // renumbering slide costs
    if (fresh.size() < (1u << 16)) {
        std::sort(fresh.begin(), fresh.end());
        return;
    }
    const int RB = fresh.size() > (1u << 19) ? 21 : 16; // (hot blocks take the top 5 bits: the cold block needs the rest)
    std::vector<uint32_t> start((size_t)(1 << RB) + 1, 0);
    for (auto &kv : fresh) start[(size_t)(kv.first >> (64 - RB)) + 1]++;
    for (size_t b = 0; b < ((size_t)1 << RB); ++b) start[b + 1] += start[b];
    std::vector<std::pair<uint64_t, int32_t>> out(fresh.size());
    {
        std::vector<uint32_t> pos(start.begin(), start.end() - 1);
        for (auto &kv : fresh) out[pos[(size_t)(kv.first >> (64 - RB))]++] = kv;
    }
    for (size_t b = 0; b < ((size_t)1 << RB); ++b)
        if (start[b + 1] - start[b] > 1) std::sort(out.begin() + start[b], out.begin() + start[b + 1]);
    fresh.swap(out);
}

} // namespace dppr
