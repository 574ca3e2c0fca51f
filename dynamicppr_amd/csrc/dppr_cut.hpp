// dppr_cut.hpp -- how the sweeps' work is dealt: the window's vertices, in 64-vertex TILES, are cut into GROUPS of
// consecutive tiles, one group per workgroup. Pure host code (the engine calls it from cut_sweep_groups with the tile
// edge prefix it reads back from the device; tests/native/cut_test.cpp drives it on the CPU).
//   prefix[t] = out-edges of the vertices below tile t (prefix[n_tiles] = all of them); a cut is the list of first tiles,
//   cut[0] = 0 < cut[1] < ... < cut[G] = n_tiles, group g = tiles [cut[g], cut[g + 1]).
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace dppr {

// Groups of about equal weight (edges + tile_w per tile), closed as soon as they reach total / want_groups or max_tiles
// tiles: per-iteration sweeps, where many more groups than workgroup slots exist and an even spread is what matters.
inline void cut_greedy(const int32_t *prefix, int n_tiles, int max_tiles, long long want_groups, long long tile_w,
                       std::vector<int32_t> &cut) {
    const long long total_w = (n_tiles ? (long long)prefix[n_tiles] : 0) + tile_w * n_tiles;
    const long long target = std::max<long long>(1, total_w / std::max<long long>(1, want_groups));
    cut.clear();
    cut.push_back(0);
    long long acc = 0;
    int first = 0;
    for (int t = 0; t < n_tiles; ++t) {
        acc += (long long)(prefix[t + 1] - prefix[t]) + tile_w;
        if (acc >= target || t + 1 - first == max_tiles) {
            cut.push_back(t + 1);
            first = t + 1;
            acc = 0;
        }
    }
    if (cut.back() != n_tiles) cut.push_back(n_tiles);
}

// At most `cap` groups of at most max_tiles tiles whose LARGEST weight (edges + tile_w per tile) is as small as
// possible: a resident launch has one workgroup per group and every iteration waits for the slowest one. Bisection on
// the bound, first-fit inside (optimal for consecutive ranges). False (cut unspecified) if no such cut exists.
inline bool cut_minmax(const int32_t *prefix, int n_tiles, int max_tiles, int cap, long long tile_w, std::vector<int32_t> &cut) {
    if (cap <= 0 || max_tiles <= 0 || (long long)n_tiles > (long long)cap * max_tiles) return false;
    auto pack = [&](long long bound) { // first-fit with groups of weight <= bound (a tile alone may exceed it)
        cut.clear();
        cut.push_back(0);
        long long acc = 0;
        int first = 0;
        for (int t = 0; t < n_tiles; ++t) {
            const long long wt = (long long)(prefix[t + 1] - prefix[t]) + tile_w;
            if (t > first && (acc + wt > bound || t - first == max_tiles)) {
                cut.push_back(t);
                first = t;
                acc = 0;
            }
            acc += wt;
        }
        cut.push_back(n_tiles);
        return (int)cut.size() - 1;
    };
    long long lo = 1, hi = (n_tiles ? (long long)prefix[n_tiles] : 0) + tile_w * n_tiles + 1;
    while (lo < hi) { // smallest bound that needs at most cap groups
        const long long mid = (lo + hi) / 2;
        if (pack(mid) <= cap) hi = mid; else lo = mid + 1;
    }
    return pack(lo) <= cap;
}

} // namespace dppr
