// CPU test driver for dppr::numbering_order (dynamicppr_amd/csrc/dppr_idspace.hpp): the order in which a window's
// vertices are numbered -- hashed, in blocks of falling in-degree on large windows, sorted by one counting pass plus
// small bucket sorts -- against a plain restatement (full sort of the degrees for the block thresholds, std::sort of
// the keyed pairs).   numbering_test <seed> <n> <hot_blocks>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../dynamicppr_amd/csrc/dppr_idspace.hpp"

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : 300000;
    const bool hot = argc > 3 ? atoi(argv[3]) != 0 : true;
    std::mt19937_64 rng(seed);
    // tags = internal ids 0..n-1 (what a renumbering passes), external ids a random injection, in-degrees heavy-tailed with ties
    std::vector<int32_t> indeg(n);
    std::vector<std::pair<uint64_t, int32_t>> a(n), want(n);
    for (size_t i = 0; i < n; ++i) {
        const double u = (double)(rng() >> 11) / (double)(1ull << 53);
        indeg[i] = (int32_t)(1.0 / (u * u * 4.0 + 1e-4)) % 5000;
        a[i] = {dppr::id_hash((int)(rng() % 2000000000ull)), (int32_t)i};
    }
    want = a;
    // restatement
    if (n > dppr::HOT_WINDOW_MIN) {
        std::vector<int32_t> d(indeg);
        std::sort(d.begin(), d.end(), std::greater<int32_t>());
        std::vector<int32_t> thr;
        for (size_t k = dppr::HOT_SET; k >= (hot ? dppr::HOT_MIN : dppr::HOT_SET); k >>= 1)
            if (k < n) thr.push_back(d[k]);
        for (auto &kv : want) {
            uint64_t block = 0;
            for (int32_t t : thr) block += indeg[(size_t)kv.second] <= t ? 1u : 0u;
            kv.first = (kv.first >> 5) | (block << 59);
        }
    }
    std::sort(want.begin(), want.end());
    dppr::numbering_order(a, indeg.data(), hot);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += a[i] != want[i];
    // the hottest block really holds the vertices of largest in-degree
    long long top = 0, all = 0;
    if (n > dppr::HOT_WINDOW_MIN && hot) {
        for (size_t i = 0; i < n; ++i) all += indeg[i];
        for (size_t i = 0; i < dppr::HOT_MIN && i < n; ++i) top += indeg[(size_t)a[i].second];
    }
    printf("seed %u n %zu hot %d: %zu mismatches; first %zu positions hold %.1f %% of the in-degree\n", seed, n, (int)hot, bad,
           (size_t)dppr::HOT_MIN, all ? 100.0 * (double)top / (double)all : 0.0);
    return bad ? 1 : 0;
}
