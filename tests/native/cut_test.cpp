// CPU test driver for dynamicppr_amd/csrc/dppr_cut.hpp (how the sweeps' work is dealt to workgroups): structure of
// the cuts on random tile weights (empty tiles, hub tiles), the bound of the greedy cut, and the min-max cut against
// an exhaustive dynamic program on small inputs.   cut_test <seed> <cases>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../dynamicppr_amd/csrc/dppr_cut.hpp"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 10) { printf("FAILED %s (line %d): ", #c, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static bool well_formed(const std::vector<int32_t> &cut, int n_tiles, int max_tiles) {
    if (cut.empty() || cut.front() != 0 || cut.back() != n_tiles) return false;
    for (size_t g = 0; g + 1 < cut.size(); ++g)
        if (cut[g + 1] <= cut[g] || cut[g + 1] - cut[g] > max_tiles) return false;
    return true;
}

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int cases = argc > 2 ? atoi(argv[2]) : 2000;
    std::mt19937 rng(seed);
    std::vector<int32_t> cut;
    long long dp_checked = 0;
    for (int c = 0; c < cases && fails == 0; ++c) {
        const bool small = c % 2 == 0;
        const int n = small ? (int)(rng() % 14) : (int)(rng() % 3000);
        const int max_tiles = 1 + (int)(rng() % (small ? 5 : 16));
        const long long tile_w = (long long)(rng() % 3 == 0 ? 0 : rng() % 200);
        std::vector<int32_t> prefix((size_t)n + 1, 0);
        for (int t = 0; t < n; ++t) {
            const unsigned k = rng() % 10;
            const int w = k == 0 ? 0 : k == 1 ? (int)(rng() % 50000) : (int)(rng() % 300); // empty tiles and hub tiles
            prefix[(size_t)t + 1] = prefix[(size_t)t] + w;
        }
        auto weight = [&](int a, int b) { return (long long)(prefix[(size_t)b] - prefix[(size_t)a]) + tile_w * (b - a); };
        // ---- greedy
        const long long want = 1 + (long long)(rng() % (unsigned)(n + 5));
        dppr::cut_greedy(prefix.data(), n, max_tiles, want, tile_w, cut);
        if (n == 0) CHECK(cut.size() == 1 && cut[0] == 0, "empty window: %zu entries", cut.size());
        else {
            CHECK(well_formed(cut, n, max_tiles), "greedy cut malformed (n %d, max %d)", n, max_tiles);
            const long long target = std::max<long long>(1, weight(0, n) / std::max<long long>(1, want));
            for (size_t g = 0; g + 1 < cut.size(); ++g) {
                // a group is closed by the tile that takes it to the target: without its last tile it is below it
                const int a = cut[g], b = cut[g + 1];
                CHECK(b - a == 1 || weight(a, b - 1) < target, "greedy group [%d,%d) weighs %lld before its last tile, target %lld", a, b,
                      weight(a, b - 1), target);
            }
        }
        // ---- min-max
        const int cap = 1 + (int)(rng() % (unsigned)(n + 3));
        const bool ok = dppr::cut_minmax(prefix.data(), n, max_tiles, cap, tile_w, cut);
        const bool feasible = (long long)n <= (long long)cap * max_tiles;
        CHECK(ok == feasible, "min-max cut: returned %d, feasible %d (n %d cap %d max %d)", (int)ok, (int)feasible, n, cap, max_tiles);
        if (ok && n > 0) {
            CHECK(well_formed(cut, n, max_tiles) && (int)cut.size() - 1 <= cap, "min-max cut malformed: %zu groups, cap %d", cut.size() - 1, cap);
            long long worst = 0;
            for (size_t g = 0; g + 1 < cut.size(); ++g) worst = std::max(worst, weight(cut[g], cut[g + 1]));
            if (small) { // best[k][t]: smallest possible largest weight when tiles [0, t) form k groups
                std::vector<std::vector<long long>> best((size_t)cap + 1, std::vector<long long>((size_t)n + 1, LLONG_MAX));
                best[0][0] = 0;
                for (int k = 1; k <= cap; ++k)
                    for (int t = 1; t <= n; ++t)
                        for (int a = std::max(0, t - max_tiles); a < t; ++a)
                            if (best[(size_t)k - 1][(size_t)a] != LLONG_MAX)
                                best[(size_t)k][(size_t)t] = std::min(best[(size_t)k][(size_t)t], std::max(best[(size_t)k - 1][(size_t)a], weight(a, t)));
                long long opt = LLONG_MAX;
                for (int k = 1; k <= cap; ++k) opt = std::min(opt, best[(size_t)k][(size_t)n]);
                CHECK(worst == opt, "min-max cut: largest group %lld, optimum %lld (n %d cap %d max %d)", worst, opt, n, cap, max_tiles);
                ++dp_checked;
            }
        }
    }
    printf("seed %u: %d cases, %lld checked against the exhaustive optimum, %d failures\n", seed, cases, dp_checked, fails);
    return fails ? 1 : 0;
}
