// rmat_gen.cpp -- seeded R-MAT stand-in streams in the reference's .bin format, multi-threaded.
//
// The compiled twin of dynamicppr_amd/datagen.py::rmat_stream: byte-identical output for the same
// (scale, edges, seed) -- tests/test_datagen.py compares the two -- but fast enough for the large
// BASELINE.json configs (the reference's own data tools are compiled C++ too,
// encoder/GraphEncoder.h:20-98; file format GraphVec.h:43-70: little-endian int32 V, then
// (int32 v1, int32 v2) per stream edge).
//
// The stream is "the first `edges` candidates with src != dst, in candidate-index order", every
// candidate a pure function of (seed, index) (counter-based splitmix64), so
//   * blocks of candidates are generated independently by all cores and appended in block order;
//   * --limit N writes only the first N edges: the prefix of the full stream, which is all a
//     sliding-window run of W + batches*c edges ever reads.
//
//   rmat_gen --scale S --edges E --seed K --out FILE [--limit N] [--threads T]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <class F>
static void parallel_for(int threads, size_t n, F f) { // f(thread, begin, end)
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        const size_t b = std::min(n, per * (size_t)t), e = std::min(n, b + per);
        if (b < e) th.emplace_back([=] { f(t, b, e); });
    }
    for (auto &x : th) x.join();
}

// perm = stable argsort of splitmix64(i ^ splitmix64(seed ^ 0xA5A5A5A5)): sort (key, i) pairs.
static std::vector<int32_t> permutation(size_t n, uint64_t seed, int threads) {
    struct KV {
        uint64_t k;
        uint32_t i;
    };
    std::vector<KV> a(n), b(n);
    const uint64_t mix = splitmix64(seed ^ 0xA5A5A5A5ull);
    parallel_for(threads, n, [&](int, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) a[i] = KV{splitmix64((uint64_t)i ^ mix), (uint32_t)i};
    });
    auto less = [](const KV &x, const KV &y) { return x.k < y.k || (x.k == y.k && x.i < y.i); };
    // sorted runs, then pairwise merges
    int runs = 1;
    while (runs < threads && (size_t)runs * 2 * 4096 <= n) runs *= 2;
    const size_t per = (n + (size_t)runs - 1) / (size_t)runs;
    parallel_for(runs, (size_t)runs, [&](int, size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) std::sort(a.begin() + (ptrdiff_t)std::min(n, r * per),
                                                   a.begin() + (ptrdiff_t)std::min(n, (r + 1) * per), less);
    });
    KV *src = a.data(), *dst = b.data();
    for (size_t width = per; width < n; width *= 2) {
        const size_t pairs = (n + 2 * width - 1) / (2 * width);
        parallel_for(threads, pairs, [&](int, size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; ++p) {
                const size_t b0 = p * 2 * width, m = std::min(n, b0 + width), e = std::min(n, b0 + 2 * width);
                std::merge(src + b0, src + m, src + m, src + e, dst + b0, less);
            }
        });
        std::swap(src, dst);
    }
    std::vector<int32_t> perm(n);
    parallel_for(threads, n, [&](int, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) perm[i] = (int32_t)src[i].i;
    });
    return perm;
}

int main(int argc, char **argv) {
    long long scale = -1, edges = -1, limit = -1, seed = 1;
    int threads = (int)std::thread::hardware_concurrency();
    std::string out;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "rmat_gen: %s needs a value\n", a.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--scale") scale = atoll(val());
        else if (a == "--edges") edges = atoll(val());
        else if (a == "--seed") seed = atoll(val());
        else if (a == "--limit") limit = atoll(val());
        else if (a == "--threads") threads = atoi(val());
        else if (a == "--out") out = val();
        else {
            fprintf(stderr, "rmat_gen: unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (scale < 1 || scale > 30 || edges < 0 || out.empty()) {
        fprintf(stderr, "usage: rmat_gen --scale S --edges E --seed K --out FILE [--limit N] [--threads T]\n");
        return 2;
    }
    if (limit < 0 || limit > edges) limit = edges;
    threads = std::max(1, std::min(threads, 64));
    const size_t V = (size_t)1 << scale;
    // quadrant thresholds of datagen.py (a, b, c = 0.57, 0.19, 0.19) on 16-bit draws
    const double pa = 0.57, pb = 0.19, pc = 0.19;
    const uint64_t ta = (uint64_t)(pa * 65536), tb = (uint64_t)((pa + pb) * 65536), tc = (uint64_t)((pa + pb + pc) * 65536);
    const std::vector<int32_t> perm = permutation(V, (uint64_t)seed, threads);
    const uint64_t seed_mix = splitmix64((uint64_t)seed);
    const int groups = (int)((scale + 3) / 4);

    FILE *f = fopen(out.c_str(), "wb");
    if (!f) {
        fprintf(stderr, "rmat_gen: cannot open %s\n", out.c_str());
        return 1;
    }
    const int32_t v32 = (int32_t)V;
    fwrite(&v32, sizeof(v32), 1, f);

    const size_t BLOCK = (size_t)1 << 18; // candidates per block
    std::vector<std::vector<int32_t>> buf((size_t)threads);
    for (auto &b : buf) b.reserve(2 * BLOCK);
    uint64_t base = 0;
    long long written = 0;
    while (written < limit) {
        // one round: block t covers candidates [base + t*BLOCK, base + (t+1)*BLOCK)
        parallel_for(threads, (size_t)threads, [&](int, size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; ++t) {
                std::vector<int32_t> &o = buf[t];
                o.clear();
                const uint64_t i0 = base + (uint64_t)t * BLOCK;
                for (uint64_t idx = i0; idx < i0 + BLOCK; ++idx) {
                    uint64_t src = 0, dst = 0;
                    int level = 0;
                    for (int g = 0; g < groups; ++g) {
                        const uint64_t h = splitmix64(idx * 64ull + (uint64_t)g + seed_mix);
                        for (int k = 0; k < 4 && level < scale; ++k, ++level) {
                            const uint64_t u = (h >> (16 * k)) & 0xFFFFull;
                            src = (src << 1) | (uint64_t)(u >= tb);
                            dst = (dst << 1) | (uint64_t)((u >= ta && u < tb) || u >= tc);
                        }
                    }
                    if (src != dst) {
                        o.push_back(perm[src]);
                        o.push_back(perm[dst]);
                    }
                }
            }
        });
        for (int t = 0; t < threads && written < limit; ++t) {
            const long long have = (long long)buf[(size_t)t].size() / 2;
            const long long take = std::min(have, limit - written);
            if (fwrite(buf[(size_t)t].data(), sizeof(int32_t) * 2, (size_t)take, f) != (size_t)take) {
                fprintf(stderr, "rmat_gen: short write to %s\n", out.c_str());
                return 1;
            }
            written += take;
        }
        base += (uint64_t)threads * BLOCK;
    }
    if (fclose(f) != 0) {
        fprintf(stderr, "rmat_gen: close failed on %s\n", out.c_str());
        return 1;
    }
    printf("wrote %s: V=%zu E=%lld (of %lld) threads=%d\n", out.c_str(), V, written, edges, threads);
    return 0;
}
