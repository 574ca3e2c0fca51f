// CPU test driver for dynamicppr_amd/csrc/dppr_idspace.hpp (the host side of the engine's vertex numbering): random
// sequences of first sightings, revivals of parked vertices, flushes of the pending row moves and renumberings, against
// plain host arrays that play the engine's device state rows. Built by tests/test_idspace.py with the address and
// undefined-behaviour sanitizers.   idspace_test <seed> <cap> <ops>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../dynamicppr_amd/csrc/dppr_idspace.hpp"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 10) { printf("FAILED %s (line %d): ", #c, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

struct Rows { // one state array of w doubles per position
    int w;
    std::vector<double> a;
    Rows(int cap, int w_) : w(w_), a((size_t)cap * (size_t)w_, 0.0) {}
    double *row(int pos) { return a.data() + (size_t)pos * (size_t)w; }
    void apply_moves(const std::vector<int32_t> &src, const std::vector<int32_t> &dst, const std::vector<int32_t> &zero) {
        std::vector<double> tmp(src.size() * (size_t)w); // gather everything first, then scatter, then zero (k_rows_*)
        for (size_t i = 0; i < src.size(); ++i)
            for (int k = 0; k < w; ++k) tmp[i * (size_t)w + (size_t)k] = row(src[i])[k];
        for (size_t i = 0; i < dst.size(); ++i)
            for (int k = 0; k < w; ++k) row(dst[i])[k] = tmp[i * (size_t)w + (size_t)k];
        for (int32_t z : zero)
            for (int k = 0; k < w; ++k) row(z)[k] = 0.0;
    }
    void apply_perm(const std::vector<int32_t> &perm) { // k_permute_rows into a zeroed array
        std::vector<double> out(a.size(), 0.0);
        for (size_t v = 0; v < perm.size(); ++v)
            if (perm[v] >= 0)
                for (int k = 0; k < w; ++k) out[(size_t)perm[v] * (size_t)w + (size_t)k] = row((int)v)[k];
        a.swap(out);
    }
};

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int cap = argc > 2 ? atoi(argv[2]) : 40;
    const int ops = argc > 3 ? atoi(argv[3]) : 20000;
    std::mt19937 rng(seed);
    dppr::IdSpace ids;
    ids.init_ids(cap);
    Rows p(cap, 1), r(cap, 3);
    std::vector<char> has_state((size_t)cap, 0); // the vertex has been given rows (a source / an endpoint of some batch)
    auto tag = [](int ext, int k) { return 1000.0 * (ext + 1) + k; };
    std::vector<int32_t> src, dst, zero, perm;

    auto flush = [&]() {
        ids.take_moves(src, dst, zero);
        p.apply_moves(src, dst, zero);
        r.apply_moves(src, dst, zero);
    };
    auto verify = [&](const char *when) {
        CHECK(ids.n_int >= 0 && ids.n_parked >= 0 && ids.n_int + ids.n_parked <= cap, "%s: zones %d + %d > %d", when, ids.n_int, ids.n_parked, cap);
        int with_id = 0;
        for (int x = 0; x < cap; ++x) {
            const int m = ids.ext2int[(size_t)x];
            if (m < 0) continue;
            ++with_id;
            CHECK(ids.int2ext[(size_t)m] == x, "%s: maps disagree at ext %d -> %d -> %d", when, x, m, ids.int2ext[(size_t)m]);
            CHECK(m < ids.n_int || m >= cap - ids.n_parked, "%s: ext %d sits in the gap (%d)", when, x, m);
            const double want0 = has_state[(size_t)x] ? tag(x, 0) : 0.0;
            CHECK(p.row(m)[0] == want0, "%s: p row of ext %d at %d holds %g, want %g", when, x, m, p.row(m)[0], want0);
            for (int k = 0; k < 3; ++k) {
                const double want = has_state[(size_t)x] ? tag(x, k) + 0.5 : 0.0;
                CHECK(r.row(m)[k] == want, "%s: r row of ext %d at %d lane %d holds %g, want %g", when, x, m, k, r.row(m)[k], want);
            }
        }
        CHECK(with_id == ids.n_int + ids.n_parked, "%s: %d vertices with an id, zones hold %d", when, with_id, ids.n_int + ids.n_parked);
        for (int pos = 0; pos < cap; ++pos) {
            const bool in_zone = pos < ids.n_int || pos >= cap - ids.n_parked;
            CHECK((ids.int2ext[(size_t)pos] >= 0) == in_zone, "%s: position %d: occupant %d, in a zone: %d", when, pos, ids.int2ext[(size_t)pos], (int)in_zone);
            if (!in_zone) CHECK(p.row(pos)[0] == 0.0 && r.row(pos)[0] == 0.0 && r.row(pos)[2] == 0.0, "%s: gap position %d is not zero", when, pos);
        }
    };

    long long n_revive = 0, n_renumber = 0, touched = 0;
    for (int op = 0; op < ops && fails == 0; ++op) {
        const unsigned what = rng() % 100;
        if (what < 70) { // a batch: some vertices are named (first sightings and revivals mixed), one flush, then they get state
            const int n = 1 + (int)(rng() % 6);
            std::vector<int> named;
            const long long before = ids.revivals;
            if (rng() & 1) { // the whole array at once (IdSpace::translate: parallel lookups, the new / parked ids serially) ...
                std::vector<int32_t> in, out, want;
                for (int i = 0; i < n + 8; ++i) in.push_back((int32_t)(rng() % (unsigned)cap));
                dppr::IdSpace ref = ids; // ... must equal one to_int after the other
                for (int32_t x : in) (void)ref.to_int(x);
                for (int32_t x : in) want.push_back(ref.ext2int[(size_t)x]);
                out.assign(in.size(), -7);
                CHECK(ids.translate(in.data(), in.size(), out.data()), "translate refused ids in range");
                for (size_t i = 0; i < in.size(); ++i) { // (an id named early may have moved when a later one was revived: compare at the end)
                    CHECK(ids.ext2int[(size_t)in[i]] == want[i], "translate: ext %d -> %d, one by one %d", in[i], ids.ext2int[(size_t)in[i]], want[i]);
                    CHECK(out[i] >= 0 && out[i] < ids.n_int, "translate gave %d for ext %d (n_int %d)", out[i], in[i], ids.n_int);
                }
                CHECK(ids.ext2int == ref.ext2int && ids.int2ext == ref.int2ext && ids.n_int == ref.n_int && ids.n_parked == ref.n_parked &&
                      ids.mv_origin == ref.mv_origin, "translate and to_int leave different maps");
                for (int32_t x : in) named.push_back(x);
                // an id out of range: refused, nothing changed
                std::vector<int32_t> bad = in;
                bad.push_back(rng() & 1 ? cap : -1);
                const dppr::IdSpace snap = ids;
                std::vector<int32_t> o2(bad.size());
                CHECK(!ids.translate(bad.data(), bad.size(), o2.data()), "translate accepted an id out of range");
                CHECK(ids.ext2int == snap.ext2int && ids.n_int == snap.n_int && ids.n_parked == snap.n_parked && ids.revivals == snap.revivals, "a refused translate changed the maps");
            }
            if (rng() % 3 == 0) { // the lookahead form (dppr_hint_next_batch): read-only lookups of an array, OTHER ids named in between
                                  // (the batch before it), then only the entries the lookup left open are resolved, in array order
                std::vector<int32_t> in, between, out;
                std::vector<uint32_t> miss;
                for (int i = 0; i < n + 8; ++i) in.push_back((int32_t)(rng() % (unsigned)cap));
                for (int i = 0; i < n; ++i) between.push_back((int32_t)(rng() % (unsigned)cap));
                out.assign(in.size(), -7);
                const dppr::IdSpace before_lookup = ids;
                const unsigned long long epoch = ids.renumber_epoch;
                CHECK(static_cast<const dppr::IdSpace &>(ids).lookup_only(in.data(), in.size(), out.data(), miss), "lookup_only refused ids in range");
                CHECK(ids.ext2int == before_lookup.ext2int && ids.int2ext == before_lookup.int2ext && ids.n_int == before_lookup.n_int &&
                      ids.n_parked == before_lookup.n_parked && ids.mv_origin == before_lookup.mv_origin, "lookup_only changed the maps");
                size_t n_open = 0;
                for (size_t i = 0; i < in.size(); ++i) n_open += out[i] < 0 ? 1 : 0;
                CHECK(n_open == miss.size(), "lookup_only: %zu open entries, %zu listed", n_open, miss.size());
                for (size_t i = 0; i + 1 < miss.size(); ++i) CHECK(miss[i] < miss[i + 1], "lookup_only: the open entries are not in array order");
                dppr::IdSpace ref = ids;
                for (int32_t x : between) (void)ref.to_int(x);
                for (int32_t x : in) (void)ref.to_int(x);
                for (int32_t x : between) (void)ids.to_int(x);
                CHECK(ids.renumber_epoch == epoch, "naming ids bumped the renumbering epoch");
                for (const uint32_t i : miss) out[i] = ids.to_int(in[i]);
                CHECK(ids.ext2int == ref.ext2int && ids.int2ext == ref.int2ext && ids.n_int == ref.n_int && ids.n_parked == ref.n_parked &&
                      ids.mv_origin == ref.mv_origin, "lookahead + resolve and to_int one by one leave different maps");
                for (size_t i = 0; i < in.size(); ++i) // (a LIVE vertex never moves outside a renumbering: every entry is final)
                    CHECK(out[i] == ids.ext2int[(size_t)in[i]] && out[i] < ids.n_int, "lookahead: ext %d -> %d, the map says %d", in[i], out[i], ids.ext2int[(size_t)in[i]]);
                for (int32_t x : in) named.push_back(x);
                for (int32_t x : between) named.push_back(x);
            }
            for (int i = 0; i < n; ++i) {
                const int x = (int)(rng() % (unsigned)cap);
                const int m = ids.to_int(x);
                CHECK(m >= 0 && m < ids.n_int, "to_int(%d) = %d is not a live id (n_int %d)", x, m, ids.n_int);
                named.push_back(x);
            }
            n_revive += ids.revivals - before;
            for (int x : named) CHECK(ids.ext2int[(size_t)x] < ids.n_int, "ext %d named earlier in the batch is no longer live", x);
            flush();
            for (int x : named) {
                const int m = ids.ext2int[(size_t)x];
                if (!has_state[(size_t)x]) { // a fresh vertex finds zero rows
                    CHECK(p.row(m)[0] == 0.0 && r.row(m)[1] == 0.0, "fresh ext %d at %d does not find zero rows", x, m);
                    has_state[(size_t)x] = 1;
                    p.row(m)[0] = tag(x, 0);
                    for (int k = 0; k < 3; ++k) r.row(m)[k] = tag(x, k) + 0.5;
                }
            }
            verify("batch");
        } else if (what < 85) { // a renumbering with a random live set, fresh order or old order
            flush();
            std::vector<uint8_t> live((size_t)std::max(ids.n_int, 1), 0);
            std::vector<int32_t> order;
            for (int v = 0; v < ids.n_int; ++v) live[(size_t)v] = (rng() % 100) < 60;
            if (rng() & 1) {
                for (int v = 0; v < ids.n_int; ++v)
                    if (live[(size_t)v]) order.push_back(v);
                std::shuffle(order.begin(), order.end(), rng);
            }
            ids.renumber(live, order, perm);
            p.apply_perm(perm);
            r.apply_perm(perm);
            ++n_renumber;
            verify("renumbering");
        } else { // names without a flush in between: the moves compose
            for (int i = 0; i < 3; ++i) (void)ids.to_int((int)(rng() % (unsigned)cap));
            // (state for first sightings is only given after a flush, in a later batch)
        }
        touched += ids.n_int + ids.n_parked == cap;
    }
    flush();
    verify("end");
    printf("seed %u cap %d: %d ops, %lld renumberings, %lld revivals, zones filled the capacity after %lld ops, %d failures\n", seed, cap, ops,
           n_renumber, ids.revivals, touched, fails);
    (void)n_revive;
    return fails ? 1 : 0;
}
