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
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

namespace dppr {

struct IdSpace {
    int cap = 0; // id capacity = the external id range V
    std::vector<int32_t> ext2int, int2ext; // both cap long; -1: no id / position holds no vertex
    int n_int = 0, n_parked = 0;
    long long revivals = 0;
    bool map_dirty = true; // a device copy of ext2int is stale
    // pending row moves: position -> position whose rows it will receive (-1: zero rows). Composed on the host while
    // ids are assigned, applied to every state array in one gather + scatter + zero pass (take_moves).
    std::unordered_map<int32_t, int32_t> mv_origin;

    void init_ids(int V) {
        cap = V;
        ext2int.assign((size_t)V, -1);
        int2ext.assign((size_t)V, -1);
        n_int = n_parked = 0;
        revivals = 0;
        map_dirty = true;
        mv_origin.clear();
    }

    bool is_parked(int pos) const { return pos >= cap - n_parked; }

    // external -> internal id, assigning a new one on first sight, bringing a parked vertex back to the live zone
    int to_int(int ext) {
        int32_t &m = ext2int[(size_t)ext];
        if (m < 0) {
            m = n_int++;
            int2ext[(size_t)m] = ext;
            map_dirty = true;
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
        map_dirty = true;
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
        perm.assign((size_t)cap, -1);
        int nl = 0, np = 0;
        if (!order.empty()) {
            for (int32_t v : order) perm[(size_t)v] = nl++;
            for (int v = 0; v < n_old; ++v)
                if (!live[(size_t)v]) perm[(size_t)v] = base + np++;
        } else {
            for (int v = 0; v < n_old; ++v) perm[(size_t)v] = live[(size_t)v] ? nl++ : base + np++;
        }
        for (int v = cap - R_old; v < cap; ++v) perm[(size_t)v] = base + np++;
        std::vector<int32_t> new_i2e((size_t)cap, -1);
        for (int v = 0; v < cap; ++v) {
            const int m = perm[(size_t)v];
            if (m < 0) continue;
            const int ext = int2ext[(size_t)v];
            new_i2e[(size_t)m] = ext;
            ext2int[(size_t)ext] = m;
        }
        int2ext.swap(new_i2e);
        n_int = n_live;
        n_parked = R_new;
        map_dirty = true;
    }
};

} // namespace dppr
