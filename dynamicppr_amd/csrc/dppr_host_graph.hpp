// dppr_host_graph.hpp -- host side, part 2 of 4: the BUILDER. Everything the reference does in its untimed region
// (gpu/SlidingGraphBuilder.cuh:117-221, gpu/PPRGPU.cuh:114-135): the id space and its renumbering, the sorted key arrays and their
// one-pass merge (f1), hub directory + in-CSR + out-CSR of an epoch, the sweep-group cuts with their row / slot tables, the
// binned-sweep tables. All of it runs on the engine's builder stream `bs` with the builder's own scratch, so that a slide may run
// beside a solver call on an older epoch (dppr_slide_concurrent).
#pragma once

namespace {

int cut_sweep_groups(dppr_engine *e, Epoch &ep);
struct BinBatch { // out-orientation keys (row << bits | head) of a slide's retired and inserted edges, unsorted: what the binned tables are patched with
    uint64_t *del = nullptr, *ins = nullptr;
    int nd = 0, ni = 0;
};
int build_bins(dppr_engine *e, Epoch &ep, const BinBatch *batch = nullptr);
bool resident_arena(dppr_engine *e, const Epoch &ep);
int res_record_ranges(dppr_engine *e, Epoch &ep);

// A vertex that got its internal id AFTER an epoch was built (a source outside the window, a
// dppr_write to an unseen vertex) is not covered by that epoch's sweep groups: re-cut them.
int recut_stale_groups(dppr_engine *e) {
    for (auto &ep : e->epochs)
        if (ep.id >= 0 && (ep.grp_n_int != e->n_int || (e->any_groups && ep.n_ggroups == 0) ||
                           (e->wide_groups && ep.ggrp_max_tiles > 512 / WAVE))) {
            int rc = cut_sweep_groups(e, ep);
            if (rc) return rc;
            // (binned tables stay valid: k_bin_reduce takes the ids beyond bin_n_int, which have no edge in this epoch, on the side)
        }
    return DPPR_OK;
}

int sync_map(dppr_engine *e) {
    const unsigned gen = e->map_gen.load(std::memory_order_acquire); // (read BEFORE the copy: an id assigned during it leaves the copy stale)
    if (gen == e->map_gen_on_device) return DPPR_OK;
    HIP_TRY(hipMemcpyAsync(e->d_ext2int, e->ext2int.data(), sizeof(int) * (size_t)e->V, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->map_gen_on_device = gen;
    return DPPR_OK;
}

// Apply the row moves that revivals queued (to_int): every solver state's p / r rows, in one gather + scatter + zero
// per array. States that lag behind the newest epoch may be moved too: a parked row is not touched by any epoch,
// and the fresh id lies beyond the ids every older epoch sweeps.
int flush_moves(dppr_engine *e) {
    if (e->mv_origin.empty()) return DPPR_OK;
    e->take_moves(e->mv_src, e->mv_dst, e->mv_zero);
    const int n = (int)e->mv_src.size(), nz = (int)e->mv_zero.size();
    if (e->slots.empty() && e->groups.empty()) return DPPR_OK;
    const size_t need_idx = (size_t)2 * n + nz + 1;
    if (need_idx > e->mv_idx_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(e->mv_idx);
        e->mv_idx = nullptr;
        e->mv_idx_cap = 0;
        HIP_TRY(hipMalloc((void **)&e->mv_idx, sizeof(int) * (need_idx * 2 + 1024)));
        e->mv_idx_cap = need_idx * 2 + 1024;
    }
    const size_t need_tmp = (size_t)std::max(n, 1) * GS_MAX;
    if (need_tmp > e->mv_tmp_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(e->mv_tmp);
        e->mv_tmp = nullptr;
        e->mv_tmp_cap = 0;
        HIP_TRY(hipMalloc((void **)&e->mv_tmp, sizeof(double) * (need_tmp * 2 + 4096)));
        e->mv_tmp_cap = need_tmp * 2 + 4096;
    }
    int *d_src = e->mv_idx, *d_dst = e->mv_idx + n, *d_zero = e->mv_idx + 2 * n;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(d_src, e->mv_src.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
        HIP_TRY(hipMemcpyAsync(d_dst, e->mv_dst.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
    }
    if (nz > 0) HIP_TRY(hipMemcpyAsync(d_zero, e->mv_zero.data(), sizeof(int) * (size_t)nz, hipMemcpyHostToDevice, e->bs));
    auto move = [&](double *a, int w) -> int {
        if (n > 0) {
            hipLaunchKernelGGL(k_rows_gather<double>, dim3(grid_for((int64_t)n * w)), dim3(BLOCK), 0, e->bs, e->mv_tmp, a, d_src, n, w);
            hipLaunchKernelGGL(k_rows_scatter<double>, dim3(grid_for((int64_t)n * w)), dim3(BLOCK), 0, e->bs, a, e->mv_tmp, d_dst, n, w);
        }
        if (nz > 0)
            hipLaunchKernelGGL(k_rows_zero<double>, dim3(grid_for((int64_t)nz * w)), dim3(BLOCK), 0, e->bs, a, d_zero, nz, w);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    };
    for (auto &s : e->slots) {
        if (int rc = move(s.p, 1)) return rc;
        if (int rc = move(s.r, 1)) return rc;
    }
    for (auto &g : e->groups) {
        if (int rc = move(g.p, g.gw)) return rc;
        if (int rc = move(g.r, g.gw)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->bs)); // the host index vectors are reused
    return DPPR_OK;
}

// Parked rows were inert under the eps they were parked with; a solve with a smaller one pushes them first.
int settle_parked(dppr_engine *e, double *p, double *r, int w, double eps, double *park_eps, dppr_stats_t *st) {
    if (!(eps < *park_eps)) return DPPR_OK; // (the steady state of a stream: the parked zone is not even looked at)
    std::lock_guard<std::mutex> map_lk(e->map_mu); // the parked zone's extent changes under a concurrent slide's revivals
    if (e->n_parked == 0) return DPPR_OK;
    const size_t base = (size_t)(e->V - e->n_parked) * (size_t)w;
    const int64_t n = (int64_t)e->n_parked * w;
    int *cnt = e->hub_hist + 41; // scratch word
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL(k_settle_parked, dim3(grid_for(n)), dim3(BLOCK), 0, e->stream, p + base, r + base, n, eps, cnt);
    HIP_TRY(hipGetLastError());
    int pushed = 0;
    HIP_TRY(hipMemcpyAsync(&pushed, cnt, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (st) st->sum_F += pushed;
    *park_eps = eps;
    return DPPR_OK;
}

// n_int at which a slide looks at the live count again: grown by growth_pct % (64-bit: n * pct overflows an int at friendster scale)
inline int renumber_threshold(int n, int growth_pct) {
    const long long t = (long long)n + std::max<long long>((long long)n * growth_pct / 100, 1);
    return (int)std::min<long long>(t, 0x7fffffff);
}

// The live vertices of a renumbering in the order they are to be numbered: hashed, in blocks of falling in-degree on
// large windows -- what dppr::numbering_order does for dppr_load_window on the host, here as device keys and one
// radix sort (a host sort of a million pairs was most of a renumbering slide; of thirty million it would stall the
// stream). live[v] for the old ids v < n_old; order receives the n_live ids.
int device_numbering_order(dppr_engine *e, const std::vector<uint8_t> &live, int n_old, int n_live, std::vector<int32_t> &order) {
    order.clear();
    if (n_live <= 0) return DPPR_OK;
    const size_t n = (size_t)n_old;
    uint8_t *d_live = nullptr;
    int *d_i2e = nullptr, *d_vals = nullptr, *d_vals2 = nullptr, *d_deg2 = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_live); (void)hipFree(d_i2e); (void)hipFree(d_vals); (void)hipFree(d_vals2); (void)hipFree(d_deg2);
        (void)hipFree(d_keys); (void)hipFree(d_keys2); (void)hipFree(d_tmp);
    };
#define NO_TRY(call)                                                          \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) {                                               \
            cleanup();                                                        \
            e->err = std::string("renumbering order: ") + hipGetErrorString(_e); \
            return _e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP; \
        }                                                                     \
    } while (0)
    NO_TRY(hipMalloc((void **)&d_live, n));
    NO_TRY(hipMalloc((void **)&d_i2e, sizeof(int) * n));
    NO_TRY(hipMalloc((void **)&d_vals, sizeof(int) * n));
    NO_TRY(hipMalloc((void **)&d_vals2, sizeof(int) * n));
    NO_TRY(hipMalloc((void **)&d_keys, sizeof(uint64_t) * n));
    NO_TRY(hipMalloc((void **)&d_keys2, sizeof(uint64_t) * n));
    NO_TRY(hipMemcpyAsync(d_live, live.data(), n, hipMemcpyHostToDevice, e->bs));
    NO_TRY(hipMemcpyAsync(d_i2e, e->int2ext.data(), sizeof(int) * n, hipMemcpyHostToDevice, e->bs));
    HotThresholds ht{};
    int *d_deg = e->hub_slot_of; // (scratch of the CSR build, V ints)
    if ((size_t)n_live > HOT_WINDOW_MIN) {
        NO_TRY(hipMalloc((void **)&d_deg2, sizeof(int) * n));
        NO_TRY(hipMemsetAsync(d_deg, 0, sizeof(int) * n, e->bs));
        hipLaunchKernelGGL(k_in_degree, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, e->W, e->directed, d_deg);
        hipLaunchKernelGGL(k_live_degree, dim3(grid_for(n_old)), dim3(BLOCK), 0, e->bs, d_live, d_deg, n_old, d_vals);
        size_t tb = 0;
        NO_TRY(rocprim::radix_sort_keys_desc(nullptr, tb, d_vals, d_deg2, n, 0u, 32u, e->bs));
        NO_TRY(hipMalloc(&d_tmp, tb));
        NO_TRY(rocprim::radix_sort_keys_desc(d_tmp, tb, d_vals, d_deg2, n, 0u, 32u, e->bs));
        // in-degree of rank k among the live vertices (the non-live ones sorted last as -1)
        for (size_t k = HOT_SET; k >= (e->hot_blocks ? HOT_MIN : HOT_SET); k >>= 1) {
            if (k >= (size_t)n_live) continue;
            NO_TRY(hipMemcpyAsync(&ht.thr[ht.n], d_deg2 + k, sizeof(int), hipMemcpyDeviceToHost, e->bs));
            ht.n++;
        }
        NO_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(d_tmp);
        d_tmp = nullptr;
    }
    hipLaunchKernelGGL(k_number_keys, dim3(grid_for(n_old)), dim3(BLOCK), 0, e->bs, d_live, d_i2e, d_deg, ht, n_old, d_keys, d_vals);
    {
        size_t tb = 0;
        NO_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0u, 64u, e->bs));
        NO_TRY(hipMalloc(&d_tmp, tb));
        NO_TRY(rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0u, 64u, e->bs));
    }
    order.resize((size_t)n_live);
    NO_TRY(hipMemcpyAsync(order.data(), d_vals2, sizeof(int) * (size_t)n_live, hipMemcpyDeviceToHost, e->bs));
    NO_TRY(hipStreamSynchronize(e->bs));
    NO_TRY(hipGetLastError());
#undef NO_TRY
    cleanup();
    return DPPR_OK;
}

// Renumber the internal ids (dppr_builder.hpp has the why). Called by dppr_slide before anything of the new batch is
// looked at; does nothing unless every solver state is converged on the newest epoch (older epochs and their CSRs
// are in the old numbering: nothing may still need them) and enough ids would be parked. On success every epoch is
// invalidated, the ring, the out-degrees, the id maps, the staged batch and every state row are in the new
// numbering, and *did tells the caller to sort the whole window for the epoch it is about to build.
int compact_ids(dppr_engine *e, bool *did) {
    *did = false;
    pre_join(e);
    if (e->build_concurrent) return DPPR_OK; // (a renumbering moves every state row: only an exclusive slide may; dppr_renumbering_due tells the host)
    if (!e->renumber_on || e->W == 0 || e->n_int < e->renumber_next) return DPPR_OK;
    if (e->slots.empty() && e->groups.empty()) return DPPR_OK;
    for (const auto &s : e->slots)
        if (!s.converged || s.last_epoch != e->newest) return DPPR_OK;
    for (const auto &g : e->groups)
        if (!g.converged || g.last_epoch != e->newest) return DPPR_OK;
    if (int rc = flush_moves(e)) return rc;
    const int V = e->V, n_old = e->n_int;
    static const bool trace = getenv("DPPR_RENUMBER_TRACE") != nullptr; // (diagnostic: where a renumbering's time goes)
    timespec t_mark;
    clock_gettime(CLOCK_MONOTONIC, &t_mark);
    auto mark = [&](const char *what) {
        if (!trace) return;
        timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        fprintf(stderr, "[renumber] %-28s %8.2f ms\n", what, (now.tv_sec - t_mark.tv_sec) * 1e3 + (now.tv_nsec - t_mark.tv_nsec) * 1e-6);
        t_mark = now;
    };
    // which ids have an edge in the window
    uint8_t *d_live = nullptr;
    HIP_TRY(hipMalloc((void **)&d_live, (size_t)std::max(n_old, 1)));
    HIP_TRY(hipMemsetAsync(d_live, 0, (size_t)std::max(n_old, 1), e->bs));
    hipLaunchKernelGGL(k_mark_live, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, e->W, d_live);
    std::vector<uint8_t> live((size_t)std::max(n_old, 1));
    hipError_t herr = hipMemcpyAsync(live.data(), d_live, (size_t)n_old, hipMemcpyDeviceToHost, e->bs);
    if (herr == hipSuccess) herr = hipStreamSynchronize(e->bs);
    (void)hipFree(d_live);
    HIP_TRY(herr);
    if (e->batch_staged) { // the staged records are in internal ids already: their vertices stay where they are
        for (int v : e->st_b1) live[(size_t)v] = 1;
        for (int v : e->st_b2) live[(size_t)v] = 1;
    }
    for (const auto &s : e->slots) live[(size_t)s.source] = 1;
    for (const auto &g : e->groups)
        for (int k = 0; k < g.n; ++k) live[(size_t)g.src.s[k]] = 1;
    int n_live = 0;
    for (int v = 0; v < n_old; ++v) n_live += live[(size_t)v];
    mark("live flags");
    const int to_park = n_old - n_live;
    if (to_park < e->renumber_min_parked || (long long)to_park * 200 < (long long)n_live * e->renumber_growth_pct) {
        e->renumber_next = renumber_threshold(n_old, 12); // look again after some more growth
        return DPPR_OK;
    }
    // old position -> new position. Live vertices are numbered afresh the way dppr_load_window numbers a window
    // (hashed, hot blocks on large windows): arrivals are appended in arrival order between two renumberings, and a
    // tail of low-degree late-comers next to each other unbalances the sweep groups (configs[1] stand-in in step,
    // survivors kept in their old order instead: 0.52 ms per batch at the start, 0.61 after 400 batches of the
    // same work). The order is computed on the device (hash + in-degree blocks as keys, one radix sort).
    // Renumbering is an optimisation: whatever can fail for lack of memory is obtained BEFORE anything is changed, and
    // then the slide simply goes on in the old numbering (and looks again after some more growth).
    int *d_perm = nullptr;
    double *tmp = nullptr;
    int maxw = 1;
    for (const auto &g : e->groups) maxw = std::max(maxw, g.gw);
    auto cleanup = [&]() {
        (void)hipFree(d_perm);
        (void)hipFree(tmp);
    };
    if (hipMalloc((void **)&d_perm, sizeof(int) * (size_t)V) != hipSuccess ||
        hipMalloc((void **)&tmp, sizeof(double) * (size_t)V * (size_t)maxw) != hipSuccess) {
        (void)hipGetLastError();
        cleanup();
        e->renumber_next = renumber_threshold(n_old, 12);
        return DPPR_OK;
    }
    mark("scratch allocation");
    std::vector<int32_t> perm, order;
    if (int rc = device_numbering_order(e, live, n_old, n_live, order)) {
        cleanup();
        if (rc != DPPR_ERR_NOMEM) return rc;
        e->renumber_next = renumber_threshold(n_old, 12);
        return DPPR_OK;
    }
    mark("numbering order (device)");
    e->renumber(live, order, perm); // (IdSpace: perm, the maps, n_int, n_parked)
    mark("host maps");
    // From here on the host maps are in the NEW numbering: a failure below leaves ring, degrees and state rows part old,
    // part new -- the engine refuses all further work (`broken`).
#define RN_TRY(call)                                                     \
    do {                                                                 \
        hipError_t _e = (call);                                          \
        if (_e != hipSuccess) {                                          \
            cleanup();                                                   \
            e->broken = true;                                            \
            for (auto &ep : e->epochs) ep.id = -1;                       \
            e->err = std::string("renumbering failed half way (") + hipGetErrorString(_e) + "): the engine is unusable, destroy it"; \
            return DPPR_ERR_HIP;                                         \
        }                                                                \
    } while (0)
    RN_TRY(hipMemcpyAsync(d_perm, perm.data(), sizeof(int) * (size_t)V, hipMemcpyHostToDevice, e->bs));
    hipLaunchKernelGGL(k_remap_ids, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w1, e->W, d_perm);
    hipLaunchKernelGGL(k_remap_ids, dim3(grid_for(e->W)), dim3(BLOCK), 0, e->bs, e->w2, e->W, d_perm);
    { // out-degrees (ints) through the row scratch
        int *itmp = reinterpret_cast<int *>(tmp);
        RN_TRY(hipMemsetAsync(itmp, 0, sizeof(int) * (size_t)V, e->bs));
        hipLaunchKernelGGL(k_permute_rows<int>, dim3(grid_for(V)), dim3(BLOCK), 0, e->bs, itmp, e->outdeg, d_perm, V, 1);
        RN_TRY(hipMemcpyAsync(e->outdeg, itmp, sizeof(int) * (size_t)V, hipMemcpyDeviceToDevice, e->bs));
    }
    auto permute = [&](double *&a, int w) -> hipError_t { // a's rows in the new order; the old array becomes the scratch
        hipError_t r = hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)V * (size_t)w, e->bs);
        if (r != hipSuccess) return r;
        hipLaunchKernelGGL(k_permute_rows<double>, dim3(grid_for((int64_t)V * w)), dim3(BLOCK), 0, e->bs, tmp, a, d_perm, V, w);
        r = hipMemcpyAsync(a, tmp, sizeof(double) * (size_t)V * (size_t)w, hipMemcpyDeviceToDevice, e->bs);
        return r != hipSuccess ? r : hipGetLastError();
    };
    for (auto &s : e->slots) {
        RN_TRY(permute(s.p, 1));
        RN_TRY(permute(s.r, 1));
        // between two loops the snapshot vectors are all zero and the lists empty: nothing to carry over
        RN_TRY(hipMemsetAsync(s.x, 0, sizeof(double) * (size_t)V, e->bs));
        RN_TRY(hipMemsetAsync(s.x2, 0, sizeof(double) * (size_t)V, e->bs));
        RN_TRY(hipMemsetAsync(s.act[0], 0, s.act_bytes, e->bs));
        RN_TRY(hipMemsetAsync(s.act[1], 0, s.act_bytes, e->bs));
        s.source = perm[(size_t)s.source];
        s.seed_lists_valid = false;
        s.phase0_done = false;
        s.park_eps = std::max(s.park_eps, s.conv_eps);
    }
    for (auto &g : e->groups) {
        RN_TRY(permute(g.p, g.gw));
        RN_TRY(permute(g.r, g.gw));
        // (snapshot rows mean something only where an activity bit is set, and between loops none is)
        RN_TRY(hipMemsetAsync(g.act[0], 0, g.act_bytes, e->bs));
        RN_TRY(hipMemsetAsync(g.act[1], 0, g.act_bytes, e->bs));
        for (int k = 0; k < g.n; ++k) g.src.s[k] = perm[(size_t)g.src.s[k]];
        g.park_eps = std::max(g.park_eps, g.conv_eps);
    }
    RN_TRY(hipStreamSynchronize(e->bs));
#undef RN_TRY
    mark("ring, degrees, state rows");
    cleanup();
    mark("scratch release");
    if (e->batch_staged) {
        for (auto &v : e->st_b1) v = perm[(size_t)v];
        for (auto &v : e->st_b2) v = perm[(size_t)v];
    }
    for (auto &ep : e->epochs) ep.id = -1; // CSRs, group tables and batch records of the old numbering
    e->renumber_next = renumber_threshold(n_live, e->renumber_growth_pct);
    e->renumberings++;
    *did = true;
    return DPPR_OK;
}

// A state that has seen the batches up to epoch `last` can only take epoch last + 1 next: skipping or
// replaying one would leave the batch delta of a whole epoch out of (or twice in) p / r and still
// "converge" (n_epochs > 1 keeps many epochs resident, so nothing else would notice).
bool epoch_in_sequence(int last, int id) { return last < 0 || id == last + 1; }

Epoch *find_epoch(dppr_engine *e, int epoch) {
    if (e->newest < 0) return nullptr;
    if (epoch < 0) epoch = e->newest;
    Epoch &ep = e->epochs[epoch % e->n_epochs];
    return ep.id == epoch ? &ep : nullptr;
}

// Sort the whole window into the persistent key arrays (load_window; also the non-incremental
// slide = what gpu/SlidingGraphBuilder.cuh:203-221 does every batch).
int sort_window_full(dppr_engine *e) {
    const int W = e->W, Ed = e->Ed;
    if (W == 0) return DPPR_OK;
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, W, e->directed, e->bits,
                       e->keys_a);
    HIP_TRY(hipGetLastError());
    size_t tmp = e->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, e->keys_a, e->in_sorted, (size_t)Ed, 0u, (unsigned)(2 * e->bits),
                                     e->bs));
    if (e->directed) { // undirected: the out-orientation is the same multiset, out_sorted aliases in_sorted
        hipLaunchKernelGGL(k_make_out_keys, dim3(grid_for(W)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, W, e->directed,
                           e->bits, e->keys_a);
        tmp = e->sort_tmp_bytes;
        HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, e->keys_a, e->out_sorted, (size_t)Ed, 0u,
                                         (unsigned)(2 * e->bits), e->bs));
    }
    return DPPR_OK;
}

// One orientation of the incremental update: sorted' = (sorted minus deleted instances) merged with inserted.
int merge_batch_keys(dppr_engine *e, uint64_t *&sorted, uint64_t *del_unsorted, uint64_t *del_sorted, int nd,
                     uint64_t *ins_unsorted, uint64_t *ins_sorted, int ni, unsigned key_bits = 0, int miss_word = MERGE_MISS_WORD) {
    const int Ed = e->Ed;
    if (key_bits == 0) key_bits = (unsigned)(2 * e->bits); // (the CSR keys; the binned tables' words say how many bits they use)
    size_t tmp = e->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, del_unsorted, del_sorted, (size_t)nd, 0u, key_bits, e->bs));
    tmp = e->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(e->sort_tmp, tmp, ins_unsorted, ins_sorted, (size_t)ni, 0u, key_bits, e->bs));
    // retired positions, then one pass: every kept and every inserted key straight to its place (dppr_builder.hpp k_merge_tiles)
    hipLaunchKernelGGL(k_del_positions, dim3(grid_for(nd)), dim3(BLOCK), 0, e->bs, sorted, Ed, del_sorted, nd, e->delpos,
                       e->hub_hist + miss_word);
    const int n_tiles = (Ed + CMP_TILE - 1) / CMP_TILE;
    hipLaunchKernelGGL(k_merge_tiles, dim3(n_tiles), dim3(BLOCK), 0, e->bs, sorted, Ed, e->delpos, nd, ins_sorted, ni, e->keys_b,
                       (size_t)Ed - (size_t)nd + (size_t)ni);
    HIP_TRY(hipGetLastError());
    std::swap(sorted, e->keys_b); // the merged array is the new persistent one; the old becomes scratch
    return DPPR_OK;
}

// Workgroup size of the sweeps: 1024 unless pinned (dppr_set_tuning; 512 was measured on the LiveJournal
// and twitter stand-ins and is not better once two 1024-thread workgroups fit a CU).
int sweep_block(const dppr_engine *e) { return e->pull_block ? e->pull_block : 1024; }

// workgroups of the resident sweep that the device holds at once (0: resident sweeps are off)
int persist_capacity(const dppr_engine *e) {
    const int pb = sweep_block(e);
    if (!e->persist_mode || !e->persist_ok || (pb != 256 && pb != 512 && pb != 1024)) return 0;
    return e->persist_cap;
}

// How many workgroups of the resident sweep the device holds at once, from the runtime's occupancy
// figure for the instantiation the engine will launch.
int query_persist_cap(dppr_engine *e) {
    e->persist_cap = 0;
    int per_cu = 0;
    const int pb = sweep_block(e);
    if (pb != 256 && pb != 512 && pb != 1024) return DPPR_OK; // other block sizes (tuning only): per-iteration launches
    if (pb == 256) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pull_resident<256>, 256, 0));
    else if (pb == 512) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pull_resident<512>, 512, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pull_resident<1024>, 1024, 0));
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
    e->persist_cap = std::min(per_cu * cus, STAT_SLOTS);
    return DPPR_OK;
}

// Cut the vertex range into sweep groups of at most (workgroup size / 64) consecutive tiles with
// about equal weight (edges + a per-vertex term), so that no workgroup of k_pull_iter is the
// straggler because a hub's long row happens to sit in its range. Host greedy over the tile
// prefix; part of the (untimed) graph build.
int cut_sweep_groups(dppr_engine *e, Epoch &ep) {
    const int NV = e->n_int;
    const int n_tiles = (NV + WAVE - 1) / WAVE;
    const int max_tiles = sweep_block(e) / WAVE;
    e->h_tiles.resize((size_t)n_tiles + 2);
    if (n_tiles > 0) {
        int *scratch = reinterpret_cast<int *>(e->keys_a); // Ed * 8 bytes >= (n_tiles + 1) * 4 unless the graph is tiny
        const bool fits = (size_t)e->Ed * sizeof(uint64_t) >= ((size_t)n_tiles + 1) * sizeof(int);
        if (!fits) scratch = e->hub_slot_of;               // V ints: always large enough
        hipLaunchKernelGGL(k_tile_prefix, dim3(grid_for(n_tiles + 1)), dim3(BLOCK), 0, e->bs, ep.out_row_ptr, NV,
                           n_tiles, scratch);
        HIP_TRY(hipMemcpyAsync(e->h_tiles.data(), scratch, sizeof(int) * ((size_t)n_tiles + 1), hipMemcpyDeviceToHost,
                               e->bs));
        HIP_TRY(hipStreamSynchronize(e->bs));
    }
    std::vector<int32_t> cut;
    const int32_t *prefix = e->h_tiles.data();
    // A window small enough for one workgroup per group to be resident at once gets at most that
    // many groups (then runs of dense iterations are single launches, dppr_resident.hpp). A resident
    // workgroup's time is its edge count (every iteration all workgroups wait for the slowest one's
    // values), so this cut MINIMISES THE LARGEST group (dppr_cut.hpp); per tile the per-vertex work of a
    // resident workgroup is small and fixed (weight 8). Otherwise: many groups of about equal weight.
    const int cap = persist_capacity(e);
    bool fitted = false;
    if (cap > 0 && (long long)n_tiles <= (long long)cap * max_tiles * 7 / 8) fitted = cut_minmax(prefix, n_tiles, max_tiles, cap, 8, cut);
    if (!fitted)
        cut_greedy(prefix, n_tiles, max_tiles,
                   std::max<long long>(252, (n_tiles + max_tiles * 3 / 4 - 1) / std::max(1, max_tiles * 3 / 4)), 2 * WAVE, cut);
    ep.n_groups = (int)cut.size() - 1;
    ep.grp_n_int = NV;
    HIP_TRY(hipMemcpyAsync(ep.grp_tile, cut.data(), sizeof(int) * cut.size(), hipMemcpyHostToDevice, e->bs));
    // slot tables for resident launches (a window that got the resident cut; every group must fit the table build's sort)
    ep.res_valid = false;
    // (a single-source slot exists: its launches will want the arena -- grown here unless a solver call may be using it right now:
    // dppr_update grows it itself before its first resident launch)
    if (fitted && !e->slots.empty() && !e->build_concurrent) (void)resident_arena(e, ep);
    if (fitted && e->res_slots && ep.Ed > 0 && NV <= RES_ID_LIMIT) {
        long long largest = 0;
        for (size_t g = 0; g + 1 < cut.size(); ++g) largest = std::max<long long>(largest, (long long)prefix[cut[g + 1]] - prefix[cut[g]]);
        if (largest <= RES_SORT_MAX) {
            if ((size_t)ep.Ed > ep.res_pk_cap) {
                HIP_TRY(hipStreamSynchronize(e->bs));
                (void)hipFree(ep.res_pk);
                ep.res_pk = nullptr;
                ep.res_pk_cap = 0;
                HIP_TRY(hipMalloc((void **)&ep.res_pk, sizeof(uint32_t) * ((size_t)ep.Ed + (size_t)ep.Ed / 8 + 1024)));
                ep.res_pk_cap = (size_t)ep.Ed + (size_t)ep.Ed / 8 + 1024;
            }
            hipLaunchKernelGGL(k_res_slots, dim3(ep.n_groups), dim3(1024), 0, e->bs, NV, ep.grp_tile, ep.out_row_ptr, ep.out_col, ep.res_pk);
            HIP_TRY(hipGetLastError());
            ep.res_valid = true;
        }
    }
    HIP_TRY(hipStreamSynchronize(e->bs)); // `cut` is a local
    ep.su_inline = false;
    if (fitted)
        if (int rrc = res_record_ranges(e, ep)) return rrc;
    ep.n_ggroups = 0;
    if (e->any_groups) { // groups of at most 16 (8) tiles for k_gsweep<1, 1024> (<2, 512>)
        const int gmax = (e->wide_groups ? 512 : 1024) / WAVE;
        const long long want = std::max<long long>(e->ggroups_min, (n_tiles + gmax * 3 / 4 - 1) / std::max(1, gmax * 3 / 4));
        ep.ggrp_max_tiles = gmax;
        cut_greedy(prefix, n_tiles, gmax, want, 2 * WAVE, cut);
        ep.n_ggroups = (int)cut.size() - 1;
        HIP_TRY(hipMemcpyAsync(ep.ggrp_tile, cut.data(), sizeof(int) * cut.size(), hipMemcpyHostToDevice, e->bs));
        // the groups' row tables, once per epoch (every sweep of every source group of this epoch loads them)
        const int nvx = gmax * WAVE;
        const size_t need = (size_t)ep.n_ggroups * (size_t)GT_STRIDE(nvx);
        if (need > ep.gtab_cap) {
            HIP_TRY(hipStreamSynchronize(e->bs));
            (void)hipFree(ep.gtab);
            ep.gtab = nullptr;
            ep.gtab_cap = 0;
            HIP_TRY(hipMalloc((void **)&ep.gtab, sizeof(int) * (need + need / 8 + 1024)));
            ep.gtab_cap = need + need / 8 + 1024;
        }
        if (ep.n_ggroups <= 0) {
            // (no vertex has an id yet: nothing to sweep)
        } else if (nvx == 512)
            hipLaunchKernelGGL(k_gtables<512>, dim3(std::min(ep.n_ggroups, 1024)), dim3(GNT), 0, e->bs, NV, ep.ggrp_tile,
                               ep.n_ggroups, ep.out_row_ptr, ep.gtab);
        else
            hipLaunchKernelGGL(k_gtables<1024>, dim3(std::min(ep.n_ggroups, 1024)), dim3(GNT), 0, e->bs, NV, ep.ggrp_tile,
                               ep.n_ggroups, ep.out_row_ptr, ep.gtab);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(e->bs));
    }
    return DPPR_OK;
}


// ---- binned sweep tables of an epoch (dppr_binned.hpp). Part of the (untimed) graph build; needs the sorted
// out-orientation keys, i.e. `ep` must be the epoch the persistent key arrays describe (the newest one).
bool bin_wanted(const dppr_engine *e) {
    if (e->bin_mode == 2) return true;
    return e->bin_mode == 1 && !e->slots.empty() && (long long)e->n_int >= e->bin_min_ids;
}

// The largest block shapes dppr_set_binned_sweep admits -- ONE pair of constants for the validation, its message and the
// kernels' dynamic-LDS attribute (the attribute said 272 tiles, the validation 288).
constexpr int BIN_MAX_HA_TILES = 272, BIN_MAX_HB_TILES = 120;
static_assert(BIN_MAX_HA_TILES * WAVE * (int)sizeof(double) + 4096 <= 160 * 1024, "k_bin_scatter: the largest A-block's slice of x + static LDS fits a gfx950 CU");
static_assert(BIN_MAX_HB_TILES * WAVE * 20 + 4096 <= 160 * 1024, "k_bin_reduce: the largest B-block's rows + static LDS fit a gfx950 CU");
static_assert(BIN_MAX_HB_TILES * WAVE <= (1 << BIN_RL) && BIN_MAX_HA_TILES * WAVE <= (1 << BIN_HL), "a row / head index inside its block fits its field of the sort words");
static_assert(BIN_RL + BIN_HL + 32 <= 64, "table words: two block numbers of <= 32 bits together above the in-block row and head indices");

// An allocation of the (optional) binned-sweep tables that fails is not an error of the call that wanted them: the
// partial allocations are released, the sticky HIP error is cleared and the epoch sweeps with k_pull_iter.
static bool bin_alloc(void **p, size_t bytes) {
    if (*p) return true;
    if (getenv("DPPR_TEST_BIN_OOM")) { // (test hook: these allocations fail as if the device were out of memory)
        *p = nullptr;
        return false;
    }
    if (hipMalloc(p, bytes) == hipSuccess) return true;
    *p = nullptr;
    (void)hipGetLastError();
    return false;
}

int bin_prepare(dppr_engine *e, bool *have) { // engine-level scratch, once (idempotent per pointer: a failed attempt may be repeated)
    *have = false;
    if (e->bin_ready) {
        *have = true;
        return DPPR_OK;
    }
    const size_t Edn = (size_t)std::max(e->Ed, 1);
    bool ok = true;
    ok = ok && bin_alloc((void **)&e->bin_vblk_a, sizeof(int) * (size_t)e->V);
    ok = ok && bin_alloc((void **)&e->bin_small, sizeof(int) * BIN_SMALL_INTS);
    ok = ok && bin_alloc((void **)&e->bin_scan, sizeof(unsigned long long) * ((size_t)4 * (Edn / WAVE + 3)));
    ok = ok && bin_alloc((void **)&e->bin_vals, sizeof(double) * (Edn + 64));
    ok = ok && bin_alloc((void **)&e->bin_wb, sizeof(uint64_t) * Edn) && bin_alloc((void **)&e->bin_wa, sizeof(uint64_t) * Edn); // (rotate with keys_a / keys_b: same size)
    if (ok && !e->bin_tmp) {
        HIP_TRY(rocprim::radix_sort_keys(nullptr, e->bin_tmp_bytes, e->keys_a, e->keys_b, Edn, 0u, 64u, e->bs));
        size_t scan_bytes = 0, scan_bytes32 = 0; // (the scans of the tables' tail share the scratch: sized for whichever needs most)
        HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, Edn / WAVE + 3,
                                        rocprim::plus<unsigned long long>(), e->bs));
        HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes32, (int *)nullptr, (int *)nullptr, 0, Edn / WAVE + 3, rocprim::plus<int>(), e->bs));
        e->bin_tmp_bytes = std::max(e->bin_tmp_bytes, std::max(scan_bytes, scan_bytes32));
        ok = bin_alloc(&e->bin_tmp, std::max<size_t>(e->bin_tmp_bytes, 16));
    }
    if (!ok) { // out of memory: nothing half-built stays behind, the sweeps of this engine gather (k_pull_iter)
        (void)hipFree(e->bin_vblk_a); (void)hipFree(e->bin_small); (void)hipFree(e->bin_vals); (void)hipFree(e->bin_tmp);
        (void)hipFree(e->bin_wb); (void)hipFree(e->bin_wa); (void)hipFree(e->bin_scan);
        e->bin_vblk_a = e->bin_small = nullptr;
        e->bin_scan = nullptr;
        e->bin_vals = nullptr;
        e->bin_tmp = nullptr;
        e->bin_wb = e->bin_wa = nullptr;
        return DPPR_OK;
    }
    // (the attribute belongs to the kernel, not to this engine: the largest shapes dppr_set_binned_sweep admits, so that engines
    // with different block shapes can share a process)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BIN_MAX_HA_TILES * WAVE * (int)sizeof(double)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bin_reduce), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BIN_MAX_HB_TILES * WAVE * 20));
    e->bin_ready = true;
    *have = true;
    return DPPR_OK;
}

// One cut (dppr_binned.hpp: every multiple of `cap` vertices, the first vertex behind every `target` edges, both sides of
// every row of >= target / 4 edges); device searches, the merge of the few thousand boundaries on the host.
int bin_cut(dppr_engine *e, const int *row_ptr, int NV, int cap, long long target, std::vector<int32_t> &cut) {
    const int Ed = e->Ed;
    target = std::max<long long>(target, 64);
    const int K = (int)std::min<long long>((Ed + target - 1) / target, BIN_MAX_BLOCKS);
    int *d_q = e->bin_small, *d_big = e->bin_small + BIN_MAX_BLOCKS, *d_cnt = d_big + BIN_MAX_BIG;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int), e->bs));
    if (K > 1) hipLaunchKernelGGL(k_bin_quantiles, dim3(grid_for(K)), dim3(BLOCK), 0, e->bs, row_ptr, NV, target, K, d_q);
    hipLaunchKernelGGL(k_bin_big_rows, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, row_ptr, NV, (int)std::max<long long>(target / 4, 1),
                       BIN_MAX_BIG, d_big, d_cnt);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h((size_t)BIN_MAX_BLOCKS + BIN_MAX_BIG + 1);
    HIP_TRY(hipMemcpyAsync(h.data(), e->bin_small, sizeof(int) * h.size(), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipStreamSynchronize(e->bs));
    cut.clear();
    for (long long v = 0; v < NV; v += cap) cut.push_back((int32_t)v);
    for (int k = 0; k + 1 < K; ++k) cut.push_back(h[(size_t)k]);
    const int nbig = std::min(h[(size_t)BIN_MAX_BLOCKS + BIN_MAX_BIG], BIN_MAX_BIG);
    for (int k = 0; k < nbig; ++k) {
        cut.push_back(h[(size_t)BIN_MAX_BLOCKS + k]);
        cut.push_back(h[(size_t)BIN_MAX_BLOCKS + k] + 1);
    }
    cut.push_back(NV);
    std::sort(cut.begin(), cut.end());
    cut.erase(std::unique(cut.begin(), cut.end()), cut.end());
    while (!cut.empty() && cut.back() > NV) cut.pop_back();
    if (cut.empty() || cut.front() != 0) cut.insert(cut.begin(), 0);
    if (cut.back() != NV) cut.push_back(NV);
    return DPPR_OK;
}

// New ids since the cuts were frozen: the last block grows up to its vertex cap, further blocks are appended. Block numbers and
// in-block indices of every existing vertex stay what they are -- the persistent words stay valid.
static void bin_extend_cut(std::vector<int32_t> &cut, int cap, int NV) {
    const int old_end = cut.back(), last_start = cut[cut.size() - 2];
    if (NV <= old_end) return;
    cut.pop_back();
    int end = old_end - last_start < cap ? std::min(last_start + cap, NV) : old_end;
    cut.push_back(end);
    while (end < NV) {
        end = std::min(end + cap, NV);
        cut.push_back(end);
    }
}

int build_bins(dppr_engine *e, Epoch &ep, const BinBatch *batch) {
    ep.bin_valid = false;
    const bool force_full = e->bin_force_full;
    e->bin_force_full = false;
    if (!bin_wanted(e) || e->Ed <= 0 || ep.grp_n_int <= 0) {
        e->bin_words_valid = false; // (not maintained through this slide)
        return DPPR_OK;
    }
    bool have = false;
    if (int rc = bin_prepare(e, &have)) return rc;
    if (!have) return DPPR_OK;
    const int Ed = e->Ed, NV = ep.grp_n_int;
    const int n_blk = (Ed + WAVE - 1) / WAVE; // aligned blocks of 64 words: the per-block tables have n_blk + 1 entries
    if (!ep.hl || !ep.dl || !ep.vb || !ep.tb) { // all four or none (a partial set from a failed attempt is released first)
        const size_t Edn = (size_t)Ed, nb1 = (size_t)n_blk + 2;
        const bool ok = bin_alloc((void **)&ep.hl, sizeof(uint16_t) * (Edn + WAVE)) && bin_alloc((void **)&ep.dl, sizeof(uint16_t) * (Edn + WAVE)) &&
                        bin_alloc((void **)&ep.vb, sizeof(int) * nb1) && bin_alloc((void **)&ep.tb, sizeof(int) * nb1);
        if (!ok) {
            (void)hipFree(ep.hl); (void)hipFree(ep.dl); (void)hipFree(ep.vb); (void)hipFree(ep.tb);
            ep.hl = ep.dl = nullptr;
            ep.vb = ep.tb = nullptr;
            e->bin_words_valid = false;
            return DPPR_OK; // (bin_valid stays false: this epoch's sweeps gather)
        }
    }
    // While the patch below is in progress the persistent words describe NEITHER epoch (one order merged, the other not yet): any
    // early return leaves them marked invalid, and the next slide sorts afresh
    const bool words_were_valid = e->bin_words_valid;
    e->bin_words_valid = false;
    // ---- the block cuts: frozen ones (extended by the ids that arrived since) while the tables are being patched, fresh ones otherwise
    const int cap_a = e->bin_ha_tiles * WAVE, cap_b = e->bin_hb_tiles * WAVE;
    bool patch = e->bin_incremental && words_were_valid && batch && !force_full && e->bin_slides_since_cut < e->bin_recut_every &&
                 !e->bin_cut_a.empty() && NV >= e->bin_cut_ids;
    bool keep_cuts = patch || (e->bin_frozen_rebuild && words_were_valid && !force_full && !e->bin_cut_a.empty() && NV >= e->bin_cut_ids &&
                               e->bin_slides_since_cut < e->bin_recut_every); // (tests: the sorts under the frozen cuts -- what the patched tables must equal)
    if (keep_cuts) {
        bin_extend_cut(e->bin_cut_a, cap_a, NV);
        bin_extend_cut(e->bin_cut_b, cap_b, NV);
        if ((int)e->bin_cut_a.size() - 1 > (1 << e->bin_abits) || (int)e->bin_cut_b.size() - 1 > (1 << e->bin_bbits) ||
            (size_t)(e->bin_cut_a.size() - 1) * (size_t)(e->bin_cut_b.size() - 1) > e->bin_first_cap)
            patch = keep_cuts = false; // the block numbers outgrew their fields / the tile table: cut afresh
    }
    if (!keep_cuts) {
        if (int rc = bin_cut(e, ep.row_ptr, NV, cap_a, e->bin_target_a, e->bin_cut_a)) return rc;
        if (int rc = bin_cut(e, ep.out_row_ptr, NV, cap_b,
                             e->bin_target > 0 ? e->bin_target : std::min<long long>(std::max<long long>(Ed / 256, 16384), 393216), e->bin_cut_b)) return rc;
        // fields with room for the blocks that new ids will append before the next re-cut
        const long long ra = (long long)e->bin_cut_a.size() - 1, rb = (long long)e->bin_cut_b.size() - 1;
        e->bin_abits = e->bin_bbits = 1;
        while ((1ll << e->bin_abits) < ra + ra / 8 + 16) e->bin_abits++;
        while ((1ll << e->bin_bbits) < rb + rb / 8 + 16) e->bin_bbits++;
        e->bin_slides_since_cut = 0;
    }
    e->bin_cut_ids = NV;
    const std::vector<int32_t> &cut_a = e->bin_cut_a, &cut_b = e->bin_cut_b;
    ep.n_a = (int)cut_a.size() - 1;
    ep.n_b = (int)cut_b.size() - 1;
    const int abits = e->bin_abits, bbits = e->bin_bbits;
    if (abits + bbits > 32 || ep.n_a + 2 > BIN_MAX_BLOCKS || ep.n_b + 2 > BIN_MAX_BLOCKS) return DPPR_OK; // (a window of that many blocks: the sweep stays k_pull_iter)
    if (!keep_cuts) { // the tile table (run index of every block pair's first run, B-major), with the fields' headroom
        const size_t need = ((size_t)ep.n_a + ep.n_a / 8 + 16) * ((size_t)ep.n_b + ep.n_b / 8 + 16);
        if (need > e->bin_first_cap) {
            HIP_TRY(hipStreamSynchronize(e->bs));
            (void)hipFree(e->bin_first);
            e->bin_first = nullptr;
            e->bin_first_cap = 0;
            if (hipMalloc((void **)&e->bin_first, sizeof(int) * need) != hipSuccess) {
                (void)hipGetLastError();
                return DPPR_OK; // (no tables: this epoch's sweeps gather)
            }
            e->bin_first_cap = need;
        }
    }
    // per epoch: acut | arun | bcut (block tables), then the chunk table
    const size_t tab_ints = (size_t)2 * (ep.n_a + 1) + (ep.n_b + 1);
    if (tab_ints > ep.bin_tab_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(ep.acut);
        ep.acut = nullptr;
        ep.bin_tab_cap = 0;
        HIP_TRY(hipMalloc((void **)&ep.acut, sizeof(int) * (tab_ints + tab_ints / 4 + 1024)));
        ep.bin_tab_cap = tab_ints + tab_ints / 4 + 1024;
    }
    int *d_arun = ep.acut + (ep.n_a + 1);
    ep.bcut = d_arun + (ep.n_a + 1);
    HIP_TRY(hipMemcpyAsync(ep.acut, cut_a.data(), sizeof(int) * cut_a.size(), hipMemcpyHostToDevice, e->bs));
    HIP_TRY(hipMemcpyAsync(ep.bcut, cut_b.data(), sizeof(int) * cut_b.size(), hipMemcpyHostToDevice, e->bs));
    int *d_astart = e->bin_small; // first in-CSR entry of every A-block (the full build's segments; not kept)
    hipLaunchKernelGGL(k_bin_vertex_block, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, ep.acut, ep.n_a, NV, ep.row_ptr, e->bin_vblk_a, d_astart);
    const bool cuts_in_lds = (size_t)(ep.n_b + 1) * sizeof(int) <= 48 * 1024;
    const size_t cut_lds = cuts_in_lds ? (size_t)(ep.n_b + 1) * sizeof(int) : 0;
    const unsigned word_bits = (unsigned)(BIN_LO + abits + bbits);
    if (patch) {
        // ---- the slide's retired and inserted edges as words of both orders, merged into the two persistent arrays
        HIP_TRY(hipMemsetAsync(e->hub_hist + MERGE_MISS_WORD + 1, 0, sizeof(int), e->bs));
        for (int amajor = 0; amajor < 2; ++amajor) {
            if (batch->nd > 0)
                hipLaunchKernelGGL(k_bin_keys, dim3(grid_for(batch->nd)), dim3(BLOCK), cut_lds, e->bs, batch->del, batch->nd, e->bits, ep.bcut, ep.n_b,
                                   cuts_in_lds ? 1 : 0, e->bin_vblk_a, ep.acut, abits, bbits, e->bks[0], amajor);
            if (batch->ni > 0)
                hipLaunchKernelGGL(k_bin_keys, dim3(grid_for(batch->ni)), dim3(BLOCK), cut_lds, e->bs, batch->ins, batch->ni, e->bits, ep.bcut, ep.n_b,
                                   cuts_in_lds ? 1 : 0, e->bin_vblk_a, ep.acut, abits, bbits, e->bks[1], amajor);
            HIP_TRY(hipGetLastError());
            if (int rc = merge_batch_keys(e, amajor ? e->bin_wa : e->bin_wb, e->bks[0], e->bks[2], batch->nd, e->bks[1], e->bks[3], batch->ni, word_bits,
                                          MERGE_MISS_WORD + 1)) return rc;
        }
        HIP_TRY(hipMemcpyAsync(&e->bin_miss_host, e->hub_hist + MERGE_MISS_WORD + 1, sizeof(int), hipMemcpyDeviceToHost, e->bs));
    } else {
        // ---- both orders from the sorted in-orientation keys ((head, row) order = A-block-major): the A-major word of every edge,
        // every A-block's segment grouped by B-block, then the B-major form sorted by B-block
        hipLaunchKernelGGL(k_bin_keys, dim3(grid_for(Ed)), dim3(BLOCK), cut_lds, e->bs, e->in_sorted, Ed, e->bits, ep.bcut, ep.n_b, cuts_in_lds ? 1 : 0,
                           e->bin_vblk_a, ep.acut, abits, bbits, e->keys_b, 1);
        HIP_TRY(hipGetLastError());
        size_t tmp = e->bin_tmp_bytes; // A-major: stable by (A-block, B-block); the words are in (head, row) order
        const char *placement = getenv("DPPR_BIN_PLACEMENT"); // (tests / A-B runs: "counting" wherever it can run -- small windows never qualify by themselves --, "radix" never)
        const bool cs_force = placement && !strcmp(placement, "counting"), cs_never = placement && !strcmp(placement, "radix");
        // every A-block's segment grouped by B-block in one pass (k_bin_group) where the radix sort would need FOUR passes over its
        // 8-bit digits (friendster stand-in, 26 bits: 6.0 ms against 8.9; with three -- twitter, 23 bits -- the sort wins, 3.0 against 3.8:
        // the single pass scatters 8-byte words over thousands of runs, a radix pass over 256)
        if (ep.n_b <= BIN_CS_MAX && (abits + bbits > 24 || cs_force) && !cs_never) {
            int n_pad = WAVE;
            while (n_pad < ep.n_b) n_pad *= 2;
            hipLaunchKernelGGL(k_bin_group, dim3(ep.n_a), dim3(BIN_CS_NT), sizeof(int) * (size_t)n_pad, e->bs, e->keys_b, d_astart, ep.acut, ep.n_b,
                               n_pad, bbits, e->bin_wa);
        } else {
            HIP_TRY(rocprim::radix_sort_keys(e->bin_tmp, tmp, e->keys_b, e->bin_wa, (size_t)Ed, (unsigned)BIN_LO, word_bits, e->bs));
        }
        hipLaunchKernelGGL(k_bin_swap_blocks, dim3(grid_for(Ed)), dim3(BLOCK), 0, e->bs, e->bin_wa, Ed, bbits, abits, e->keys_b);
        HIP_TRY(hipGetLastError());
        tmp = e->bin_tmp_bytes;        // B-major: the A-major sequence, stable by B-block (the top field of the B-major form)
        HIP_TRY(rocprim::radix_sort_keys(e->bin_tmp, tmp, e->keys_b, e->bin_wb, (size_t)Ed, (unsigned)(BIN_LO + abits), word_bits, e->bs));
        e->bin_miss_host = 0;
    }
    // ---- the tail both ways (dppr_binned.hpp): runs and tiles counted per aligned block of 64 words, scanned, then the tables
    unsigned long long *cnt_b = e->bin_scan, *x_b = cnt_b + (n_blk + 1), *cnt_a = x_b + (n_blk + 1), *x_a = cnt_a + (n_blk + 1);
    const int wgrid = std::min(grid_for((long long)(n_blk + 1) * WAVE), 4096);
    hipLaunchKernelGGL(k_bin_count, dim3(wgrid), dim3(BLOCK), 0, e->bs, e->bin_wb, Ed, cnt_b);
    hipLaunchKernelGGL(k_bin_count, dim3(wgrid), dim3(BLOCK), 0, e->bs, e->bin_wa, Ed, cnt_a);
    HIP_TRY(hipGetLastError());
    {
        size_t tmp = e->bin_tmp_bytes;
        HIP_TRY(rocprim::exclusive_scan(e->bin_tmp, tmp, cnt_b, x_b, 0ull, (size_t)n_blk + 1, rocprim::plus<unsigned long long>(), e->bs));
        tmp = e->bin_tmp_bytes;
        HIP_TRY(rocprim::exclusive_scan(e->bin_tmp, tmp, cnt_a, x_a, 0ull, (size_t)n_blk + 1, rocprim::plus<unsigned long long>(), e->bs));
    }
    unsigned long long totals[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&totals[0], x_a + n_blk, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipMemcpyAsync(&totals[1], x_b + n_blk, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipStreamSynchronize(e->bs)); // the totals (and the patch's miss count) have arrived
    if (patch && e->bin_miss_host != 0) {
        // a retired edge's word was not in the persistent arrays (never on a consistent window): they cannot be trusted -- sort afresh
        e->bin_force_full = true;
        return build_bins(e, ep, nullptr);
    }
    const int R = (int)(totals[0] >> 32), T = (int)(totals[0] & 0xffffffffull);
    if (R != (int)(totals[1] >> 32) || T != (int)(totals[1] & 0xffffffffull) || R <= 0 || T <= 0)
        return fail(e, DPPR_ERR_HIP, "binned tables: the two orders of the window's edges disagree about their runs / tiles");
    if ((size_t)T + WAVE > ep.tdelta_cap) {
        (void)hipFree(ep.tdelta);
        ep.tdelta = nullptr;
        ep.tdelta_cap = 0;
        const size_t want = (size_t)T + (size_t)T / 4 + 1024;
        if (hipMalloc((void **)&ep.tdelta, sizeof(int) * want) != hipSuccess) {
            (void)hipGetLastError();
            return DPPR_OK; // (this epoch's sweeps gather; the words were not kept either: the next slide sorts)
        }
        ep.tdelta_cap = want;
    }
    ep.n_runs = R;
    ep.n_tiles = T;
    const int n_rb = (R + WAVE - 1) / WAVE;
    HIP_TRY(hipMemsetAsync(d_arun, 0xff, sizeof(int) * (size_t)(ep.n_a + 1), e->bs)); // -1: a block without an edge
    hipLaunchKernelGGL(k_bin_btables, dim3(wgrid), dim3(BLOCK), 0, e->bs, e->bin_wb, Ed, x_b, ep.n_a, abits, ep.dl, ep.vb, e->bin_first);
    hipLaunchKernelGGL(k_bin_atables, dim3(wgrid), dim3(BLOCK), 0, e->bs, e->bin_wa, Ed, x_a, ep.n_a, bbits, e->bin_first, ep.hl, ep.tdelta, d_arun);
    int *cnt_r = reinterpret_cast<int *>(cnt_b), *x_r = cnt_r + (n_rb + 1); // (the B-major counts are done with)
    hipLaunchKernelGGL(k_bin_count16, dim3(std::min(grid_for((long long)(n_rb + 1) * WAVE), 4096)), dim3(BLOCK), 0, e->bs, ep.hl, R, cnt_r);
    HIP_TRY(hipGetLastError());
    {
        size_t tmp = e->bin_tmp_bytes;
        HIP_TRY(rocprim::exclusive_scan(e->bin_tmp, tmp, cnt_r, x_r, 0, (size_t)n_rb + 1, rocprim::plus<int>(), e->bs));
    }
    hipLaunchKernelGGL(k_bin_tb, dim3(grid_for(n_rb + 1)), dim3(BLOCK), 0, e->bs, ep.hl, R, x_r, ep.tb);
    HIP_TRY(hipGetLastError());
    // chunks of the A-major run list (a block of many runs is dealt to several workgroups of k_bin_scatter; chunk starts inside a
    // block are multiples of 64 runs: a wave works on aligned blocks of the tables)
    std::vector<int32_t> arun((size_t)ep.n_a + 1);
    HIP_TRY(hipMemcpyAsync(arun.data(), d_arun, sizeof(int) * arun.size(), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipStreamSynchronize(e->bs));
    arun[(size_t)ep.n_a] = R;
    for (int a = ep.n_a - 1; a >= 0; --a)
        if (arun[(size_t)a] < 0) arun[(size_t)a] = arun[(size_t)a + 1]; // (no edge: an empty range)
    HIP_TRY(hipMemcpyAsync(d_arun, arun.data(), sizeof(int) * arun.size(), hipMemcpyHostToDevice, e->bs));
    if (patch) e->bin_patched++; else e->bin_rebuilt++;
    e->bin_words_valid = true;
    e->bin_slides_since_cut++;
    std::vector<BinChunk> chunks;
    const int csize = (int)std::max<long long>((e->bin_chunk + WAVE - 1) / WAVE * WAVE, WAVE);
    for (int a = 0; a < ep.n_a; ++a) {
        const int j0 = arun[(size_t)a], j1 = arun[(size_t)a + 1];
        for (int lo = j0; lo < j1;) {
            const int hi = std::min(j1, (lo / WAVE) * WAVE + csize); // (ends on a multiple of 64 unless the block does)
            chunks.push_back(BinChunk{a, lo, hi});
            lo = hi;
        }
    }
    ep.n_chunks = (int)chunks.size();
    if (chunks.size() > ep.chunk_cap) {
        (void)hipFree(ep.chunks);
        ep.chunks = nullptr;
        ep.chunk_cap = 0;
        HIP_TRY(hipMalloc((void **)&ep.chunks, sizeof(BinChunk) * (chunks.size() + chunks.size() / 4 + 256)));
        ep.chunk_cap = chunks.size() + chunks.size() / 4 + 256;
    }
    if (!chunks.empty()) HIP_TRY(hipMemcpy(ep.chunks, chunks.data(), sizeof(BinChunk) * chunks.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(e->bs));
    ep.bin_n_int = NV;
    ep.bin_valid = true;
    return DPPR_OK;
}

// Hub directory + in-CSR + out-CSR of `ep` from the persistent sorted keys and outdeg.
int build_epoch(dppr_engine *e, Epoch &ep, const BinBatch *batch = nullptr) {
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    const int Ed = e->Ed;
    const int NV = e->n_int; // only vertices that ever had an edge (or are a source) exist internally
    // hub directory: the (at most HUB_CAP) vertices of largest out-degree, at least hub_min_degree
    {
        HIP_TRY(hipMemsetAsync(e->hub_hist, 0, sizeof(int) * 33, e->bs));
        hipLaunchKernelGGL(k_deg_hist, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, e->outdeg, NV, e->hub_min_degree,
                           e->hub_hist);
        int hist[32];
        HIP_TRY(hipMemcpyAsync(hist, e->hub_hist, sizeof(hist), hipMemcpyDeviceToHost, e->bs));
        HIP_TRY(hipStreamSynchronize(e->bs));
        long long above = 0;
        int k = 31;
        for (; k >= 0; --k) {
            if (above + hist[k] > HUB_CAP) break;
            above += hist[k];
        }
        // every bucket > k fits; threshold = lower edge of bucket k+1
        const long long thresh = (long long)e->hub_min_degree << (k + 1);
        const int th = (int)std::min<long long>(thresh, 0x7fffffff);
        hipLaunchKernelGGL(k_assign_hubs, dim3(grid_for(NV)), dim3(BLOCK), 0, e->bs, e->outdeg, NV, th,
                           e->hub_slot_of, ep.hub_v, ep.hub_degp1, e->hub_hist + 32);
        HIP_TRY(hipGetLastError());
        ep.n_hubs = (int)above;
    }
    // row pointers are filled for the whole id capacity: ids assigned later read as empty rows
    hipLaunchKernelGGL(k_build_csr, dim3(grid_for(std::max(Ed, e->V + 1))), dim3(BLOCK), 0, e->bs, e->in_sorted, Ed,
                       e->V, e->bits, e->hub_slot_of, ep.row_ptr, ep.adj);
    hipLaunchKernelGGL(k_build_out_csr, dim3(grid_for(std::max(Ed, e->V + 1))), dim3(BLOCK), 0, e->bs,
                       e->directed ? e->out_sorted : e->in_sorted, Ed, e->V, e->bits, ep.out_row_ptr, ep.out_col);
    HIP_TRY(hipGetLastError());
    ep.Ed = Ed;
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    if (int rc = cut_sweep_groups(e, ep)) return rc;
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    const int brc = build_bins(e, ep, batch);
    e->heartbeat.fetch_add(1, std::memory_order_relaxed);
    return brc;
}

// The batch's records, grouped by tail at slide time, cut into the sweep groups' ranges (dppr_resident.hpp, PLAN_UPDATE). Needs
// both the grouping and a resident-size group cut: called by whichever of the two is made last. Untimed (graph build / slide).
int res_record_ranges(dppr_engine *e, Epoch &ep) {
    ep.su_inline = false;
    const int pb = sweep_block(e);
    if (!e->res_update || !ep.grouped || ep.L <= 0 || ep.L >= SU_SPLIT_MIN || ep.n_groups <= 0 || ep.n_groups > persist_capacity(e)) return DPPR_OK;
    const size_t need = (size_t)ep.n_groups + 3;
    if (need > ep.su_rng_cap) {
        HIP_TRY(hipStreamSynchronize(e->bs));
        (void)hipFree(ep.su_rng);
        ep.su_rng = nullptr;
        ep.su_rng_cap = 0;
        HIP_TRY(hipMalloc((void **)&ep.su_rng, sizeof(int) * (need + 1024)));
        ep.su_rng_cap = need + 1024;
    }
    int *stat = ep.su_rng + ep.n_groups + 1;
    HIP_TRY(hipMemsetAsync(stat, 0, sizeof(int) * 2, e->bs));
    hipLaunchKernelGGL(k_res_rec_ranges, dim3((ep.n_groups + 256) / 256), dim3(256), 0, e->bs, ep.sk, ep.L, ep.grp_tile, ep.n_groups,
                       ep.su_rng, stat);
    HIP_TRY(hipGetLastError());
    int h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, stat, sizeof(h), hipMemcpyDeviceToHost, e->bs));
    HIP_TRY(hipStreamSynchronize(e->bs));
    ep.su_inline = h[0] <= pb && h[1] == ep.L; // (a tail beyond the last group: an id assigned after the cut -- the cut is redone then)
    return DPPR_OK;
}

// The arena of a resident launch (dppr_resident.hpp, FRESH VECTORS): RES_VECTORS vectors of `stride` doubles, scratch between
// launches, one per engine (the engine's launches are serial on its stream).
// Grown when a larger window is cut (graph build) or, failing that, before the first launch that needs it; without it (out of
// memory) the window's sweeps simply run as per-iteration launches.
bool resident_arena(dppr_engine *e, const Epoch &ep) {
    const long long stride = ((long long)ep.grp_n_int + 1023) / 1024 * 1024;
    if (stride <= e->res_arena_stride) return true;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    (void)hipFree(e->res_arena);
    e->res_arena = nullptr;
    e->res_arena_stride = 0;
    const long long want = std::min<long long>(((long long)e->V + 1023) / 1024 * 1024, stride + stride / 4);
    if (hipMalloc((void **)&e->res_arena, sizeof(double) * (size_t)want * RES_VECTORS) != hipSuccess) {
        (void)hipGetLastError();
        e->res_arena = nullptr;
        return false;
    }
    e->res_arena_stride = want;
    return true;
}

} // namespace
