// dppr_host_group.hpp -- host side, part 4 of 4: source groups (f2). The frontier loop of up to 16 sources solved together:
// one-sweep and multi-sweep launches of k_gsweep, the push form of a loop's tail (dppr_gpush.hpp), the group's stream update.
#pragma once

namespace {

// ---------------------------------------------------------------------------- f2: groups
// The group kernels are instantiated per row width (dppr_multi.hpp: GW = 2, 4, .. 16 doubles; one double per lane of
// an octet up to 8, two beyond): f(SPL, GW) is called with the two as compile-time constants.
template <int N> using IC = std::integral_constant<int, N>;
template <class F>
void with_row(int gw, F &&f) {
    switch (gw) {
    case 2: f(IC<1>{}, IC<2>{}); break;
    case 4: f(IC<1>{}, IC<4>{}); break;
    case 6: f(IC<1>{}, IC<6>{}); break;
    case 8: f(IC<1>{}, IC<8>{}); break;
    case 10: f(IC<2>{}, IC<10>{}); break;
    case 12: f(IC<2>{}, IC<12>{}); break;
    case 14: f(IC<2>{}, IC<14>{}); break;
    default: f(IC<2>{}, IC<16>{}); break;
    }
}

// workgroups of the multi-sweep form of k_gsweep that the device holds at once
int group_multi_capacity(dppr_engine *e, int spl) {
    int &cap = e->gmulti_cap[spl - 1];
    if (cap < 0) {
        int per_cu = 0, cus = 0; // (the widest row of each lane split: narrower ones need no more)
        hipError_t rc = spl == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gsweep<1, 8, 1024, true, 2>, GNT, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gsweep<2, 16, 512, true, 2>, GNT, 0);
        if (rc != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess)
            per_cu = 0;
        cap = std::min(per_cu * cus, STAT_SLOTS);
    }
    return cap;
}

// One frontier loop of a source group. `tails`: the state was converged before the batch's stream
// update, so only the batch tails (sorted in su_k[1]) can be legal -- no pass over all vertices.
// The tail of a group's loop in push form (dppr_gpush.hpp). Called between two chunks of sweeps when the frontier is
// small: g.act[0] / g.x hold the frontier the last sweep left. Returns with *converged set (the loop is over; state as
// a finished loop leaves it) or cleared (the mode gave up -- an iteration too large for it -- and put the frontier back
// in sweep form: g.act[0], g.x, frontier sizes in row 0 of g.cnt, the other rows zero), or with *entered false if it
// did not start (nothing changed). Iterations run are added to *iters and to the group's statistics.
// *owed: the handed-over snapshot's pagerank share is still to be credited (the last sweep was a deferring one,
// dppr_multi.hpp); on a return in sweep form it says the same about the snapshot handed back.
int group_push_tail(dppr_engine *e, Group &g, const Epoch &ep, int phase, double eps, long long pairs_at_entry, int *iters, bool *entered,
                    bool *converged, bool *owed) {
    *entered = false;
    *converged = false;
    const int GWM = GS_MAX;
    const int cap = std::max(1024, std::min(e->gpush_list_cap, e->V));
    if (g.plist_cap != cap) {
        HIP_TRY(loop_wait(e));
        (void)hipFree(g.plist[0]); (void)hipFree(g.plist[1]); (void)hipFree(g.ppre); (void)hipFree(g.pctl);
        g.plist[0] = g.plist[1] = g.ppre = nullptr;
        g.pctl = nullptr;
        g.plist_cap = 0;
        HIP_TRY(hipMalloc((void **)&g.plist[0], sizeof(int) * (size_t)cap));
        HIP_TRY(hipMalloc((void **)&g.plist[1], sizeof(int) * (size_t)cap));
        HIP_TRY(hipMalloc((void **)&g.ppre, sizeof(int) * ((size_t)cap + 1)));
        HIP_TRY(hipMalloc((void **)&g.pctl, sizeof(GPushCtl)));
        g.plist_cap = cap;
    }
    static thread_local GPushCtl h;
    // no host round trip on the way in: a list that does not fit (overflow) moves nothing and makes the first scan call
    // the mode off, which the read-back of the first chunk shows
    HIP_TRY(hipMemsetAsync(g.pctl, 0, sizeof(GPushCtl), e->stream));
    const int n_words = (ep.grp_n_int + 31) / 32;
    hipLaunchKernelGGL(k_gpush_list, dim3(grid_for(n_words)), dim3(BLOCK), 0, e->stream, g.act[0], n_words, g.plist[0], cap, g.pctl);
    // the frontier's rows move from the snapshot back to residual[]; its bits stay set (they queue it for iteration 0)
    const int rows_grid = grid_for(std::min<long long>(pairs_at_entry, cap), BLOCK / OCT);
    with_row(g.gw, [&](auto spl, auto gw) {
        hipLaunchKernelGGL((k_gpush_rows<decltype(spl)::value, decltype(gw)::value>), dim3(rows_grid), dim3(BLOCK), 0, e->stream, g.plist[0],
                           g.pctl, 0, g.x, g.r, false);
    });
    HIP_TRY(hipGetLastError());
    *entered = true;
    const int credit_first = *owed ? 1 : 0; // (iteration 0 of this mode settles it; every later one credits as it snapshots)
    // what an iteration may cost here: a sweep's floor is ~0.02 us per sweep group, a returning f64 atomic ~1 / 20 000 us
    // (round 6: 20 in-edges per sweep group, was 200 -- on the headline's 3 075 groups a push iteration of up to 615 K in-edges x 10 sources cost up
    // to 540 us where a near-empty sweep costs 50-85: bound 615 K / 150 K / 60 K / 40 K / 20 K -> 12.37 / 12.24 / 12.18 / 12.19 / 12.23 ms per batch,
    // two runs each on one box; twitter / friendster groups and resident-size windows: unchanged)
    const long long max_edges = e->gpush_max_edges > 0 ? e->gpush_max_edges : std::max<long long>(4096, 20ll * std::max(ep.n_ggroups, 1));
    const int grid = 256;
    static const bool trace = getenv("DPPR_GROUP_TRACE") != nullptr;
    int it_done = 0;
    long long known_n = pairs_at_entry; // (an upper bound of the frontier's vertices until the first read-back)
    bool tiny_declined = false;
    long long last_adds = pairs_at_entry <= 64 ? 0 : -1; // edge x source adds of the last iteration run (-1: not known yet)
    for (;;) {
        if (known_n <= TINY_N && last_adds >= 0 && last_adds <= TINY_E / 2 && !tiny_declined) {
            // a frontier of a few hundred vertices: a run of iterations as ONE single-workgroup launch
            with_row(g.gw, [&](auto spl, auto gw) {
                hipLaunchKernelGGL((k_gpush_tiny<decltype(spl)::value, decltype(gw)::value>), dim3(1), dim3(1024), 0, e->stream, g.pctl, g.plist[0],
                                   g.plist[1], ep.row_ptr, ep.adj, ep.hub_degp1, g.r, g.p, g.act[0], phase, eps, g.dstats, GPUSH_LOG, credit_first);
            });
        } else {
        // iterations per chunk (<= GPUSH_LOG): down here the frontier about halves per iteration, so the first chunk is
        // sized to reach the single-workgroup form (an iteration that finds nothing is three empty dispatches)
        int m = 2;
        if (it_done == 0)
            for (long long f = pairs_at_entry; f > 128 && m < GPUSH_LOG; f >>= 2) ++m;
        tiny_declined = false;
        for (int k = 0; k < m; ++k) {
            hipLaunchKernelGGL(k_gpush_scan, dim3(1), dim3(1024), 0, e->stream, g.pctl, g.plist[0], g.plist[1], ep.row_ptr, g.ppre, cap - 1, max_edges);
            with_row(g.gw, [&](auto spl, auto gw) {
                constexpr int SPL = decltype(spl)::value, GW = decltype(gw)::value;
                hipLaunchKernelGGL((k_gpush_snap<SPL, GW>), dim3(grid), dim3(BLOCK), 0, e->stream, g.plist[0], g.plist[1], g.pctl, g.x, g.r, g.p,
                                   g.act[0], phase, eps, credit_first);
                hipLaunchKernelGGL((k_gpush_expand<SPL, GW>), dim3(grid), dim3(BLOCK), 0, e->stream, g.plist[0], g.plist[1], g.pctl, g.ppre,
                                   ep.row_ptr, ep.adj, ep.hub_degp1, g.x, g.r, g.act[0], g.plist[0], g.plist[1], cap, phase, eps, g.dstats);
            });
        }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h, g.pctl, sizeof(GPushCtl), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(loop_wait(e));
        for (int i = it_done; i < h.it; ++i) {
            long long F = 0;
            for (int s = 0; s < GWM; ++s) F += h.F[i & (GPUSH_LOG - 1)][s];
            if (F == 0) continue;
            g.st.iterations++;
            g.st.sum_F += F;
            ++*iters;
            if (trace)
                fprintf(stderr, "[gpush ] phase %d iteration +%d  frontier pairs %9lld  adds %lld\n", phase, i, F, h.atomics[i & (GPUSH_LOG - 1)]);
        }
        if (h.it == it_done && !h.stop && known_n <= TINY_N) tiny_declined = true; // (too many vertices or in-edges for one workgroup)
        if (h.it > it_done) last_adds = h.atomics[(h.it - 1) & (GPUSH_LOG - 1)];
        it_done = h.it;
        known_n = h.n[h.it & 1];
        if (h.stop && h.it == 0 && h.overflow) { // the frontier did not fit the lists: nothing was moved, the sweeps go on
            *entered = false;
            return DPPR_OK;
        }
        if (h.stop) { // an iteration too large for this form: the queued vertices go back to sweep form
            if (trace) fprintf(stderr, "[gpush ] phase %d: an iteration of %d vertices called itself off after %d iterations\n", phase, h.n[h.it & 1], h.it);
            HIP_TRY(hipMemsetAsync(g.cnt, 0, sizeof(int) * 3 * GWM, e->stream));
            with_row(g.gw, [&](auto spl, auto gw) {
                hipLaunchKernelGGL((k_gpush_leave<decltype(spl)::value, decltype(gw)::value>), dim3(grid_for(ep.grp_n_int, BLOCK / OCT)), dim3(BLOCK), 0,
                                   e->stream, ep.grp_n_int, g.act[0], g.x, g.r, g.p, h.it > 0 ? 1 : 0, phase, eps, g.cnt);
            });
            HIP_TRY(hipGetLastError());
            if (h.it > 0) *owed = false; // (iteration 0 settled the hand-over, k_gpush_leave credited what it queued)
            return DPPR_OK;
        }
        if (h.n[h.it & 1] == 0) {
            *converged = true;
            return DPPR_OK;
        }
        if (it_done >= e->max_iters) return fail(e, DPPR_ERR_NOT_CONVERGED, "iteration cap hit");
    }
}

int group_loop(dppr_engine *e, Group &g, const Epoch &ep, int phase, double eps, bool tails) {
    const int hp = phase == PHASE_BOTH ? 0 : phase; // (loop histories: the merged loop uses slot 0)
    int cur = 0;
    const int GWM = GS_MAX;
    HIP_TRY(hipMemsetAsync(g.cnt, 0, sizeof(int) * 3 * GWM, e->stream));
    if (tails) {
        HIP_TRY(hipMemsetAsync(g.act[0], 0, g.act_bytes, e->stream));
        if (ep.L > 0) {
            with_row(g.gw, [&](auto spl, auto gw) {
                hipLaunchKernelGGL((k_gseed_tails<decltype(spl)::value, decltype(gw)::value>), dim3(grid_for(ep.L, BLOCK / OCT)), dim3(BLOCK), 0,
                                   e->stream, batch_tails(e, ep), ep.L, g.r, g.x, g.p, g.act[0], phase, eps, g.cnt + cur * GWM);
            });
        }
    } else {
        // dense seeding: every legal vertex of every source enters, snapshot taken
        with_row(g.gw, [&](auto spl, auto gw) {
            hipLaunchKernelGGL((k_gseed_dense<decltype(spl)::value, decltype(gw)::value>), dim3(grid_for(ep.grp_n_int, BLOCK / OCT)), dim3(BLOCK), 0,
                               e->stream, ep.grp_n_int, g.r, g.x, g.p, g.act[0], phase, eps, g.cnt + cur * GWM);
        });
        g.st.inspected += (int64_t)ep.grp_n_int * g.n;
    }
    HIP_TRY(hipGetLastError());
    int *log = g.cnt + 5 * GWM;
    auto any_left = [&](const int *c) {
        for (int s = 0; s < GWM; ++s)
            if (c[s] > 0) return true;
        return false;
    };
    HIP_TRY(hipMemcpyAsync(e->pinned, g.cnt + cur * GWM, sizeof(int) * GWM, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(loop_wait(e));
    bool more = any_left(e->pinned);
    int active_iters = 0;
    if (e->gsweep_grid_cap <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0) cus = 256;
        e->gsweep_grid_cap = std::min(2 * cus, STAT_SLOTS);
    }
    const int sweep_grid = std::min(std::max(ep.n_ggroups, 1), e->gsweep_grid_cap);
    int follow = 4; // size of the next follow-up chunk of one-sweep launches
    // pagerank is credited every other sweep (dppr_multi.hpp): the seeding credited its snapshot, so the first sweep defers;
    // `owed` = the live snapshot's share has not been added yet, the next sweep is a crediting one
    bool owed = false;
    // the tail of the loop as pushes (dppr_gpush.hpp): below push_thr frontier pairs, one-sweep launches only
    long long push_thr = e->gpush_enter_pairs == 0 ? 0 : e->gpush_enter_pairs > 0 ? e->gpush_enter_pairs : std::max(64, ep.n_ggroups * e->gpush_auto_factor);
    bool push_gave_up = false;
    int dense_len = -1; // sweeps of this loop before the frontier was that small
    const int nvx = ep.ggrp_max_tiles * WAVE; // vertices per sweep group of this epoch's tables: 1024, or 512 once a 16-wide group exists
    for (int it = 0; more;) {
        if (it >= e->max_iters) return fail(e, DPPR_ERR_NOT_CONVERGED, "iteration cap hit");
        // ---- a window whose sweep groups are all resident at once: a run of sweeps as ONE launch (k_gsweep<.., true>)
        const int mcap = e->group_resident && e->persist_mode && e->persist_ok && e->chunk_iters > 1 ? group_multi_capacity(e, g.spl) : 0;
        if (mcap > 0 && ep.n_ggroups > 0 && ep.n_ggroups <= mcap) {
            int n = g.iter_hint[hp] > it ? g.iter_hint[hp] - it + RESIDENT_MARGIN : 2 * e->chunk_iters;
            n = std::max(2, std::min(n, GMULTI_MAX));
            if (e->chunk_explicit) n = std::min(n, std::max(e->chunk_iters, 2)); // (tests: launches that stop mid-loop and are resumed)
            HIP_TRY(hipMemsetAsync(g.mlog, 0, sizeof(int) * (size_t)(n + 2) * GWM, e->stream));
            HIP_TRY(hipMemsetAsync(e->bar, 0, sizeof(GridBar), e->stream));
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[0], e->stream));
            int *status = g.mlog, *rows = g.mlog + GWM;
#define DPPR_LAUNCH_GMULTI(SPL, GW, NVX)                                                                                 \
    hipLaunchKernelGGL((k_gsweep<SPL, GW, NVX, true, 2>), dim3(ep.n_ggroups), dim3(GNT), 0, e->stream, ep.grp_n_int, ep.gtab,   \
                       ep.n_ggroups, g.cnt + cur * GWM, e->gsweep_hot_rows, ep.out_col, g.x, g.x2, g.act[0], g.act[1], g.r, g.p, \
                       g.cnt + 3 * GWM, g.cnt + 4 * GWM, phase, eps, g.dstats + 1, rows, n, e->bar, status, e->persist_ticks,    \
                       e->persist_rollcall_extra, owed ? 1 : 0, (int *)nullptr, (int *)nullptr)
            with_row(g.gw, [&](auto spl, auto gw) {
                constexpr int SPL = decltype(spl)::value, GW = decltype(gw)::value;
                if constexpr (SPL == 2) DPPR_LAUNCH_GMULTI(2, GW, 512);
                else if (nvx == 512) DPPR_LAUNCH_GMULTI(1, GW, 512); // (a narrow group on an engine that also has a wide one)
                else DPPR_LAUNCH_GMULTI(1, GW, 1024);
            });
#undef DPPR_LAUNCH_GMULTI
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[1], e->stream));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(e->pinned, g.mlog, sizeof(int) * (size_t)(n + 2) * GWM, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(loop_wait(e));
            const int st = e->pinned[0];
            g.st.persist_launches++;
            if (st & GSM_FAULT) return fail(e, DPPR_ERR_HIP, "a grid barrier of the multi-sweep group launch timed out");
            if (st & GSM_ABORTED) { // not co-resident: nothing was changed; one-sweep launches from here on (re-armed later)
                g.st.persist_aborts++;
                e->persist_ok = false;
                e->persist_retry = PERSIST_RETRY_BATCHES;
                continue;
            }
            const int sweeps = st & GSM_SWEEPS;
            for (int k = 0; k < sweeps; ++k) {
                const int *f = e->pinned + GWM + k * GWM;
                g.st.iterations++;
                g.st.pull_iterations++;
                for (int s = 0; s < GWM; ++s) g.st.sum_F += f[s];
                for (int s = 0; s < GWM; ++s) g.st.sweep_F += f[s];
                active_iters = it + k + 1;
            }
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
                g.st.push_ms += ms;
                g.st.push_launches++;
            }
            if (sweeps & 1) {
                std::swap(g.x, g.x2);
                std::swap(g.act[0], g.act[1]);
                owed = !owed;
            }
            it += sweeps;
            if (st & GSM_CONVERGED) break;
            // out of sweeps: the live frontier sizes are in row `sweeps`; the launch left them in cnt[3] -- make them cnt[0]
            HIP_TRY(hipMemcpyAsync(g.cnt, g.cnt + 3 * GWM, sizeof(int) * GWM, hipMemcpyDeviceToDevice, e->stream));
            HIP_TRY(hipMemsetAsync(g.cnt + GWM, 0, sizeof(int) * 2 * GWM, e->stream));
            cur = 0;
            more = any_left(e->pinned + GWM + sweeps * GWM);
            continue;
        }
        // One-sweep launches are enqueued in chunks; a launch that finds every frontier empty returns at once, but it
        // still costs a dispatch (~4 us + gap). Consecutive batches take about the same number of sweeps, so the first
        // chunk is the SHORTEST of the last four loops of this phase (almost surely needed in full), and what follows
        // doubles from 4: a boundary (read-back + relaunch) costs about three empty dispatches.
        int n;
        if (it == 0) {
            int lo = 0;
            for (int h : (push_thr > 0 ? g.dense_hist : g.iter_hist)[hp]) lo = h > 0 && (lo == 0 || h < lo) ? h : lo;
            n = lo > 0 ? lo : e->chunk_iters;
            follow = 4;
        } else if (push_thr > 0 && !push_gave_up) {
            // the push form takes over below push_thr pairs and a sweep of the tail costs its floor whatever it finds: go
            // only as far as the frontier is sure to stay above the threshold (it shrinks by <= ~4x per sweep down there)
            long long F = 0;
            for (int s = 0; s < GWM; ++s) F += e->pinned[cur * GWM + s];
            n = 1;
            for (long long f = F / 4; f > push_thr && n < e->chunk_iters; f /= 4) ++n;
        } else {
            n = std::min(follow, e->chunk_iters);
            follow *= 2;
        }
        if (e->chunk_explicit) n = std::min(n, std::max(e->chunk_iters, 1));
        n = std::max(1, std::min(n, MAX_CHUNK));
        for (int k = 0; k < n; ++k) {
            const int nxt = (cur + 1) % 3, zer = (cur + 2) % 3;
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[2 * k], e->stream));
#define DPPR_LAUNCH_GSWEEP(SPL, GW, NVX) do { if (owed) DPPR_LAUNCH_GSWEEP_CM(SPL, GW, NVX, 1); else DPPR_LAUNCH_GSWEEP_CM(SPL, GW, NVX, 0); } while (0)
#define DPPR_LAUNCH_GSWEEP_CM(SPL, GW, NVX, CM)                                                                          \
    hipLaunchKernelGGL((k_gsweep<SPL, GW, NVX, false, CM>), dim3(sweep_grid), dim3(GNT), 0, e->stream, ep.grp_n_int, ep.gtab,  \
                       ep.n_ggroups, g.cnt + cur * GWM, e->gsweep_hot_rows, ep.out_col, g.x, g.x2, g.act[0], g.act[1], g.r, g.p,  \
                       g.cnt + nxt * GWM, g.cnt + zer * GWM, phase, eps, g.dstats + 1, log + k * GWM, 1, (GridBar *)nullptr,      \
                       (int *)nullptr, 0ull, 0, owed ? 1 : 0, g.gq + (g.gq_seq % 3) * GQ_PAD, g.gq + ((g.gq_seq + 1) % 3) * GQ_PAD)
            with_row(g.gw, [&](auto spl, auto gw) {
                constexpr int SPL = decltype(spl)::value, GW = decltype(gw)::value;
                if constexpr (SPL == 2) DPPR_LAUNCH_GSWEEP(2, GW, 512);
                else if (nvx == 512) DPPR_LAUNCH_GSWEEP(1, GW, 512);
                else DPPR_LAUNCH_GSWEEP(1, GW, 1024);
            });
#undef DPPR_LAUNCH_GSWEEP
#undef DPPR_LAUNCH_GSWEEP_CM
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[2 * k + 1], e->stream));
            std::swap(g.x, g.x2);
            std::swap(g.act[0], g.act[1]);
            cur = nxt;
            g.gq_seq++;
            owed = !owed; // (if the frontier emptied on the way, the later launches do nothing and nothing is owed: `more` is false below)
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(e->pinned, g.cnt, sizeof(int) * (size_t)(5 * GWM + n * GWM), hipMemcpyDeviceToHost,
                               e->stream));
        HIP_TRY(loop_wait(e));
        for (int k = 0; k < n; ++k) {
            const int *f = e->pinned + 5 * GWM + k * GWM;
            if (!any_left(f)) continue;
            if (push_thr > 0 && dense_len < 0) { // (the sweep that FOUND the frontier this small could have been a push iteration)
                long long F = 0;
                for (int s = 0; s < GWM; ++s) F += f[s];
                if (F <= push_thr) dense_len = it + k;
            }
            g.st.iterations++;
            g.st.pull_iterations++;
            for (int s = 0; s < GWM; ++s) g.st.sum_F += f[s];
            for (int s = 0; s < GWM; ++s) g.st.sweep_F += f[s];
            active_iters = it + k + 1;
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[2 * k], e->evpool[2 * k + 1]));
                g.st.push_ms += ms;
                g.st.push_launches++;
                g.st.sweep_ms += ms;
                g.st.sweep_launches++;
                static const bool trace = getenv("DPPR_GROUP_TRACE") != nullptr; // (diagnostic: one line per sweep)
                if (trace) {
                    long long F = 0;
                    for (int s = 0; s < GWM; ++s) F += f[s];
                    fprintf(stderr, "[gsweep] phase %d sweep %3d  frontier pairs %9lld  %7.1f us\n", phase, it + k, F, ms * 1e3);
                }
            }
        }
        more = any_left(e->pinned + cur * GWM);
        it += n;
        if (more && push_thr > 0 && !push_gave_up) {
            long long F = 0;
            for (int s = 0; s < GWM; ++s) F += e->pinned[cur * GWM + s];
            if (F <= push_thr) {
                if (dense_len < 0) dense_len = it;
                int pushed = 0;
                bool entered = false, conv = false;
                int rc = group_push_tail(e, g, ep, phase, eps, F, &pushed, &entered, &conv, &owed);
                if (rc) return rc;
                if (entered) {
                    active_iters = it + pushed;
                    it += pushed;
                    if (conv) more = false;
                    else { // back in sweep form: frontier sizes in row 0; the next try waits for a much smaller frontier
                        cur = 0;
                        HIP_TRY(hipMemcpyAsync(e->pinned, g.cnt, sizeof(int) * GWM, hipMemcpyDeviceToHost, e->stream));
                        HIP_TRY(loop_wait(e));
                        more = any_left(e->pinned);
                        push_thr = std::max<long long>(F / 8, 1);
                        dense_len = -1;
                    }
                } else { // (the frontier did not fit the lists)
                    push_thr = std::max<long long>(F / 8, 1);
                    dense_len = -1;
                }
            }
        }
    }
    if (push_thr > 0) {
        for (int k = 3; k > 0; --k) g.dense_hist[hp][k] = g.dense_hist[hp][k - 1];
        g.dense_hist[hp][0] = dense_len >= 0 ? std::max(dense_len, 1) : std::max(active_iters, 1);
    }
    g.iter_hint[hp] = active_iters;
    for (int k = 3; k > 0; --k) g.iter_hist[hp][k] = g.iter_hist[hp][k - 1];
    g.iter_hist[hp][0] = active_iters;
    return DPPR_OK;
}

int group_stream_update(dppr_engine *e, Group &g, const Epoch &ep) {
    const int L = ep.L;
    if (L == 0) return DPPR_OK;
    int rc = group_records_by_tail(e, ep, nullptr, 0, nullptr, 0);
    if (rc) return rc;
    SuSources srcs{};
    for (int s = 0; s < GS_MAX; ++s) srcs.s[s] = g.src.s[s];
    // blockIdx.y = source lane; state element (v, lane) at base[v * gw + lane]
    hipLaunchKernelGGL(k_su_terms, dim3(grid_for(L), g.n), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), ep.b2,
                       ep.ins, L, g.p, g.gw, e->su_term, e->su_ins);
    hipLaunchKernelGGL(k_su_apply, dim3(grid_for(L), g.n), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), e->su_term,
                       e->su_ins, ep.deg_after, L, g.r, g.gw, srcs, 0.0, (int *)nullptr, (int *)nullptr, (int *)nullptr,
                       (int *)nullptr);
    HIP_TRY(hipGetLastError());
    g.st.records += (int64_t)L * g.n;
    return DPPR_OK;
}

} // namespace
