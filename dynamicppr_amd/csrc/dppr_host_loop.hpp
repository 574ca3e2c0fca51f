// dppr_host_loop.hpp -- host side, part 3 of 4: the single-source SOLVER. IncrementalBatchUpdate's grouping and replay
// (gpu/StreamUpdate.cuh:7-76), the frontier loop of PPRRevPushGPU::ExecuteOptimized (gpu/PPRRevPushGPU.cuh:97-131) with its launch
// forms -- push iterations, per-iteration sweeps (gather or binned), resident launches of a run of sweeps or of a whole batch --
// and the policies that choose between them. Runs on the engine's solver stream.
#pragma once

namespace {

int read_count(dppr_engine *e, const int *dptr, int *out) {
    HIP_TRY(hipMemcpyAsync(e->pinned, dptr, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(loop_wait(e));
    *out = e->pinned[0];
    return DPPR_OK;
}

// Frontier loop: PPRRevPushGPU::ExecuteOptimized's while(1) (gpu/PPRRevPushGPU.cuh:106-130).
// On entry s.ft[buf] holds the frontier and s.cnt[cur] its size; cnt[(cur+1)%3] is zero and
// the dense vectors s.x / s.x2 are all zero (no snapshot taken yet) -- unless `entry` says
// otherwise.
//
// The reference reads the frontier count back after EVERY iteration (blocking 4-byte D2H,
// :107). Here iterations are enqueued in CHUNKS: every kernel takes F from device memory,
// rotates the three counters itself and exits at once when F == 0, so the host only reads
// the count (and the per-iteration log of F) once per chunk. The host also picks, per chunk,
// how the iterations are evaluated: SPARSE (push kernels, atomics) or DENSE (pull sweep, no
// atomics; as ONE resident launch for the whole chunk when the epoch's sweep groups fit the chip,
// dppr_resident.hpp) -- the same sums either way.
//
// `entry` describes a loop that is picked up in the middle (after a launch of batch_ahead that
// ended before the loop did): iterations already done, the frontier size if the host knows it,
// and whether s.x already holds the frontier's dense snapshot.
struct LoopEntry {
    int it = 0;
    int F = -1; // -1: read cnt[cur]
    bool dense = false;
    bool any_pull = false;
};

int pull_min_frontier(const dppr_engine *e) {
    return e->pull_min_frontier > 0 ? e->pull_min_frontier : e->pull_min_frontier < 0 ? 0x7fffffff : std::max(1024, e->Ed / 192);
}

int run_frontier_loop(dppr_engine *e, Slot &s, const Epoch &ep, int phase, double eps, int buf, int cur,
                      LoopEntry entry = LoopEntry()) {
    const int hp = phase == PHASE_BOTH ? 0 : phase; // (loop histories: the merged loop uses slot 0)
    const int pull_min = pull_min_frontier(e);
    const bool sync_sched = e->schedule == DPPR_SCHEDULE_SYNC;
    const HubTable hubs{ep.hub_v, ep.hub_degp1, ep.n_hubs};
    // the sparse grid must cover the largest frontier a push chunk can meet
    const int push_grid = pull_min == 0x7fffffff ? 2048 : std::min(2048, std::max(64, (pull_min * 4 / WAVE + 3) / 4));
    // Sweeps on a window that cannot run resident carry the activity bitmap of their snapshot (k_pull_iter<.., true>)
    const int pcap0 = persist_capacity(e);
    const bool binned = ep.bin_valid && ep.bin_n_int <= ep.grp_n_int && (pcap0 <= 0 || ep.n_groups > pcap0 || e->bin_mode == 2);
    const bool use_bits = e->sweep_bits && !binned && !entry.dense && (pcap0 <= 0 || ep.n_groups > pcap0);
    // The merged loop always filters through the status array: adds of both signs can take a residual across the threshold more
    // than once per iteration, and with the crossing test every crossing would append -- the next-frontier list (V entries) could
    // overflow. One entry per vertex and launch keeps it bounded.
    const bool use_status = e->status_dedup || phase == PHASE_BOTH;
    if (use_status && !s.status) { // (first use: -1 everywhere = "never queued")
        HIP_TRY(hipMalloc((void **)&s.status, sizeof(int) * (size_t)e->V));
        HIP_TRY(hipMemsetAsync(s.status, 0xff, sizeof(int) * (size_t)e->V, e->stream));
    }
    bool extracted = false;         // ... and that snapshot zeroed the residuals it took (InspectExtra): the push needs no repair
    bool dense_valid = entry.dense; // s.x holds the snapshot of the current frontier (p already updated)
    bool list_valid = !entry.dense; // s.ft[buf] holds the frontier as a list (sweeps only count it)
    bool any_pull = entry.any_pull;
    bool x_clean = false;           // a resident launch ended the loop and left s.x / s.x2 all zero
    auto make_list = [&]() -> int { // dense snapshot -> sparse list (after a sweep)
        HIP_TRY(hipMemsetAsync(s.cnt + 7, 0, sizeof(int), e->stream));
        hipLaunchKernelGGL(k_list_from_dense, dim3(grid_for(ep.grp_n_int, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream,
                           s.x, ep.grp_n_int, s.cnt + cur, s.ft[buf], s.cnt + 7);
        HIP_TRY(hipGetLastError());
        list_valid = true;
        return DPPR_OK;
    };
    int F = entry.F, prevF = 0, active_iters = entry.it;
    long long D = -1; // in-edges of the current frontier (binned windows), -1 = not counted
    unsigned long long *dsum = reinterpret_cast<unsigned long long *>(s.cnt + 8); // three slots beside the rotating counters
    int follow = 4; // size of the next follow-up chunk of per-iteration sweeps
    int rc = DPPR_OK;
    if (F < 0 && (rc = read_count(e, s.cnt + cur, &F))) return rc;
    if (entry.it == 0) {
        s.start_dense[hp] = F >= pull_min;
        s.last_F0[hp] = F;
    }
    for (int it = entry.it; F > 0;) {
        if (it >= e->max_iters) return fail(e, DPPR_ERR_NOT_CONVERGED, "iteration cap hit");
        if (s.trace) {
            if (!list_valid && (rc = make_list())) return rc;
            size_t old = s.trace_ids.size();
            s.trace_ids.resize(old + (size_t)F);
            HIP_TRY(hipMemcpyAsync(s.trace_ids.data() + old, s.ft[buf], sizeof(int) * (size_t)F,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(loop_wait(e));
            for (size_t i = old; i < s.trace_ids.size(); ++i) s.trace_ids[i] = e->int2ext[(size_t)s.trace_ids[i]];
            s.trace_off.push_back((int64_t)s.trace_ids.size());
        }
        bool pull = F >= pull_min;
        // a window whose iterations cost hundreds of microseconds and more (twitter / friendster size): decisions per iteration
        const bool costly = binned && !e->chunk_explicit && (s.sweep_us > 0 ? s.sweep_us : 6.5e-6 * (double)ep.Ed) >= 300.0;
        if (costly && e->cost_model && !sync_sched && !s.trace && e->pull_min_frontier == 0) {
            // Push or sweep by what each would cost. A push is one returning atomic per in-edge of the
            // frontier, executed at the memory side at ~23.5 G/s chip-wide whatever the locality (profiles/r03_atomics_probe.json);
            // a sweep of this window costs what the last ones did. The frontier's in-edges are counted by the sweep that left it
            // (k_bin_reduce) or, for a list, by k_front_degree. (Round 3 switched on the vertex count: a late frontier of 1.7 M
            // low-degree vertices is pushed in 0.23 ms and was swept for 2.6, the 156 K batch tails -- hubs -- cost a sweep's time.)
            if (D < 0 && F >= 1024) {
                if (!list_valid && (rc = make_list())) return rc;
                HIP_TRY(hipMemsetAsync(dsum + cur, 0, sizeof(unsigned long long), e->stream));
                hipLaunchKernelGGL(k_front_degree, dim3(grid_for(F)), dim3(BLOCK), 0, e->stream, s.ft[buf], s.cnt + cur, ep.row_ptr, dsum + cur);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(e->pinned, dsum + cur, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(loop_wait(e));
                unsigned long long d;
                memcpy(&d, e->pinned, sizeof(d));
                D = (long long)d;
            }
            if (D >= 0 || F < 1024) {
                const double sweep_us = s.sweep_us > 0 ? s.sweep_us : 6.5e-6 * (double)ep.Ed; // (no sweep timed yet: ~6.5 ps per edge)
                const double push_us = 15.0 + (double)std::max<long long>(D, 0) * s.atomic_ns * 1e-3; // (measured on this slot's own pushes)
                pull = F >= 1024 && push_us > 0.9 * sweep_us;
            }
        }
        int n;
        if (s.trace || e->chunk_iters <= 1) n = 1;
        else if (costly)
            // A window on binned sweeps: an iteration costs milliseconds (friendster stand-in: 2.6 ms a sweep, 11-13 ms the push
            // of a 3-10 M-vertex frontier), a read-back tens of microseconds. Nothing is enqueued blind: round 3 ran the second
            // iteration of every loop as a push of ten million vertices (decided at 156 K) and ended every loop with three to
            // seven sweeps over frontiers of a few hundred vertices (enqueued from the last batches' lengths) -- 40 of 183 ms.
            n = (!pull && F < 4096 && F <= prevF) ? e->chunk_iters : 1;
        else if (pull) // consecutive batches take almost the same number of iterations: aim just past the end
            n = s.iter_hint[hp] > it ? s.iter_hint[hp] - it + 1 : e->chunk_iters;
        else if ((long long)F * 4 >= pull_min) n = 1;          // about to turn dense: re-decide next iteration
        else n = F > prevF ? 2 : e->chunk_iters;                // growing: short chunks; decaying tail: long
        const int pcap = persist_capacity(e);
        const bool resident = pull && n >= 2 && !s.trace && pcap > 0 && ep.n_groups > 0 && ep.n_groups <= pcap && resident_arena(e, ep);
        if (resident && s.iter_hint[hp] > it) n += RESIDENT_MARGIN - 1;
        if (!resident && pull && n > 1) {
            // per-iteration sweeps: a launch that finds the frontier empty is still a dispatch, a chunk boundary (read-back
            // + relaunch) costs about three of them -- go as far as the SHORTEST of the last four loops of this phase went
            // (almost surely needed in full), then in chunks that double from 4 (group_loop sizes its chunks the same way)
            int lo = 0;
            for (int h : s.iter_hist[hp]) lo = h > 0 && (lo == 0 || h < lo) ? h : lo;
            if (lo > it) n = lo - it;
            else if (lo > 0) {
                n = std::min(follow, e->chunk_iters);
                follow *= 2;
            }
        }
        n = std::min(n, MAX_CHUNK);
        if (e->chunk_explicit) n = std::min(n, std::max(e->chunk_iters, 1));
        if (!pull && !list_valid && (rc = make_list())) return rc;
        if (resident) {
            // ---- a run of dense iterations as ONE resident launch (dppr_resident.hpp)
            if (!dense_valid) {
                hipLaunchKernelGGL(k_snapshot_dense, dim3(std::min(grid_for(std::max(F, 1 << 14)), 1024)), dim3(BLOCK), 0,
                                   e->stream, s.ft[buf], s.cnt + cur, s.r, s.p, s.x, (uint32_t *)nullptr, phase == PHASE_BOTH ? 1 : 0, 0);
                dense_valid = true;
            }
            HIP_TRY(hipMemsetAsync(e->bar, 0, sizeof(GridBar), e->stream));
            n = std::min(n, RES_MAX_SWEEPS);
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[0], e->stream));
#define DPPR_LAUNCH_PERSIST(PB)                                                                                       \
    hipLaunchKernelGGL(k_pull_resident<PB>, dim3(ep.n_groups), dim3(PB), 0, e->stream, ep.grp_n_int, ep.grp_tile,       \
                       ep.out_row_ptr, ep.out_col, s.x, e->res_arena, e->res_arena_stride, s.r, s.p, s.cnt, cur, phase, eps, s.dstats, s.log,  \
                       n, e->bar, s.cnt + 7, e->persist_ticks, e->persist_rollcall_extra, 0,                          \
                       ep.res_valid ? ep.res_pk : nullptr, ResUpdate{})
            switch (sweep_block(e)) {
            case 256: DPPR_LAUNCH_PERSIST(256); break;
            case 512: DPPR_LAUNCH_PERSIST(512); break;
            default: DPPR_LAUNCH_PERSIST(1024); break;
            }
#undef DPPR_LAUNCH_PERSIST
            if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[1], e->stream));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(e->pinned, s.cnt, sizeof(int) * (size_t)(CNT_HDR + n), hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(loop_wait(e));
            const int status = e->pinned[7];
            s.st.persist_launches++;
            if (status & PERSIST_FAULT) return fail(e, DPPR_ERR_HIP, "grid barrier of the resident sweep timed out");
            if (status & PERSIST_ABORTED) {
                // the roll-call failed (the grid was not co-resident): nothing was changed; this engine
                // goes on with per-iteration launches
                s.st.persist_aborts++;
                e->persist_ok = false;
                e->persist_retry = PERSIST_RETRY_BATCHES;
                continue;
            }
            for (int k = 0; k < n; ++k) {
                const int f = e->pinned[CNT_HDR + k];
                if (f <= 0) continue;
                s.st.iterations++;
                s.st.pull_iterations++;
                s.st.sum_F += f;
                active_iters = it + k + 1;
            }
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
                s.st.push_ms += ms;
                s.st.push_launches++;
            }
            // (s.x holds the snapshot the last sweep wrote)
            cur = 0;                              // the launch leaves the live count in cnt[0]
            list_valid = false;
            any_pull = true;
            x_clean = (status & PERSIST_CONVERGED) != 0;
            prevF = F;
            F = e->pinned[0];
            it += n;
            continue;
        }
        for (int k = 0; k < n; ++k) {
            const int nxt = (cur + 1) % 3, zer = (cur + 2) % 3;
            int *log_slot = s.log + k;
            if ((pull || sync_sched) && !dense_valid) {
                // grid-stride over a frontier whose size is only known on the device (k > 0): sized for
                // the last size the host saw, capped
                const bool bm = use_bits && pull;
                if (bm) HIP_TRY(hipMemsetAsync(s.act[0], 0, s.act_bytes, e->stream));
                extracted = e->pre_extract && !pull; // (a sweep repairs by itself: rn -= x[v])
                hipLaunchKernelGGL(k_snapshot_dense, dim3(std::min(grid_for(std::max(F, 1 << 14)), 1024)), dim3(BLOCK), 0,
                                   e->stream, s.ft[buf], s.cnt + cur, s.r, s.p, s.x, bm ? s.act[0] : (uint32_t *)nullptr, phase == PHASE_BOTH ? 1 : 0,
                                   extracted ? 1 : 0);
                dense_valid = true;
            }
            if (costly) HIP_TRY(hipMemsetAsync(dsum + nxt, 0, sizeof(unsigned long long), e->stream));
            const bool timed = e->profiling || (costly && n == 1); // (the push / sweep decision prices both by what the last ones took)
            if (timed) HIP_TRY(hipEventRecord(e->evpool[2 * k], e->stream));
            if (pull && binned) {
                // the sweep as two streaming passes over the epoch's binned edge layout (dppr_binned.hpp)
                if (ep.n_chunks > 0)
                    hipLaunchKernelGGL(k_bin_scatter, dim3(ep.n_chunks), dim3(BIN_NT), (size_t)e->bin_ha_tiles * WAVE * sizeof(double), e->stream,
                                       ep.bin_n_int, s.cnt + cur, ep.acut, ep.chunks, ep.hl, ep.tb, ep.tdelta, ep.n_runs, s.x, e->bin_vals);
                const int rows_cap = e->bin_hb_tiles * WAVE;
                hipLaunchKernelGGL(k_bin_reduce, dim3(ep.n_b + (ep.grp_n_int - ep.bin_n_int + rows_cap - 1) / rows_cap), dim3(BIN_NT),
                                   (size_t)rows_cap * 20, e->stream, ep.grp_n_int, ep.bin_n_int, ep.n_b, s.cnt + cur, ep.bcut, rows_cap,
                                   ep.out_row_ptr, ep.dl, ep.vb, ep.Ed, e->bin_vals, s.x,
                                   s.x2, s.r, s.p, s.cnt + nxt, s.cnt + zer, phase, eps, s.dstats + 1, log_slot, e->directed ? ep.row_ptr : (const int *)nullptr,
                                   costly ? dsum + nxt : (unsigned long long *)nullptr);
                std::swap(s.x, s.x2);
                dense_valid = true;
                extracted = false;
                list_valid = false;
                any_pull = true;
            } else if (pull) {
                // workgroup size = max tiles per group x 64 (the groups themselves were cut by the builder)
                const int pb = sweep_block(e);
#define DPPR_LAUNCH_PULL(PB, BITS)                                                                                    \
    hipLaunchKernelGGL((k_pull_iter<PB, BITS>), dim3(std::min(std::max(ep.n_groups, 1), 1024)), dim3(PB), 0, e->stream, \
                       ep.grp_n_int, ep.grp_tile, ep.n_groups, s.cnt + cur, ep.out_row_ptr, ep.out_col, s.x, s.x2, s.r, \
                       s.p, s.cnt + nxt, s.cnt + zer, phase, eps, s.dstats + 1, log_slot,                               \
                       std::min(e->big_row, PULL_BIG_ROW_DEFAULT), s.act[0], s.act[1])
                if (use_bits) {
                    switch (pb) {
                    case 256: DPPR_LAUNCH_PULL(256, true); break;
                    case 384: DPPR_LAUNCH_PULL(384, true); break;
                    case 512: DPPR_LAUNCH_PULL(512, true); break;
                    case 576: DPPR_LAUNCH_PULL(576, true); break;
                    case 640: DPPR_LAUNCH_PULL(640, true); break;
                    case 768: DPPR_LAUNCH_PULL(768, true); break;
                    case 896: DPPR_LAUNCH_PULL(896, true); break;
                    default: DPPR_LAUNCH_PULL(1024, true); break;
                    }
                    std::swap(s.act[0], s.act[1]);
                } else { // (block sizes that are not 256 / 512 / 1024 never run resident: they always take the form above)
                    switch (pb) {
                    case 256: DPPR_LAUNCH_PULL(256, false); break;
                    case 512: DPPR_LAUNCH_PULL(512, false); break;
                    default: DPPR_LAUNCH_PULL(1024, false); break;
                    }
                }
#undef DPPR_LAUNCH_PULL
                std::swap(s.x, s.x2); // the sweep wrote every entry of x2: it is the next snapshot
                dense_valid = true;
                extracted = false;
                list_valid = false;
                any_pull = true;
            } else {
                int *big_cnt = s.cnt + 5 + (int)(s.iter_seq & 1), *big_zero = s.cnt + 5 + (int)((s.iter_seq + 1) & 1);
                s.iter_seq++;
                const Dedup dd{use_status ? s.status : nullptr, (int)(s.iter_seq & 0x3fffffff)};
                if (dense_valid)
                    hipLaunchKernelGGL(k_push_iter<true>, dim3(push_grid), dim3(BLOCK), 0, e->stream, s.ft[buf],
                                       s.cnt + cur, s.ft[buf ^ 1], s.cnt + nxt, s.cnt + zer, s.x, ep.row_ptr, ep.adj, hubs,
                                       s.big, big_cnt, big_zero, e->big_row, s.r, s.p, phase, eps, s.dstats, log_slot, dd, extracted ? 1 : 0);
                else
                    hipLaunchKernelGGL(k_push_iter<false>, dim3(push_grid), dim3(BLOCK), 0, e->stream, s.ft[buf],
                                       s.cnt + cur, s.ft[buf ^ 1], s.cnt + nxt, s.cnt + zer, s.x, ep.row_ptr, ep.adj, hubs,
                                       s.big, big_cnt, big_zero, e->big_row, s.r, s.p, phase, eps, s.dstats, log_slot, dd, 0);
                hipLaunchKernelGGL(k_push_big, dim3(512), dim3(BLOCK), 0, e->stream, s.big, big_cnt, s.ft[buf ^ 1],
                                   s.cnt + nxt, ep.adj, hubs, s.r, phase, eps, s.dstats, dd);
                dense_valid = false; // the push consumed (and zeroed) the snapshot
                extracted = false;
                list_valid = true;
            }
            if (timed) HIP_TRY(hipEventRecord(e->evpool[2 * k + 1], e->stream));
            buf ^= 1;
            cur = nxt;
        }
        HIP_TRY(hipGetLastError());
        // one read-back per chunk: the new frontier size and the F of each iteration just run
        HIP_TRY(hipMemcpyAsync(e->pinned, s.cnt, sizeof(int) * (size_t)(CNT_HDR + n), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(loop_wait(e));
        for (int k = 0; k < n; ++k) {
            const int f = e->pinned[CNT_HDR + k];
            if (f <= 0) continue; // the frontier emptied inside the chunk: the rest were no-ops
            s.st.iterations++;
            s.st.sum_F += f;
            if (pull) s.st.pull_iterations++;
            if (pull) s.st.sweep_F += f;
            if (pull && binned) s.st.binned_sweeps++;
            active_iters = it + k + 1;
            if (e->profiling) {
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e->evpool[2 * k], e->evpool[2 * k + 1]));
                s.st.push_ms += ms;
                s.st.push_launches++;
                if (pull) {
                    s.st.sweep_ms += ms;
                    s.st.sweep_launches++;
                }
                static const bool trace = getenv("DPPR_LOOP_TRACE") != nullptr; // (diagnostic: one line per iteration of a profiled batch)
                if (trace)
                    fprintf(stderr, "[loop  ] phase %d iteration %3d  %-6s frontier %9d  %8.1f us\n", phase, it + k,
                            pull ? (binned ? "binned" : "sweep") : "push", f, ms * 1e3);
            }
        }
        if (costly && n == 1 && e->pinned[CNT_HDR] > 0) { // what a sweep of this window costs / what an atomic of a push does (running means)
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
            if (pull) s.sweep_us = s.sweep_us > 0 ? 0.75 * s.sweep_us + 0.25 * ms * 1e3 : ms * 1e3;
            else if (D >= (1 << 20)) s.atomic_ns = 0.75 * s.atomic_ns + 0.25 * std::min(1.0, std::max(0.02, (ms * 1e6 - 15e3) / (double)D));
        }
        prevF = F;
        F = e->pinned[cur];
        if (binned && pull) { // the sweep counted the in-edges of the frontier it left
            unsigned long long d;
            memcpy(&d, e->pinned + 8 + 2 * cur, sizeof(d));
            D = (long long)d;
        } else {
            D = -1;
        }
        it += n;
    }
    s.iter_hint[hp] = active_iters;
    for (int k = 3; k > 0; --k) s.iter_hist[hp][k] = s.iter_hist[hp][k - 1];
    s.iter_hist[hp][0] = active_iters;
    if (any_pull && !x_clean) { // leave both dense vectors all-zero for the next loop
        // only internal ids below n_int are ever written
        HIP_TRY(hipMemsetAsync(s.x, 0, sizeof(double) * (size_t)ep.grp_n_int, e->stream));
        HIP_TRY(hipMemsetAsync(s.x2, 0, sizeof(double) * (size_t)ep.grp_n_int, e->stream));
    }
    return DPPR_OK;
}

// ---------------------------------------------------------------------------------------------
// Both frontier loops of one batch as ONE resident launch, without a read-back in between.
//
// When consecutive batches behave alike (both phases start with a frontier worth a sweep -- the
// steady state of a sliding-window stream), the host knows what it will launch before it has seen
// any count. After a converged solve the frontier of a phase is {v : legal(residual[v])}, which the
// resident kernel reads off its registers (PLAN_SEED), and when phase 0 is over it seeds phase 1
// the same way and goes on (PLAN_BOTH): Inspect / snapshot / phase 0 / Inspect / snapshot / phase 1
// of gpu/PPRGPU.cuh:138-164 are one kernel. One copy of the counters, the status word and the log
// comes back at the end. Whatever did not go as expected (a phase needed more sweeps than the
// launch was given, the roll-call failed) leaves the state at a well-defined point from which the
// ordinary host-driven loop resumes (`stage`, `en0`, `en1`).
// The reference pays a blocking read-back per ITERATION (gpu/PPRRevPushGPU.cuh:107).
// ---------------------------------------------------------------------------------------------
bool can_batch_ahead(const dppr_engine *e, const Slot &s, const Epoch &ep) {
    const int cap = persist_capacity(e);
    if (e->persist_mode != 1 || cap <= 0 || ep.n_groups <= 0 || ep.n_groups > cap || s.trace || e->chunk_iters <= 1 || ep.L <= 0)
        return false;
    // A resident sweep costs the same ~5 us whatever the frontier size, less than one push iteration's
    // launches: with the automatic push/pull threshold a window that can run resident always does.
    // With an explicit threshold (tests) only if the last batch's phases both started above it.
    if (e->merge_phases && e->schedule == DPPR_SCHEDULE_EAGER) // (the merged loop keeps its history in slot 0)
        return e->pull_min_frontier == 0 || (s.iter_hint[0] > 0 && s.start_dense[0]);
    return e->pull_min_frontier == 0 ||
           (s.iter_hint[0] > 0 && s.iter_hint[1] > 0 && s.start_dense[0] && s.start_dense[1]);
}

// stage (out): 0 = phase 0 still open (resume with en0), 1 = phase 0 done, phase 1 open (resume with
// en1; *p1_seeded tells whether its snapshot exists), 2 = both phases done
int batch_ahead(dppr_engine *e, Slot &s, const Epoch &ep, double eps, int *stage, LoopEntry *en0, LoopEntry *en1,
                bool *p1_seeded, bool merged = false, bool inline_update = false) {
    // merged (dppr_set_phase_merge): ONE loop over residuals of both signs -- the launch seeds it (PLAN_SEED) and runs it to the
    // end; stage 0 + en0 if it ran out of sweeps, stage 2 when it converged (histories in slot 0)
    const int pull_min = pull_min_frontier(e);
    // a resident launch stops by itself when the frontier empties: a generous allowance costs nothing,
    // a short one costs a read-back and another launch (+1: the step that seeds phase 1)
    int n = merged ? (s.iter_hint[0] > 0 ? std::min(s.iter_hint[0] + 2 * RESIDENT_MARGIN, 2 * MAX_CHUNK) : 2 * MAX_CHUNK)
            : s.iter_hint[0] > 0 && s.iter_hint[1] > 0
                      ? std::min(s.iter_hint[0] + s.iter_hint[1] + 1 + 2 * RESIDENT_MARGIN, 2 * MAX_CHUNK)
                      : 2 * MAX_CHUNK; // no history yet
    if (e->chunk_explicit) n = std::min(n, e->chunk_iters); // (tests: launches that stop mid-phase and are resumed)
    n = std::min(n, RES_MAX_SWEEPS);
    int *status = s.cnt + 7; // (the GridBar was zeroed by the batch's first kernel, k_su_keys)
    const ResUpdate upd = !inline_update ? ResUpdate{}
                          : ep.grouped   ? ResUpdate{ep.su_rng, ep.sk, ep.sv, ep.b2, ep.ins, ep.deg_after, s.source, nullptr, 0}
                                         : ResUpdate{nullptr, nullptr, nullptr, ep.b2, ep.ins, nullptr, s.source, ep.b1, ep.L}; // raw records
    const int plan = (merged ? PLAN_SEED : (PLAN_SEED | PLAN_BOTH)) | (inline_update ? PLAN_UPDATE : 0);
    if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[0], e->stream));
#define DPPR_LAUNCH_PERSIST(PB)                                                                                        \
    hipLaunchKernelGGL(k_pull_resident<PB>, dim3(ep.n_groups), dim3(PB), 0, e->stream, ep.grp_n_int, ep.grp_tile,        \
                       ep.out_row_ptr, ep.out_col, s.x, e->res_arena, e->res_arena_stride, s.r, s.p, s.cnt, 0,                  \
                       merged ? PHASE_BOTH : 0, eps, s.dstats,                                                             \
                       s.log, n, e->bar, status, e->persist_ticks, e->persist_rollcall_extra,                             \
                       plan, ep.res_valid ? ep.res_pk : nullptr, upd)
    switch (sweep_block(e)) {
    case 256: DPPR_LAUNCH_PERSIST(256); break;
    case 512: DPPR_LAUNCH_PERSIST(512); break;
    default: DPPR_LAUNCH_PERSIST(1024); break;
    }
#undef DPPR_LAUNCH_PERSIST
    if (e->profiling) HIP_TRY(hipEventRecord(e->evpool[1], e->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e->pinned, s.cnt, sizeof(int) * (size_t)(CNT_HDR + n), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(loop_wait(e));

    const int st = e->pinned[7];
    *stage = 0;
    *p1_seeded = false;
    *en0 = LoopEntry();
    *en1 = LoopEntry();
    s.st.persist_launches++;
    if (st & PERSIST_FAULT) return fail(e, DPPR_ERR_HIP, "a wait inside the resident sweep timed out");
    e->launch_called_off = false;
    if ((st & PERSIST_ABORTED) && inline_update && !ep.grouped && e->pinned[4] == 1) {
        // a sweep group owns more of the batch's records than it has threads: the launch called itself off before anything was
        // changed -- not a residency problem. The caller applies the update with its own kernels; the next batches do so at once.
        e->launch_called_off = true;
        e->raw_backoff = 16;
        return DPPR_OK;
    }
    if (st & PERSIST_ABORTED) { // roll-call failed: nothing was changed, the lists of the stream update stand
        s.st.persist_aborts++;
        e->launch_called_off = true;
        e->persist_ok = false;
        e->persist_retry = PERSIST_RETRY_BATCHES;
        return DPPR_OK;
    }
    if (e->profiling) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e->evpool[0], e->evpool[1]));
        s.st.push_ms += ms;
        s.st.push_launches++;
    }
    // the log: frontier sizes of phase 0, a zero (phase 0 over), those of phase 1, a zero
    const int *log = e->pinned + CNT_HDR;
    const int pos = st & PERSIST_SWEEPS; // loop position the launch stopped at
    int act[2] = {0, 0}, ph = 0;
    for (int k = 0; k < std::min(pos + 1, n) && ph < 2; ++k) {
        if (log[k] <= 0) {
            ++ph;
            continue;
        }
        if (act[ph] == 0) {
            s.start_dense[ph] = log[k] >= pull_min;
            s.last_F0[ph] = log[k];
        }
        s.st.iterations++;
        s.st.pull_iterations++;
        s.st.sum_F += log[k];
        act[ph]++;
    }
    if (merged) {
        if (!(st & PERSIST_CONVERGED)) { // out of sweeps: the host-driven loop goes on from here
            en0->it = act[0];
            en0->F = e->pinned[0];
            en0->dense = true;
            en0->any_pull = true;
            return DPPR_OK;
        }
        s.iter_hint[0] = act[0];
        for (int k = 3; k > 0; --k) s.iter_hist[0][k] = s.iter_hist[0][k - 1];
        s.iter_hist[0][0] = act[0];
        if (act[0] == 0) s.start_dense[0] = false;
        *stage = 2;
        return DPPR_OK;
    }
    if (!(st & PERSIST_PHASE1)) { // phase 0 needs more sweeps than the launch had; phase 1 has not started
        en0->it = act[0];
        en0->F = e->pinned[0];
        en0->dense = true;
        en0->any_pull = true;
        return DPPR_OK;
    }
    s.iter_hint[0] = act[0];
    if (act[0] == 0) s.start_dense[0] = false;
    *stage = 1;
    *p1_seeded = true;
    if (!(st & PERSIST_CONVERGED)) {
        en1->it = act[1];
        en1->F = e->pinned[0];
        en1->dense = true;
        en1->any_pull = true;
        return DPPR_OK;
    }
    s.iter_hint[1] = act[1];
    if (act[1] == 0) s.start_dense[1] = false;
    *stage = 2;
    return DPPR_OK;
}

// full Inspect seeding + loop = ExecuteMainLoop(phase)
int main_loop_inspect(dppr_engine *e, Slot &s, const Epoch &ep, int phase, double eps) {
    s.seed_lists_valid = false;
    HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * 3, e->stream));
    hipLaunchKernelGGL(k_inspect, dim3(grid_for(ep.grp_n_int, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream, s.r,
                       ep.grp_n_int, phase, eps, s.ft[0], s.cnt + 0);
    HIP_TRY(hipGetLastError());
    s.st.inspected += ep.grp_n_int;
    return run_frontier_loop(e, s, ep, phase, eps, 0, 0);
}

// Stable grouping of the epoch's batch records by tail: su_k[1] = tails ascending, su_v[1] = record indices
// (ascending inside a tail): key extraction + the device radix sort. `zero` / `zero_ints` are cleared on the
// way (the counters of what follows).
inline const uint32_t *batch_tails(const dppr_engine *e, const Epoch &ep) { return ep.grouped ? ep.sk : e->su_k[1]; }
inline const uint32_t *batch_order(const dppr_engine *e, const Epoch &ep) { return ep.grouped ? ep.sv : e->su_v[1]; }

// The grouping is a function of the batch's records alone (not of any solver state): by default it is done once, when the batch
// is uploaded (dppr_slide -> epoch_group_records; the reference uploads its GPUEdgeBatch untimed as well, gpu/PPRGPU.cuh:131-135),
// and the timed region starts with a kernel that only clears the loop's counters. dppr_set_batch_grouping(e, 0) keeps it inside
// dppr_update (the accounting of rounds 1-2: + 5 dispatches of the device radix sort per batch).
int epoch_group_records(dppr_engine *e, Epoch &ep) {
    ep.grouped = false;
    ep.su_inline = false;
    if (!e->group_at_slide || ep.L <= 0) return DPPR_OK;
    if (ep.id >= 0) { // an epoch built while the default accounting was on, grouped outside the bracket after all (prepare_epoch): its degrees too
        hipLaunchKernelGGL(k_copy_out_degree, dim3(grid_for(ep.L)), dim3(BLOCK), 0, e->bs, ep.b1, ep.L, ep.out_row_ptr, ep.deg_after);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_su_keys, dim3(grid_for(ep.L)), dim3(BLOCK), 0, e->bs, ep.b1, ep.L, e->su_k[0], e->su_v[0],
                       (unsigned long long *)nullptr, 0, (int *)nullptr, 0);
    size_t tmp = e->su_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(e->su_tmp, tmp, e->su_k[0], ep.sk, e->su_v[0], ep.sv, (size_t)ep.L, 0u, (unsigned)e->bits, e->bs));
    ep.grouped = true;
    return res_record_ranges(e, ep);
}

// CopyOutDegree (gpu/StreamUpdate.cuh:7-17; a tail's post-batch out-degree = the length of its row in the epoch's out-CSR, written to
// `deg`) and the stable grouping of the L records by tail into su_k[1] / su_v[1], as the timed region runs them: ranked in one launch up
// to SU_RANK_MAX records, bucketed + ranked (three launches, dppr_update.hpp) up to SU_GRP_MAX_RECORDS, the device radix sort beyond.
int enqueue_grouping(dppr_engine *e, const Epoch &ep, int *deg, unsigned long long *zero, int nz, int *zero_ints, int nzi) {
    const int L = ep.L;
    if (L <= SU_RANK_MAX && !e->force_radix_grouping) {
        hipLaunchKernelGGL(k_su_group_rank, dim3((L + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, e->stream, ep.b1, L, ep.out_row_ptr, e->su_k[1],
                           e->su_v[1], deg, zero, nz, zero_ints, nzi);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    }
    if (L <= SU_GRP_MAX_RECORDS && !e->force_radix_grouping && e->su_grp) {
        // hand-written: bucket histogram (+ CopyOutDegree + the counters), unordered scatter into the buckets, ranking inside them:
        // no library sort between the bracket's events
        int nb = 64;
        while (nb < SU_GRP_MAX_BUCKETS && nb * 512 < L) nb *= 2;
        int *hist = e->su_grp, *cursor = hist + SU_GRP_MAX_BUCKETS, *ctl = cursor + SU_GRP_MAX_BUCKETS;
        const int wgs = (L + SU_GRP_PER_WG - 1) / SU_GRP_PER_WG;
        hipLaunchKernelGGL(k_su_grp_hist, dim3(wgs), dim3(BLOCK), 0, e->stream, ep.b1, L, nb, ep.out_row_ptr, deg, hist, zero, nz, zero_ints, nzi);
        hipLaunchKernelGGL(k_su_grp_scatter, dim3(wgs), dim3(BLOCK), 0, e->stream, ep.b1, L, nb, hist, cursor, ctl, e->su_k[0], e->su_v[0]);
        hipLaunchKernelGGL(k_su_grp_rank, dim3((L + BLOCK - 1) / BLOCK + nb), dim3(BLOCK), 0, e->stream, e->su_k[0], e->su_v[0], ctl, nb, e->su_k[1],
                           e->su_v[1], hist, cursor);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    }
    // (batches beyond 4 Mi records, or DPPR_GROUPING_RADIX=1: the device radix sort of rounds 1-5)
    hipLaunchKernelGGL(k_copy_out_degree, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, ep.b1, L, ep.out_row_ptr, deg);
    hipLaunchKernelGGL(k_su_keys, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, ep.b1, L, e->su_k[0], e->su_v[0], zero, nz, zero_ints, nzi);
    size_t tmp = e->su_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(e->su_tmp, tmp, e->su_k[0], e->su_k[1], e->su_v[0], e->su_v[1], (size_t)L, 0u, (unsigned)e->bits, e->stream));
    return DPPR_OK;
}

int group_records_by_tail(dppr_engine *e, const Epoch &ep, unsigned long long *zero, int nz, int *zero_ints, int nzi) {
    if (ep.grouped) { // only the counters (and the GridBar of a resident launch enqueued ahead) are cleared here
        if (nz > 0 || nzi > 0)
            hipLaunchKernelGGL(k_su_keys, dim3(1), dim3(BLOCK), 0, e->stream, ep.b1, 0, e->su_k[0], e->su_v[0], zero, nz, zero_ints, nzi);
        HIP_TRY(hipGetLastError());
        return DPPR_OK;
    }
    return enqueue_grouping(e, ep, ep.deg_after, zero, nz, zero_ints, nzi); // inside the timed region (default)
}

// dppr_set_batch_grouping(1) after epochs were built: their records are grouped now, BEFORE the caller's event bracket opens
inline int prepare_epoch(dppr_engine *e, Epoch &ep) {
    if (e->group_at_slide && !ep.grouped && ep.L > 0) {
        if (int rc = epoch_group_records(e, ep)) return rc;
        HIP_TRY(hipStreamSynchronize(e->bs)); // (the grouping ran on the builder's stream; what follows reads it on the solver's)
    }
    return DPPR_OK;
}

int stream_update(dppr_engine *e, Slot &s, const Epoch &ep, double eps, bool seed, bool zero_bars = false) {
    const int L = ep.L;
    if (L == 0) {
        HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * 5, e->stream));
        return DPPR_OK;
    }
    // (the batch's first kernel also clears cnt[0..4] and, for a resident launch enqueued ahead, its GridBar)
    int rc = group_records_by_tail(e, ep, zero_bars ? reinterpret_cast<unsigned long long *>(e->bar) : nullptr,
                                   zero_bars ? (int)(sizeof(GridBar) / sizeof(unsigned long long)) : 0, s.cnt, 5);
    if (rc) return rc;
    // without seeding the lists go to scratch space (cnt[4] / neg) and are ignored
    if (L >= SU_SPLIT_MIN) {
        // Large batches: a hub's tail owns thousands of records, and the fused kernel's leader walks what lies beyond its
        // 1 024-record LDS window through three dependent gathers per record (twitter stand-in, 2.9 M records: 3.0 ms of a batch).
        // The terms of ALL records are computed in parallel first; the leaders then walk contiguous arrays (same expressions,
        // same order: bit-identical, the form source groups use).
        SuSources srcs{};
        srcs.s[0] = s.source;
        hipLaunchKernelGGL(k_su_terms, dim3(grid_for(L), 1), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), ep.b2, ep.ins, L, s.p, 1,
                           e->su_term, e->su_ins);
        hipLaunchKernelGGL(k_su_apply, dim3(grid_for(L), 1), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), e->su_term, e->su_ins,
                           ep.deg_after, L, s.r, 1, srcs, seed ? eps : 1e300, s.ft[0], s.cnt + 0, s.neg, s.cnt + 3);
    } else {
        hipLaunchKernelGGL(k_su_apply_fused, dim3(grid_for(L)), dim3(BLOCK), 0, e->stream, batch_tails(e, ep), batch_order(e, ep), ep.b2, ep.ins,
                           ep.deg_after, L, s.p, s.r, s.source, seed ? eps : 1e300, s.ft[0], s.cnt + 0, s.neg, s.cnt + 3);
    }
    HIP_TRY(hipGetLastError());
    s.st.records += L;
    return DPPR_OK;
}

int pull_device_stats(dppr_engine *e, Slot &s) {
    static thread_local IterStats h[2];
    HIP_TRY(hipMemcpyAsync(h, s.dstats, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long t = 0, ts = 0;
    for (int i = 0; i < STAT_SLOTS; ++i) {
        t += h[0].blk_E[i];
        ts += h[1].blk_E[i];
    }
    s.st.sum_E = (int64_t)(t + ts);
    s.st.sweep_E = (int64_t)ts;
    return DPPR_OK;
}

} // namespace
