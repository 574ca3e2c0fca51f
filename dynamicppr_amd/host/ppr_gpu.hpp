// ppr_gpu.hpp -- driver classes of ./pagerank over the C ABI (include/dppr.h).
//
// PPRGPU keeps the reference's driver shape (gpu/PPRGPU.cuh:21-198): DynamicExecute(),
// SlidingWindowExecuteMainLoop() and the four virtuals GPUBuildSlidingGraph /
// IncrementalBatchUpdate / ExecuteMainLoop(phase) / ValidateResult, the same timed scope
// (gpu/PPRGPU.cuh:138-164) and the same stdout contract (:116-124, :170-176). What differs:
//   * one engine (= one device, one window-graph replica) serves SEVERAL source vertices;
//   * no CUDA types here -- everything device-side sits behind libdppr_hip.so;
//   * validation is a run-time flag (--validate) instead of -DVALIDATE;
//   * errors come back as status codes; this layer prints and exits like CUDA_ERROR did
//     (gpu/GPUUtil.cuh:7-19).
#pragma once

#include <atomic>
#include <future>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/dppr.h"
#include "graph_vec.hpp"
#include "meta.hpp"

#define DPPR_CHECK(eng, call)                                                                       \
    do {                                                                                            \
        int _rc = (call);                                                                           \
        if (_rc != DPPR_OK) {                                                                       \
            std::cout << dppr_strerror(_rc) << " (" << dppr_last_error(eng) << ") in " << __FILE__  \
                      << " at line " << __LINE__ << std::endl;                                      \
            std::exit(-1);                                                                          \
        }                                                                                           \
    } while (0)

// --profile: what the reference prints when it is compiled with -DPROFILE (ProfileCommon.h:6-31 phase and counter
// names, GPUProfiler::ReportProfile gpu/GPUProfiler.cuh:42-55, the per-iteration line of gpu/PPRRevPushGPU.cuh:109-111).
// Phases are timed on the host clock around calls that synchronise the stream, scoped as gpu/PPRGPU.cuh:47-163 scopes
// them. Inspect, Expand and RepairFrontier are ONE kernel per iteration here: their event-timed total is reported as
// expand_time, inspect_time / repair_frontier_time stay 0 (as do the phases the reference's GPU path never starts).
struct HostProfile {
    enum Phase { INSPECT, EXPAND, INIT_GRAPH_CALC, DYNA_GRAPH_CALC, EXCLUDE_GRAPH_UPDATE, SORT, REDUCE, REPAIR_FRONTIER,
                 INC_UPDATE, PUSH, TOTAL, PPR, PPR_UPDATE, PPR_QUERY, N_PHASES };
    double ms[N_PHASES] = {0};
    timespec t0[N_PHASES];
    void Start(Phase p) { clock_gettime(CLOCK_MONOTONIC, &t0[p]); }
    void End(Phase p) {
        timespec b;
        clock_gettime(CLOCK_MONOTONIC, &b);
        ms[p] += (b.tv_sec - t0[p].tv_sec) * 1e3 + (b.tv_nsec - t0[p].tv_nsec) * 1e-6;
    }
    static const char *Name(int p) {
        static const char *names[N_PHASES] = {"inspect_time", "expand_time", "init_graph_calculation_time",
                                              "dynamic_graph_calculation_time", "exclude_graph_update_time", "sort_time",
                                              "reduce_time", "repair_frontier_time", "inc_update_time", "push_time",
                                              "total_time", "ppr_time", "ppr_update_time", "ppr_query_time"};
        return names[p];
    }
};

class PPRGPU {
public:
    PPRGPU(SlidingGraphVec *g, int device, const std::vector<IndexType> &sources, bool quiet)
        : graph(g), device_id(device), source_vertex_ids(sources), quiet_(quiet) {
        if (!quiet_)
            for (IndexType s : sources) std::cout << "choose " << s << " as source vertex id" << std::endl;
        // two resident epochs: the graph of batch k + 1 is built (dppr_slide_concurrent) while batch k is being solved
        overlap_ = !gValidate && !gSplitInterface && std::getenv("DPPR_NO_OVERLAP") == nullptr;
        int rc = dppr_create(&engine, device, g->vertex_count, g->sliding_window_size, g->directed ? 1 : 0,
                             (int32_t)gStreamUpdateCountPerBatch, overlap_ ? 2 : 1);
        if (rc != DPPR_OK) {
            std::cout << "dppr_create: " << dppr_strerror(rc) << std::endl;
            std::exit(-1);
        }
        DPPR_CHECK(engine, dppr_set_variant(engine, gVariant)); // (-o: the variant's mechanisms and its schedule)
        if (gSchedule) DPPR_CHECK(engine, dppr_set_schedule(engine, gSchedule)); // (--sync on an eager variant)
        if (gPushOnly) DPPR_CHECK(engine, dppr_set_tuning(engine, 256, 512, -1, 0, 0)); // (--push-only: every iteration through the push kernels)
        if (gMergePhases) DPPR_CHECK(engine, dppr_set_phase_merge(engine, 1, 0)); // (--merge-phases: include/dppr.h; not the reference's schedule)
        ppr_time.assign(sources.size(), 0.0f);
    }
    virtual ~PPRGPU() { dppr_destroy(engine); }

    // gpu/PPRGPU.cuh:64-108
    virtual void DynamicExecute() {
        prof.Start(HostProfile::TOTAL);
        prof.Start(HostProfile::INIT_GRAPH_CALC);
        {
            EdgeBatch init_stream(graph->sliding_window_size);
            graph->SerializeEdgeStream(&init_stream);
            DPPR_CHECK(engine, dppr_load_window(engine, init_stream.edge1, init_stream.edge2, init_stream.length));
        }
        // several sources on one device are solved together, up to 16 per group (multi-source batched
        // sweeps); --split keeps the reference's one-source-at-a-time driver flow
        use_groups = source_vertex_ids.size() > 1 && !gSplitInterface && !gNoGroups;
        // two or three sources on a large window: one after the other on the single-source path (binned sweeps) beats a group, whose
        // sweep costs about the same for 2 as for 8 sources (friendster stand-in, 3 sources: 422 ms per batch as a group, 3 x 100 in series)
        if (use_groups && source_vertex_ids.size() <= 3 && graph->sliding_window_size >= 4000000) use_groups = false;
        if (gProfile) DPPR_CHECK(engine, dppr_set_profiling(engine, 1));
        if (!quiet_) std::cout << "start..." << std::endl;
        if (use_groups) {
            for (size_t i = 0; i < source_vertex_ids.size(); i += kGroupMax) {
                const int32_t n = (int32_t)std::min<size_t>(kGroupMax, source_vertex_ids.size() - i);
                int32_t gid = -1;
                DPPR_CHECK(engine, dppr_add_source_group(engine, source_vertex_ids.data() + i, n, &gid));
                groups.push_back(gid);
            }
            ppr_time.assign(groups.size(), 0.0f);
            for (size_t k = 0; k < groups.size(); ++k) {
                float ms = 0;
                DPPR_CHECK(engine, dppr_group_init_solve(engine, groups[k], gTolerance, &ms));
                progress++;
                if (!quiet_) std::cout << "elapsed time=" << ms << "ms" << std::endl;
            }
            if (gValidate)
                for (size_t i = 0; i < source_vertex_ids.size(); ++i) ValidateResult(i);
        } else {
            slots.resize(source_vertex_ids.size());
            for (size_t i = 0; i < slots.size(); ++i)
                DPPR_CHECK(engine, dppr_add_source(engine, source_vertex_ids[i], &slots[i]));
            for (size_t i = 0; i < slots.size(); ++i) { // Init + ExecuteMainLoop(0)
                float ms = 0;
                if (gProfile) DPPR_CHECK(engine, dppr_trace_enable(engine, slots[i], 1));
                DPPR_CHECK(engine, dppr_init_solve(engine, slots[i], gTolerance, &ms));
                progress++;
                if (gProfile) PrintIterations(i, 0);
                if (!quiet_) std::cout << "elapsed time=" << ms << "ms" << std::endl;
                if (gValidate) ValidateResult(i);
            }
        }
        prof.End(HostProfile::INIT_GRAPH_CALC);
        prof.Start(HostProfile::DYNA_GRAPH_CALC);
        if (overlap_) SlidingWindowExecuteOverlapped();
        else SlidingWindowExecuteMainLoop();
        prof.End(HostProfile::DYNA_GRAPH_CALC);
        prof.End(HostProfile::TOTAL);
        if (!quiet_) std::cout << "finish!" << std::endl;
        if (std::getenv("DPPR_HOST_TIMES")) // (stderr, not a line of the reference: the loop's wall time beside the ppr_time it reports)
            std::cerr << "host_times batches=" << batches_done << " dynamic_ms=" << prof.ms[HostProfile::DYNA_GRAPH_CALC]
                      << " graph_update_ms=" << prof.ms[HostProfile::EXCLUDE_GRAPH_UPDATE] << " ppr_ms=" << prof.ms[HostProfile::PPR]
                      << " graph_update_beside_ppr_ms=" << build_beside_ms << " overlap=" << (overlap_ ? 1 : 0) << std::endl;
        if (gProfile && !quiet_) ReportProfile();
    }

    // gpu/PPRRevPushGPU.cuh:109-111: one line per frontier iteration of the loop that just ran (the engine's iteration
    // trace holds the frontier of every iteration; it is cleared for the next loop)
    void PrintIterations(size_t i, size_t phase_id) {
        int64_t n_iters = 0, n_ids = 0;
        DPPR_CHECK(engine, dppr_trace_get(engine, slots[i], &n_iters, &n_ids, nullptr, nullptr));
        std::vector<int64_t> off((size_t)n_iters + 1, 0);
        if (n_iters > 0) DPPR_CHECK(engine, dppr_trace_get(engine, slots[i], nullptr, nullptr, off.data(), nullptr));
        if (!quiet_)
            for (int64_t it = 0; it < n_iters; ++it)
                std::cout << "phase_id=" << phase_id << ",iteration_id=" << it << ",frontier_count=" << off[(size_t)it + 1] - off[(size_t)it]
                          << std::endl;
        DPPR_CHECK(engine, dppr_trace_enable(engine, slots[i], 1)); // (clears it)
    }

    // GPUProfiler::ReportProfile (gpu/GPUProfiler.cuh:42-55)
    void ReportProfile() {
        long long traverse = 0, expand = 0;
        for (size_t i = 0; i < slots.size(); ++i) {
            dppr_stats_t st;
            DPPR_CHECK(engine, dppr_stats(engine, slots[i], &st));
            prof.ms[HostProfile::EXPAND] += st.push_ms;
            traverse += st.sum_E;
            expand += st.sum_F;
        }
        std::cout << "****************** profile time **********************" << std::endl;
        for (int j = 0; j < HostProfile::N_PHASES; ++j) std::cout << "[" << HostProfile::Name(j) << "]=" << prof.ms[j] << "ms ";
        std::cout << std::endl;
        std::cout << "[traverse_count]=" << traverse << " [expand_count]=" << expand
                  << " [update_pos_residual_count]=0 [update_neg_residual_count]=0 [update_random_walk_count]=0 " << std::endl;
        std::cout << "****************** end profile  **********************" << std::endl;
    }

    // gpu/PPRGPU.cuh:109-177
    virtual void SlidingWindowExecuteMainLoop() {
        size_t stream_batch_count = 0;
        if (std::getenv("DPPR_TEST_STALL")) // (test hook: a driver that stops making progress -- the watchdog's post-mortem path)
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
        // Lookahead (include/dppr.h dppr_hint_next_batch): once batch k's graph is built, a helper thread advances the host stream to
        // batch k+1 and tells the engine which id arrays it will get next -- the stream advance and the id lookups then run while
        // this thread waits for batch k's update (the reference's untimed region, gpu/PPRGPU.cuh:114-135, is serial). Not with
        // --validate (its power iteration reads the host graph at batch k); DPPR_NO_LOOKAHEAD=1 for A/B runs. Results are identical.
        const bool lookahead = !gValidate && std::getenv("DPPR_NO_LOOKAHEAD") == nullptr;
        bool ahead = false, ahead_end = false; // the next batch is staged in graph->edge_batch / new_stream already | the stream ended there
        std::future<int> ahead_task;
        while (stream_batch_count++ < gStreamBatchCount) {
            if (!quiet_ && (gStreamUpdateCountPerBatch > 100 || stream_batch_count % 100 == 0))
                Report(stream_batch_count);
            progress++;
            prof.Start(HostProfile::EXCLUDE_GRAPH_UPDATE);
            if (ahead ? ahead_end : graph->StreamUpdates(gStreamUpdateCountPerBatch)) break; // partial batch: dropped
            ahead = false;
            // ---- untimed: batch upload + device graph rebuild ----
            DPPR_CHECK(engine, dppr_set_batch(engine, graph->edge_batch->edge1, graph->edge_batch->edge2,
                                              graph->edge_batch->is_insert, graph->edge_batch->length));
            GPUBuildSlidingGraph();
            progress++;
            prof.End(HostProfile::EXCLUDE_GRAPH_UPDATE);
            if (lookahead && stream_batch_count < gStreamBatchCount) { // (both calls above have consumed the host arrays of batch k)
                ahead = true;
                ahead_task = std::async(std::launch::async, [this, &ahead_end] {
                    ahead_end = graph->StreamUpdates(gStreamUpdateCountPerBatch);
                    if (ahead_end) return (int)DPPR_OK;
                    return dppr_hint_next_batch(engine, graph->edge_batch->edge1, graph->edge_batch->edge2, graph->edge_batch->length,
                                                graph->new_stream->edge1, graph->new_stream->edge2, graph->new_stream->length);
                });
            }
            prof.Start(HostProfile::PPR);
            // ---- timed: IncrementalBatchUpdate + ExecuteMainLoop(0) + (1), per source or per group ----
            for (size_t k = 0; k < groups.size(); ++k) {
                float ms = 0;
                DPPR_CHECK(engine, dppr_group_update(engine, groups[k], -1, gTolerance, &ms));
                ppr_time[k] += ms;
            }
            if (use_groups && gValidate)
                for (size_t i = 0; i < source_vertex_ids.size(); ++i) ValidateResult(i);
            for (size_t i = 0; i < slots.size(); ++i) {
                float ms = 0;
                if (gSplitInterface) {
                    // timed on the host: the three calls each synchronise the stream
                    struct timespec a, b;
                    clock_gettime(CLOCK_MONOTONIC, &a);
                    prof.Start(HostProfile::INC_UPDATE);
                    IncrementalBatchUpdate(i);
                    prof.End(HostProfile::INC_UPDATE);
                    prof.Start(HostProfile::PUSH);
                    ExecuteMainLoop(i, 0);
                    if (gProfile) PrintIterations(i, 0);
                    ExecuteMainLoop(i, 1);
                    if (gProfile) PrintIterations(i, 1);
                    prof.End(HostProfile::PUSH);
                    clock_gettime(CLOCK_MONOTONIC, &b);
                    ms = (float)((b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6);
                } else {
                    DPPR_CHECK(engine, dppr_update(engine, slots[i], -1, gTolerance, &ms));
                }
                ppr_time[i] += ms;
                if (gValidate) ValidateResult(i);
            }
            prof.End(HostProfile::PPR);
            if (ahead_task.valid()) {
                const int ahead_rc = ahead_task.get();
                DPPR_CHECK(engine, ahead_rc);
            }
        }
        batches_done = stream_batch_count - 1;
        if (!quiet_) Report(stream_batch_count);
        if (!quiet_) { // (not a line of the reference: how far the engine's internal id space is from the window's vertices)
            int32_t ids = 0, parked = 0, renumberings = 0;
            int64_t revivals = 0;
            DPPR_CHECK(engine, dppr_id_space(engine, &ids, &parked, &renumberings, &revivals));
            std::cout << "id_space ids=" << ids << " parked=" << parked << " renumberings=" << renumberings
                      << " revivals=" << revivals << std::endl;
        }
    }

    // The same loop with the untimed region of batch k + 1 (host stream advance, batch upload, device graph build: gpu/PPRGPU.cuh:114-135)
    // running on a helper thread WHILE batch k is solved. The engine keeps two epochs; the helper builds epoch
    // k + 1 through dppr_set_batch + dppr_slide_concurrent (own HIP stream and scratch inside the engine), this thread solves the
    // explicitly named epoch k. What this thread still waits for -- the first batch's graph, the remainder of a build that outlasts
    // its solve, and any slide that has to be exclusive (a renumbering of the id space moves every state row: dppr_renumbering_due)
    // -- is what exclude_graph_update_time reports; the helper's own time is reported beside it. Same stdout lines, same results
    // (DPPR_NO_OVERLAP=1: the serial loop above with its id lookahead; --validate and --split use that one too).
    struct Built {
        bool end = false;      // the stream ended: no such batch (a partial last batch is dropped, like the reference's)
        bool deferred = false; // host stream advanced, but batch upload + slide wait for the solver (exclusive slide)
        int32_t epoch = -1;
        double ms = 0;
    };
    Built BuildNext(bool concurrent) {
        Built b;
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        b.end = graph->StreamUpdates(gStreamUpdateCountPerBatch);
        if (!b.end) {
            if (concurrent && dppr_renumbering_due(engine)) {
                b.deferred = true;
            } else {
                DPPR_CHECK(engine, dppr_set_batch(engine, graph->edge_batch->edge1, graph->edge_batch->edge2, graph->edge_batch->is_insert,
                                                  graph->edge_batch->length));
                if (concurrent)
                    DPPR_CHECK(engine, dppr_slide_concurrent(engine, graph->new_stream->edge1, graph->new_stream->edge2, graph->new_stream->length, &b.epoch));
                else
                    DPPR_CHECK(engine, dppr_slide(engine, graph->new_stream->edge1, graph->new_stream->edge2, graph->new_stream->length, &b.epoch));
            }
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        b.ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
        progress++;
        return b;
    }

    virtual void SlidingWindowExecuteOverlapped() {
        size_t stream_batch_count = 0;
        if (std::getenv("DPPR_TEST_STALL")) // (test hook, as in the serial loop)
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
        prof.Start(HostProfile::EXCLUDE_GRAPH_UPDATE);
        Built nxt = gStreamBatchCount > 0 ? BuildNext(false) : Built(); // batch 1's graph: nothing to run beside
        prof.End(HostProfile::EXCLUDE_GRAPH_UPDATE);
        if (gStreamBatchCount == 0) nxt.end = true;
        while (stream_batch_count++ < gStreamBatchCount) {
            if (!quiet_ && (gStreamUpdateCountPerBatch > 100 || stream_batch_count % 100 == 0))
                Report(stream_batch_count);
            progress++;
            if (nxt.end) break;
            if (nxt.deferred) { // the host arrays hold this batch; the solver is idle now
                prof.Start(HostProfile::EXCLUDE_GRAPH_UPDATE);
                DPPR_CHECK(engine, dppr_set_batch(engine, graph->edge_batch->edge1, graph->edge_batch->edge2, graph->edge_batch->is_insert,
                                                  graph->edge_batch->length));
                DPPR_CHECK(engine, dppr_slide(engine, graph->new_stream->edge1, graph->new_stream->edge2, graph->new_stream->length, &nxt.epoch));
                prof.End(HostProfile::EXCLUDE_GRAPH_UPDATE);
                progress++;
            }
            const int32_t epoch = nxt.epoch;
            const bool more = stream_batch_count < gStreamBatchCount;
            std::future<Built> task;
            if (more) task = std::async(std::launch::async, [this] { return BuildNext(true); });
            prof.Start(HostProfile::PPR);
            for (size_t k = 0; k < groups.size(); ++k) {
                float ms = 0;
                DPPR_CHECK(engine, dppr_group_update(engine, groups[k], epoch, gTolerance, &ms));
                ppr_time[k] += ms;
            }
            for (size_t i = 0; i < slots.size(); ++i) {
                float ms = 0;
                DPPR_CHECK(engine, dppr_update(engine, slots[i], epoch, gTolerance, &ms));
                ppr_time[i] += ms;
            }
            prof.End(HostProfile::PPR);
            if (more) {
                prof.Start(HostProfile::EXCLUDE_GRAPH_UPDATE); // (what is left of the build once the solve is over)
                nxt = task.get();
                prof.End(HostProfile::EXCLUDE_GRAPH_UPDATE);
                build_beside_ms += nxt.ms;
            }
        }
        batches_done = stream_batch_count - 1;
        if (!quiet_) Report(stream_batch_count);
        if (!quiet_) {
            int32_t ids = 0, parked = 0, renumberings = 0;
            int64_t revivals = 0;
            DPPR_CHECK(engine, dppr_id_space(engine, &ids, &parked, &renumberings, &revivals));
            std::cout << "id_space ids=" << ids << " parked=" << parked << " renumberings=" << renumberings
                      << " revivals=" << revivals << std::endl;
        }
    }

    // the reference's four virtuals (gpu/PPRGPU.cuh:179-182), per source slot
    virtual void GPUBuildSlidingGraph() {
        DPPR_CHECK(engine, dppr_slide(engine, graph->new_stream->edge1, graph->new_stream->edge2,
                                      graph->new_stream->length, nullptr));
    }
    virtual void IncrementalBatchUpdate(size_t i) { DPPR_CHECK(engine, dppr_incremental_batch_update(engine, slots[i], -1)); }
    virtual void ExecuteMainLoop(size_t i, size_t phase_id) {
        DPPR_CHECK(engine, dppr_execute_main_loop(engine, slots[i], -1, (int)phase_id, gTolerance));
    }

    // gpu/PPRRevPushGPU.cuh:134-164: residual bound, then |p - p_pow| < 100 eps with the
    // power iteration of cpu/PPRCPUPowVec.h:55-83 on the current window graph.
    void ReadSource(size_t i, double *p, double *r) {
        if (use_groups) DPPR_CHECK(engine, dppr_group_read(engine, groups[i / kGroupMax], (int32_t)(i % kGroupMax), p, r));
        else DPPR_CHECK(engine, dppr_read(engine, slots[i], p, r));
    }

    virtual void ValidateResult(size_t i) {
        const IndexType V = graph->vertex_count, s = source_vertex_ids[i];
        std::vector<double> p((size_t)V), r((size_t)V);
        ReadSource(i, p.data(), r.data());
        for (IndexType u = 0; u < V; ++u) {
            if (!(r[u] < gTolerance && r[u] > -gTolerance)) {
                std::cout << "VALIDATE FAILED: residual[" << u << "]=" << r[u] << std::endl;
                std::exit(-1);
            }
        }
        std::vector<IndexType> row, col;
        graph->ConstructGraph(row, col);
        std::vector<double> a((size_t)V), b((size_t)V);
        for (IndexType u = 0; u < V; ++u) a[u] = (u == s) ? 1 : 0;
        size_t iters = 0;
        for (;;) { // Jacobi sweep until no component moves by more than 1e-14
            bool stop = true;
            for (IndexType u = 0; u < V; ++u) {
                double acc = 0.0;
                const size_t d = (size_t)(row[u + 1] - row[u]);
                for (IndexType j = row[u]; j < row[u + 1]; ++j) acc += a[col[j]] / (d + 1);
                acc = (1.0 - ALPHA) * acc;
                if (u == s) acc += ALPHA * 1.0;
                b[u] = acc;
                if (std::fabs(acc - a[u]) > 1e-14) stop = false;
            }
            if (stop) break;
            a.swap(b);
            ++iters;
        }
        const double bound = gTolerance * 100;
        for (IndexType u = 0; u < V; ++u) {
            const double err = std::fabs(a[u] - p[u]);
            if (!(err < bound)) {
                std::cout << "VALIDATE FAILED: " << err << "," << bound << " at vertex " << u << std::endl;
                std::exit(-1);
            }
        }
        if (!quiet_) std::cout << "validate ok (power iterations=" << iters << ")" << std::endl;
    }

    void Dump(const std::string &path) {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) return;
        const IndexType V = graph->vertex_count;
        std::vector<double> p((size_t)V), r((size_t)V);
        for (size_t i = 0; i < source_vertex_ids.size(); ++i) {
            ReadSource(i, p.data(), r.data());
            std::fwrite(&source_vertex_ids[i], sizeof(IndexType), 1, f);
            std::fwrite(&V, sizeof(IndexType), 1, f);
            std::fwrite(p.data(), sizeof(double), (size_t)V, f);
            std::fwrite(r.data(), sizeof(double), (size_t)V, f);
        }
        std::fclose(f);
    }

    double TotalPprTime() const {
        double t = 0;
        for (float x : ppr_time) t += x;
        return t;
    }

    SlidingGraphVec *graph;
    dppr_engine *engine = nullptr;
    int device_id;
    std::vector<IndexType> source_vertex_ids;
    std::vector<int32_t> slots;  // one per source (single-source mode)
    static constexpr size_t kGroupMax = 16; // sources per group (dppr_add_source_group)
    std::vector<int32_t> groups; // one per 16 sources (group mode: several sources per device)
    bool use_groups = false;
    std::vector<float> ppr_time; // ms per source (single mode) or per group, timed region only
    HostProfile prof;
    size_t batches_done = 0;
    bool overlap_ = false;       // the graph of batch k + 1 is built while batch k is solved (SlidingWindowExecuteOverlapped)
    double build_beside_ms = 0;  // ... time the helper thread spent on those builds
    std::atomic<unsigned long long> progress{0}; // bumped after every engine call that can take long (the watchdog of pagerank_main.cpp looks at it)

protected:
    // the progress / summary lines scripts/extract_gpu.py scrapes (gpu/PPRGPU.cuh:116-124,170-176);
    // with several sources, time is summed over sources and edges counted once per source
    void Report(size_t stream_batch_count) const {
        const double t = TotalPprTime();
        long long cur_edge_num = (long long)gStreamUpdateCountPerBatch * (long long)(stream_batch_count - 1) *
                                 (long long)source_vertex_ids.size();
        const size_t solves = (stream_batch_count - 1) * source_vertex_ids.size();
        std::cout << "coming stream_batch_count=" << stream_batch_count << std::endl;
        std::cout << "ppr_time " << t << std::endl;
        std::cout << "edge_num " << cur_edge_num << std::endl;
        std::cout << "ppr_latency " << (solves > 0 ? t / solves : 0) << std::endl;
        std::cout << "ppr_throughput " << cur_edge_num / t * 1000.0 << std::endl;
        // which loop produced the two figures above: with the graph of batch k + 1 built BESIDE the solve of batch k the
        // builder shares the device with the timed region, and ppr_latency comes out a few per cent above the serial loop's
        // (DPPR_NO_OVERLAP=1: the reference's loop shape, what tools/sweep.py and bench.py measure)
        std::cout << "graph_update " << (overlap_ ? "overlapped" : "serial") << std::endl;
    }
    bool quiet_;
};

// The reference has one driver class per variant (gpu/PPRRevPushGPU.cuh, gpu/PPRRevPushGPUVariants.cuh); here all four variants
// (-o 0 OPTIMIZED, 1 FAST_FRONTIER, 2 EAGER, 3 VANILLA) are mechanisms of one engine, selected by dppr_set_variant in the base
// class's constructor (gVariant): the duplicate filter (threshold crossing | status array) and the residual read (eager | pre-extracted)
// of the push kernels. One class therefore serves every -o.
class PPRRevPushGPU : public PPRGPU {
public:
    using PPRGPU::PPRGPU;
};
