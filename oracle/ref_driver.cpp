// ref_driver.cpp -- ORACLE-SIDE TOOL (test infrastructure, not product code).
//
// Drives the REAL reference headers, included from where they lie under
// /root/reference (nothing is copied, no stand-in headers are written), to dump
// golden vectors for the parts of the hot path that compile with the plain
// toolchain of this image:
//   SlidingGraphVec.h  (window / EdgeBatch derivation, Inc + Scratch construct)
//   cpu/PPRCPURev.h    (single-thread FIFO reverse push + stream update rule)
//   cpu/PPRCPUPowVec.h (power-iteration ground truth, CalPPRRev)
// The Cilk path (cpu/PPRCPUMTCilkRev.h) needs <cilk/cilk.h>, which this image
// lacks, so it is NOT built.
//
// Built by oracle/Makefile into oracle/_ref/ref_driver (git-ignored).
// Usage: ref_driver <reference CLI flags> --dump <file>
// Dump format: repeated records  [u32 name_len][name][u8 kind 0=i32 1=f64 2=u8][u64 n][payload]
#include "Meta.h"
#include "GraphVec.h"
#include "SlidingGraphVec.h"
#include "Profiler.h"
#include "PPRCPURev.h"
#include "PPRCPUPowVec.h"
#include "Arguments.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

static FILE *g_out = NULL;

static void put(const std::string &name, int kind, const void *data, uint64_t n) {
    uint32_t len = (uint32_t)name.size();
    fwrite(&len, 4, 1, g_out);
    fwrite(name.data(), 1, len, g_out);
    uint8_t k = (uint8_t)kind;
    fwrite(&k, 1, 1, g_out);
    fwrite(&n, 8, 1, g_out);
    size_t esz = kind == 0 ? 4 : (kind == 1 ? 8 : 1);
    if (n) fwrite(data, esz, n, g_out);
}

static void put_adj(const std::string &name, const std::vector<std::vector<IndexType> > &adj) {
    std::vector<int> row(adj.size() + 1, 0), col;
    for (size_t u = 0; u < adj.size(); ++u) {
        row[u] = (int)col.size();
        col.insert(col.end(), adj[u].begin(), adj[u].end());
    }
    row[adj.size()] = (int)col.size();
    put(name + ".row", 0, row.data(), row.size());
    put(name + ".col", 0, col.data(), col.size());
}

static void put_batch(const std::string &name, EdgeBatch *b) {
    std::vector<uint8_t> ins(b->length);
    for (int i = 0; i < b->length; ++i) ins[i] = b->is_insert[i] ? 1 : 0;
    put(name + ".e1", 0, b->edge1, b->length);
    put(name + ".e2", 0, b->edge2, b->length);
    put(name + ".ins", 2, ins.data(), b->length);
}

int main(int argc, char *argv[]) {
    std::string dump;
    for (int i = 1; i + 1 < argc; ++i)
        if (std::string(argv[i]) == "--dump") dump = argv[i + 1];
    if (dump.empty()) { fprintf(stderr, "need --dump <file>\n"); return 2; }
    ArgumentsParser(argc, argv);
    PrintArguments();
    g_out = fopen(dump.c_str(), "wb");
    if (!g_out) { perror("dump"); return 2; }

    // g drives the algorithm with IncConstructWindowGraph (production flow,
    // cpu/PPRCPU.h:76-121); g2 tracks the same stream position and is rebuilt from
    // scratch each batch (the -DVALIDATE flow) for the ground truth.
    SlidingGraphVec *g = new SlidingGraphVec(gDataFileName, gIsDirected);
    SlidingGraphVec *g2 = new SlidingGraphVec(gDataFileName, gIsDirected);
    Profiler::InitProfiler(1, PROFILE_PHASE_NUM, PROFILE_COUNT_TYPE_NUM);

    int cfg[8] = {g->vertex_count, g->sliding_window_size, (int)gStreamUpdateCountPerBatch,
                  (int)gStreamBatchCount, (int)gStreamUpdateCountTotal, g->edge_count,
                  gSourceVertexId, gIsDirected};
    put("config", 0, cfg, 8);
    double tol = gTolerance;
    put("tolerance", 1, &tol, 1);

    // --dump /dev/null = timing only: run the reference's FIFO push code, skip dumps and ground truth
    const bool timing_only = dump == "/dev/null";
    PPRCPURev *ppr = new PPRCPURev(g);
    if (!timing_only) {
        put_adj("b0.inc.out", g->col_ind);
        put_adj("b0.inc.in", g->in_col_ind);
        put("b0.deg", 0, g->deg.data(), g->deg.size());
    }
    ppr->ExecuteImpl();
    put("b0.fifo.p", 1, ppr->pagerank, g->vertex_count);
    put("b0.fifo.r", 1, ppr->residual, g->vertex_count);
    if (!timing_only) {
        PPRCPUPowVec pow(g2);
        pow.CalPPRRev(gSourceVertexId);
        put("b0.pow.p", 1, pow.pagerank, g->vertex_count);
    }

    double fifo_inc_ms = 0.0; // wall time of IncExecuteImpl only (the scope cpu/PPRCPUMTCilk.h:131-137 times)
    size_t k = 0;
    while (k++ < gStreamBatchCount) {
        bool over = g->StreamUpdates(gStreamUpdateCountPerBatch);
        bool over2 = g2->StreamUpdates(gStreamUpdateCountPerBatch);
        if (over || over2) break;
        std::string b = "b" + std::to_string(k);
        g->IncConstructWindowGraph();
        if (!timing_only) {
            put_batch(b + ".batch", g->edge_batch);
            put_batch(b + ".new", g->new_stream);
            g2->ScratchConstructWindowGraph();
            put_adj(b + ".inc.out", g->col_ind);
            put_adj(b + ".inc.in", g->in_col_ind);
            put_adj(b + ".scr.out", g2->col_ind);
            put_adj(b + ".scr.in", g2->in_col_ind);
            put(b + ".deg", 0, g->deg.data(), g->deg.size());
            put(b + ".scr.deg", 0, g2->deg.data(), g2->deg.size());
        }
        const auto t0 = std::chrono::steady_clock::now();
        ppr->IncExecuteImpl();
        fifo_inc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (timing_only) continue;
        put(b + ".fifo.p", 1, ppr->pagerank, g->vertex_count);
        put(b + ".fifo.r", 1, ppr->residual, g->vertex_count);
        PPRCPUPowVec pow(g2);
        pow.CalPPRRev(gSourceVertexId);
        put(b + ".pow.p", 1, pow.pagerank, g->vertex_count);
    }
    int done = (int)(k - 1);
    put("batches_done", 0, &done, 1);
    std::cout << "ref_fifo_inc_ms " << fifo_inc_ms << " batches " << done << " per_batch "
              << gStreamUpdateCountPerBatch << std::endl;
    fclose(g_out);
    return 0;
}
