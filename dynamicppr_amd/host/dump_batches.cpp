// dump_batches.cpp -- test tool: streams a .bin through the PRODUCT's host graph types
// (graph_vec.hpp: SlidingGraphVec / EdgeBatch, the code ./pagerank feeds the engine with) and writes
// the workload derivation, the serialised window and every batch's edge_batch / new_stream arrays to
// a file, so tests can compare them with what the reference's SlidingGraphVec.h produced on the same
// input (tests/golden/*.npz: config, b{k}.batch.*, b{k}.new.*). No GPU, no engine.
//   dump_batches <pagerank flags...> --out FILE
// Records: u32 name length, name, u64 count, int32 values.
#include <cstdio>
#include <string>
#include <vector>

#include "arguments.hpp"
#include "graph_vec.hpp"
#include "meta.hpp"

static FILE *g_out = nullptr;

static void put(const std::string &name, const int32_t *v, size_t n) {
    const uint32_t ln = (uint32_t)name.size();
    const uint64_t cnt = n;
    fwrite(&ln, 4, 1, g_out);
    fwrite(name.data(), 1, ln, g_out);
    fwrite(&cnt, 8, 1, g_out);
    if (n) fwrite(v, 4, n, g_out);
}

static void put_bytes(const std::string &name, const uint8_t *v, size_t n) {
    std::vector<int32_t> w(v, v + n);
    put(name, w.data(), n);
}

int main(int argc, char **argv) {
    const char *out = args_detail::find(argc, argv, "--out");
    if (!out) {
        std::printf("usage: dump_batches <pagerank flags> --out FILE\n");
        return 2;
    }
    ArgumentsParser(argc, argv);
    SlidingGraphVec g(gDataFileName, gIsDirected != 0);
    g_out = std::fopen(out, "wb");
    if (!g_out) return 1;
    const int32_t cfg[6] = {g.vertex_count, g.sliding_window_size, (int32_t)gStreamUpdateCountPerBatch,
                            (int32_t)gStreamBatchCount, (int32_t)gStreamUpdateCountTotal, g.edge_count};
    put("config", cfg, 6);
    EdgeBatch win(g.sliding_window_size);
    g.SerializeEdgeStream(&win);
    put("window.e1", win.edge1, (size_t)win.length);
    put("window.e2", win.edge2, (size_t)win.length);
    int32_t done = 0;
    for (size_t k = 1; k <= gStreamBatchCount; ++k) { // the driver loop of gpu/PPRGPU.cuh:112-129
        if (g.StreamUpdates(gStreamUpdateCountPerBatch)) break;
        const std::string b = "b" + std::to_string(k);
        put(b + ".batch.e1", g.edge_batch->edge1, (size_t)g.edge_batch->length);
        put(b + ".batch.e2", g.edge_batch->edge2, (size_t)g.edge_batch->length);
        put_bytes(b + ".batch.ins", g.edge_batch->is_insert, (size_t)g.edge_batch->length);
        put(b + ".new.e1", g.new_stream->edge1, (size_t)g.new_stream->length);
        put(b + ".new.e2", g.new_stream->edge2, (size_t)g.new_stream->length);
        put_bytes(b + ".new.ins", g.new_stream->is_insert, (size_t)g.new_stream->length);
        ++done;
    }
    put("batches_done", &done, 1);
    std::fclose(g_out);
    return 0;
}
