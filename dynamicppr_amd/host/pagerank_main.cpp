// pagerank_main.cpp -- ./pagerank: the reference's CLI (gpu/PPRGPUMain.cu:8-39) on the MI355X
// engine. `./pagerank -d g.bin -a 0 -i 0 -y 1 -w 0.1 -n 0 -r 0.01 -b 100 -s 1` behaves like the
// reference binary; `-g N` spreads the sources of `--sources <file>` round-robin over N GPUs
// (one host thread and one full window-graph replica per device, no collective).
#include <chrono>
#include <fstream>
#include <iostream>
#include <memory>
#include <thread>
#include <vector>

#include "arguments.hpp"
#include "graph_vec.hpp"
#include "meta.hpp"
#include "ppr_gpu.hpp"

static std::vector<IndexType> LoadSources() {
    std::vector<IndexType> s;
    if (gSourcesFile.empty()) {
        s.push_back(gSourceVertexId);
        return s;
    }
    std::ifstream in(gSourcesFile);
    if (!in) {
        std::cout << "cannot open " << gSourcesFile << std::endl;
        std::exit(-1);
    }
    long long v;
    while (in >> v) s.push_back((IndexType)v);
    if (s.empty()) {
        std::cout << "no source ids in " << gSourcesFile << std::endl;
        std::exit(-1);
    }
    return s;
}

int main(int argc, char *argv[]) {
    ArgumentsParser(argc, argv);
    PrintArguments();
    const std::vector<IndexType> sources = LoadSources();
    const int ngpu = std::min<int>(gNumGpus, (int)sources.size());

    // Every device thread streams the same file through its own SlidingGraphVec (a position in
    // a shared read-only mapping of the page cache) and owns one engine.
    // --share-device / DPPR_DEVICE_ALIAS=1: the N device threads (each with its own engine, stream and graph replica) run on the
    // devices that exist -- the thread-per-device flow of -g N on a node with fewer GPUs (one, on the test pool)
    const int present = gShareDevice ? std::max(dppr_device_count(), 1) : 0;
    std::vector<std::unique_ptr<SlidingGraphVec>> graphs((size_t)ngpu);
    std::vector<std::unique_ptr<PPRGPU>> drivers((size_t)ngpu);
    for (int d = 0; d < ngpu; ++d) {
        std::vector<IndexType> mine;
        for (size_t i = (size_t)d; i < sources.size(); i += (size_t)ngpu) mine.push_back(sources[i]);
        for (IndexType s : mine) {
            if (s < 0) {
                std::cout << "negative source id" << std::endl;
                return -1;
            }
        }
        graphs[(size_t)d].reset(new SlidingGraphVec(gDataFileName, gIsDirected != 0));
        drivers[(size_t)d].reset(new PPRRevPushGPU(graphs[(size_t)d].get(), present ? d % present : d, mine, /*quiet=*/d != 0 && ngpu > 1));
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (ngpu == 1) {
        drivers[0]->DynamicExecute();
    } else {
        std::vector<std::thread> th;
        for (int d = 0; d < ngpu; ++d) th.emplace_back([&, d] { drivers[(size_t)d]->DynamicExecute(); });
        for (auto &t : th) t.join();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    if (ngpu > 1 || sources.size() > 1) {
        // aggregate: sum over sources of c * batches / slowest device's timed total
        double slowest = 0;
        long long edges = 0;
        for (int d = 0; d < ngpu; ++d) {
            slowest = std::max(slowest, drivers[(size_t)d]->TotalPprTime());
            edges += (long long)gStreamUpdateCountPerBatch * (long long)drivers[(size_t)d]->batches_done *
                     (long long)drivers[(size_t)d]->source_vertex_ids.size();
        }
        std::cout << "gpus " << ngpu << " sources " << sources.size() << std::endl;
        std::cout << "aggregate_edge_num " << edges << std::endl;
        std::cout << "aggregate_ppr_time_slowest_gpu " << slowest << std::endl;
        std::cout << "aggregate_ppr_throughput " << edges / slowest * 1000.0 << std::endl;
        std::cout << "wall_ms " << wall_ms << std::endl;
    }
    if (!gDumpPath.empty())
        for (int d = 0; d < ngpu; ++d) drivers[(size_t)d]->Dump(gDumpPath + (ngpu > 1 ? "." + std::to_string(d) : ""));
    return 0;
}
