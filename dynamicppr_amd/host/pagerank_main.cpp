// pagerank_main.cpp -- ./pagerank: the reference's CLI (gpu/PPRGPUMain.cu:8-39) on the MI355X
// engine. `./pagerank -d g.bin -a 0 -i 0 -y 1 -w 0.1 -n 0 -r 0.01 -b 100 -s 1` behaves like the
// reference binary; `-g N` spreads the sources of `--sources <file>` round-robin over N GPUs
// (one host thread and one full window-graph replica per device, no collective).
#include <atomic>
#include <chrono>
#include <cstdlib>
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
    // DPPR_WATCHDOG_S=<seconds>: a thread that ends the process -- after printing dppr_debug_dump of every engine -- when no
    // driver has made progress (a from-scratch solve, a slide, a batch) for that long. The engine waits on device-side
    // barriers with their own time limits (dppr_resident.hpp), so this should never fire; the test-suite sets it so that a
    // hang, should one happen again (one unexplained 300-second guard in round 3), leaves a post-mortem instead
    // of a silent time-out. Exit code 125.
    std::atomic<bool> finished{false};
    std::thread watchdog;
    if (const char *w = std::getenv("DPPR_WATCHDOG_S")) {
        const double limit = std::atof(w);
        if (limit > 0)
            watchdog = std::thread([&, limit] {
                unsigned long long last = ~0ull;
                auto since = std::chrono::steady_clock::now();
                while (!finished.load()) {
                    std::this_thread::sleep_for(std::chrono::milliseconds(50));
                    unsigned long long now = 0;
                    for (auto &d : drivers) now += d->progress.load() + dppr_heartbeat(d->engine); // (the engine's own read-backs count: one long call is not a hang)
                    if (now != last) {
                        last = now;
                        since = std::chrono::steady_clock::now();
                    } else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - since).count() > limit) {
                        // what the host knows first, flushed; then the engines' dumps (bounded inside the library) under a hard
                        // deadline of their own -- a wedged device must not turn the post-mortem into a silent time-out.
                        // Exit code 125: 124 is what `timeout` returns, and the two causes must be told apart.
                        std::cerr << "[watchdog] no progress for " << limit << " s (" << last << " progress marks so far) -- engine state:" << std::endl;
                        std::cout.flush();
                        std::thread([] {
                            std::this_thread::sleep_for(std::chrono::seconds(10));
                            std::cerr << "[watchdog] the dump did not finish within 10 s" << std::endl;
                            std::_Exit(125);
                        }).detach();
                        std::vector<char> buf(1 << 16);
                        for (auto &d : drivers) {
                            dppr_debug_dump(d->engine, buf.data(), (int32_t)buf.size());
                            std::cerr << buf.data();
                            std::cerr.flush();
                        }
                        std::_Exit(125);
                    }
                }
            });
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
    finished.store(true);
    if (watchdog.joinable()) watchdog.join();

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
