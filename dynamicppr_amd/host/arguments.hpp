// arguments.hpp -- flag parsing for ./pagerank. Same flags and defaults as the reference
// (Arguments.h:66-86); unlike it, a flag at the end of argv without a value is an error instead
// of an out-of-bounds read (util/CommandLine.h:52-55), and -a is range-checked exactly.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "meta.hpp"

namespace args_detail {
inline const char *find(int argc, char **argv, const char *flag) {
    for (int i = 1; i < argc; ++i) {
        if (std::strcmp(argv[i], flag) == 0) {
            if (i + 1 >= argc) {
                std::cout << "missing value after " << flag << std::endl;
                std::exit(-1);
            }
            return argv[i + 1];
        }
    }
    return nullptr;
}
inline bool has(int argc, char **argv, const char *flag) {
    for (int i = 1; i < argc; ++i)
        if (std::strcmp(argv[i], flag) == 0) return true;
    return false;
}
inline int as_int(int argc, char **argv, const char *flag, int dflt) {
    const char *v = find(argc, argv, flag);
    return v ? std::atoi(v) : dflt;
}
inline double as_double(int argc, char **argv, const char *flag, double dflt) {
    const char *v = find(argc, argv, flag);
    if (!v) return dflt;
    char *end = nullptr;
    const double x = std::strtod(v, &end);
    if (end == v) {
        std::cout << "bad number after " << flag << std::endl;
        std::exit(-1);
    }
    return x;
}
} // namespace args_detail

inline void PrintUsage() {
    std::cout << "==========[USAGE]==========\n"
              << "-d: gDataFileName\n-a: gAppType\n" << REVERSE_PUSH << ":rev push\n"
              << "-i: gIsDirected\n-y: gIsDynamic\n-w: gWindowRatio\n-n: gWorkloadConfigType\n"
              << SLIDE_WINDOW_RATIO << ": SLIDE_WINDOW_RATIO, " << SLIDE_BATCH_SIZE << ": SLIDE_BATCH_SIZE\n"
              << "-r: gStreamUpdateCountVersusWindowRatio\n-b: gStreamBatchCount\n"
              << "-c: gStreamUpdateCountPerBatch\n-l: gStreamUpdateCountTotal\n"
              << "-s: gSourceVertexId\n-t: gThreadNum (ignored on the GPU)\n-o: gVariant\n-e: error tolerance\n"
              << "-g: number of GPUs (sources are dealt round-robin)\n"
              << "--sources <file>: one source vertex id per line (overrides -s)\n"
              << "--dump <path>: write pagerank/residual of every source after the last batch\n"
              << "--validate: residual bound + power-iteration check after every solve\n"
              << "--split: drive each batch through IncrementalBatchUpdate/ExecuteMainLoop(0)/(1)\n"
              << "--sync: synchronous (deterministic) push schedule\n"
              << "--share-device: with -g N on a node of fewer devices, device thread d uses device d % (devices present) (also DPPR_DEVICE_ALIAS=1)\n"
              << "--push-only: every iteration as a push iteration (no pull sweeps): what the -o variants differ in\n"
              << "--merge-phases: push the residuals of both signs in ONE loop, to eps / 4 (not the reference's schedule; same pushes, |p - p_reference| < 1e-9)\n"
              << "--no-groups: with several sources per GPU, solve them one at a time (default: up to 16 together)\n"
              << "--profile: per-iteration frontier lines and the phase-time report of the reference's -DPROFILE build (implies --split)\n"
              << "EXAMPLE: ./pagerank -d ../data/com-dblp.ungraph.bin -a 0 -i 0 -y 1 -w 0.1 -n 0 -r 0.01 -b 1000 -s 1\n"
              << "EXAMPLE: ./pagerank -d ../data/com-dblp.ungraph.bin -a 0 -i 0 -y 1 -w 0.1 -n 1 -c 100 -l 10000 -s 1"
              << std::endl;
}

inline void PrintArguments() {
    std::cout << "gAppType=" << gAppType << ",gIsDirected=" << gIsDirected << ",gIsDynamic=" << gIsDynamic << std::endl;
    std::cout << "gWindowRatio=" << gWindowRatio << ",gWorkloadConfigType=" << gWorkloadConfigType
              << ",gStreamUpdateCountVersusWindowRatio=" << gStreamUpdateCountVersusWindowRatio
              << ",gStreamBatchCount=" << gStreamBatchCount << ",gStreamUpdateCountPerBatch="
              << gStreamUpdateCountPerBatch << ",gStreamUpdateCountTotal=" << gStreamUpdateCountTotal << std::endl;
    std::cout << "gSourceVertexId=" << gSourceVertexId << std::endl;
    std::cout << "gThreadNum=" << gThreadNum << ",gVariant=" << gVariant << std::endl;
    std::cout << "error=" << gTolerance << ",ALPHA=" << ALPHA << std::endl;
}

inline void ArgumentsChecker() {
    bool ok = gAppType >= 0 && gAppType < kAlgoTypeSize && gIsDirected >= 0 && gIsDynamic >= 0 &&
              !gDataFileName.empty() && gTolerance > 0 && gNumGpus >= 1;
    if (gWorkloadConfigType == SLIDE_WINDOW_RATIO)
        ok = ok && gStreamUpdateCountVersusWindowRatio >= 0.0 && gStreamBatchCount != 0;
    else if (gWorkloadConfigType == SLIDE_BATCH_SIZE)
        ok = ok && gStreamUpdateCountPerBatch != 0 && gStreamUpdateCountTotal != 0;
    else
        ok = false;
    if (gIsDynamic == 0) {
        std::cout << "-y 0 (static mode) is deprecated in the reference (README.md:82) and not supported" << std::endl;
        ok = false;
    }
    if (gVariant < 0 || gVariant >= kVariantTypeSize) ok = false;
    if (!ok) {
        std::cout << "invalid arguments" << std::endl;
        PrintUsage();
        std::exit(-1);
    }
}

inline void ArgumentsParser(int argc, char **argv) {
    using namespace args_detail;
    if (const char *d = find(argc, argv, "-d")) gDataFileName = d;
    gAppType = as_int(argc, argv, "-a", 0);
    gIsDirected = as_int(argc, argv, "-i", -1);
    gIsDynamic = as_int(argc, argv, "-y", -1);
    gWindowRatio = as_double(argc, argv, "-w", 0.1);
    gWorkloadConfigType = as_int(argc, argv, "-n", SLIDE_WINDOW_RATIO);
    gStreamUpdateCountVersusWindowRatio = as_double(argc, argv, "-r", -1.0);
    gStreamBatchCount = (size_t)as_int(argc, argv, "-b", 0);
    gStreamUpdateCountPerBatch = (size_t)as_int(argc, argv, "-c", 0);
    gStreamUpdateCountTotal = (size_t)as_int(argc, argv, "-l", 0);
    gSourceVertexId = as_int(argc, argv, "-s", 1);
    gThreadNum = as_int(argc, argv, "-t", 1);
    gVariant = as_int(argc, argv, "-o", 0);
    gTolerance = as_double(argc, argv, "-e", 1e-9);
    gNumGpus = as_int(argc, argv, "-g", 1);
    if (const char *f = find(argc, argv, "--sources")) gSourcesFile = f;
    if (const char *f = find(argc, argv, "--dump")) gDumpPath = f;
    gValidate = has(argc, argv, "--validate");
    gSplitInterface = has(argc, argv, "--split");
    gSchedule = has(argc, argv, "--sync") ? 1 : 0;
    gNoGroups = has(argc, argv, "--no-groups");
    gMergePhases = has(argc, argv, "--merge-phases");
    gPushOnly = has(argc, argv, "--push-only");
    gShareDevice = has(argc, argv, "--share-device") || (getenv("DPPR_DEVICE_ALIAS") && atoi(getenv("DPPR_DEVICE_ALIAS")) != 0);
    gProfile = has(argc, argv, "--profile");
    if (gProfile) gSplitInterface = true; // the phases are timed and traced around the three virtual calls
    // -o: the reference's four variants (gpu/PPRRevPushGPUVariants.cuh) = {eager residual read | pre-extracted residuals
    // (InspectExtra)} x {threshold-crossing | status-array duplicate filter}; the engine has both mechanisms of both kinds
    // (dppr_set_variant). The pre-extracting variants (1 FAST_FRONTIER, 3 VANILLA) run the synchronous schedule.
    if (gVariant == FAST_FRONTIER || gVariant == VANILLA) gSchedule = 1;
    ArgumentsChecker();
}
