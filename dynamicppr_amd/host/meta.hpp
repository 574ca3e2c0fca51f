// meta.hpp -- constants, flag globals and small helpers of the host program.
//
// The names of the flag globals (gDataFileName ... gTolerance) and of the enums are the
// reference's CLI contract (Meta.h:5-67, Meta.cpp); everything else is new code.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

using IndexType = int32_t; // Meta.h:26
using ValueType = double;  // Meta.h:25

enum AlgoType { REVERSE_PUSH = 0, kAlgoTypeSize };
enum VariantType { OPTIMIZED = 0, FAST_FRONTIER = 1, EAGER = 2, VANILLA = 3, kVariantTypeSize };
enum WorkloadConfigType {
    SLIDE_WINDOW_RATIO, // batch = ratio of the window (-r, -b)
    SLIDE_BATCH_SIZE    // batch = fixed number of edges (-c, -l)
};

constexpr ValueType ALPHA = 0.15; // Meta.h:31

// ---- flags (same letters and defaults as Arguments.h:66-86) -------------------------------
inline std::string gDataFileName;
inline int gAppType = -1;
inline int gIsDirected = -1;
inline int gIsDynamic = -1;
inline double gWindowRatio = 0.1;
inline int gWorkloadConfigType = SLIDE_WINDOW_RATIO;
inline double gStreamUpdateCountVersusWindowRatio = -1.0;
inline size_t gStreamBatchCount = 0;
inline size_t gStreamUpdateCountPerBatch = 0;
inline size_t gStreamUpdateCountTotal = 0;
inline int gSourceVertexId = 1;
inline int gThreadNum = 1;
inline int gVariant = OPTIMIZED;
inline ValueType gTolerance = 1e-9;
// ---- additions of this build ------------------------------------------------------------
inline int gNumGpus = 1;              // -g : devices; sources are dealt round-robin over them
inline std::string gSourcesFile;      // --sources : file with one source vertex id per line
inline std::string gDumpPath;         // --dump : write p/r of every source after the last batch
inline bool gValidate = false;        // --validate : the reference's -DVALIDATE checks at run time
inline bool gShareDevice = false;     // --share-device (or DPPR_DEVICE_ALIAS=1): the -g N device threads share the devices that exist (d % count)
inline bool gPushOnly = false;        // --push-only : no pull sweeps (the ablation of the -o variants times their push mechanisms)
inline bool gMergePhases = false;     // --merge-phases : one loop for residuals of both signs, to eps / 4 (dppr_set_phase_merge; off: the reference's two loops)
inline bool gSplitInterface = false;  // --split : drive the timed region through the 3 virtual calls
inline int gSchedule = 0;             // --sync : deterministic synchronous schedule
inline bool gProfile = false;         // --profile : the reference's -DPROFILE output (per-iteration frontier lines, phase times)
inline bool gNoGroups = false;        // --no-groups : solve several sources one at a time instead of up to 16 together
