// dppr_kernels.hpp -- all hand-written HIP kernels of the engine (gfx950 / CDNA4, wave64).
//
// Hot path of guowentian/dynamicppr's gpu/ tree, re-designed for MI355X:
//   * no CUB/Thrust in any kernel: frontier compaction uses wave64 ballot +
//     mbcnt prefix ranks and LDS staging, one global counter atomic per workgroup;
//   * neighbour lists are expanded by a per-wavefront load-balanced search (64
//     frontier vertices per wave tile staged in LDS, consecutive lanes read
//     consecutive CSR entries -> coalesced bursts), replacing the 32-lane
//     CTA/warp/scan tiers of gpu/ExpandRev.cuh:44-176;
//   * residual pushes are native returning global_atomic_add_f64 (the reference
//     emulates them with a CAS loop, gpu/GPUUtil.cuh:21-30);
//   * the out-degree of the edge tail rides in the CSR entry ({src, outdeg+1}), so
//     an edge costs one coalesced 8-byte read + one atomic instead of the
//     reference's col_ind read + two random row_ptr reads + atomic;
//   * Repair (gpu/ExpandRev.cuh:708-743) is fused into the push kernel.
// Compiled with -ffp-contract=off: the double arithmetic is the same sequence of
// IEEE operations as the reference's expressions (cited per kernel).
//
//   dppr_common.hpp   wave64 primitives (ballot/mbcnt ranks, DPP scans, f64 atomics), statistics slots
//   dppr_push.hpp     Init, Inspect, dense snapshot, sparse frontier iteration (push atomics)
//   dppr_pull.hpp     dense frontier iteration (pull sweep, no global atomics)
//   dppr_resident.hpp a run of dense iterations as one resident launch (state on chip, data-flow synchronisation)
//   dppr_update.hpp   IncrementalBatchUpdate (lock-free, batch-index order)
//   dppr_builder.hpp  sliding-window graph builder (full sort / incremental merge), id translation
//   dppr_multi.hpp    multi-source batched sweeps (included separately by the engine)
#pragma once

#include "dppr_common.hpp"
#include "dppr_push.hpp"
#include "dppr_pull.hpp"
#include "dppr_resident.hpp"
#include "dppr_update.hpp"
#include "dppr_builder.hpp"
