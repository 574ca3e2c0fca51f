// dppr_engine.hip -- host side of libdppr_hip.so: the C ABI of include/dppr.h (the extern "C" entry points), over
//   dppr_host_state.hpp   what an engine owns: epochs, slots, groups, the solver's and the builder's streams and scratch
//   dppr_host_graph.hpp   the BUILDER: id space, key merge, CSRs, group cuts + tables, binned tables   (untimed region)
//   dppr_host_loop.hpp    the single-source SOLVER: IncrementalBatchUpdate + the frontier loop's launch forms (timed region)
//   dppr_host_group.hpp   source groups: the same loop for up to 16 sources at once
// One translation unit (the kernels are templates instantiated by the host code that launches them).
//
// Owns device memory (replaces gpu/DeviceMemory.cuh, gpu/GPUEdgeBatch.cuh,
// gpu/SlidingGraphBuilder.cuh), drives the frontier loop (replaces
// PPRRevPushGPU::ExecuteOptimized, gpu/PPRRevPushGPU.cuh:97-131) and times the
// region the reference times (gpu/PPRGPU.cuh:138-164).
//
// There is NO CPU fallback: without a HIP device dppr_create fails with
// DPPR_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <ctime>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp> // device radix sort only (CSR rebuild, batch grouping); no CUB/Thrust in kernels

#include "../../include/dppr.h"
#include "dppr_cut.hpp"
#include "dppr_idspace.hpp"
#include "dppr_kernels.hpp"
#include "dppr_multi.hpp"
#include "dppr_gpush.hpp"
#include "dppr_binned.hpp"
#include "dppr_calib.hpp"

using namespace dppr;

#include "dppr_host_state.hpp"
#include "dppr_host_graph.hpp"
#include "dppr_host_loop.hpp"
#include "dppr_host_group.hpp"


extern "C" {

int dppr_abi_version(void) { return DPPR_ABI_VERSION; }

const char *dppr_strerror(int status) {
    switch (status) {
    case DPPR_OK: return "ok";
    case DPPR_ERR_INVALID: return "invalid argument or call order";
    case DPPR_ERR_HIP: return "HIP runtime error";
    case DPPR_ERR_NOMEM: return "out of device memory";
    case DPPR_ERR_NO_DEVICE: return "no usable HIP device (the HIP path is mandatory; there is no CPU fallback)";
    case DPPR_ERR_NOT_CONVERGED: return "iteration cap hit before the frontier emptied";
    default: return "unknown status";
    }
}

const char *dppr_last_error(const dppr_engine *e) { return e ? e->err.c_str() : ""; }

int dppr_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int dppr_create(dppr_engine **out, int device, int32_t V, int32_t W, int directed, int32_t c, int32_t n_epochs) {
    if (!out || V <= 0 || W < 0 || c < 0 || n_epochs < 1) return DPPR_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    dppr_engine *e = new dppr_engine();
    dppr::g_live_engines.fetch_add(1, std::memory_order_relaxed);
    auto bail = [&](int code) {
        dppr_destroy(e);
        return code;
    };
#define HIP_TRY_C(call)                                                                 \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            fprintf(stderr, "dppr_create: %s at line %d\n", hipGetErrorString(_e), __LINE__); \
            return bail(_e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP);     \
        }                                                                               \
    } while (0)
    // Diagnostic switches (A/B runs, tests): every one of them is listed in INTEGRATION.md ("Environment variables"), none is needed
    // in production, and one that is set says so on stderr -- a stray variable must not change the product's behaviour silently
    auto diag_env = [](const char *name) -> const char * {
        const char *v = getenv(name);
        if (v && !getenv("DPPR_QUIET_SWITCHES")) fprintf(stderr, "dppr_create: diagnostic switch %s=%s in effect (INTEGRATION.md, Environment variables)\n", name, v);
        return v;
    };
    if (const char *v = diag_env("DPPR_SWEEP_BITS")) e->sweep_bits = atoi(v) != 0; // diagnostic A/B switches
    if (const char *v = diag_env("DPPR_HOT_BLOCKS")) e->hot_blocks = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_GSWEEP_HOT")) e->gsweep_hot_rows = std::max(0, atoi(v));
    if (const char *v = diag_env("DPPR_GSWEEP_GRID")) e->gsweep_grid_cap = std::max(1, std::min(atoi(v), STAT_SLOTS));
    if (const char *v = diag_env("DPPR_RENUMBER")) e->renumber_on = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_RENUMBER_PCT")) e->renumber_growth_pct = std::max(1, atoi(v));
    if (const char *v = diag_env("DPPR_RENUMBER_MIN")) e->renumber_min_parked = std::max(1, atoi(v));
    if (const char *v = diag_env("DPPR_GROUP_PUSH")) e->gpush_enter_pairs = std::max(-1, atoi(v));
    if (const char *v = diag_env("DPPR_GROUP_PUSH_FACTOR")) e->gpush_auto_factor = std::max(1, atoi(v));
    if (const char *v = diag_env("DPPR_GGROUPS_MIN")) e->ggroups_min = std::max(1, atoi(v));
    if (const char *v = diag_env("DPPR_COST_MODEL")) e->cost_model = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_GROUPING_RADIX")) e->force_radix_grouping = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_TEST_MERGE_MISS")) e->test_force_merge_miss = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_BIN_INCREMENTAL")) e->bin_incremental = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_BIN_RECUT_EVERY")) e->bin_recut_every = std::max(1, atoi(v));
    if (const char *v = diag_env("DPPR_BIN_FROZEN_REBUILD")) e->bin_frozen_rebuild = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_GROUP_AT_SLIDE")) e->group_at_slide = atoi(v) != 0;
    if (const char *v = diag_env("DPPR_GROUP_FULL_ROWS")) e->group_full_rows = atoi(v) != 0;
    e->device = device;
    e->V = V;
    e->W = W;
    e->c = c;
    e->directed = directed ? 1 : 0;
    e->n_epochs = n_epochs;
    e->Ed = directed ? W : 2 * W;
    e->bits = 1;
    while ((1ll << e->bits) < (long long)V) e->bits++;
    HIP_TRY_C(hipSetDevice(device));
    HIP_TRY_C(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    {
        int least = 0, greatest = 0; // (numerically: least priority >= greatest priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        HIP_TRY_C(hipStreamCreateWithPriority(&e->bs, hipStreamNonBlocking, least));
    }
    HIP_TRY_C(hipEventCreate(&e->ev0));
    HIP_TRY_C(hipEventCreate(&e->ev1));
    for (auto &ev : e->evpool) HIP_TRY_C(hipEventCreate(&ev));
    HIP_TRY_C(hipHostMalloc((void **)&e->dump_pin, dppr_engine::DUMP_PIN_BYTES, hipHostMallocDefault));
    HIP_TRY_C(hipHostMalloc((void **)&e->pinned, sizeof(int) * ((GMULTI_MAX + 2) * GS_MAX + 3 * GS_MAX + MAX_CHUNK * GS_MAX + 16), hipHostMallocDefault));
    const size_t Wn = (size_t)std::max(W, 1), Edn = (size_t)std::max(e->Ed, 1), Ln = (size_t)std::max(4 * c, 1);
    HIP_TRY_C(hipMalloc((void **)&e->w1, sizeof(int) * Wn));
    HIP_TRY_C(hipMalloc((void **)&e->w2, sizeof(int) * Wn));
    HIP_TRY_C(hipMalloc((void **)&e->outdeg, sizeof(int) * (size_t)V));
    // every memset / copy of the engine goes on ITS stream: the stream is non-blocking, so work on
    // the null stream (plain hipMemset / hipMemcpy) is not ordered with it
    HIP_TRY_C(hipMemsetAsync(e->outdeg, 0, sizeof(int) * (size_t)V, e->stream));
    HIP_TRY_C(hipMalloc((void **)&e->hub_slot_of, sizeof(int) * (size_t)V));
    HIP_TRY_C(hipMalloc((void **)&e->d_ext2int, sizeof(int) * (size_t)V));
    HIP_TRY_C(hipMalloc((void **)&e->d_xfer, sizeof(double) * (size_t)V));
    e->init_ids(V);
    HIP_TRY_C(hipMalloc((void **)&e->hub_hist, sizeof(int) * 64));
    HIP_TRY_C(hipMalloc((void **)&e->bar, sizeof(GridBar)));
    HIP_TRY_C(hipMalloc((void **)&e->keys_a, sizeof(uint64_t) * Edn));
    HIP_TRY_C(hipMalloc((void **)&e->keys_b, sizeof(uint64_t) * Edn));
    HIP_TRY_C(hipMalloc((void **)&e->in_sorted, sizeof(uint64_t) * Edn));
    if (e->directed) HIP_TRY_C(hipMalloc((void **)&e->out_sorted, sizeof(uint64_t) * Edn));
    HIP_TRY_C(hipMalloc((void **)&e->delpos, sizeof(int) * ((size_t)2 * (size_t)std::max(c, 1) + 16)));
    {
        const size_t bn = (size_t)std::max(2 * c, 1);
        for (int k = 0; k < 4; ++k) {
            HIP_TRY_C(hipMalloc((void **)&e->bk[k], sizeof(uint64_t) * bn));
            HIP_TRY_C(hipMalloc((void **)&e->bks[k], sizeof(uint64_t) * bn));
        }
    }
    HIP_TRY_C(rocprim::radix_sort_keys(nullptr, e->sort_tmp_bytes, e->keys_a, e->keys_b, Edn, 0u,
                                       (unsigned)(2 * e->bits), e->stream));
    HIP_TRY_C(hipMalloc(&e->sort_tmp, std::max<size_t>(e->sort_tmp_bytes, 16)));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY_C(hipMalloc((void **)&e->su_k[k], sizeof(uint32_t) * Ln));
        HIP_TRY_C(hipMalloc((void **)&e->su_v[k], sizeof(uint32_t) * Ln));
    }
    HIP_TRY_C(hipMalloc((void **)&e->su_grp, sizeof(int) * (4 * SU_GRP_MAX_BUCKETS + 8)));
    HIP_TRY_C(hipMemsetAsync(e->su_grp, 0, sizeof(int) * (4 * SU_GRP_MAX_BUCKETS + 8), e->stream)); // (the ranking launch leaves histogram and cursors cleared for the next batch)
    HIP_TRY_C(hipMalloc((void **)&e->su_term, sizeof(double) * Ln * GS_MAX)); // one term array per source lane of a group
    HIP_TRY_C(hipMalloc((void **)&e->su_ins, Ln));
    HIP_TRY_C(rocprim::radix_sort_pairs(nullptr, e->su_tmp_bytes, e->su_k[0], e->su_k[1], e->su_v[0], e->su_v[1], Ln,
                                        0u, (unsigned)e->bits, e->stream));
    HIP_TRY_C(hipMalloc(&e->su_tmp, std::max<size_t>(e->su_tmp_bytes, 16)));
    e->epochs.resize((size_t)n_epochs);
    for (auto &ep : e->epochs) {
        HIP_TRY_C(hipMalloc((void **)&ep.row_ptr, sizeof(int) * ((size_t)V + 1)));
        HIP_TRY_C(hipMalloc((void **)&ep.adj, sizeof(Adj) * Edn));
        HIP_TRY_C(hipMalloc((void **)&ep.out_row_ptr, sizeof(int) * ((size_t)V + 1)));
        HIP_TRY_C(hipMalloc((void **)&ep.out_col, sizeof(int) * Edn));
        HIP_TRY_C(hipMalloc((void **)&ep.b1, sizeof(int) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.b2, sizeof(int) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.deg_after, sizeof(int) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.ins, Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.sk, sizeof(uint32_t) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.sv, sizeof(uint32_t) * Ln));
        HIP_TRY_C(hipMalloc((void **)&ep.grp_tile, sizeof(int) * ((size_t)V / WAVE + 3)));
        HIP_TRY_C(hipMalloc((void **)&ep.ggrp_tile, sizeof(int) * ((size_t)V / WAVE + 3)));
        HIP_TRY_C(hipMalloc((void **)&ep.hub_v, sizeof(int) * HUB_CAP));
        HIP_TRY_C(hipMalloc((void **)&ep.hub_degp1, sizeof(int) * HUB_CAP));
    }
    HIP_TRY_C(hipStreamSynchronize(e->stream)); // (the builder's stream is another one: nothing of this set-up may still be in flight)
#undef HIP_TRY_C
    *out = e;
    return DPPR_OK;
}

void dppr_destroy(dppr_engine *e) {
    if (!e) return;
    if (e->pre.task.valid()) e->pre.task.wait();
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->bs) (void)hipStreamSynchronize(e->bs);
    for (auto &s : e->slots) {
        (void)hipFree(s.p); (void)hipFree(s.r); (void)hipFree(s.x); (void)hipFree(s.x2);
        (void)hipFree(s.ft[0]); (void)hipFree(s.ft[1]); (void)hipFree(s.neg); (void)hipFree(s.status); (void)hipFree(s.act[0]); (void)hipFree(s.act[1]);
        (void)hipFree(s.cnt); (void)hipFree(s.dstats); (void)hipFree(s.big);
    }
    for (auto &g : e->groups) {
        (void)hipFree(g.p); (void)hipFree(g.r); (void)hipFree(g.x); (void)hipFree(g.x2);
        (void)hipFree(g.act[0]); (void)hipFree(g.act[1]);
        (void)hipFree(g.cnt); (void)hipFree(g.mlog); (void)hipFree(g.dstats); (void)hipFree(g.gq);
    }
    for (auto &ep : e->epochs) {
        (void)hipFree(ep.row_ptr); (void)hipFree(ep.adj); (void)hipFree(ep.out_row_ptr); (void)hipFree(ep.out_col); (void)hipFree(ep.b1); (void)hipFree(ep.b2);
        (void)hipFree(ep.deg_after); (void)hipFree(ep.ins); (void)hipFree(ep.sk); (void)hipFree(ep.sv); (void)hipFree(ep.hub_v); (void)hipFree(ep.hub_degp1); (void)hipFree(ep.grp_tile); (void)hipFree(ep.ggrp_tile); (void)hipFree(ep.gtab);
        (void)hipFree(ep.acut); (void)hipFree(ep.chunks); (void)hipFree(ep.hl); (void)hipFree(ep.dl); (void)hipFree(ep.vb); (void)hipFree(ep.tb); (void)hipFree(ep.tdelta);
        (void)hipFree(ep.res_pk); (void)hipFree(ep.su_rng);
    }
    (void)hipFree(e->bin_vblk_a); (void)hipFree(e->bin_small); (void)hipFree(e->bin_scan); (void)hipFree(e->bin_vals); (void)hipFree(e->bin_tmp);
    (void)hipFree(e->bin_wb); (void)hipFree(e->bin_wa); (void)hipFree(e->bin_first);
    (void)hipFree(e->w1); (void)hipFree(e->w2); (void)hipFree(e->outdeg);
    (void)hipFree(e->bar);
    (void)hipFree(e->res_arena);
    (void)hipFree(e->hub_slot_of); (void)hipFree(e->hub_hist); (void)hipFree(e->d_ext2int); (void)hipFree(e->d_xfer);
    (void)hipFree(e->mv_idx); (void)hipFree(e->mv_tmp);
    (void)hipFree(e->keys_a); (void)hipFree(e->keys_b); (void)hipFree(e->sort_tmp);
    (void)hipFree(e->in_sorted); (void)hipFree(e->out_sorted); (void)hipFree(e->delpos);
    for (int k = 0; k < 4; ++k) { (void)hipFree(e->bk[k]); (void)hipFree(e->bks[k]); }
    for (int k = 0; k < 2; ++k) { (void)hipFree(e->su_k[k]); (void)hipFree(e->su_v[k]); }
    (void)hipFree(e->su_term); (void)hipFree(e->su_ins); (void)hipFree(e->su_tmp); (void)hipFree(e->su_grp);
    if (e->pinned) (void)hipHostFree(e->pinned);
    if (e->dump_pin) (void)hipHostFree(e->dump_pin);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    for (auto &ev : e->evpool)
        if (ev) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->bs) (void)hipStreamDestroy(e->bs);
    dppr::g_live_engines.fetch_sub(1, std::memory_order_relaxed);
    delete e;
}

int dppr_set_schedule(dppr_engine *e, int schedule) {
    if (!e || (schedule != DPPR_SCHEDULE_EAGER && schedule != DPPR_SCHEDULE_SYNC)) return DPPR_ERR_INVALID;
    e->schedule = schedule;
    return DPPR_OK;
}

int dppr_set_batch_grouping(dppr_engine *e, int at_slide) {
    if (!e) return DPPR_ERR_INVALID;
    e->group_at_slide = at_slide != 0; // (applies to the epochs dppr_slide builds from now on)
    return DPPR_OK;
}

int dppr_set_variant(dppr_engine *e, int variant) {
    if (!e || variant < 0 || variant > 3) return fail(e, DPPR_ERR_INVALID, "set_variant: 0 OPTIMIZED, 1 FAST_FRONTIER, 2 EAGER, 3 VANILLA");
    e->schedule = (variant == 1 || variant == 3) ? DPPR_SCHEDULE_SYNC : DPPR_SCHEDULE_EAGER; // pre-extracted residuals = the synchronous schedule
    e->pre_extract = variant == 1 || variant == 3;
    e->status_dedup = variant == 2 || variant == 3;
    return DPPR_OK;
}

int dppr_set_phase_merge(dppr_engine *e, int on, int eps_divisor) {
    if (!e || eps_divisor < 0 || eps_divisor > 1024) return fail(e, DPPR_ERR_INVALID, "set_phase_merge: eps_divisor 1..1024 (0 keeps it)");
    e->merge_phases = on != 0;
    if (eps_divisor > 0) e->merge_div = eps_divisor;
    return DPPR_OK;
}

int dppr_set_profiling(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->profiling = on != 0;
    return DPPR_OK;
}

int dppr_set_persistent(dppr_engine *e, int mode, int64_t timeout_us) {
    if (!e || mode < 0 || mode > 2 || e->loaded || !e->slots.empty())
        return fail(e, DPPR_ERR_INVALID, "set_persistent: call right after dppr_create, mode 0, 1 or 2");
    e->persist_mode = mode;
    if (timeout_us > 0) e->persist_ticks = (unsigned long long)timeout_us * 100ull; // wall_clock64 runs at 100 MHz
    if (timeout_us < 0) { // tests: a roll-call that cannot succeed, given up after 200 us
        e->persist_ticks = 20000ull;
        e->persist_rollcall_extra = 1;
    }
    return DPPR_OK;
}

int dppr_set_tuning(dppr_engine *e, int hub_min_degree, int big_row_edges, int pull_min_frontier, int chunk_iters,
                    int pull_block) {
    if (!e || hub_min_degree < 1 || big_row_edges < 1 || e->loaded || !e->slots.empty())
        return fail(e, DPPR_ERR_INVALID, "set_tuning: call right after dppr_create, values >= 1");
    e->hub_min_degree = hub_min_degree;
    e->big_row = big_row_edges;
    e->pull_min_frontier = pull_min_frontier;
    if (chunk_iters > 0) {
        e->chunk_iters = std::min(chunk_iters, MAX_CHUNK);
        e->chunk_explicit = true;
    }
    if (pull_block >= 256 && pull_block <= 1024 && pull_block % 64 == 0) e->pull_block = pull_block;
    return DPPR_OK;
}

int dppr_set_sweep_bitmap(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->sweep_bits = on != 0;
    return DPPR_OK;
}

int dppr_set_binned_sweep(dppr_engine *e, int mode, int ha_tiles, int hb_tiles, int64_t target_edges, int64_t min_ids, int64_t chunk_edges,
                          int64_t target_a_edges) {
    if (!e || mode < 0 || mode > 2 || ha_tiles < 0 || ha_tiles > BIN_MAX_HA_TILES || hb_tiles < 0 || hb_tiles > BIN_MAX_HB_TILES || target_edges < 0 || min_ids < 0 ||
        chunk_edges < 0 || target_a_edges < 0 || e->bin_ready || e->loaded)
        return fail(e, DPPR_ERR_INVALID, "set_binned_sweep: call right after dppr_create; mode 0..2, ha_tiles <= 272, hb_tiles <= 120");
    e->bin_mode = mode;
    if (ha_tiles > 0) e->bin_ha_tiles = ha_tiles;
    if (hb_tiles > 0) e->bin_hb_tiles = hb_tiles;
    if (target_edges > 0) e->bin_target = target_edges;
    if (min_ids > 0) e->bin_min_ids = min_ids;
    if (chunk_edges > 0) e->bin_chunk = chunk_edges;
    if (target_a_edges > 0) e->bin_target_a = target_a_edges;
    return DPPR_OK;
}

int dppr_set_resident_slots(dppr_engine *e, int sorted) {
    if (!e) return DPPR_ERR_INVALID;
    e->res_slots = sorted != 0;
    // epochs already cut keep their tables until the next cut; switching OFF takes effect at once
    if (!e->res_slots)
        for (auto &ep : e->epochs) ep.res_valid = false;
    return DPPR_OK;
}

int dppr_set_resident_update(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->res_update = on != 0;
    if (!e->res_update)
        for (auto &ep : e->epochs) ep.su_inline = false;
    return DPPR_OK;
}

int dppr_set_group_resident(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->group_resident = on != 0;
    return DPPR_OK;
}

int dppr_set_group_push(dppr_engine *e, int enter_pairs, int list_cap, int64_t max_edges) {
    if (!e || enter_pairs < -1 || list_cap < 0 || max_edges < 0) return fail(e, DPPR_ERR_INVALID, "set_group_push: bad argument");
    e->gpush_enter_pairs = enter_pairs;
    if (list_cap > 0) e->gpush_list_cap = list_cap;
    e->gpush_max_edges = max_edges;
    return DPPR_OK;
}

int dppr_set_group_seeding(dppr_engine *e, int from_tails) {
    if (!e) return DPPR_ERR_INVALID;
    e->group_tail_seeding = from_tails != 0;
    return DPPR_OK;
}

int dppr_set_renumbering(dppr_engine *e, int on, int growth_pct, int min_parked) {
    if (!e || growth_pct < 0 || min_parked < 0) return fail(e, DPPR_ERR_INVALID, "set_renumbering: bad argument");
    e->renumber_on = on != 0;
    if (growth_pct > 0) e->renumber_growth_pct = growth_pct;
    if (min_parked > 0) e->renumber_min_parked = min_parked;
    e->renumber_next = std::min(e->renumber_next, renumber_threshold(e->n_int, e->renumber_growth_pct));
    return DPPR_OK;
}

int dppr_id_space(dppr_engine *e, int32_t *n_ids, int32_t *n_parked, int32_t *renumberings, int64_t *revivals) {
    if (!e) return DPPR_ERR_INVALID;
    if (n_ids) *n_ids = e->n_int;
    if (n_parked) *n_parked = e->n_parked;
    if (renumberings) *renumberings = e->renumberings;
    if (revivals) *revivals = e->revivals;
    return DPPR_OK;
}

int dppr_set_incremental_graph(dppr_engine *e, int on) {
    if (!e) return DPPR_ERR_INVALID;
    e->incremental = on != 0;
    return DPPR_OK;
}

int dppr_synchronize(dppr_engine *e) {
    if (!e) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipStreamSynchronize(e->bs));
    return DPPR_OK;
}

int dppr_load_window(dppr_engine *e, const int32_t *e1, const int32_t *e2, int32_t n) {
    if (!e || n != e->W || (n > 0 && (!e1 || !e2))) return fail(e, DPPR_ERR_INVALID, "load_window: n must equal W");
    HIP_TRY(hipSetDevice(e->device));
    {
        // Number the window's vertices in a pseudo-random order (hash of the external id), not by
        // first appearance: high-degree vertices show up early in a stream, and packing them into
        // the first tiles would serialise the sweeps on a few workgroups.
        std::vector<std::pair<uint64_t, int32_t>> fresh;
        for (int k = 0; k < 2; ++k) {
            const int32_t *a = k ? e2 : e1;
            for (int i = 0; i < n; ++i) {
                const int v = a[i];
                if (v < 0 || v >= e->V) return fail(e, DPPR_ERR_INVALID, "load_window: vertex id out of range");
                if (e->ext2int[(size_t)v] == -1) {
                    e->ext2int[(size_t)v] = -2; // seen, not numbered yet
                    fresh.emplace_back(id_hash(v), v);
                }
            }
        }
        std::vector<int32_t> indeg;
        if (fresh.size() > HOT_WINDOW_MIN) {
            indeg.assign((size_t)e->V, 0);
            for (int i = 0; i < n; ++i) {
                indeg[(size_t)e2[i]]++;
                if (!e->directed) indeg[(size_t)e1[i]]++;
            }
        }
        numbering_order(fresh, indeg.empty() ? nullptr : indeg.data(), e->hot_blocks);
        for (auto &kv : fresh) {
            e->ext2int[(size_t)kv.second] = -1;
            (void)to_int(e, kv.second);
        }
    }
    if (!translate(e, e1, n, e->h_tmp1) || !translate(e, e2, n, e->h_tmp2))
        return fail(e, DPPR_ERR_INVALID, "load_window: vertex id out of range");
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(e->w1, e->h_tmp1.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
        HIP_TRY(hipMemcpyAsync(e->w2, e->h_tmp2.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->bs));
    }
    e->head = 0;
    HIP_TRY(hipMemsetAsync(e->outdeg, 0, sizeof(int) * (size_t)e->V, e->bs));
    if (n > 0) {
        hipLaunchKernelGGL(k_deg_update, dim3(grid_for(n)), dim3(BLOCK), 0, e->bs, e->w1, e->w2, n, e->directed, 1,
                           e->outdeg);
        HIP_TRY(hipGetLastError());
    }
    for (auto &ep : e->epochs) ep.id = -1;
    Epoch &ep = e->epochs[0];
    ep.L = 0;
    ep.grouped = false;
    ep.su_inline = false;
    int rc = query_persist_cap(e);
    if (rc) return rc;
    rc = sort_window_full(e);
    if (rc) return rc;
    rc = build_epoch(e, ep);
    if (rc) return rc;
    ep.id = 0;
    e->newest = 0;
    e->renumber_next = renumber_threshold(e->n_int, e->renumber_growth_pct);
    e->loaded = true;
    e->batch_staged = false;
    HIP_TRY(hipStreamSynchronize(e->bs));
    return DPPR_OK;
}

int dppr_hint_next_batch(dppr_engine *e, const int32_t *b1, const int32_t *b2, int32_t L, const int32_t *n1, const int32_t *n2, int32_t c) {
    // (no message through fail(): this call may come from a helper thread while the engine's own thread writes e->err -- ADVICE r04)
    if (!e || e->broken || L < 0 || L > 4 * e->c || c < 0 || c > e->c || (L > 0 && (!b1 || !b2)) || (c > 0 && (!n1 || !n2))) return DPPR_ERR_INVALID;
    pre_join(e);
    dppr_engine::Pre &pr = e->pre;
    const int32_t *src[4] = {b1, b2, n1, n2};
    const int n[4] = {L, L, c, c};
    for (int k = 0; k < 4; ++k) {
        pr.src[k] = src[k];
        pr.n[k] = n[k];
        pr.out[k].resize((size_t)std::max(n[k], 1));
    }
    pr.epoch = e->renumber_epoch;
    pr.armed = true;
    pr.ok = false;
    pr.task = std::async(std::launch::async, [e] {
        bool ok = true;
        for (int k = 0; k < 4; ++k)
            if (e->pre.n[k] > 0) ok = e->lookup_only(e->pre.src[k], (size_t)e->pre.n[k], e->pre.out[k].data(), e->pre.miss[k]) && ok;
        return ok;
    });
    return DPPR_OK;
}

int dppr_set_batch(dppr_engine *e, const int32_t *b1, const int32_t *b2, const uint8_t *ins, int32_t L) {
    if (!e || e->broken || L < 0 || L > 4 * e->c || (L > 0 && (!b1 || !b2 || !ins)))
        return fail(e, DPPR_ERR_INVALID, "set_batch: length exceeds 4*max_batch");
    std::lock_guard<std::mutex> map_lk(e->map_mu); // (ids are assigned and rows moved below: not beside a dppr_read of the solver thread)
    if (!ids_in_range(e, b1, L) || !ids_in_range(e, b2, L) || !translate(e, b1, L, e->st_b1) || !translate(e, b2, L, e->st_b2))
        return fail(e, DPPR_ERR_INVALID, "set_batch: vertex id out of range");
    e->st_b1.resize((size_t)L);
    e->st_b2.resize((size_t)L);
    e->st_ins.assign(ins, ins + L);
    e->batch_staged = true;
    HIP_TRY(hipSetDevice(e->device));
    return flush_moves(e); // (a record may have named a parked vertex)
}

static int slide_impl(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch);

int dppr_slide(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch) {
    if (e) e->build_concurrent = false;
    return slide_impl(e, n1, n2, c, out_epoch);
}

int dppr_slide_concurrent(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch) {
    if (!e || e->n_epochs < 2) return fail(e, DPPR_ERR_INVALID, "slide_concurrent: needs n_epochs >= 2 (the epoch being built must not be the one being solved)");
    e->build_concurrent = true;
    const int rc = slide_impl(e, n1, n2, c, out_epoch);
    e->build_concurrent = false;
    return rc;
}

int dppr_renumbering_due(const dppr_engine *e) {
    return e && e->renumber_on && e->W > 0 && e->n_int >= e->renumber_next ? 1 : 0;
}

static int slide_impl(dppr_engine *e, const int32_t *n1, const int32_t *n2, int32_t c, int32_t *out_epoch) {
    // c is bounded by max_batch of dppr_create: the batch key buffers (2 * max_batch keys each) and the
    // merge scratch are sized for it
    if (!e || e->broken || !e->loaded || c < 0 || c > e->W || c > e->c || (c > 0 && (!n1 || !n2)))
        return fail(e, DPPR_ERR_INVALID, "slide: window not loaded, or c exceeds the window / max_batch of dppr_create");
    if (!ids_in_range(e, n1, c) || !ids_in_range(e, n2, c)) // before anything is touched: a rejected slide is a no-op
        return fail(e, DPPR_ERR_INVALID, "slide: vertex id out of range");
    HIP_TRY(hipSetDevice(e->device));
    const int W = e->W;
    static const bool slide_trace = getenv("DPPR_SLIDE_TRACE") != nullptr; // (diagnostic: phases of a slide; each mark synchronises)
    timespec t_mark;
    clock_gettime(CLOCK_MONOTONIC, &t_mark);
    auto mark = [&](const char *what) {
        if (!slide_trace) return;
        (void)hipStreamSynchronize(e->bs);
        timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        fprintf(stderr, "[slide] %-34s %8.1f us\n", what, (now.tv_sec - t_mark.tv_sec) * 1e6 + (now.tv_nsec - t_mark.tv_nsec) * 1e-3);
        t_mark = now;
    };
    bool renumbered = false;
    {
        std::lock_guard<std::mutex> map_lk(e->map_mu); // the id-assigning head of the slide: maps and row moves change together
        if (int rc = compact_ids(e, &renumbered)) return rc;
        if (renumbered) e->bin_words_valid = false; // (the binned tables' words and cuts are in the old numbering)
        mark("renumbering check");
        if (!translate(e, n1, c, e->h_tmp1) || !translate(e, n2, c, e->h_tmp2))
            return fail(e, DPPR_ERR_INVALID, "slide: vertex id out of range");
        if (int rc = flush_moves(e)) return rc;
    }
    mark("translate new edges");
    n1 = e->h_tmp1.data();
    n2 = e->h_tmp2.data();
    // the c oldest edges sit at ring positions head .. head+c (mod W): retire their degrees (and
    // note their keys), overwrite them with the new edges, add the new degrees (and note those keys)
    const bool inc = e->incremental && c > 0 && 2 * c <= e->Ed && !renumbered; // (renumbered: the sorted keys are stale)
    const int per = e->directed ? 1 : 2; // keys per stream edge in the in-orientation array
    int done = 0;
    while (done < c) {
        const int pos = (e->head + done) % W;
        const int len = std::min(c - done, W - pos);
        hipLaunchKernelGGL(k_deg_update, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos, len,
                           e->directed, -1, e->outdeg);
        if (inc)
            hipLaunchKernelGGL(k_make_keys_seg, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos,
                               len, e->directed, e->bits, e->bk[0] + (size_t)done * per, e->bk[2] + done);
        HIP_TRY(hipMemcpyAsync(e->w1 + pos, n1 + done, sizeof(int) * (size_t)len, hipMemcpyHostToDevice, e->bs));
        HIP_TRY(hipMemcpyAsync(e->w2 + pos, n2 + done, sizeof(int) * (size_t)len, hipMemcpyHostToDevice, e->bs));
        hipLaunchKernelGGL(k_deg_update, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos, len,
                           e->directed, 1, e->outdeg);
        if (inc)
            hipLaunchKernelGGL(k_make_keys_seg, dim3(grid_for(len)), dim3(BLOCK), 0, e->bs, e->w1 + pos, e->w2 + pos,
                               len, e->directed, e->bits, e->bk[1] + (size_t)done * per, e->bk[3] + done);
        HIP_TRY(hipGetLastError());
        done += len;
    }
    if (W > 0) e->head = (e->head + c) % W;
    mark("ring, degrees, batch keys");
    const int id = e->newest + 1;
    Epoch &ep = e->epochs[id % e->n_epochs];
    ep.id = -1;
    ep.L = 0;            // (before build_epoch: its group cut looks at the epoch's records, and the ring entry still holds
    ep.grouped = false;  //  the previous occupant's -- possibly in an older numbering)
    ep.su_inline = false;
    int rc;
    e->merge_miss_host = 0;
    if (inc) { // f1: merge the batch into the previous sorted keys
        HIP_TRY(hipMemsetAsync(e->hub_hist + MERGE_MISS_WORD, 0, sizeof(int), e->bs));
        rc = merge_batch_keys(e, e->in_sorted, e->bk[0], e->bks[0], c * per, e->bk[1], e->bks[1], c * per);
        if (!rc && e->directed) rc = merge_batch_keys(e, e->out_sorted, e->bk[2], e->bks[2], c, e->bk[3], e->bks[3], c);
        // (read with the build's own synchronisations below: no extra wait on the path that finds every key)
        if (!rc) HIP_TRY(hipMemcpyAsync(&e->merge_miss_host, e->hub_hist + MERGE_MISS_WORD, sizeof(int), hipMemcpyDeviceToHost, e->bs));
    } else {
        rc = sort_window_full(e);
    }
    if (rc) return rc;
    mark("sorted keys (merge / full sort)");
    BinBatch bb; // the slide's retired / inserted edges in the in-orientation ((head, row) keys): what the binned tables are patched with
    if (inc) {
        bb.del = e->bk[0];
        bb.ins = e->bk[1];
        bb.nd = bb.ni = c * per;
    }
    rc = build_epoch(e, ep, inc ? &bb : nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->bs));
    if (inc && (e->merge_miss_host != 0 || e->test_force_merge_miss)) {
        e->bin_force_full = true;
        // a retired key was not among the kept sorted keys (an inconsistent window: never seen): the merged arrays
        // cannot be trusted -- the ring itself is right, so sort it afresh and build the epoch again
        e->merge_fallbacks++;
        rc = sort_window_full(e);
        if (rc) return rc;
        rc = build_epoch(e, ep);
        if (rc) return rc;
    }
    mark("hubs, CSRs, group cut + tables");
    ep.L = 0;
    ep.grouped = false;
    if (e->batch_staged) {
        const int L = (int)e->st_b1.size();
        ep.L = L;
        if (L > 0) {
            HIP_TRY(hipMemcpyAsync(ep.b1, e->st_b1.data(), sizeof(int) * (size_t)L, hipMemcpyHostToDevice, e->bs));
            HIP_TRY(hipMemcpyAsync(ep.b2, e->st_b2.data(), sizeof(int) * (size_t)L, hipMemcpyHostToDevice, e->bs));
            HIP_TRY(hipMemcpyAsync(ep.ins, e->st_ins.data(), (size_t)L, hipMemcpyHostToDevice, e->bs));
            if (e->group_at_slide) {
                // CopyOutDegree (gpu/StreamUpdate.cuh:7-17): post-batch out-degree of every tail
                hipLaunchKernelGGL(k_gather_deg, dim3(grid_for(L)), dim3(BLOCK), 0, e->bs, ep.b1, L, e->outdeg,
                                   ep.deg_after);
                HIP_TRY(hipGetLastError());
                if (int grc = epoch_group_records(e, ep)) return grc; // the records grouped by tail, for IncrementalBatchUpdate
            } // (default: both are part of the timed region -- group_records_by_tail, or the resident launch itself)
        }
    }
    HIP_TRY(hipStreamSynchronize(e->bs)); // staged host vectors may be reused now
    mark("batch records");
    e->batch_staged = false;
    ep.id = id;
    e->newest = id;
    if (out_epoch) *out_epoch = id;
    return DPPR_OK;
}

int dppr_add_source(dppr_engine *e, int32_t source, int32_t *out_slot) {
    if (!e || e->broken || source < 0 || source >= e->V) return fail(e, DPPR_ERR_INVALID, "add_source: vertex out of range");
    HIP_TRY(hipSetDevice(e->device));
    Slot s;
    s.source_ext = source;
    s.source = to_int(e, source);
    source = s.source;
    if (int rc = flush_moves(e)) return rc;
    const size_t V = (size_t)e->V;
    HIP_TRY(hipMalloc((void **)&s.p, sizeof(double) * V));
    HIP_TRY(hipMalloc((void **)&s.r, sizeof(double) * V));
    HIP_TRY(hipMalloc((void **)&s.x, sizeof(double) * V));
    HIP_TRY(hipMalloc((void **)&s.x2, sizeof(double) * V));
    HIP_TRY(hipMemsetAsync(s.x, 0, sizeof(double) * V, e->stream));
    HIP_TRY(hipMemsetAsync(s.x2, 0, sizeof(double) * V, e->stream));
    s.act_bytes = (V / 32 + 64) * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void **)&s.act[0], s.act_bytes));
    HIP_TRY(hipMalloc((void **)&s.act[1], s.act_bytes));
    HIP_TRY(hipMemsetAsync(s.act[0], 0, s.act_bytes, e->stream));
    HIP_TRY(hipMemsetAsync(s.act[1], 0, s.act_bytes, e->stream));
    HIP_TRY(hipMalloc((void **)&s.ft[0], sizeof(int) * V));
    HIP_TRY(hipMalloc((void **)&s.ft[1], sizeof(int) * V));
    HIP_TRY(hipMalloc((void **)&s.neg, sizeof(int) * (size_t)std::max(4 * e->c, 1)));
    HIP_TRY(hipMalloc((void **)&s.cnt, sizeof(int) * (CNT_HDR + 2 * MAX_CHUNK)));
    s.log = s.cnt + CNT_HDR; // the per-chunk log sits right behind the counters: one read-back fetches both
    // a row is deferred only if it has >= big_row edges, so at most Ed / big_row of them exist
    // (pieces of <= 1024 edges of rows of >= big_row edges: a row of d edges has ceil(d / 1024) <= d / min(big_row, 512) of them)
    HIP_TRY(hipMalloc((void **)&s.big, sizeof(BigItem) * ((size_t)e->Ed / (size_t)std::min(std::max(e->big_row, 1), 512) + 64)));
    HIP_TRY(hipMalloc((void **)&s.dstats, 2 * sizeof(IterStats)));
    HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * (CNT_HDR + 2 * MAX_CHUNK), e->stream));
    HIP_TRY(hipMemsetAsync(s.dstats, 0, 2 * sizeof(IterStats), e->stream));
    hipLaunchKernelGGL(k_init, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, s.p, s.r, e->V, source);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->slots.push_back(std::move(s));
    if (int rc = recut_stale_groups(e)) return rc; // the source may have received a fresh internal id
    if (Epoch *nw = find_epoch(e, -1)) // a window that sweeps binned: the newest epoch's tables, if it was built before any slot existed
        if (!nw->bin_valid && bin_wanted(e))
            if (int rc = build_bins(e, *nw)) return rc;
    if (out_slot) *out_slot = (int)e->slots.size() - 1;
    return DPPR_OK;
}

#define GET_SLOT(e, slot)                                                                       \
    if (!(e) || (slot) < 0 || (slot) >= (int)(e)->slots.size()) return fail((e), DPPR_ERR_INVALID, "bad slot"); \
    if ((e)->broken) return fail((e), DPPR_ERR_INVALID, "engine unusable after a failed renumbering");            \
    Slot &s = (e)->slots[(size_t)(slot)]
#define GET_EPOCH(e, epoch)                                                       \
    Epoch *epp = find_epoch((e), (epoch));                                        \
    if (!epp) return fail((e), DPPR_ERR_INVALID, "epoch not resident (evicted or never built)"); \
    Epoch &ep = *epp

int dppr_time_batch_grouping(dppr_engine *e, int32_t epoch, int32_t reps, float *out_ms) {
    if (!e || e->broken || reps < 1 || !out_ms) return fail(e, DPPR_ERR_INVALID, "time_batch_grouping: reps >= 1");
    GET_EPOCH(e, epoch);
    HIP_TRY(hipSetDevice(e->device));
    *out_ms = 0.0f;
    const int L = ep.L;
    if (L <= 0) return DPPR_OK;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    int *deg_scratch = reinterpret_cast<int *>(e->su_term);
    for (int k = 0; k < reps; ++k) // what group_records_by_tail enqueues inside the timed region, the degrees into scratch
        if (int rc = enqueue_grouping(e, ep, deg_scratch, nullptr, 0, nullptr, 0)) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *out_ms = ms / (float)reps;
    return DPPR_OK;
}

int dppr_debug_dump(dppr_engine *e, char *buf, int32_t cap) {
    if (!e || !buf || cap < 2) return 0;
    std::string o;
    char line[512];
#define add(...)                                  \
    do {                                          \
        snprintf(line, sizeof(line), __VA_ARGS__); \
        o += line;                                \
    } while (0)
    add("dppr engine %p device %d: V %d W %d c %d directed %d n_int %d newest epoch %d broken %d\n", (void *)e, e->device, e->V, e->W, e->c,
        e->directed, e->n_int, e->newest, (int)e->broken);
    add("last error: %s\n", e->err.empty() ? "(none)" : e->err.c_str());
    add("slides that discarded their key merge and re-sorted the window (a retired key was missing): %lld\n", e->merge_fallbacks);
    add("binned-sweep tables: %lld epochs patched, %lld built by the sorts (%d slides since the cuts were made)\n", e->bin_patched, e->bin_rebuilt, e->bin_slides_since_cut);
    add("id lookahead (dppr_hint_next_batch): %lld id arrays taken from it so far (%lld entries resolved at the call), renumberings %llu\n", e->pre_hits, e->pre_misses, e->renumber_epoch);
    add("resident launches: mode %d ok %d retry %d time limit %llu ticks (100 MHz) rollcall_extra %d; schedule %d merge %d\n", e->persist_mode,
        (int)e->persist_ok, e->persist_retry, e->persist_ticks, e->persist_rollcall_extra, e->schedule, (int)e->merge_phases);
    // Device words through a stream of their own, waited for at most ~2 s in total. The copies land in a PINNED buffer the engine
    // owns for its whole life (a copy into pageable memory is staged and may block inside the call on a wedged device,
    // and one that completes after its stack destination is gone writes into dead memory); once a copy has not completed in time no
    // further one is issued and the side stream is abandoned, not destroyed (hipStreamDestroy would wait for it).
    hipStream_t side = nullptr;
    const bool have_side = e->dump_pin && hipSetDevice(e->device) == hipSuccess && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess;
    bool side_stuck = false;
    int polls_left = 2000; // x 1 ms, shared by all fetches of this dump
    auto fetch = [&](void *dst, const void *src, size_t bytes) -> bool {
        if (!have_side || side_stuck || !src || bytes > dppr_engine::DUMP_PIN_BYTES) return false;
        if (hipMemcpyAsync(e->dump_pin, src, bytes, hipMemcpyDeviceToHost, side) != hipSuccess) return false;
        while (polls_left-- > 0) {
            const hipError_t q = hipStreamQuery(side);
            if (q == hipSuccess) {
                memcpy(dst, e->dump_pin, bytes);
                return true;
            }
            if (q != hipErrorNotReady) return false;
            timespec ts{0, 1000000};
            nanosleep(&ts, nullptr);
        }
        side_stuck = true; // the copy stays queued: its destination outlives it
        return false;
    };
    add("engine stream: %s\n", hipStreamQuery(e->stream) == hipSuccess ? "idle" : "BUSY (work enqueued or running)");
    {
        static thread_local GridBar hb;
        if (fetch(&hb, e->bar, sizeof(GridBar))) {
            unsigned long long roll = 0, sub[2] = {0, 0};
            for (int s = 0; s < BAR_SUBS; ++s) roll += hb.roll[s].w >> 32;
            for (int par = 0; par < 2; ++par)
                for (int s = 0; s < BAR_SUBS; ++s) sub[par] += hb.sub[par][0][s].w >> 32;
            add("GridBar: gen %llu (0 pending, %llu ready, %llu abort) roll-call check-ins %llu, arrivals (replica 0) even sweeps %llu odd sweeps %llu\n",
                hb.gen.w, (unsigned long long)BAR_READY, (unsigned long long)BAR_ABORT, roll, sub[0], sub[1]);
        } else {
            add("GridBar: not readable (copy did not complete within 2 s)\n");
        }
    }
    for (size_t i = 0; i < e->slots.size(); ++i) {
        const Slot &s = e->slots[i];
        int h[CNT_HDR] = {0};
        const bool ok = fetch(h, s.cnt, sizeof(h));
        add("slot %zu: source %d (internal %d) converged %d last_epoch %d iterations %lld persist launches %lld aborts %lld binned sweeps %lld; "
            "device counters %s[%d %d %d | cand %d | big %d %d | status 0x%x]\n", i, s.source_ext, s.source, (int)s.converged, s.last_epoch,
            (long long)s.st.iterations, (long long)s.st.persist_launches, (long long)s.st.persist_aborts, (long long)s.st.binned_sweeps,
            ok ? "" : "(unreadable) ", h[0], h[1], h[2], h[3], h[5], h[6], (unsigned)h[7]);
    }
    for (size_t i = 0; i < e->groups.size(); ++i) {
        const Group &g = e->groups[i];
        int h[3 * GS_MAX] = {0}, st = 0;
        const bool ok = fetch(h, g.cnt, sizeof(h)) && fetch(&st, g.mlog, sizeof(int));
        long long f[3] = {0, 0, 0};
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < GS_MAX; ++k) f[r] += h[r * GS_MAX + k];
        add("group %zu: %d sources, rows of %d doubles, converged %d last_epoch %d iterations %lld multi-sweep launches %lld aborts %lld; device: %sfrontier "
            "pairs in the three rotating rows %lld %lld %lld, multi-sweep status 0x%x\n", i, g.n, g.gw, (int)g.converged, g.last_epoch,
            (long long)g.st.iterations, (long long)g.st.persist_launches, (long long)g.st.persist_aborts, ok ? "" : "(unreadable) ", f[0], f[1], f[2],
            (unsigned)st);
    }
    if (side_stuck) add("(a device read did not complete within 2 s: the remaining ones were skipped, the side stream is abandoned)\n");
    if (have_side && !side_stuck) (void)hipStreamDestroy(side);
#undef add
    const size_t n = std::min(o.size(), (size_t)cap - 1);
    memcpy(buf, o.data(), n);
    buf[n] = 0;
    return (int)n;
}

int dppr_init_solve(dppr_engine *e, int32_t slot, double eps, float *out_ms) { return dppr_init_solve_at(e, slot, -1, eps, out_ms); }

int dppr_init_solve_at(dppr_engine *e, int32_t slot, int32_t epoch, double eps, float *out_ms) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(k_init, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, s.p, s.r, e->V, s.source);
    HIP_TRY(hipGetLastError());
    s.converged = false;
    s.park_eps = 0.0; // (parked rows are zero again)
    int rc = main_loop_inspect(e, s, ep, 0, eps);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    s.converged = true;
    s.conv_eps = eps;
    s.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_incremental_batch_update(dppr_engine *e, int32_t slot, int32_t epoch) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if (!epoch_in_sequence(s.last_epoch, ep.id)) return fail(e, DPPR_ERR_INVALID, "epoch out of sequence for this source");
    HIP_TRY(hipSetDevice(e->device));
    // after a converged solve the update also lists the tails that left [-eps, eps] (dppr_seed_lists)
    const bool seeded = s.converged;
    int rc = prepare_epoch(e, ep);
    if (rc) return rc;
    rc = stream_update(e, s, ep, s.conv_eps, seeded);
    if (rc) return rc;
    s.seed_lists_valid = seeded;
    s.last_epoch = ep.id;
    s.converged = false;
    s.phase0_done = false;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_execute_main_loop(dppr_engine *e, int32_t slot, int32_t epoch, int phase, double eps) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if ((phase != 0 && phase != 1) || !(eps > 0)) return fail(e, DPPR_ERR_INVALID, "phase must be 0/1, eps > 0");
    HIP_TRY(hipSetDevice(e->device));
    int rc = settle_parked(e, s.p, s.r, 1, eps, &s.park_eps, &s.st);
    if (rc) return rc;
    rc = main_loop_inspect(e, s, ep, phase, eps);
    if (rc) return rc;
    if (phase == 0) {
        s.phase0_done = true;
        s.phase0_eps = eps;
    } else if (s.phase0_done && s.phase0_eps == eps) { // both phases done: |r| <= eps everywhere
        s.converged = true;
        s.conv_eps = eps;
    }
    return DPPR_OK;
}

int dppr_update(dppr_engine *e, int32_t slot, int32_t epoch, double eps, float *out_ms) {
    GET_SLOT(e, slot);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    if (!epoch_in_sequence(s.last_epoch, ep.id)) return fail(e, DPPR_ERR_INVALID, "epoch out of sequence for this source");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->persist_ok && e->persist_mode && e->persist_retry > 0 && --e->persist_retry == 0)
        e->persist_ok = true; // a resident launch gave up a while ago (the CUs were shared): try them again
    s.seed_lists_valid = false;
    // Seeding from the batch tails is exact only if every |r| <= eps beforehand
    // (the state a completed solve leaves). Otherwise fall back to full Inspect passes.
    // Merged loop (dppr_set_phase_merge, eager schedule): residuals of both signs are pushed in ONE loop, to eps / merge_div.
    const bool merged = e->merge_phases && e->schedule == DPPR_SCHEDULE_EAGER;
    if (merged) eps = eps / e->merge_div;
    const bool seeded = s.converged && s.conv_eps <= eps;
    const bool ahead = seeded && can_batch_ahead(e, s, ep) && resident_arena(e, ep);
    int rc = settle_parked(e, s.p, s.r, 1, eps, &s.park_eps, &s.st);
    if (rc) return rc;
    rc = prepare_epoch(e, ep);
    if (rc) return rc;
    if (e->raw_backoff > 0) --e->raw_backoff;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    // A whole-batch resident launch applies the records itself (PLAN_UPDATE) -- grouped at slide time and cut into the sweep groups'
    // ranges, or (default accounting) RAW: the launch finds, orders and applies every group's records itself. Only the counters
    // and the GridBar are cleared here. Should the launch call itself off, nothing was changed and the update runs as its own
    // kernels after all.
    const bool raw_ok = !ep.grouped && ep.L > 0 && ep.L <= RES_RAW_STEPS * sweep_block(e) && e->raw_backoff == 0;
    bool inline_su = ahead && e->res_update && ((ep.su_inline && ep.grouped) || raw_ok);
    if (inline_su) {
        hipLaunchKernelGGL(k_su_keys, dim3(1), dim3(BLOCK), 0, e->stream, ep.b1, 0, e->su_k[0], e->su_v[0],
                           reinterpret_cast<unsigned long long *>(e->bar), (int)(sizeof(GridBar) / sizeof(unsigned long long)), s.cnt, 5);
        HIP_TRY(hipGetLastError());
    } else {
        rc = stream_update(e, s, ep, eps, seeded, ahead);
    }
    if (rc) return rc;
    s.converged = false;
    auto update_after_abort = [&]() -> int { // (inline_su only) the launch called itself off
        if (!inline_su) return DPPR_OK;
        if (e->launch_called_off) {
            inline_su = false;
            return stream_update(e, s, ep, eps, seeded, false);
        }
        s.st.records += ep.L;
        return DPPR_OK;
    };
    if (merged && ahead) { // a window that runs resident: the whole merged loop as ONE launch that seeds itself
        int stage = 0;
        bool p1 = false;
        LoopEntry en0, en1;
        rc = batch_ahead(e, s, ep, eps, &stage, &en0, &en1, &p1, true, inline_su);
        if (rc) return rc;
        rc = update_after_abort();
        if (rc) return rc;
        if (stage != 2) { // out of sweeps, or the roll-call failed (then the update's lists stand: add the negative tails)
            if (en0.it == 0 && !en0.dense) {
                hipLaunchKernelGGL(k_filter, dim3(grid_for(std::max(ep.L, 1))), dim3(BLOCK), 0, e->stream, s.neg, s.cnt + 3, s.r, 1, eps,
                                   s.ft[0], s.cnt + 0);
                HIP_TRY(hipGetLastError());
            }
            rc = run_frontier_loop(e, s, ep, PHASE_BOTH, eps, 0, 0, en0);
            if (rc) return rc;
        }
    } else if (merged) {
        if (seeded) { // the frontier: the tails the update left above eps (ft[0]) and those it left below -eps (the candidates)
            hipLaunchKernelGGL(k_filter, dim3(grid_for(std::max(ep.L, 1))), dim3(BLOCK), 0, e->stream, s.neg, s.cnt + 3, s.r, 1, eps,
                               s.ft[0], s.cnt + 0);
            HIP_TRY(hipGetLastError());
            rc = run_frontier_loop(e, s, ep, PHASE_BOTH, eps, 0, 0);
        } else {
            rc = main_loop_inspect(e, s, ep, PHASE_BOTH, eps);
        }
        if (rc) return rc;
    } else if (seeded) {
        int stage = 0;
        bool p1_seeded = false;
        LoopEntry en0, en1;
        if (ahead) {
            rc = batch_ahead(e, s, ep, eps, &stage, &en0, &en1, &p1_seeded, false, inline_su);
            if (rc) return rc;
            rc = update_after_abort();
            if (rc) return rc;
        }
        if (stage == 0) {
            rc = run_frontier_loop(e, s, ep, 0, eps, 0, 0, en0);
            if (rc) return rc;
        }
        if (stage <= 1 && !p1_seeded && inline_su) {
            // the update ran inside the launch and recorded no candidates: phase 1 starts from a full Inspect
            rc = main_loop_inspect(e, s, ep, 1, eps);
            if (rc) return rc;
        } else if (stage <= 1) {
            if (!p1_seeded) { // phase 1: candidates recorded by the update, re-checked now
                HIP_TRY(hipMemsetAsync(s.cnt, 0, sizeof(int) * 3, e->stream));
                hipLaunchKernelGGL(k_filter, dim3(grid_for(std::max(ep.L, 1))), dim3(BLOCK), 0, e->stream, s.neg,
                                   s.cnt + 3, s.r, 1, eps, s.ft[0], s.cnt + 0);
                HIP_TRY(hipGetLastError());
            }
            rc = run_frontier_loop(e, s, ep, 1, eps, 0, 0, en1);
            if (rc) return rc;
        }
    } else {
        rc = main_loop_inspect(e, s, ep, 0, eps);
        if (rc) return rc;
        rc = main_loop_inspect(e, s, ep, 1, eps);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    s.st.gpu_ms += ms;
    s.st.batches++;
    s.converged = true;
    s.conv_eps = eps; // (the merged loop's eps / merge_div)
    s.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_read(dppr_engine *e, int32_t slot, double *p, double *r) {
    GET_SLOT(e, slot);
    HIP_TRY(hipSetDevice(e->device));
    std::lock_guard<std::mutex> map_lk(e->map_mu); // (beside a concurrent slide: the map copy and the gathers see one state of the id space)
    int rc = sync_map(e);
    if (rc) return rc;
    const double *src[2] = {s.p, s.r};
    double *dst[2] = {p, r};
    for (int k = 0; k < 2; ++k) {
        if (!dst[k]) continue;
        hipLaunchKernelGGL(k_int_to_ext, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, src[k], e->d_ext2int, e->V,
                           e->d_xfer);
        HIP_TRY(hipMemcpyAsync(dst[k], e->d_xfer, sizeof(double) * (size_t)e->V, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return DPPR_OK;
}

int dppr_write(dppr_engine *e, int32_t slot, const double *p, const double *r) {
    GET_SLOT(e, slot);
    HIP_TRY(hipSetDevice(e->device));
    // vertices that carry a value get an internal id first
    for (int v = 0; v < e->V; ++v)
        if ((p && p[v] != 0.0) || (r && r[v] != 0.0)) (void)to_int(e, v);
    int rc = flush_moves(e);
    if (rc) return rc;
    rc = sync_map(e);
    if (rc) return rc;
    rc = recut_stale_groups(e);
    if (rc) return rc;
    const double *src[2] = {p, r};
    double *dst[2] = {s.p, s.r};
    for (int k = 0; k < 2; ++k) {
        if (!src[k]) continue;
        HIP_TRY(hipMemcpyAsync(e->d_xfer, src[k], sizeof(double) * (size_t)e->V, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemsetAsync(dst[k], 0, sizeof(double) * (size_t)e->V, e->stream));
        hipLaunchKernelGGL(k_ext_to_int, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, e->d_xfer, e->d_ext2int, e->V,
                           dst[k]);
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    s.converged = false;
    s.phase0_done = false;
    if (r) s.park_eps = 0.0; // (every vertex that carries a value has a live id now; parked rows were written as zeros)
    s.last_epoch = -2; // the caller supplied the state: which batches it contains is the caller's business
    s.seed_lists_valid = false;
    return DPPR_OK;
}

int dppr_seed_lists(dppr_engine *e, int32_t slot, int phase, int32_t *out_ids, int32_t *out_count) {
    GET_SLOT(e, slot);
    if (!out_ids || !out_count || (phase != 0 && phase != 1)) return DPPR_ERR_INVALID;
    if (!s.seed_lists_valid)
        return fail(e, DPPR_ERR_INVALID, "seed_lists: only right after dppr_incremental_batch_update on a converged state");
    HIP_TRY(hipSetDevice(e->device));
    int n = 0;
    int rc = read_count(e, s.cnt + (phase == 0 ? 0 : 3), &n);
    if (rc) return rc;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(out_ids, phase == 0 ? s.ft[0] : s.neg, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    for (int i = 0; i < n; ++i) out_ids[i] = e->int2ext[(size_t)out_ids[i]];
    *out_count = n;
    return DPPR_OK;
}

int dppr_stats(dppr_engine *e, int32_t slot, dppr_stats_t *out) {
    GET_SLOT(e, slot);
    if (!out) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    int rc = pull_device_stats(e, s);
    if (rc) return rc;
    // every enqueued vertex is a frontier member of a later iteration, except the seeds
    s.st.sum_N = s.st.sum_F;
    // SURVEY.md 8(d); its Inspect term (8 bytes per vertex and pass) is counted for the passes that RAN:
    // after a converged solve the frontier is seeded from the batch tails and no vertex is scanned
    s.st.algorithmic_bytes = 8ll * s.st.inspected + 45ll * s.st.records + 72ll * s.st.sum_F + 24ll * s.st.sum_E +
                             4ll * s.st.sum_N;
    *out = s.st;
    return DPPR_OK;
}

int dppr_reset_stats(dppr_engine *e, int32_t slot) {
    GET_SLOT(e, slot);
    HIP_TRY(hipSetDevice(e->device));
    s.st = dppr_stats_t{};
    HIP_TRY(hipMemsetAsync(s.dstats, 0, 2 * sizeof(IterStats), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_inspect(dppr_engine *e, int32_t slot, int phase, double eps, int32_t *out_ids, int32_t *out_count) {
    GET_SLOT(e, slot);
    if (!out_ids || !out_count || (phase != 0 && phase != 1)) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemsetAsync(s.cnt + 4, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL(k_inspect, dim3(grid_for(e->n_int, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream, s.r,
                       e->n_int, phase, eps, s.ft[1], s.cnt + 4);
    HIP_TRY(hipGetLastError());
    int n = 0;
    int rc = read_count(e, s.cnt + 4, &n);
    if (rc) return rc;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(out_ids, s.ft[1], sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    if (e->n_parked > 0 && eps < s.park_eps) { // parked rows are inert down to park_eps only
        const int base = e->V - e->n_parked;
        int m = 0;
        HIP_TRY(hipMemsetAsync(s.cnt + 4, 0, sizeof(int), e->stream));
        hipLaunchKernelGGL(k_inspect, dim3(grid_for(e->n_parked, BLOCK * INSPECT_ITEMS)), dim3(BLOCK), 0, e->stream, s.r + base,
                           e->n_parked, phase, eps, s.ft[1], s.cnt + 4);
        HIP_TRY(hipGetLastError());
        rc = read_count(e, s.cnt + 4, &m);
        if (rc) return rc;
        if (m > 0) {
            HIP_TRY(hipMemcpyAsync(out_ids + n, s.ft[1], sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            for (int i = n; i < n + m; ++i) out_ids[i] += base;
        }
        n += m;
    }
    for (int i = 0; i < n; ++i) out_ids[i] = e->int2ext[(size_t)out_ids[i]];
    *out_count = n;
    return DPPR_OK;
}

int dppr_graph_edges(dppr_engine *e, int32_t epoch, int32_t *out) {
    if (!e || !out) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    *out = ep.Ed;
    return DPPR_OK;
}

// internal CSR (rows by internal id, columns internal) -> external CSR with ascending rows
static void csr_to_external(const dppr_engine *e, const std::vector<int> &irow, const std::vector<int> &icol,
                            int32_t *row_ptr, int32_t *col) {
    int off = 0;
    std::vector<int> tmp;
    for (int v = 0; v < e->V; ++v) {
        if (row_ptr) row_ptr[v] = off;
        const int m = e->ext2int[(size_t)v];
        if (m >= 0) {
            tmp.clear();
            for (int j = irow[(size_t)m]; j < irow[(size_t)m + 1]; ++j) tmp.push_back(e->int2ext[(size_t)icol[(size_t)j]]);
            std::sort(tmp.begin(), tmp.end());
            if (col) std::copy(tmp.begin(), tmp.end(), col + off);
            off += (int)tmp.size();
        }
    }
    if (row_ptr) row_ptr[e->V] = off;
}

int dppr_read_graph(dppr_engine *e, int32_t epoch, int32_t *row_ptr, int32_t *col, int32_t *out_degree) {
    if (!e) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    HIP_TRY(hipSetDevice(e->device));
    std::vector<int> irow((size_t)e->V + 1), icol((size_t)std::max(ep.Ed, 1)), ideg((size_t)e->V);
    HIP_TRY(hipMemcpyAsync(irow.data(), ep.row_ptr, sizeof(int) * ((size_t)e->V + 1), hipMemcpyDeviceToHost, e->stream));
    if (ep.Ed > 0) {
        int *tmp = reinterpret_cast<int *>(e->keys_a); // scratch
        hipLaunchKernelGGL(k_split_adj, dim3(grid_for(ep.Ed)), dim3(BLOCK), 0, e->stream, ep.adj, ep.Ed, tmp);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(icol.data(), tmp, sizeof(int) * (size_t)ep.Ed, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipMemcpyAsync(ideg.data(), e->outdeg, sizeof(int) * (size_t)e->V, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    csr_to_external(e, irow, icol, row_ptr, col);
    if (out_degree)
        for (int v = 0; v < e->V; ++v) {
            const int m = e->ext2int[(size_t)v];
            out_degree[v] = m >= 0 ? ideg[(size_t)m] : 0;
        }
    return DPPR_OK;
}

int dppr_read_out_graph(dppr_engine *e, int32_t epoch, int32_t *row_ptr, int32_t *col) {
    if (!e) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    HIP_TRY(hipSetDevice(e->device));
    std::vector<int> irow((size_t)e->V + 1), icol((size_t)std::max(ep.Ed, 1));
    HIP_TRY(hipMemcpyAsync(irow.data(), ep.out_row_ptr, sizeof(int) * ((size_t)e->V + 1), hipMemcpyDeviceToHost, e->stream));
    if (ep.Ed > 0)
        HIP_TRY(hipMemcpyAsync(icol.data(), ep.out_col, sizeof(int) * (size_t)ep.Ed, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    csr_to_external(e, irow, icol, row_ptr, col);
    return DPPR_OK;
}

int dppr_trace_enable(dppr_engine *e, int32_t slot, int on) {
    GET_SLOT(e, slot);
    s.trace = on != 0;
    s.trace_ids.clear();
    s.trace_off.assign(1, 0);
    return DPPR_OK;
}

int dppr_trace_get(dppr_engine *e, int32_t slot, int64_t *n_iters, int64_t *n_ids, int64_t *offsets, int32_t *ids) {
    GET_SLOT(e, slot);
    const int64_t ni = s.trace_off.empty() ? 0 : (int64_t)s.trace_off.size() - 1;
    if (n_iters) *n_iters = ni;
    if (n_ids) *n_ids = (int64_t)s.trace_ids.size();
    if (offsets && !s.trace_off.empty()) memcpy(offsets, s.trace_off.data(), sizeof(int64_t) * s.trace_off.size());
    if (ids && !s.trace_ids.empty()) memcpy(ids, s.trace_ids.data(), sizeof(int32_t) * s.trace_ids.size());
    return DPPR_OK;
}

#define GET_GROUP(e, gid)                                                                             \
    if (!(e) || (gid) < 0 || (gid) >= (int)(e)->groups.size()) return fail((e), DPPR_ERR_INVALID, "bad group"); \
    if ((e)->broken) return fail((e), DPPR_ERR_INVALID, "engine unusable after a failed renumbering");           \
    Group &g = (e)->groups[(size_t)(gid)]

int dppr_add_source_group(dppr_engine *e, const int32_t *sources, int32_t n, int32_t *out_group) {
    if (!e || e->broken || !sources || n < 1 || n > GS_MAX) return fail(e, DPPR_ERR_INVALID, "add_source_group: 1..16 sources");
    if (!ids_in_range(e, sources, n)) return fail(e, DPPR_ERR_INVALID, "add_source_group: vertex out of range"); // before anything is touched
    HIP_TRY(hipSetDevice(e->device));
    Group g;
    g.n = n;
    g.gw = e->group_full_rows ? (n > OCT ? 2 * OCT : OCT) : row_width(n); // doubles per vertex: the sources + at most one of padding
    g.spl = row_spl(g.gw);
    for (int s = 0; s < GS_MAX; ++s) g.src.s[s] = -1;
    for (int s = 0; s < n; ++s) g.src_ext[s] = sources[s];
    const size_t V = (size_t)e->V, row = sizeof(double) * (size_t)g.gw;
    g.act_bytes = (V / 32 + 1024 / 32 + 4) * sizeof(uint32_t);
    auto release = [&]() { // (an allocation failed: nothing of this group stays behind)
        (void)hipFree(g.p); (void)hipFree(g.r); (void)hipFree(g.x); (void)hipFree(g.x2); (void)hipFree(g.act[0]); (void)hipFree(g.act[1]);
        (void)hipFree(g.cnt); (void)hipFree(g.mlog); (void)hipFree(g.dstats); (void)hipFree(g.gq);
    };
#define GRP_TRY(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            release();                                                                                  \
            e->err = std::string("add_source_group: ") + hipGetErrorString(_e);                         \
            return _e == hipErrorOutOfMemory ? DPPR_ERR_NOMEM : DPPR_ERR_HIP;                           \
        }                                                                                               \
    } while (0)
    GRP_TRY(hipMalloc((void **)&g.p, row * V));
    GRP_TRY(hipMalloc((void **)&g.r, row * V));
    const size_t xrow = sizeof(double) * (size_t)x_stride(g.gw); // snapshot rows never straddle a 128-byte line (dppr_multi.hpp)
    GRP_TRY(hipMalloc((void **)&g.x, xrow * V));
    GRP_TRY(hipMalloc((void **)&g.x2, xrow * V));
    GRP_TRY(hipMalloc((void **)&g.act[0], g.act_bytes));
    GRP_TRY(hipMalloc((void **)&g.act[1], g.act_bytes));
    GRP_TRY(hipMalloc((void **)&g.cnt, sizeof(int) * (5 * GS_MAX + MAX_CHUNK * GS_MAX))); // rows 3, 4: scratch of multi-sweep launches
    GRP_TRY(hipMalloc((void **)&g.mlog, sizeof(int) * (size_t)(GMULTI_MAX + 2) * GS_MAX));
    GRP_TRY(hipMalloc((void **)&g.dstats, 2 * sizeof(IterStats)));
    GRP_TRY(hipMalloc((void **)&g.gq, sizeof(int) * 3 * GQ_PAD));
    GRP_TRY(hipMemsetAsync(g.gq, 0, sizeof(int) * 3 * GQ_PAD, e->stream));
    GRP_TRY(hipMemsetAsync(g.act[0], 0, g.act_bytes, e->stream));
    GRP_TRY(hipMemsetAsync(g.act[1], 0, g.act_bytes, e->stream));
    GRP_TRY(hipMemsetAsync(g.cnt, 0, sizeof(int) * (5 * GS_MAX + MAX_CHUNK * GS_MAX), e->stream));
    GRP_TRY(hipMemsetAsync(g.dstats, 0, 2 * sizeof(IterStats), e->stream));
    // the memory is there: now the ids (a source outside the window receives one; a parked one is revived)
    for (int s = 0; s < n; ++s) (void)to_int(e, sources[s]);
    for (int s = 0; s < n; ++s) g.src.s[s] = e->ext2int[(size_t)sources[s]]; // (after ALL revivals: one may move another)
    if (int rc = flush_moves(e)) {
        release();
        return rc;
    }
    hipLaunchKernelGGL(k_ginit, dim3(grid_for((int64_t)e->V * g.gw)), dim3(BLOCK), 0, e->stream, g.p, g.r, e->V, g.gw, g.src);
    GRP_TRY(hipGetLastError());
    GRP_TRY(hipStreamSynchronize(e->stream));
#undef GRP_TRY
    e->groups.push_back(g);
    e->any_groups = true; // (recut_stale_groups below adds the second group table to resident epochs)
    if (g.spl == 2) e->wide_groups = true;
    if (int rc = recut_stale_groups(e)) return rc;
    if (out_group) *out_group = (int)e->groups.size() - 1;
    return DPPR_OK;
}

int dppr_group_init_solve(dppr_engine *e, int32_t group, double eps, float *out_ms) {
    return dppr_group_init_solve_at(e, group, -1, eps, out_ms);
}

int dppr_group_init_solve_at(dppr_engine *e, int32_t group, int32_t epoch, double eps, float *out_ms) {
    GET_GROUP(e, group);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    hipLaunchKernelGGL(k_ginit, dim3(grid_for((int64_t)e->V * g.gw)), dim3(BLOCK), 0, e->stream, g.p, g.r, e->V, g.gw, g.src);
    HIP_TRY(hipGetLastError());
    g.converged = false;
    g.park_eps = 0.0;
    int rc = group_loop(e, g, ep, 0, eps, false);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    g.converged = true; // r = e_s >= 0 and phase 0 left every r <= eps: nothing is below -eps
    g.conv_eps = eps;
    g.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_group_update(dppr_engine *e, int32_t group, int32_t epoch, double eps, float *out_ms) {
    GET_GROUP(e, group);
    GET_EPOCH(e, epoch);
    if (!(eps > 0)) return fail(e, DPPR_ERR_INVALID, "eps must be positive");
    if (!epoch_in_sequence(g.last_epoch, ep.id)) return fail(e, DPPR_ERR_INVALID, "epoch out of sequence for this group");
    HIP_TRY(hipSetDevice(e->device));
    // seeding from the batch tails is exact only if every |r| <= eps beforehand (dppr_update has the same rule)
    const bool merged = e->merge_phases && e->schedule == DPPR_SCHEDULE_EAGER; // (dppr_set_phase_merge)
    if (merged) eps = eps / e->merge_div;
    const bool tails = g.converged && g.conv_eps <= eps && e->group_tail_seeding;
    int rc = settle_parked(e, g.p, g.r, g.gw, eps, &g.park_eps, &g.st);
    if (rc) return rc;
    rc = prepare_epoch(e, ep);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    rc = group_stream_update(e, g, ep);
    if (rc) return rc;
    g.converged = false;
    if (merged) {
        rc = group_loop(e, g, ep, PHASE_BOTH, eps, tails);
        if (rc) return rc;
    } else {
        rc = group_loop(e, g, ep, 0, eps, tails);
        if (rc) return rc;
        rc = group_loop(e, g, ep, 1, eps, tails);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    if (out_ms) *out_ms = ms;
    g.st.gpu_ms += ms;
    g.st.batches++;
    g.converged = true;
    g.conv_eps = eps;
    g.last_epoch = ep.id;
    return DPPR_OK;
}

int dppr_group_read(dppr_engine *e, int32_t group, int32_t index, double *p, double *r) {
    GET_GROUP(e, group);
    if (index < 0 || index >= g.n) return fail(e, DPPR_ERR_INVALID, "group_read: bad source index");
    HIP_TRY(hipSetDevice(e->device));
    std::lock_guard<std::mutex> map_lk(e->map_mu); // (as dppr_read)
    int rc = sync_map(e);
    if (rc) return rc;
    const double *src[2] = {g.p, g.r};
    double *dst[2] = {p, r};
    for (int k = 0; k < 2; ++k) {
        if (!dst[k]) continue;
        hipLaunchKernelGGL(k_gint_to_ext, dim3(grid_for(e->V)), dim3(BLOCK), 0, e->stream, src[k], g.gw, index, e->d_ext2int,
                           e->V, e->d_xfer);
        HIP_TRY(hipMemcpyAsync(dst[k], e->d_xfer, sizeof(double) * (size_t)e->V, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return DPPR_OK;
}

int dppr_group_reset_stats(dppr_engine *e, int32_t group) {
    GET_GROUP(e, group);
    HIP_TRY(hipSetDevice(e->device));
    g.st = dppr_stats_t{};
    HIP_TRY(hipMemsetAsync(g.dstats, 0, 2 * sizeof(IterStats), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return DPPR_OK;
}

int dppr_group_stats(dppr_engine *e, int32_t group, dppr_stats_t *out) {
    GET_GROUP(e, group);
    if (!out) return DPPR_ERR_INVALID;
    HIP_TRY(hipSetDevice(e->device));
    static thread_local IterStats h[2];
    HIP_TRY(hipMemcpyAsync(h, g.dstats, sizeof(h), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long t = 0, ts = 0;
    for (int i = 0; i < STAT_SLOTS; ++i) {
        t += h[0].blk_E[i];
        ts += h[1].blk_E[i];
    }
    g.st.sum_E = (int64_t)(t + ts);
    g.st.sweep_E = (int64_t)ts;
    g.st.sum_N = g.st.sum_F;
    g.st.algorithmic_bytes = 8ll * g.st.inspected + 45ll * g.st.records + 72ll * g.st.sum_F + 24ll * g.st.sum_E +
                             4ll * g.st.sum_N;
    *out = g.st;
    return DPPR_OK;
}

#ifdef DPPR_STAMPS
// diagnostic build only: copy the stage stamps of the last sweep (rows x 8 clock values)
extern "C" int dppr_debug_bin_stamps(unsigned long long *out, int which, int rows) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dppr::g_bin_stamps), sizeof(unsigned long long) * 6 * (size_t)rows,
                               sizeof(unsigned long long) * 6 * 16384 * (size_t)which) == hipSuccess ? 0 : -2;
}
extern "C" int dppr_debug_stamps(unsigned long long *out, int rows) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dppr::g_stamps), sizeof(unsigned long long) * 8 * (size_t)rows) == hipSuccess
               ? 0 : -2;
}
#endif

int dppr_bench_atomics(int device, int64_t table_elems, int64_t n, int scope, int reps, float *out_ms) {
    if (table_elems <= 0 || (table_elems & (table_elems - 1)) || n <= 0 || reps <= 0 || !out_ms) return DPPR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DPPR_ERR_HIP;
    double *table = nullptr, *sink = nullptr;
    hipEvent_t a, b;
    if (hipMalloc((void **)&table, sizeof(double) * (size_t)table_elems) != hipSuccess) return DPPR_ERR_NOMEM;
    (void)hipMalloc((void **)&sink, sizeof(double));
    (void)hipMemset(table, 0, sizeof(double) * (size_t)table_elems);
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 2048;
    auto launch = [&]() {
        if (scope == 2) // (calibration only) adds whose result is not used
            hipLaunchKernelGGL(k_bench_scatter<0>, dim3(grid), dim3(BLOCK), 0, 0, table, (uint64_t)table_elems - 1, n);
        else if (scope == 3) // ... and plain scattered 8-byte stores
            hipLaunchKernelGGL(k_bench_scatter<1>, dim3(grid), dim3(BLOCK), 0, 0, table, (uint64_t)table_elems - 1, n);
        else if (scope == 0)
            hipLaunchKernelGGL(k_bench_atomics<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(BLOCK), 0, 0, table,
                               (uint64_t)table_elems - 1, n, sink);
        else
            hipLaunchKernelGGL(k_bench_atomics<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(BLOCK), 0, 0, table,
                               (uint64_t)table_elems - 1, n, sink);
    };
    launch(); // warm-up
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    hipError_t err = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    *out_ms = ms / reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(table);
    (void)hipFree(sink);
    return err == hipSuccess ? DPPR_OK : DPPR_ERR_HIP;
}

int dppr_bench_line_fills(int device, int64_t table_bytes, int64_t lines, int reps, float *out_ms) {
    if (table_bytes < 128 || (table_bytes & (table_bytes - 1)) || lines <= 0 || reps <= 0 || !out_ms) return DPPR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DPPR_ERR_HIP;
    double2 *table = nullptr;
    double *sink = nullptr;
    if (hipMalloc((void **)&table, (size_t)table_bytes) != hipSuccess) return DPPR_ERR_NOMEM;
    (void)hipMalloc((void **)&sink, sizeof(double));
    (void)hipMemset(table, 0, (size_t)table_bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 2048; // 2048 x 128 octets
    const int per = (int)std::max<int64_t>(8, (lines + (int64_t)grid * 128 - 1) / ((int64_t)grid * 128) / 8 * 8);
    auto launch = [&]() {
        hipLaunchKernelGGL(k_bench_lines<8>, dim3(grid), dim3(1024), 0, 0, table, (uint64_t)(table_bytes / 128) - 1, per, sink);
    };
    launch(); // warm-up
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    hipError_t err = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    // per launch, scaled to the number of lines asked for (the launch fetches grid * 128 * per of them)
    *out_ms = ms / reps * (float)((double)lines / ((double)grid * 128.0 * per));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(table);
    (void)hipFree(sink);
    return err == hipSuccess ? DPPR_OK : DPPR_ERR_HIP;
}

int dppr_bench_stream_copy(int device, int64_t bytes, int reps, float *out_ms) {
    if (bytes < 16 || reps <= 0 || !out_ms) return DPPR_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return DPPR_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DPPR_ERR_HIP;
    double2 *src = nullptr, *dst = nullptr;
    if (hipMalloc((void **)&src, (size_t)bytes) != hipSuccess) return DPPR_ERR_NOMEM;
    if (hipMalloc((void **)&dst, (size_t)bytes) != hipSuccess) {
        (void)hipFree(src);
        return DPPR_ERR_NOMEM;
    }
    (void)hipMemset(src, 0, (size_t)bytes);
    (void)hipMemset(dst, 0, (size_t)bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    auto launch = [&]() { hipLaunchKernelGGL(k_bench_copy, dim3(4096), dim3(1024), 0, 0, src, dst, bytes / 16); };
    launch();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    hipError_t err = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    *out_ms = ms / reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(src);
    (void)hipFree(dst);
    return err == hipSuccess ? DPPR_OK : DPPR_ERR_HIP;
}

int dppr_debug_bin_tables(dppr_engine *e, int32_t epoch, int32_t *n_a, int32_t *n_b, int32_t *n_edges, int32_t *n_runs, int32_t *n_tiles,
                          int32_t *acut, int32_t *bcut, uint16_t *hl, int32_t *tdelta, int32_t *tb, uint16_t *dl, int32_t *vb,
                          int64_t *patched, int64_t *rebuilt) {
    if (!e) return DPPR_ERR_INVALID;
    GET_EPOCH(e, epoch);
    if (patched) *patched = e->bin_patched;
    if (rebuilt) *rebuilt = e->bin_rebuilt;
    if (!ep.bin_valid) return fail(e, DPPR_ERR_INVALID, "debug_bin_tables: this epoch has no binned tables");
    HIP_TRY(hipSetDevice(e->device));
    if (n_a) *n_a = ep.n_a;
    if (n_b) *n_b = ep.n_b;
    if (n_edges) *n_edges = ep.Ed;
    if (n_runs) *n_runs = ep.n_runs;
    if (n_tiles) *n_tiles = ep.n_tiles;
    const size_t n_blk = ((size_t)ep.Ed + WAVE - 1) / WAVE, n_rb = ((size_t)ep.n_runs + WAVE - 1) / WAVE;
    if (acut) HIP_TRY(hipMemcpy(acut, ep.acut, sizeof(int) * ((size_t)ep.n_a + 1), hipMemcpyDeviceToHost));
    if (bcut) HIP_TRY(hipMemcpy(bcut, ep.bcut, sizeof(int) * ((size_t)ep.n_b + 1), hipMemcpyDeviceToHost));
    if (hl) HIP_TRY(hipMemcpy(hl, ep.hl, sizeof(uint16_t) * (size_t)ep.n_runs, hipMemcpyDeviceToHost));
    if (tdelta) HIP_TRY(hipMemcpy(tdelta, ep.tdelta, sizeof(int) * (size_t)ep.n_tiles, hipMemcpyDeviceToHost));
    if (tb) HIP_TRY(hipMemcpy(tb, ep.tb, sizeof(int) * (n_rb + 1), hipMemcpyDeviceToHost));
    if (dl) HIP_TRY(hipMemcpy(dl, ep.dl, sizeof(uint16_t) * (size_t)ep.Ed, hipMemcpyDeviceToHost));
    if (vb) HIP_TRY(hipMemcpy(vb, ep.vb, sizeof(int) * (n_blk + 1), hipMemcpyDeviceToHost));
    return DPPR_OK;
}

unsigned long long dppr_heartbeat(const dppr_engine *e) { return e ? e->heartbeat.load(std::memory_order_relaxed) : 0ull; }

#ifndef DPPR_BUILD_ID
#define DPPR_BUILD_ID "unstamped"
#endif
const char *dppr_build_id(void) { return DPPR_BUILD_ID; }

} // extern "C"
