// dppr_common.hpp -- shared device primitives of the hand-written gfx950 (CDNA4, wave64) kernels.
// Compiled with -ffp-contract=off: the double arithmetic is the same sequence of IEEE operations
// as the reference's expressions (cited per kernel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dppr {

constexpr double ALPHA = 0.15;                 // Meta.h:31
constexpr double ONE_MINUS_ALPHA = 1.0 - ALPHA; // "(1.0 - ALPHA)" of gpu/ExpandRev.cuh:72
constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int OUT_CAP = 1024; // per-wave staged next-frontier entries (4 KiB of LDS)

struct Adj { // one in-CSR entry: edge src -> (row vertex)
    int32_t v;      // tail of the edge (in-neighbour)
    int32_t degp1;  // outdeg(v) + 1 at this epoch
};

// gpu/PPRCommon.cuh:6-11 IsLegalRevPush (strict inequalities). phase 2 is not the reference's: the MERGED loop of
// dppr_set_phase_merge pushes residuals of both signs in one loop (|r| > eps).
constexpr int PHASE_BOTH = 2;
// Written without a branch on `phase`: the two thresholds are scalar selects the compiler computes once per kernel
// (+-infinity switches a side off), every test is two compares -- as a three-way conditional each test was a small
// tree of scalar branches inside the hottest loops.
__device__ __forceinline__ bool legal(double r, int phase, double eps) {
    const double hi = phase == 1 ? __builtin_huge_val() : eps;   // r > eps counts in phase 0 and in the merged loop
    const double lo = phase == 0 ? -__builtin_huge_val() : -eps; // r < -eps counts in phase 1 and in the merged loop
    return (r > hi) | (r < lo);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x / WAVE; }

// number of set bits of mask strictly below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ int mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// wave64 inclusive scans on the DPP path (ALU latency, no LDS crossbar): Hillis-Steele inside each
// row of 16 lanes (row_shr 1,2,4,8), then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.
// Lanes without a source keep `old` (the operation's identity).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_take(int identity, int x) {
    return __builtin_amdgcn_update_dpp(identity, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int x) {
    x += dpp_take<0x111, 0xf>(0, x);
    x += dpp_take<0x112, 0xf>(0, x);
    x += dpp_take<0x114, 0xf>(0, x);
    x += dpp_take<0x118, 0xf>(0, x);
    x += dpp_take<0x142, 0xa>(0, x);
    x += dpp_take<0x143, 0xc>(0, x);
    return x;
}
__device__ __forceinline__ int wave_inclusive_max(int x) { // for values >= -1
    x = max(x, dpp_take<0x111, 0xf>(-1, x));
    x = max(x, dpp_take<0x112, 0xf>(-1, x));
    x = max(x, dpp_take<0x114, 0xf>(-1, x));
    x = max(x, dpp_take<0x118, 0xf>(-1, x));
    x = max(x, dpp_take<0x142, 0xa>(-1, x));
    x = max(x, dpp_take<0x143, 0xc>(-1, x));
    return x;
}

// Streamed-once data (out_col). Measured with a non-temporal load (-DDPPR_NT_STREAM=1), on the theory that the
// stream would then not push gathered x lines out of L2: LiveJournal single source 69.5 -> 79.6 us per sweep,
// 10-source batch 20.2 -> 20.8 ms, twitter unchanged -- plain loads stay.
#ifndef DPPR_NT_STREAM
#define DPPR_NT_STREAM 0
#endif
__device__ __forceinline__ int ld_stream(const int *p) {
#if DPPR_NT_STREAM
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// device-scope returning f64 atomics (global_atomic_add_f64 / global_atomic_swap_x2)
__device__ __forceinline__ double atomic_add_ret(double *p, double v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double atomic_exch(double *p, double v) {
    unsigned long long o = __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(p),
                                                 __double_as_longlong(v), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double(o);
}

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// (1.0-ALPHA)*x/den, bit for bit, without the ~11-instruction f64 division sequence per edge and
// source: with rcp = RN(1/den) (one true division per ROW), q0 = a*rcp is within an ulp of a/den, the
// remainder r = a - q0*den is exact in an FMA, and q0 + r*rcp rounds to RN(a/den) (Markstein's
// correction step; den = outdeg+1 is a small integer, never an all-ones significand).
// tests/test_exact_division.py checks the identity on 2e7 random operands.
__device__ __forceinline__ double push_term(double x, double den, double rcp) {
    const double a = ONE_MINUS_ALPHA * x;
    const double q0 = a * rcp;
    const double rem = __builtin_fma(-q0, den, a);
    return __builtin_fma(rem, rcp, q0);
}

// wave-wide sum; every lane gets the total. Same DPP ladder as the integer scans (two 32-bit DPP moves
// + one v_add_f64 per step; the xor-butterfly over ds_bpermute it replaces cost ~10x the cycles), fixed
// order -> deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take_f64(double x) { // 0.0 where the lane has no source
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_take_f64<0x111, 0xf>(x);
    x += dpp_take_f64<0x112, 0xf>(x);
    x += dpp_take_f64<0x114, 0xf>(x);
    x += dpp_take_f64<0x118, 0xf>(x);
    x += dpp_take_f64<0x142, 0xa>(x);
    x += dpp_take_f64<0x143, 0xc>(x);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), WAVE - 1);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), WAVE - 1);
    return __hiloint2double(hi, lo);
}

// Device-side statistics. Same-address global atomics serialise at ~11 ns each, so one
// atomic per wave (or per workgroup) on a shared counter would cost more than the kernels'
// real work; every workgroup owns one slot instead (a slot stream runs one kernel at a
// time) and the host sums the slots when statistics are read.
constexpr int STAT_SLOTS = 4096; // >= the largest grid of the iteration kernels
struct IterStats {
    unsigned long long blk_E[STAT_SLOTS]; // traversed edges, per workgroup slot
};
// workgroup total of a wave-uniform per-wave value -> this workgroup's slot (call from all threads)
template <int NWAVES>
__device__ __forceinline__ void stat_add_edges(IterStats *stats, unsigned long long wave_edges,
                                               unsigned long long *s_edges) {
    if (lane_id() == 0) s_edges[wave_id()] = wave_edges;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int k = 0; k < NWAVES; ++k) t += s_edges[k];
        if (t) stats->blk_E[blockIdx.x] += t;
    }
}


} // namespace dppr
