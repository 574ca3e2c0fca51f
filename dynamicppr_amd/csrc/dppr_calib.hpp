// dppr_calib.hpp -- calibration kernels behind dppr_bench_line_fills / dppr_bench_stream_copy: the two ceilings a sweep of this
// engine is held against besides the HBM peak (SURVEY.md 8(d) "the harness should also report a calibrated ceiling";
// bench.py measures them in the run that prints the line: roofline.ceiling_*). Not part of any solve.
#pragma once

#include "dppr_common.hpp"

namespace dppr {

// Random 128-byte line fills in the access shape of the group sweep's gathers (dppr_multi.hpp): the eight lanes of an octet
// fetch the eight 16-byte pieces of ONE line at a hashed line index -- one 128-byte request per octet -- GB of them in flight
// per lane. n_per_octet line fetches per octet.
template <int GB>
__global__ __launch_bounds__(1024) void k_bench_lines(const double2 *__restrict__ table, uint64_t line_mask, int n_per_octet,
                                                      double *__restrict__ sink) {
    const int j = threadIdx.x & 7;
    const uint64_t oct = ((uint64_t)blockIdx.x * 1024 + threadIdx.x) >> 3;
    double a0 = 0.0, a1 = 0.0;
    for (int it = 0; it < n_per_octet; it += GB) {
        double2 v[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            uint64_t z = oct * (uint64_t)n_per_octet + (uint64_t)(it + k) + 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            v[k] = table[(z & line_mask) * 8 + j];
        }
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            a0 += v[k].x;
            a1 += v[k].y;
        }
    }
    if (a0 + a1 == 12345.678) sink[0] = a0; // (never true: keeps the loads alive)
}

// Streaming copy, 16 bytes per lane and step: what a kernel that only streams reaches on this device ("achievable" HBM rate)
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_bench_copy(const double2 *__restrict__ src2, double2 *__restrict__ dst2, int64_t n16) {
    const v2d *src = reinterpret_cast<const v2d *>(src2);
    v2d *dst = reinterpret_cast<v2d *>(dst2);
    const int64_t stride = (int64_t)gridDim.x * 1024;
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) { // four independent 16-byte loads in flight per lane
        const v2d a = src[i], b = src[i + stride];
        const v2d c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

} // namespace dppr
