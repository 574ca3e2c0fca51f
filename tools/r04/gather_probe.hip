// gather_probe.hip -- diagnostic (not part of the product): how many state-row gathers per second does an MI355X
// deliver in the access shape of k_gsweep's edge phase? An octet (8 lanes) fetches one row of `row_bytes` (16 bytes
// per live lane) at a random row index; GB rows are in flight per octet; 1024-thread workgroups, two per CU.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_probe tools/r04/gather_probe.hip && /tmp/gather_probe
// Output: one line per (table rows, stride, live lanes): G rows/s, TB/s of useful bytes and of 128-byte lines touched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int GB>
__global__ __launch_bounds__(1024, 8) void k_gather(const int *__restrict__ idx, long long n, const double *__restrict__ table,
                                                    int stride_d, int live, double *__restrict__ sink) {
    const int lane = threadIdx.x & 63, j = lane & 7;
    const long long n_oct = (long long)gridDim.x * 128, oct = (long long)blockIdx.x * 128 + threadIdx.x / 8;
    const long long per = ((n + n_oct - 1) / n_oct + 7) / 8 * 8;
    const long long e0 = oct * per, e1 = e0 + per < n ? e0 + per : n;
    double acc0 = 0, acc1 = 0;
    int mycol = e0 + j < e1 ? idx[e0 + j] : -1;
    for (long long e = e0; e < e1; e += 8) {
        const int ncol = e + 8 + j < e1 ? idx[e + 8 + j] : -1;
#pragma unroll
        for (int h = 0; h < 8; h += GB) {
            double2 v[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int ck = __shfl(mycol, (lane & ~7) + h + k, 64);
                v[k] = make_double2(0, 0);
                if (ck >= 0 && j < live) v[k] = *reinterpret_cast<const double2 *>(table + (size_t)ck * stride_d + j * 2);
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                acc0 += v[k].x;
                acc1 += v[k].y;
            }
        }
        mycol = ncol;
    }
    if (acc0 + acc1 == 12345.678) sink[0] = acc0; // (never true: keeps the loads alive)
}

static uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 6900000ll * 4; // gathers per launch
    const int reps = 5;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    double *sink;
    CHECK(hipMalloc(&sink, 8));
    struct Case { long long rows; int stride_d, live; double zipf; };
    std::vector<Case> cases = {
        {16384, 16, 8, 0}, {16384, 16, 5, 0}, {16384, 10, 5, 0},                 // L2-resident table: the issue path
        {300000, 16, 8, 0}, {300000, 16, 5, 0}, {300000, 10, 5, 0},              // 38 MB: Infinity Cache
        {1180000, 16, 8, 0}, {1180000, 16, 5, 0}, {1180000, 10, 5, 0}, {1180000, 8, 4, 0}, {1180000, 8, 8, 0}, // the LiveJournal stand-in's table
        {1180000, 16, 5, 1.0}, {1180000, 10, 5, 1.0},                            // ... with a skewed (hub-heavy) index distribution
        {8000000, 16, 8, 0}, {8000000, 16, 5, 0}, {8000000, 10, 5, 0}, {8000000, 8, 4, 0}, // 1 GB: HBM
    };
    for (const Case &c : cases) {
        std::vector<int> h((size_t)n);
        uint64_t s = 42;
        for (long long i = 0; i < n; ++i) {
            const double u = (double)(sm64(s) >> 11) / 9007199254740992.0;
            // zipf-ish: rank = rows^(u) concentrates the mass on small ranks
            long long r = c.zipf > 0 ? (long long)(pow((double)c.rows, u)) - 1 : (long long)(u * c.rows);
            if (r < 0) r = 0;
            if (r >= c.rows) r = c.rows - 1;
            // (ranks are scattered over the table so that hot rows do not share lines)
            h[(size_t)i] = c.zipf > 0 ? (int)((r * 2654435761ull) % (unsigned long long)c.rows) : (int)r;
        }
        int *d_idx;
        double *table;
        CHECK(hipMalloc(&d_idx, sizeof(int) * (size_t)n));
        CHECK(hipMemcpy(d_idx, h.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&table, sizeof(double) * (size_t)c.rows * c.stride_d + 256));
        CHECK(hipMemset(table, 0, sizeof(double) * (size_t)c.rows * c.stride_d + 256));
        for (int gb : {4, 8}) {
            auto launch = [&]() {
                if (gb == 4) hipLaunchKernelGGL(k_gather<4>, dim3(2048), dim3(1024), 0, 0, d_idx, n, table, c.stride_d, c.live, sink);
                else hipLaunchKernelGGL(k_gather<8>, dim3(2048), dim3(1024), 0, 0, d_idx, n, table, c.stride_d, c.live, sink);
            };
            launch();
            CHECK(hipEventRecord(a, 0));
            for (int r = 0; r < reps; ++r) launch();
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double t = ms * 1e-3 / reps;
            printf("rows %8lld (%6.1f MB) stride %3d B live %3d B zipf %.1f GB %d : %6.1f us  %6.2f G rows/s  useful %5.2f TB/s\n", c.rows,
                   c.rows * c.stride_d * 8 / 1e6, c.stride_d * 8, c.live * 16, c.zipf, gb, t * 1e6, n / t / 1e9, n / t * c.live * 16 / 1e12);
        }
        CHECK(hipFree(d_idx));
        CHECK(hipFree(table));
    }
    return 0;
}
