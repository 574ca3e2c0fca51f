// policy_probe.hip -- diagnostic (not part of the product): can a cache-policy bit on the COLD gathers of k_gsweep's edge phase keep the
// HOT rows (hubs: the 32 K hottest rows take 54 % of the LiveJournal stand-in's gathers, 4 MiB = one XCD's L2) resident in the L2s?
// Access shape of gather_probe.hip (an octet fetches one 80-byte row at a 128-byte stride); a fraction `hot_frac` of the gathers goes to
// rows [0, hot_rows), the rest uniformly to the other rows of a 151-MB table; hot rows are loaded with plain buffer loads, cold rows with
// the policy under test (aux bits of buffer_load: sc0 = 1, nt = 2, sc1 = 16).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/policy_probe tools/r04/policy_probe.hip && /tmp/policy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <int AUXC, int AUXH>
__global__ __launch_bounds__(1024, 8) void k_gather(const int *__restrict__ idx, long long n, const double *__restrict__ table, unsigned bytes,
                                                    int hot_rows, int live, double *__restrict__ sink) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63, j = lane & 7;
    const long long n_oct = (long long)gridDim.x * 128, oct = (long long)blockIdx.x * 128 + threadIdx.x / 8;
    const long long per = ((n + n_oct - 1) / n_oct + 7) / 8 * 8;
    const long long e0 = oct * per, e1 = e0 + per < n ? e0 + per : n;
    unsigned acc = 0;
    int mycol = e0 + j < e1 ? idx[e0 + j] : -1;
    for (long long e = e0; e < e1; e += 8) {
        const int ncol = e + 8 + j < e1 ? idx[e + 8 + j] : -1;
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            v4u v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ck = __shfl(mycol, (lane & ~7) + h + k, 64);
                v[k] = v4u{0, 0, 0, 0};
                if (ck >= 0 && j < live) {
                    const int off = ck * 128 + j * 16;
                    if (ck < hot_rows) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUXH);
                    else v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUXC);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
        mycol = ncol;
    }
    if (acc == 0x12345678u) sink[0] = 1.0; // (never true: keeps the loads alive)
}

static uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int AUXC, int AUXH>
static void run(const char *name, const int *d_idx, long long n, const double *table, unsigned bytes, int hot_rows, double *sink) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    auto launch = [&]() { hipLaunchKernelGGL((k_gather<AUXC, AUXH>), dim3(2048), dim3(1024), 0, 0, d_idx, n, table, bytes, hot_rows, 5, sink); };
    launch();
    CHECK(hipEventRecord(a, 0));
    for (int r = 0; r < 5; ++r) launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double t = ms * 1e-3 / 5;
    printf("    cold loads %-12s hot loads %-8s: %7.1f us  %6.2f G rows/s\n", name, AUXH == 0 ? "plain" : AUXH == 2 ? "nt" : "other", t * 1e6, n / t / 1e9);
}

int main() {
    const long long n = 6900000ll * 4, rows = 1180000;
    double *sink, *table;
    int *d_idx;
    CHECK(hipMalloc(&sink, 8));
    const unsigned bytes = (unsigned)(rows * 128);
    CHECK(hipMalloc(&table, bytes + 256));
    CHECK(hipMemset(table, 0, bytes + 256));
    CHECK(hipMalloc(&d_idx, sizeof(int) * (size_t)n));
    struct Mix { int hot_rows; double hot_frac; };
    for (const Mix &m : {Mix{32768, 0.54}, Mix{16384, 0.45}, Mix{8192, 0.35}, Mix{32768, 0.0}}) {
        std::vector<int> h((size_t)n);
        uint64_t s = 7;
        for (long long i = 0; i < n; ++i) {
            const double u = (double)(sm64(s) >> 11) / 9007199254740992.0, w = (double)(sm64(s) >> 11) / 9007199254740992.0;
            h[(size_t)i] = u < m.hot_frac ? (int)(w * m.hot_rows) : m.hot_rows + (int)(w * (rows - m.hot_rows));
        }
        CHECK(hipMemcpy(d_idx, h.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
        printf("hot rows %d (%.1f MB) take %.0f %% of %lld gathers, table %.0f MB:\n", m.hot_rows, m.hot_rows * 128 / 1e6, 100 * m.hot_frac, n, bytes / 1e6);
        run<0, 0>("plain", d_idx, n, table, bytes, m.hot_rows, sink);
        run<2, 0>("nt", d_idx, n, table, bytes, m.hot_rows, sink);
        run<1, 0>("sc0", d_idx, n, table, bytes, m.hot_rows, sink);
        run<16, 0>("sc1", d_idx, n, table, bytes, m.hot_rows, sink);
        run<17, 0>("sc0 sc1", d_idx, n, table, bytes, m.hot_rows, sink);
        run<18, 0>("sc1 nt", d_idx, n, table, bytes, m.hot_rows, sink);
        run<19, 0>("sc0 sc1 nt", d_idx, n, table, bytes, m.hot_rows, sink);
        run<3, 0>("sc0 nt", d_idx, n, table, bytes, m.hot_rows, sink);
    }
    return 0;
}
