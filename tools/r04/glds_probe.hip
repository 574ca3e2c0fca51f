// glds_probe.hip -- diagnostic: where does global_load_lds put a lane's data when some lanes are switched off, and does a
// wave-uniform base computed from an opaque thread index work? (tools/r04; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ __launch_bounds__(1024) void k(const int *g, const unsigned short *g2, int *out, int *out2, int mode) {
    __shared__ double pad[5120];   // push the targets to a 40 KB offset, like s_acc does
    __shared__ int s[1024];
    __shared__ unsigned short s2[1024];
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    pad[t] = 1.0;
    s[t] = -7;
    s2[t] = 7;
    __syncthreads();
    const int wb = t & ~63;
    const bool on = mode == 0 ? true : mode == 1 ? (t % 3 != 0) : (t < 16 && t * 32 < 300);
    if (on) __builtin_amdgcn_global_load_lds(g + t, (lds_void_t *)&s[mode == 2 ? 0 : wb], 4, 0, 0);
    if (t < 512) __builtin_amdgcn_global_load_lds(g2 + t, (lds_void_t *)&s2[wb], 2, 0, 0);
    __syncthreads();
    out[t] = s[t] + (pad[t] > 2.0 ? 1 : 0);
    out2[t] = s2[t];
}
int main() {
    int *g, *out, *out2; unsigned short *g2;
    hipMalloc(&g, 4096); hipMalloc(&out, 4096); hipMalloc(&out2, 4096); hipMalloc(&g2, 2048);
    std::vector<int> h(1024); std::vector<unsigned short> h2(1024);
    for (int i = 0; i < 1024; ++i) { h[i] = 1000 + i; h2[i] = (unsigned short)(2000 + i); }
    hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(g2, h2.data(), 2048, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 1024>>>(g, g2, out, out2, mode);
        std::vector<int> o(1024), o2(1024);
        hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), out2, 4096, hipMemcpyDeviceToHost);
        int bad = 0, bad2 = 0;
        for (int t = 0; t < 1024; ++t) {
            const bool on = mode == 0 ? true : mode == 1 ? (t % 3 != 0) : (t < 16 && t * 32 < 300);
            const int want = on ? 1000 + t : -7;
            if (o[t] != want) { if (bad < 6) printf("mode %d: s[%d] = %d, want %d\n", mode, t, o[t], want); ++bad; }
            const int want2 = t < 512 ? 2000 + t : 7;
            if (o2[t] != want2) { if (bad2 < 6) printf("mode %d: s2[%d] = %d, want %d\n", mode, t, o2[t], want2); ++bad2; }
        }
        printf("mode %d: %d wrong dwords, %d wrong ushorts\n", mode, bad, bad2);
    }
    return 0;
}
