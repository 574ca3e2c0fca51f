// Gap between two DEPENDENT kernel dispatches: plain stream launches (all enqueued ahead, as the engine's chunks do)
// against the same launches captured into a hipGraph.   hipcc --offload-arch=gfx950 -O2 tools/r02/graph_gap_probe.hip -o /tmp/gap && /tmp/gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_step(int *cnt, double *a, int n) { // reads what the previous launch wrote; ~2 us of work for a full grid
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * 0.999 + 1.0;
    if (i == 0) cnt[0] += 1;
}

int main() {
    const int N = 200, n = 1 << 20;
    int *cnt; double *a;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&a, sizeof(double) * n));
    CK(hipMemset(cnt, 0, 4)); CK(hipMemset(a, 0, sizeof(double) * n));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run_stream = [&](int grid) {
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, s, cnt, a, n);
    };
    for (int grid : {1, 4096}) {
        run_stream(grid); CK(hipStreamSynchronize(s)); // warm
        float best_s = 1e9f, best_g = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); run_stream(grid); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_s = ms < best_s ? ms : best_s;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); run_stream(grid); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_g = ms < best_g ? ms : best_g;
        }
        printf("grid %5d: %d dependent launches  stream %.1f us per launch   graph %.1f us per launch\n", grid, N, best_s * 1e3 / N, best_g * 1e3 / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
