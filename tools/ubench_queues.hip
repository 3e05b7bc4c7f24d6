// Which HIP streams overlap how: N streams created back to back; for every pair, K ragged kernels (long workgroups first, as
// the march launches them) alternate between the two streams; the time of the batch against the same batch on ONE stream.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_queues tools/ubench_queues.hip && tools/bin/ubench_queues [n_streams] [pads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void ragged(float *out, int n_wg, int base_iters) {
    // workgroup b spins base_iters * (1 + 3 (1 - b / n_wg)): the first ones four times as long as the last
    const int iters = base_iters + (int)(3.0f * base_iters * (1.0f - (float)blockIdx.x / n_wg));
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = fmaf(a, b, 1e-7f);
    if (a == 123.0f) out[0] = a;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, pads = argc > 2 ? atoi(argv[2]) : 0;
    float *d; CK(hipMalloc(&d, 64));
    for (int k = 0; k < pads; ++k) { hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    std::vector<hipStream_t> st(n);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n_wg = 256 * 8 * 4, K = 24, iters = 3000;      // 4 rounds of full occupancy at 256 threads
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](hipStream_t a, hipStream_t b) {
        for (int w = 0; w < 2; ++w) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, a));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(ragged, dim3(n_wg), dim3(256), 0, (k & 1) ? b : a, d, n_wg, iters);
            CK(hipDeviceSynchronize());
        }
        // wall clock over the batch (events on one stream do not bracket the other)
        CK(hipDeviceSynchronize());
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(ragged, dim3(n_wg), dim3(256), 0, (k & 1) ? b : a, d, n_wg, iters);
        CK(hipDeviceSynchronize());
        clock_gettime(CLOCK_MONOTONIC, &t1);
        return (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    };
    // clocks up
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(ragged, dim3(n_wg), dim3(256), 0, st[0], d, n_wg, iters);
    CK(hipDeviceSynchronize());
    const double one = run(st[0], st[0]);
    printf("one stream: %.3f ms for %d kernels (%.1f us each)\npairs, time relative to one stream:\n     ", one, K, one / K * 1e3);
    for (int j = 0; j < n; ++j) printf("  s%-3d", j);
    printf("\n");
    for (int i = 0; i < n; ++i) {
        printf("s%-3d ", i);
        for (int j = 0; j < n; ++j) {
            if (j <= i) { printf("   .  "); continue; }
            printf(" %5.3f", run(st[i], st[j]) / one);
        }
        printf("\n");
    }
    return 0;
}
