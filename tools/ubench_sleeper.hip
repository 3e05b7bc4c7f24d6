// Does ONE resident sleeping wave slow a saturating kernel on another stream?  (The early H pass of csrc/group.hip first waited
// for the march's tile count with such a wave.)  hipcc --offload-arch=gfx950 -O2 -o /tmp/ubs tools/ubench_sleeper.hip && /tmp/ubs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void heavy(float *out, int n_wg, int iters, unsigned int *count) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = fmaf(a, b, 1e-7f);
    if (a == 123.0f) out[0] = a;
    if (count && threadIdx.x == 0) atomicAdd(count, 1u);
}
template <int MODE>
__global__ void sleeper(const unsigned int *count, unsigned int n) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 10000000ull) {
        if (MODE == 0 && __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) break;
        if (MODE == 1 && *(volatile const unsigned int *)count >= n) break;
        __builtin_amdgcn_s_sleep(127);
    }
}

int main() {
    float *d; unsigned int *count;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&count, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int n_wg = 256 * 8 * 6, iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(heavy, dim3(n_wg), dim3(256), 0, a, d, n_wg, iters, (unsigned int *)nullptr);
    CK(hipDeviceSynchronize());
    for (int mode = -1; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(count, 0, 64));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, a));
            hipLaunchKernelGGL(heavy, dim3(n_wg), dim3(256), 0, a, d, n_wg, iters, mode >= 0 ? count : (unsigned int *)nullptr);
            CK(hipEventRecord(e1, a));
            if (mode == 0) hipLaunchKernelGGL(sleeper<0>, dim3(1), dim3(1), 0, b, count, (unsigned int)n_wg);
            if (mode == 1) hipLaunchKernelGGL(sleeper<1>, dim3(1), dim3(1), 0, b, count, (unsigned int)n_wg);
            if (mode == 2) CK(hipStreamWaitValue32(b, count, n_wg, hipStreamWaitValueGte, 0xffffffffu));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: heavy kernel %.3f ms\n", mode < 0 ? "alone" : mode == 0 ? "beside a sleeping wave (atomic load poll)" : mode == 1 ? "beside a sleeping wave (volatile poll)" : "beside hipStreamWaitValue32", ms);
        }
    }
    return 0;
}
