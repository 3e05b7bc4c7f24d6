// Does hipStreamWaitValue32 release a stream while ANOTHER stream's kernel is still running, and how soon after the value is
// reached?  A slow kernel raises a counter step by step; a second stream waits for the counter to reach half its final value and
// then runs a kernel that stamps the wall clock.    hipcc --offload-arch=gfx950 -O2 -o /tmp/ubw tools/ubench_waitvalue.hip && /tmp/ubw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void slow_counter(unsigned int *counter, unsigned long long *stamps, int steps, int ticks_per_step) {
    // one workgroup: every step takes ticks_per_step of the 100 MHz clock, then the counter goes up by one
    if (threadIdx.x != 0) return;
    for (int s = 0; s < steps; ++s) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)ticks_per_step) __builtin_amdgcn_s_sleep(4);
        __threadfence();
        atomicAdd(counter, 1u);
        if (s == steps / 2 - 1) stamps[0] = wall_clock64();       // the moment the waited-for value is reached
    }
    stamps[1] = wall_clock64();
}
__global__ void stamp(unsigned long long *stamps) { if (threadIdx.x == 0) stamps[2] = wall_clock64(); }

int main() {
    unsigned int *counter = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&counter, 64, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMalloc((void **)&counter, 64));
    unsigned long long *stamps, h[3];
    CK(hipMalloc((void **)&stamps, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 64));
        CK(hipMemset(stamps, 0, 64));
        CK(hipDeviceSynchronize());
        const int steps = 200, ticks = 500;                       // 200 x 5 us = 1 ms
        hipLaunchKernelGGL(slow_counter, dim3(1), dim3(64), 0, a, counter, stamps, steps, ticks);
        e = hipStreamWaitValue32(b, counter, steps / 2, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, stamps);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
        printf("rep %d: value reached at t, waiter's kernel ran %+.1f us later; the counting kernel ended %+.1f us after t\n", rep,
               ((double)h[2] - (double)h[0]) / 100.0, ((double)h[1] - (double)h[0]) / 100.0);
    }
    return 0;
}
