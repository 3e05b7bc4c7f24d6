// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction per SIMD for scalar and
// packed f32 ops and the transcendental unit, at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define N_ITERS 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 c = {1.0001f, 0.9999f}, dd = {1e-7f, -1e-7f};
    float cs = 1.0001f, ds = 1e-7f;
    for (int i = 0; i < N_ITERS; ++i) {
        if (OP == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs), "v"(ds));
        } else if (OP == 1) {  // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(dd));
        } else if (OP == 2) {  // 8 independent v_rsq_f32
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 3) {  // 8 independent v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));
        } else if (OP == 4) {  // dependent chain of v_fma_f32 (latency)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(cs), "v"(ds));
        } else if (OP == 5) {  // dependent chain of v_rsq_f32
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n"
                         "v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n" : "+v"(a0));
        } else if (OP == 6) {  // 4 fma + 4 rsq interleaved, independent (does the trans unit co-issue?)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_rsq_f32 %4, %4\n v_fma_f32 %1, %1, %8, %9\n v_rsq_f32 %5, %5\n"
                         "v_fma_f32 %2, %2, %8, %9\n v_rsq_f32 %6, %6\n v_fma_f32 %3, %3, %8, %9\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs), "v"(ds));
        } else if (OP == 7) {  // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));
        } else if (OP == 8) {  // dependent chain of v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(p0) : "v"(c), "v"(dd));
        } else if (OP == 9) {  // 8 independent v_mov_b32
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                         "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int OP>
int run(const char *name, float *d, int clock_khz) {
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        int blocks = 256 * wps;     // 256 CUs x (4 waves per block = 1 wave per SIMD) x wps
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double insts_per_simd = (double)N_ITERS * 8 * wps;
        double cycles = ms * 1e-3 * clock_khz * 1e3;
        printf("  wps%d: %6.2f cyc/inst", wps, cycles / insts_per_simd);
    }
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs %d clock %d kHz (cycles below assume that clock)\n", p.name, p.multiProcessorCount, p.clockRate);
    float *d; CHK(hipMalloc(&d, 4096));
    run<0>("v_fma_f32 x8 indep", d, p.clockRate);
    run<7>("v_mul_f32 x8 indep", d, p.clockRate);
    run<9>("v_mov_b32 x8", d, p.clockRate);
    run<1>("v_pk_fma_f32 x8 indep", d, p.clockRate);
    run<3>("v_pk_mul_f32 x8 indep", d, p.clockRate);
    run<2>("v_rsq_f32 x8 indep", d, p.clockRate);
    run<6>("fma+rsq interleaved", d, p.clockRate);
    run<4>("v_fma_f32 dependent", d, p.clockRate);
    run<8>("v_pk_fma_f32 dependent", d, p.clockRate);
    run<5>("v_rsq_f32 dependent", d, p.clockRate);
    return 0;
}
