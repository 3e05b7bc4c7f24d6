// VALU issue cost on gfx950 in SHADER CYCLES, measured inside the kernel (no assumption about the clock):
// every wave stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop, after the
// chip has run the same kernel back to back for 2 s (DVFS settled).  Reported per instruction stream and per
// occupancy (waves per SIMD):
//   cyc/inst/SIMD = median over waves of  d(s_memtime) / (instructions per wave * waves per SIMD)
//   clock         = median of d(s_memtime) / d(s_memrealtime) * 100 MHz
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_clock tools/ubench_clock.hip && /tmp/ubench_clock
// Reconciles tools/ubench.hip's "one v_fma_f32 per 1.25 ns" (which divided wall time by an ASSUMED 2.4 GHz) with
// MI355X_MICROARCH.md's "2 cycles per wave64 VALU instruction on a SIMD-32" (DESIGN.md section 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>

#define N_ITERS 8192
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Stamp { unsigned long long cyc, rt; };

#define FMA8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define REGS8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int OP>
__global__ __launch_bounds__(256) void k(Stamp *out, float seed, int iters) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float cs = 1.0001f, ds = 1e-7f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ float lds_pad[];           // occupancy limiter: blocks per CU = 160 KiB / dynamic LDS size
    if (seed == 12345.0f) lds_pad[threadIdx.x] = seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (OP == 0) {          // 8 independent v_fma_f32
            asm volatile(FMA8 : REGS8 : "v"(cs), "v"(ds));
        } else if (OP == 1) {   // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n" : REGS8 : "v"(cs));
        } else if (OP == 2) {   // 8 v_mov_b32
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                         "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" : REGS8);
        } else if (OP == 3) {   // 8 independent v_rsq_f32
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n" : REGS8);
        } else if (OP == 4) {   // dependent chain of 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(cs), "v"(ds));
        } else if (OP == 5) {   // two interleaved dependent chains (ILP 2)
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         : "+v"(a0), "+v"(a1) : "v"(cs), "v"(ds));
        } else if (OP == 6) {   // the strict march's mix: 7 plain per transcendental, as two dependent chains
            asm volatile("v_rsq_f32 %0, %0\n v_mul_f32 %0, %0, %2\n v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %0, %0, %2\n"
                         "v_rcp_f32 %1, %1\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_mul_f32 %1, %1, %2\n"
                         "v_add_f32 %0, %0, %3\n v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n v_mul_f32 %1, %1, %2\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n"
                         : "+v"(a0), "+v"(a1) : "v"(cs), "v"(ds));
        } else if (OP == 7) {   // dependent chain: rsq -> mul -> fma -> fma  (the Newton sequence of sqrt_rn), x2
            asm volatile("v_rsq_f32 %1, %0\n v_mul_f32 %2, %0, %1\n v_fma_f32 %3, -%2, %2, %0\n v_fma_f32 %0, %3, %1, %2\n"
                         "v_rsq_f32 %1, %0\n v_mul_f32 %2, %0, %1\n v_fma_f32 %3, -%2, %2, %0\n v_fma_f32 %0, %3, %1, %2\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (OP == 8) {   // 8 independent v_cndmask (VCC)
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         : REGS8 : : "vcc");
        } else if (OP == 9) {   // 4 v_pk_fma_f32 (= 8 FMAs per lane)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, c = {cs, cs}, dd = {ds, ds};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(dd));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        }
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) {
        Stamp s = {t1 - t0, r1 - r0};
        if (r == 12345.678f) s.cyc = 0;   // keeps the arithmetic alive
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
    }
}

template <int OP>
void run(const char *name, Stamp *d, int insts_per_iter, double warm_s) {
    printf("%-34s", name);
    const int iters = N_ITERS / 8, rounds = 12;
    for (int wps : {2, 4, 8}) {                    // blocks of 256 threads per CU = waves per SIMD, enforced through LDS
        size_t lds = (160 * 1024) / wps;
        if (lds > 64 * 1024) lds = 64 * 1024;
        CHK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int blocks = 256 * wps * rounds, waves = blocks * 4;
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < warm_s) {
            for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, 1.0f, iters);
            CHK(hipDeviceSynchronize());
        }
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, 1.0f, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> h(waves);
        CHK(hipMemcpy(h.data(), d, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> cyc(waves), clk(waves);
        for (int i = 0; i < waves; ++i) { cyc[i] = (double)h[i].cyc; clk[i] = (double)h[i].cyc / (double)h[i].rt * 100.0; }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        double insts_wave = (double)iters * 8 * insts_per_iter;
        double mhz = clk[waves / 2];
        // chip throughput: all wave-instructions / 1024 SIMDs over the launch, in cycles of the in-kernel clock
        double cyc_per_inst_simd = ms * 1e-3 * mhz * 1e6 / (insts_wave * waves / 1024.0);
        // what one wave sees: its own elapsed cycles per instruction
        printf("  w%d: %5.2f cyc/inst/SIMD (wave: %5.2f) %4.0f MHz %6.3f ms |", wps, cyc_per_inst_simd, cyc[waves / 2] / insts_wave, mhz, ms);
    }
    printf("\n");
}

int main(int argc, char **argv) {
    double warm = argc > 1 ? atof(argv[1]) : 2.0;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, nominal %d kHz; per column: shader cycles per wave-instruction per SIMD, in-kernel clock, launch wall time\n",
           p.name, p.multiProcessorCount, p.clockRate);
    Stamp *d; CHK(hipMalloc(&d, 256 * 8 * 4 * 12 * sizeof(Stamp)));
    run<0>("v_fma_f32 x8 independent", d, 8, warm);
    run<1>("v_mul_f32 x8 independent", d, 8, warm);
    run<2>("v_mov_b32 x8", d, 8, warm);
    run<8>("v_cndmask_b32 x8", d, 8, warm);
    run<3>("v_rsq_f32 x8 independent", d, 8, warm);
    run<9>("v_pk_fma_f32 x8 (4 regs pairs)", d, 8, warm);
    run<4>("v_fma_f32 dependent chain", d, 8, warm);
    run<5>("v_fma_f32 two chains interleaved", d, 8, warm);
    run<7>("rsq->mul->fma->fma chain (sqrt_rn)", d, 8, warm);
    run<6>("march-like mix, two chains", d, 16, warm);
    return 0;
}
