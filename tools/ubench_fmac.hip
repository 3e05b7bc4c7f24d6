// What does the bloom's inner instruction cost?  The convolution kernels issue `v_fmac_f32 vacc, sW, vIN` -- the tap
// weight as an SGPR operand -- and the PMC counters say the VALU is busy ~100 % of the time at 4 cycles per instruction,
// twice the 2.07 cycles tools/ubench_clock.hip measures for an all-VGPR v_fma_f32.  This measures the forms side by side
// (shader cycles per wave-instruction per SIMD at 4 and 8 waves per SIMD, in-kernel clock).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_fmac tools/ubench_fmac.hip && /tmp/ubench_fmac
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Stamp { unsigned long long cyc, rt; };
#define REGS8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int OP>
__global__ __launch_bounds__(256) void k(Stamp *out, float seed, int iters, float s0, float s1, float s2, float s3) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float cs = 1.0001f + threadIdx.x * 1e-9f, ds = 1e-7f + threadIdx.x * 1e-12f, es = 0.9999f + threadIdx.x * 1e-9f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ float lds_pad[];
    if (seed == 12345.0f) lds_pad[threadIdx.x] = seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (OP == 0) {          // v_fma_f32 v, v, v, v   (VOP3, all VGPR)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : REGS8 : "v"(cs), "v"(ds));
        } else if (OP == 1) {   // v_fmac_f32 vacc, v, v   (VOP2, all VGPR)
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n" : REGS8 : "v"(cs), "v"(ds));
        } else if (OP == 2) {   // v_fmac_f32 vacc, s, v   (the bloom's form: tap weight in an SGPR)
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n" : REGS8 : "s"(s0), "v"(ds));
        } else if (OP == 3) {   // v_fma_f32 v, s, v, v    (VOP3 with one SGPR)
            asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                         "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n" : REGS8 : "s"(s0), "v"(ds));
        } else if (OP == 4) {   // the bloom's pattern: 4 accumulators, 4 different SGPR weights, 2 different VGPR inputs
            asm volatile("v_fmac_f32 %0, %4, %8\n v_fmac_f32 %1, %5, %8\n v_fmac_f32 %2, %6, %8\n v_fmac_f32 %3, %7, %8\n"
                         "v_fmac_f32 %0, %5, %9\n v_fmac_f32 %1, %6, %9\n v_fmac_f32 %2, %7, %9\n v_fmac_f32 %3, %4, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(cs), "v"(ds));
        } else if (OP == 5) {   // the same with the weights in VGPRs
            asm volatile("v_fmac_f32 %0, %4, %8\n v_fmac_f32 %1, %5, %8\n v_fmac_f32 %2, %6, %8\n v_fmac_f32 %3, %7, %8\n"
                         "v_fmac_f32 %0, %5, %9\n v_fmac_f32 %1, %6, %9\n v_fmac_f32 %2, %7, %9\n v_fmac_f32 %3, %4, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(cs), "v"(ds));
        } else if (OP == 6) {   // v_fma_f32 v, v, v, v with three DIFFERENT sources per instruction
            asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %9, %10, %1\n v_fma_f32 %2, %10, %8, %2\n v_fma_f32 %3, %8, %9, %3\n"
                         "v_fma_f32 %4, %9, %10, %4\n v_fma_f32 %5, %10, %8, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %9, %10, %7\n" : REGS8 : "v"(cs), "v"(ds), "v"(es));
        } else if (OP == 7) {   // v_mul_f32 v, s, v  (VOP2 with SGPR) for comparison
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n" : REGS8 : "s"(s0));
        } else if (OP == 9) {   // inline constant operand: v_fma_f32 v, 0.5, v, v  (add_half of the strict march)
            asm volatile("v_fma_f32 %0, 0.5, %8, %0\n v_fma_f32 %1, 0.5, %8, %1\n v_fma_f32 %2, 0.5, %8, %2\n v_fma_f32 %3, 0.5, %8, %3\n"
                         "v_fma_f32 %4, 0.5, %8, %4\n v_fma_f32 %5, 0.5, %8, %5\n v_fma_f32 %6, 0.5, %8, %6\n v_fma_f32 %7, 0.5, %8, %7\n" : REGS8 : "v"(ds));
        } else if (OP == 10) {  // inline constant, VOP2: v_mul_f32 v, 0.5, v
            asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 0.5, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 0.5, %3\n"
                         "v_mul_f32 %4, 0.5, %4\n v_mul_f32 %5, 0.5, %5\n v_mul_f32 %6, 0.5, %6\n v_mul_f32 %7, 0.5, %7\n" : REGS8);
        } else if (OP == 11) {  // 32-bit literal, VOP2: v_mul_f32 v, 0x3e2aaaab, v  (div6 of the strict march)
            asm volatile("v_mul_f32 %0, 0x3f7fff00, %0\n v_mul_f32 %1, 0x3f7fff00, %1\n v_mul_f32 %2, 0x3f7fff00, %2\n v_mul_f32 %3, 0x3f7fff00, %3\n"
                         "v_mul_f32 %4, 0x3f7fff00, %4\n v_mul_f32 %5, 0x3f7fff00, %5\n v_mul_f32 %6, 0x3f7fff00, %6\n v_mul_f32 %7, 0x3f7fff00, %7\n" : REGS8);
        } else if (OP == 12) {  // constant as the addend: v_fma_f32 v, -v, v, 1.0  (Newton residual)
            asm volatile("v_fma_f32 %0, -%0, %8, 1.0\n v_fma_f32 %1, -%1, %8, 1.0\n v_fma_f32 %2, -%2, %8, 1.0\n v_fma_f32 %3, -%3, %8, 1.0\n"
                         "v_fma_f32 %4, -%4, %8, 1.0\n v_fma_f32 %5, -%5, %8, 1.0\n v_fma_f32 %6, -%6, %8, 1.0\n v_fma_f32 %7, -%7, %8, 1.0\n" : REGS8 : "v"(ds));
        } else if (OP == 13) {  // v_fmac_f32 vacc, 2.0, v  (rk_sum)
            asm volatile("v_fmac_f32 %0, 2.0, %8\n v_fmac_f32 %1, 2.0, %8\n v_fmac_f32 %2, 2.0, %8\n v_fmac_f32 %3, 2.0, %8\n"
                         "v_fmac_f32 %4, 2.0, %8\n v_fmac_f32 %5, 2.0, %8\n v_fmac_f32 %6, 2.0, %8\n v_fmac_f32 %7, 2.0, %8\n" : REGS8 : "v"(ds));
        } else if (OP == 14) {  // v_cmp with an SGPR: v_cmp_lt_f32 vcc, s, v
            asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cmp_lt_f32 vcc, %8, %1\n v_cmp_lt_f32 vcc, %8, %2\n v_cmp_lt_f32 vcc, %8, %3\n"
                         "v_cmp_lt_f32 vcc, %8, %4\n v_cmp_lt_f32 vcc, %8, %5\n v_cmp_lt_f32 vcc, %8, %6\n v_cmp_lt_f32 vcc, %8, %7\n" : REGS8 : "s"(s0) : "vcc");
        } else if (OP == 15) {  // v_cmp all VGPR
            asm volatile("v_cmp_lt_f32 vcc, %8, %0\n v_cmp_lt_f32 vcc, %8, %1\n v_cmp_lt_f32 vcc, %8, %2\n v_cmp_lt_f32 vcc, %8, %3\n"
                         "v_cmp_lt_f32 vcc, %8, %4\n v_cmp_lt_f32 vcc, %8, %5\n v_cmp_lt_f32 vcc, %8, %6\n v_cmp_lt_f32 vcc, %8, %7\n" : REGS8 : "v"(ds) : "vcc");
        } else if (OP == 16) {  // v_cndmask with an SGPR-pair mask (VOP3)
            asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %8\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %8\n"
                         "v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %8\n" : REGS8 : "s"(0x5555555555555555ull));
        } else if (OP == 8) {   // v_pk_fma_f32 with the weight pair in SGPRs? (packed needs VGPR pairs) -- VGPR weights, 2 FMAs per lane
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, c = {cs, es}, dd = {ds, ds};
            asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                         "v_pk_fma_f32 %0, %5, %4, %0\n v_pk_fma_f32 %1, %5, %4, %1\n v_pk_fma_f32 %2, %5, %4, %2\n v_pk_fma_f32 %3, %5, %4, %3\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(dd));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        }
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) {
        Stamp s = {t1 - t0, r1 - r0};
        if (r == 12345.678f) s.cyc = 0;
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
    }
}

template <int OP>
void run(const char *name, Stamp *d, double warm_s) {
    printf("%-52s", name);
    const int iters = 1024, rounds = 12;
    for (int wps : {2, 4, 8}) {
        size_t lds = (160 * 1024) / wps;
        if (lds > 64 * 1024) lds = 64 * 1024;
        CHK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int blocks = 256 * wps * rounds, waves = blocks * 4;
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < warm_s) {
            for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, 1.0f, iters, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
            CHK(hipDeviceSynchronize());
        }
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, 1.0f, iters, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<Stamp> h(waves);
        CHK(hipMemcpy(h.data(), d, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> clk(waves);
        for (int i = 0; i < waves; ++i) clk[i] = (double)h[i].cyc / (double)h[i].rt * 100.0;
        std::sort(clk.begin(), clk.end());
        double insts_wave = (double)iters * 8 * 8, mhz = clk[waves / 2];
        printf("  w%d: %5.2f cyc/inst/SIMD %4.0f MHz |", wps, ms * 1e-3 * mhz * 1e6 / (insts_wave * waves / 1024.0), mhz);
    }
    printf("\n");
}

int main(int argc, char **argv) {
    double warm = argc > 1 ? atof(argv[1]) : 1.0;
    Stamp *d; CHK(hipMalloc(&d, 256 * 8 * 4 * 12 * sizeof(Stamp)));
    run<0>("v_fma_f32 v,v,v,v", d, warm);
    run<1>("v_fmac_f32 vacc,v,v", d, warm);
    run<2>("v_fmac_f32 vacc,s,v  (bloom's form)", d, warm);
    run<3>("v_fma_f32 v,s,v,v", d, warm);
    run<4>("v_fmac_f32: 4 acc x 4 SGPR weights x 2 inputs", d, warm);
    run<5>("v_fmac_f32: 4 acc x 4 VGPR weights x 2 inputs", d, warm);
    run<6>("v_fma_f32 v,v,v,v three different sources", d, warm);
    run<7>("v_mul_f32 v,s,v", d, warm);
    run<8>("v_pk_fma_f32 (counts 1 inst = 2 FMAs/lane)", d, warm);
    run<9>("v_fma_f32 v, 0.5, v, v   (inline constant)", d, warm);
    run<10>("v_mul_f32 v, 0.5, v      (inline constant, VOP2)", d, warm);
    run<11>("v_mul_f32 v, literal32, v", d, warm);
    run<12>("v_fma_f32 v, -v, v, 1.0  (constant addend)", d, warm);
    run<13>("v_fmac_f32 vacc, 2.0, v", d, warm);
    run<14>("v_cmp_lt_f32 vcc, s, v", d, warm);
    run<15>("v_cmp_lt_f32 vcc, v, v", d, warm);
    run<16>("v_cndmask_b32 v, v, v, s[mask]", d, warm);
    return 0;
}
