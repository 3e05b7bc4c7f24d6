// Exhaustive search for short exactly-rounded sqrt / reciprocal / divide sequences on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ float sqrtA(float x) { float y = __builtin_amdgcn_rsqf(x); float s = x * y; float h = 0.5f * y; float r = fmaf(-s, s, x); return fmaf(r, h, s); }
__device__ float sqrtB(float x) { float y = __builtin_amdgcn_rsqf(x); float s = x * y; float h = 0.5f * y; float r = fmaf(-s, s, x); s = fmaf(r, h, s); r = fmaf(-s, s, x); return fmaf(r, h, s); }
__device__ float sqrtC(float x) { float s = __builtin_amdgcn_sqrtf(x); float h = 0.5f * __builtin_amdgcn_rcpf(s); float r = fmaf(-s, s, x); return fmaf(r, h, s); }
__device__ float sqrtD(float x) { float s = __builtin_amdgcn_sqrtf(x); float y = __builtin_amdgcn_rsqf(x); float r = fmaf(-s, s, x); return fmaf(r, 0.5f * y, s); }
__device__ float rcpA(float x) { float y = __builtin_amdgcn_rcpf(x); float e = fmaf(-x, y, 1.0f); return fmaf(e, y, y); }
__device__ float rcpB(float x) { float y = __builtin_amdgcn_rcpf(x); float e = fmaf(-x, y, 1.0f); y = fmaf(e, y, y); e = fmaf(-x, y, 1.0f); return fmaf(e, y, y); }
__device__ float divA(float a, float b) { float y = rcpA(b); float q = a * y; float r = fmaf(-b, q, a); return fmaf(r, y, q); }
__device__ float divB(float a, float b) { float y = rcpB(b); float q = a * y; float r = fmaf(-b, q, a); return fmaf(r, y, q); }
__device__ unsigned int lcg(unsigned int &s) { s = s * 1664525u + 1013904223u; return s; }

__global__ void k(unsigned long long *out, unsigned int rounds) {
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    unsigned long long b[10] = {0};
    for (unsigned long long i = tid; i < 160ull << 23; i += nt) {   // exponents 47..206: [2^-80, 2^80)
        float x = __uint_as_float((unsigned int)(i + (47ull << 23)));
        float s = sqrtf(x), rc = 1.0f / x;
        b[0] += sqrtA(x) != s; b[1] += sqrtB(x) != s; b[2] += sqrtC(x) != s; b[3] += sqrtD(x) != s;
        b[4] += rcpA(x) != rc; b[5] += rcpB(x) != rc; b[4] += rcpA(-x) != -rc;
    }
    unsigned int st = tid * 2654435761u + 777u;
    for (unsigned int r = 0; r < rounds; ++r) {
        unsigned int ra = lcg(st), rb = lcg(st), re = lcg(st);
        unsigned int ea = 97u + (re & 0xffffu) % 60u, eb = 97u + (re >> 16) % 60u;
        float a = __uint_as_float((ra & 0x807fffffu) | (ea << 23)), bb = __uint_as_float((rb & 0x807fffffu) | (eb << 23));
        float q = a / bb;
        b[6] += divA(a, bb) != q; b[7] += divB(a, bb) != q; b[8] += 1;
        // adversarial: divisor significand near all-ones / quotient near 1
        float b2 = __uint_as_float((rb | 0x007ff000u) & 0x7fffffffu | 0u); b2 = __uint_as_float((__float_as_uint(b2) & 0x007fffffu) | (eb << 23));
        float q2 = a / b2; b[6] += divA(a, b2) != q2; b[7] += divB(a, b2) != q2; b[8] += 1;
    }
    for (int i = 0; i < 9; ++i) atomicAdd(out + i, b[i]);
}

int main() {
    unsigned long long *d, h[10] = {0};
    CHK(hipMalloc(&d, sizeof(h))); CHK(hipMemset(d, 0, sizeof(h)));
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, 8192u);
    CHK(hipDeviceSynchronize()); CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("sqrtA(rsq,1NR) %llu  sqrtB(rsq,2NR) %llu  sqrtC(sqrt,rcp) %llu  sqrtD(sqrt,rsq) %llu  of %llu\n", h[0], h[1], h[2], h[3], 160ull << 23);
    printf("rcpA(1NR) %llu  rcpB(2NR) %llu\n", h[4], h[5]);
    printf("divA %llu  divB %llu  of %llu pairs\n", h[6], h[7], h[8]);
    return 0;
}
