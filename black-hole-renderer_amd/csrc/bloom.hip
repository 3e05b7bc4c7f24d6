// bloom.hip -- separable RGB-dispersion bloom + final combine for gfx950.
//
// Restates _bloom_kernel (render.py:3022-3114) as the reference's render() drives it
// (render.py:3914-3918): threshold 0, radius R = int(0.02 W), weights exp(-d^2 / (sigma_c * s)) with
// sigma = {25, 80, 1600} per channel and s = (W/640)^2, out-of-image taps skipped and every channel
// divided by its own in-bounds weight sum.  With threshold 0 and a non-negative disk layer the
// "bright" copy equals the disk layer (lum > 0 fails only for all-zero pixels), so pass 1 of the
// reference is folded away.
//
// Both passes are FMA bound ((2R + 1) x 3 FMAs per pixel per pass: 77 taps at fhd, 307 at 8k), so
// the kernels are built to issue almost nothing else:
//   * every thread produces 4 adjacent outputs from a window staged in LDS and read with 16-byte
//     ds_read_b128 (1 LDS read per 16 FMAs);
//   * tap weights are uniform across the wave: they are fetched with scalar loads into SGPRs and
//     used as the scalar operand of v_fma, never through VGPRs or LDS;
//   * H pass: (rows, W, 3) disk layer -> planar (3, rows + 2R, W) intermediate, 1024 pixels of one row
//     per block; V pass: 32 columns x 128 rows per block, the tile transposed in LDS so that the
//     vertical window is contiguous, epilogue fuses clip(bg + disk + blur) of render.py:3918.
// The intermediate carries R halo rows on either side so that row-block tiles on different GPUs can
// exchange them (bhr_group_render).  Summation order differs from the reference's tap order
// (-R .. R) only by f32 rounding (tests: 2e-6 against the CPU restatement in the tests).
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "bhr_internal.h"

namespace {

constexpr int WPAD = 8;       // zero entries behind w[R] in the weight table (window overhang <= 6)
constexpr int HB_PIX = 1024;  // pixels per H-pass block (256 threads x 4)
constexpr int VB_COLS = 32;   // columns per V-pass block
// output rows per V-pass block = 8 row-lanes x G groups x 4 rows.  Measured (v_groups): G = 1 (32 rows) up to
// fhd, where the pass needs blocks more than it needs reuse (0.095 -> 0.073 ms); G = 4 at 4k; G = 8 (256 rows) at
// 8k, where the 2R halo rows would otherwise outweigh the tile (2.0 -> 1.74 ms)

__global__ void bloom_weights_kernel(float *wtab, int R, float sigma_scale) {
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    int n = R + 1 + WPAD;
    if (d >= n) return;
    float dist_sq = (float)(d * d);
    bool in = d <= R;
    wtab[0 * n + d] = in ? expf(-dist_sq / (25.0f * sigma_scale)) : 0.0f;
    wtab[1 * n + d] = in ? expf(-dist_sq / (80.0f * sigma_scale)) : 0.0f;
    wtab[2 * n + d] = in ? expf(-dist_sq / (1600.0f * sigma_scale)) : 0.0f;
}

// wext[c][i] = w_c[|i - (R4 + 3)|], i < 2 R4 + 8 (zero beyond R): the unfolded table conv4 reads
__global__ void bloom_wext_kernel(const float *wtab, float *wext, int R) {
    const int R4 = (R + 3) & ~3, n = 2 * R4 + 8, stride = R + 1 + WPAD;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int d = i - (R4 + 3);
    d = d < 0 ? -d : d;
    for (int c = 0; c < 3; ++c) wext[c * n + i] = d <= R ? wtab[c * stride + d] : 0.0f;
}

// wsum[c][x] = sum over taps d = -R..R with 0 <= x + d < n of w_c[|d|], in tap order
__global__ void bloom_wsum_kernel(const float *wtab, float *wsum, int R, int n) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    int stride = R + 1 + WPAD;
    float s0 = 0, s1 = 0, s2 = 0;
    for (int d = -R; d <= R; ++d) {
        int q = x + d;
        if (0 <= q && q < n) {
            int ad = d < 0 ? -d : d;
            s0 += wtab[0 * stride + ad];
            s1 += wtab[1 * stride + ad];
            s2 += wtab[2 * stride + ad];
        }
    }
    wsum[0 * n + x] = s0;
    wsum[1 * n + x] = s1;
    wsum[2 * n + x] = s2;
}

// 4 adjacent outputs k = 0..3 at window index R4 + k from the 16-byte aligned window `win`
// (win[4m + e], m < M): acc[k] += sum_{m,e} win[4m + e] * w[|4m + e - k - R4|].  `wx` is the channel's
// symmetric weight table unfolded to wx[i] = w[|i - (R4 + 3)|] (zero beyond R), so the seven weights
// of block m are the consecutive entries wx[4m .. 4m + 6]: uniform across the wave, fetched with
// scalar loads from one base address, no per-weight address arithmetic.
__device__ __forceinline__ void conv4(const float *__restrict__ win, int M, const float *__restrict__ wx, float acc[4]) {
    for (int m = 0; m < M; ++m) {
        const float4 v = *reinterpret_cast<const float4 *>(win + 4 * m);
        const float *__restrict__ wp = wx + 4 * m;
        float w[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) w[t] = wp[t];
        acc[0] = fmaf(v.x, w[3], fmaf(v.y, w[4], fmaf(v.z, w[5], fmaf(v.w, w[6], acc[0]))));
        acc[1] = fmaf(v.x, w[2], fmaf(v.y, w[3], fmaf(v.z, w[4], fmaf(v.w, w[5], acc[1]))));
        acc[2] = fmaf(v.x, w[1], fmaf(v.y, w[2], fmaf(v.z, w[3], fmaf(v.w, w[4], acc[2]))));
        acc[3] = fmaf(v.x, w[0], fmaf(v.y, w[1], fmaf(v.z, w[2], fmaf(v.w, w[3], acc[3]))));
    }
}

// grid (ceil(W / 1024), rows).  hblur row index = local row + R.
__global__ __launch_bounds__(256) void bloom_h_kernel(const float *__restrict__ disk, float *__restrict__ hblur,
                                                      const float *__restrict__ wext,
                                                      const float *__restrict__ wsum_h, int W, int rows, int R, int row_begin) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int R4 = (R + 3) & ~3;
    const int span = HB_PIX + 2 * R4 + 4;          // per-channel window, multiple of 4
    const int x0 = blockIdx.x * HB_PIX;
    const int row = blockIdx.y + row_begin;        // the launch covers local rows [row_begin, row_begin + gridDim.y)
    const int tid = threadIdx.x;
    const int xstride = 2 * R4 + 8;

    // stage pixels [x0 - R4, x0 + 1024 + R4 + 4) of this row, de-interleaved to planar; zero outside
    const float *src = disk + (size_t)row * W * 3;
    for (int k = tid; k < span * 3; k += 256) {
        int p = k / 3, c = k - p * 3;
        int x = x0 - R4 + p;
        lds[c * span + p] = (x >= 0 && x < W) ? src[(size_t)x * 3 + c] : 0.0f;
    }
    __syncthreads();

    const int x = x0 + 4 * tid;
    if (x >= W) return;
    const int M = (2 * R4) / 4 + 1;                // covers window indices 4 tid .. 4 tid + 2 R4 + 3
    const size_t plane = (size_t)(rows + 2 * R) * W;
    const size_t o = (size_t)(row + R) * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        conv4(lds + c * span + 4 * tid, M, wext + c * xstride, acc);
        float *dst = hblur + c * plane + o;
        const float *ws = wsum_h + c * W + x;
        if (x + 3 < W && (W & 3) == 0) {
            *reinterpret_cast<float4 *>(dst) = make_float4(acc[0] / ws[0], acc[1] / ws[1], acc[2] / ws[2], acc[3] / ws[3]);
        } else {
            for (int k = 0; k < 4 && x + k < W; ++k) dst[k] = acc[k] / ws[k];
        }
    }
}

// grid (ceil(W / 32), ceil(rows / (32 G))).  Thread = (column, row-lane); G groups of 4 rows each.
template <int G>
__global__ __launch_bounds__(256) void bloom_v_kernel(const float *__restrict__ hblur, const float *__restrict__ bg,
                                                      const float *__restrict__ disk, float *__restrict__ blur_out,
                                                      float *__restrict__ final_out, const float *__restrict__ wext,
                                                      const float *__restrict__ wsum_v, int W, int H, int row0,
                                                      int rows, int R, int S, int with_bloom,
                                                      unsigned long long *__restrict__ zero_cell, int row_begin, int row_end,
                                                      uint8_t *__restrict__ u8_out) {
    // housekeeping folded into the frame's last kernel: clear the ray-step counter cell of the NEXT timed frame,
    // which saves a fill dispatch (and its barrier) in front of every march
    if (zero_cell && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < BHR_STEP_LANES)
        zero_cell[(size_t)threadIdx.x * BHR_STEP_STRIDE] = 0ull;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [VB_COLS][S], column-major tile
    const int R4 = (R + 3) & ~3;
    const int col = threadIdx.x & (VB_COLS - 1);
    const int lane_g = threadIdx.x >> 5;           // 0..7
    const int x = blockIdx.x * VB_COLS + col;
    constexpr int VB_ROWS = 32 * G;
    const int y0 = row_begin + blockIdx.y * VB_ROWS;   // local row of the first output of the tile; the launch covers [row_begin, row_end)
    const int xstride = 2 * R4 + 8;
    const int tile_rows = VB_ROWS + 2 * R4 + 4;
    const int M = (2 * R4) / 4 + 1;

    float res[G][4][3];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) res[g][k][0] = res[g][k][1] = res[g][k][2] = 0.0f;

    if (with_bloom) {
        const size_t plane = (size_t)(rows + 2 * R) * W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            __syncthreads();
            // stage rows [y0 - R4, y0 + VB_ROWS + R4 + 4) x 32 columns of plane c, transposed
            for (int k = threadIdx.x; k < tile_rows * VB_COLS; k += 256) {
                int r = k >> 5, cc = k & (VB_COLS - 1);
                int yl = y0 - R4 + r;              // local row
                int yg = yl + row0;                // image row
                int xx = blockIdx.x * VB_COLS + cc;
                float v = 0.0f;
                // rows outside the image are skipped taps; rows outside this context's halo cannot be
                // reached by a tap (|d| <= R) of one of its outputs
                if (yg >= 0 && yg < H && yl >= -R && yl < rows + R && xx < W) v = hblur[c * plane + (size_t)(yl + R) * W + xx];
                lds[cc * S + r] = v;
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int grp = lane_g + 8 * g;    // group of 4 rows inside the tile
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                conv4(lds + col * S + 4 * grp, M, wext + c * xstride, acc);
#pragma unroll
                for (int k = 0; k < 4; ++k) res[g][k][c] = acc[k];
            }
        }
    }
    if (x >= W) return;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int grp = lane_g + 8 * g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = y0 + 4 * grp + k;
            if (yl >= row_end) continue;
            const int yg = yl + row0;
            float b0 = 0, b1 = 0, b2 = 0;
            if (with_bloom) {
                b0 = res[g][k][0] / wsum_v[yg];
                b1 = res[g][k][1] / wsum_v[H + yg];
                b2 = res[g][k][2] / wsum_v[2 * H + yg];
            }
            const size_t o = ((size_t)yl * W + x) * 3;
            blur_out[o + 0] = b0;
            blur_out[o + 1] = b1;
            blur_out[o + 2] = b2;
            // render.py:3912 / 3918: clip(img + disk [+ blur], 0, 1)
            const float f0 = fminf(fmaxf(bg[o + 0] + disk[o + 0] + b0, 0.0f), 1.0f);
            const float f1 = fminf(fmaxf(bg[o + 1] + disk[o + 1] + b1, 0.0f), 1.0f);
            const float f2 = fminf(fmaxf(bg[o + 2] + disk[o + 2] + b2, 0.0f), 1.0f);
            final_out[o + 0] = f0;
            final_out[o + 1] = f1;
            final_out[o + 2] = f2;
            if (u8_out) {   // save_image's quantisation fused in (render.py:423): the row-block gather ships these bytes
                u8_out[o + 0] = (uint8_t)(int)(f0 * 255.0f);
                u8_out[o + 1] = (uint8_t)(int)(f1 * 255.0f);
                u8_out[o + 2] = (uint8_t)(int)(f2 * 255.0f);
            }
        }
    }
}

// ---- round 3 variants: every tap block's seven weights are fetched ONCE for all the thread's output groups ------------
// conv4 above serves one group of 4 outputs per call, so a thread with G groups walks the weight table G times and
// every 16 FMAs wait for their own scalar load.  conv4g keeps NG accumulator groups live and feeds all of them from one
// fetch: 16 NG FMAs per scalar load, NG independent dependency chains per lane.
// WLDS: the weight table lives in LDS (`wx` points into it, 16-byte aligned) and each tap block's weights arrive as two
// broadcast ds_read_b128 in VGPRs; otherwise they are scalar loads used as SGPR operands.  v_fmac_f32 with an SGPR
// operand issues at HALF the rate of the all-VGPR form on gfx950 (tools/ubench_fmac.hip) -- the round-2 kernels were
// VALU-busy ~100 % of the time at 4 cycles per FMA.
template <int NG, bool WLDS = false>
__device__ __forceinline__ void conv4g(const float *__restrict__ win, int gstride, int M, const float *__restrict__ wx,
                                       float (&acc)[NG][4]) {
#pragma unroll 2
    for (int m = 0; m < M; ++m) {
        const float *__restrict__ wp = wx + 4 * m;
        float w[8];
        if (WLDS) {
            const float4 wa = *reinterpret_cast<const float4 *>(wp), wb = *reinterpret_cast<const float4 *>(wp + 4);
            w[0] = wa.x; w[1] = wa.y; w[2] = wa.z; w[3] = wa.w; w[4] = wb.x; w[5] = wb.y; w[6] = wb.z; w[7] = wb.w;
        } else {
#pragma unroll
            for (int t = 0; t < 7; ++t) w[t] = wp[t];
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(win + g * gstride + 4 * m);
            acc[g][0] = fmaf(v.x, w[3], fmaf(v.y, w[4], fmaf(v.z, w[5], fmaf(v.w, w[6], acc[g][0]))));
            acc[g][1] = fmaf(v.x, w[2], fmaf(v.y, w[3], fmaf(v.z, w[4], fmaf(v.w, w[5], acc[g][1]))));
            acc[g][2] = fmaf(v.x, w[1], fmaf(v.y, w[2], fmaf(v.z, w[3], fmaf(v.w, w[4], acc[g][2]))));
            acc[g][3] = fmaf(v.x, w[0], fmaf(v.y, w[1], fmaf(v.z, w[2], fmaf(v.w, w[3], acc[g][3]))));
        }
    }
}

// H pass, NG groups per thread 1024 pixels apart: a block covers 1024 NG pixels of one row.
template <int NG, bool WLDS = false>
__global__ __launch_bounds__(256) void bloom_h2_kernel(const float *__restrict__ disk, float *__restrict__ hblur,
                                                       const float *__restrict__ wext, const float *__restrict__ wsum_h,
                                                       int W, int rows, int R, int row_begin) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PIX = HB_PIX * NG;
    const int R4 = (R + 3) & ~3;
    const int span = PIX + 2 * R4 + 4;
    float *wl = lds + 3 * span;                    // WLDS: the three unfolded weight tables, 2 R4 + 8 floats each
    if (WLDS)
        for (int k = threadIdx.x; k < 3 * (2 * R4 + 8); k += 256) wl[k] = wext[k];
    const int x0 = blockIdx.x * PIX;
    const int row = blockIdx.y + row_begin;
    const int tid = threadIdx.x;
    const int xstride = 2 * R4 + 8;
    const float *src = disk + (size_t)row * W * 3;
    // stage pixels [x0 - R4, x0 + PIX + R4 + 4): each thread moves whole pixels (3 consecutive floats), no division
    for (int p = tid; p < span; p += 256) {
        const int x = x0 - R4 + p;
        float a = 0.0f, b = 0.0f, c = 0.0f;
        if (x >= 0 && x < W) { a = src[(size_t)x * 3]; b = src[(size_t)x * 3 + 1]; c = src[(size_t)x * 3 + 2]; }
        lds[p] = a;
        lds[span + p] = b;
        lds[2 * span + p] = c;
    }
    __syncthreads();
    const int M = (2 * R4) / 4 + 1;
    const size_t plane = (size_t)(rows + 2 * R) * W;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g][0] = acc[g][1] = acc[g][2] = acc[g][3] = 0.0f;
        conv4g<NG, WLDS>(lds + c * span + 4 * tid, HB_PIX, M, (WLDS ? wl : wext) + c * xstride, acc);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int x = x0 + g * HB_PIX + 4 * tid;
            if (x >= W) continue;
            float *dst = hblur + c * plane + (size_t)(row + R) * W + x;
            const float *ws = wsum_h + c * W + x;
            if (x + 3 < W && (W & 3) == 0) {
                *reinterpret_cast<float4 *>(dst) = make_float4(acc[g][0] / ws[0], acc[g][1] / ws[1], acc[g][2] / ws[2], acc[g][3] / ws[3]);
            } else {
                for (int k = 0; k < 4 && x + k < W; ++k) dst[k] = acc[g][k] / ws[k];
            }
        }
    }
}

// V pass, COLS columns x (256 / COLS) row lanes x G groups x 4 rows per block.  COLS = 16 halves the LDS tile of the
// 32-column kernel (the 2 R halo rows dominate it: 73 KB at 8k, two blocks per CU), so four blocks fit a CU.
template <int COLS, int G, bool WLDS = false>
__global__ __launch_bounds__(256) void bloom_v2_kernel(const float *__restrict__ hblur, const float *__restrict__ bg,
                                                       const float *__restrict__ disk, float *__restrict__ blur_out,
                                                       float *__restrict__ final_out, const float *__restrict__ wext,
                                                       const float *__restrict__ wsum_v, int W, int H, int row0, int rows,
                                                       int R, int S, int with_bloom, unsigned long long *__restrict__ zero_cell,
                                                       int row_begin, int row_end, uint8_t *__restrict__ u8_out) {
    if (zero_cell && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < BHR_STEP_LANES)
        zero_cell[(size_t)threadIdx.x * BHR_STEP_STRIDE] = 0ull;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [COLS][S], column-major tile
    constexpr int LG = 256 / COLS;                 // row lanes
    constexpr int VROWS = LG * 4 * G;              // output rows per block
    const int R4 = (R + 3) & ~3;
    const int col = threadIdx.x % COLS;
    const int lane_g = threadIdx.x / COLS;
    const int x = blockIdx.x * COLS + col;
    const int y0 = row_begin + blockIdx.y * VROWS;
    const int xstride = 2 * R4 + 8;
    const int tile_rows = VROWS + 2 * R4 + 4;
    const int M = (2 * R4) / 4 + 1;
    float *wl = lds + COLS * S;                    // WLDS: the weight tables behind the tile
    if (WLDS && with_bloom)
        for (int k = threadIdx.x; k < 3 * xstride; k += 256) wl[k] = wext[k];

    float res[3][G][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int g = 0; g < G; ++g) res[c][g][0] = res[c][g][1] = res[c][g][2] = res[c][g][3] = 0.0f;

    if (with_bloom) {
        const size_t plane = (size_t)(rows + 2 * R) * W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            __syncthreads();
            for (int k = threadIdx.x; k < tile_rows * COLS; k += 256) {
                const int r = k / COLS, cc = k % COLS;
                const int yl = y0 - R4 + r, yg = yl + row0, xx = blockIdx.x * COLS + cc;
                float v = 0.0f;
                if (yg >= 0 && yg < H && yl >= -R && yl < rows + R && xx < W) v = hblur[c * plane + (size_t)(yl + R) * W + xx];
                lds[cc * S + r] = v;
            }
            __syncthreads();
            // group g of this lane = rows 4 (lane_g + LG g) .. + 3 of the tile
            conv4g<G, WLDS>(lds + col * S + 4 * lane_g, 4 * LG, M, (WLDS ? wl : wext) + c * xstride, res[c]);
        }
    }
    if (x >= W) return;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int grp = lane_g + LG * g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yl = y0 + 4 * grp + k;
            if (yl >= row_end) continue;
            const int yg = yl + row0;
            float b0 = 0, b1 = 0, b2 = 0;
            if (with_bloom) {
                b0 = res[0][g][k] / wsum_v[yg];
                b1 = res[1][g][k] / wsum_v[H + yg];
                b2 = res[2][g][k] / wsum_v[2 * H + yg];
            }
            const size_t o = ((size_t)yl * W + x) * 3;
            blur_out[o + 0] = b0;
            blur_out[o + 1] = b1;
            blur_out[o + 2] = b2;
            const float f0 = fminf(fmaxf(bg[o + 0] + disk[o + 0] + b0, 0.0f), 1.0f);
            const float f1 = fminf(fmaxf(bg[o + 1] + disk[o + 1] + b1, 0.0f), 1.0f);
            const float f2 = fminf(fmaxf(bg[o + 2] + disk[o + 2] + b2, 0.0f), 1.0f);
            final_out[o + 0] = f0;
            final_out[o + 1] = f1;
            final_out[o + 2] = f2;
            if (u8_out) {
                u8_out[o + 0] = (uint8_t)(int)(f0 * 255.0f);
                u8_out[o + 1] = (uint8_t)(int)(f1 * 255.0f);
                u8_out[o + 2] = (uint8_t)(int)(f2 * 255.0f);
            }
        }
    }
}

// ---- V pass on the matrix cores -------------------------------------------------------------------------------------
// The vertical blur of a 32-column strip is a banded Toeplitz product  out(y, x) = sum_i T(y, i) in(i, x),  T(y, i) =
// w[|i - y|].  v_mfma_f32_32x32x2_f32 takes exactly that shape: D(32 y x 32 x) += A(32 y x 2 i) B(2 i x 32 x), exact f32
// (bit for bit a k-ordered fmaf chain, MI355X_MICROARCH.md) at the f32 vector peak -- but with operands that cost almost
// nothing to fetch, where the VALU kernels above are bound by what feeds their FMAs (SGPR-operand FMAs issue at half
// rate, VGPR weights come through LDS, tools/ubench_fmac.hip):
//   B  lane l holds in(i0 + (l >> 5), x0 + (l & 31)): two 128-byte row segments of the planar H-blur buffer, straight
//      from global memory (L2-resident: every input row is read by (32 T + 2 R) / 32 T row groups), no LDS staging;
//   A  lane l holds w[|i0 + (l >> 5) - y - (l & 31)|]: one ds_read_b32 from a zero-padded table of 2 R + 64 T + 44 weights.
// A wave owns T stacked 32-row tiles of one strip for all three channels (48 T accumulator registers), walks the input
// rows in pairs, and each pair feeds the tiles whose band it touches.  Accumulation order per output = ascending input
// row, whatever the tiling: row blocks, chunks and whole frames give the same bits.  No dense GEMM is invented here --
// the 91 % of the issued multiply-adds that fall inside the band are the convolution's own.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 48 T accumulator registers (AGPRs) + ~30 for the loop; the epilogue is held to what is left at 4 (T <= 2) waves per SIMD
template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(T == 1 ? 4 : T == 2 ? 3 : 2, T == 1 ? 4 : T == 2 ? 3 : 2))) void bloom_v_mfma_kernel(const float *__restrict__ hblur, const float *__restrict__ bg,
                                                           const float *__restrict__ disk, float *__restrict__ blur_out,
                                                           float *__restrict__ final_out, const float *__restrict__ wtab,
                                                           const float *__restrict__ wsum_v, int W, int H, int row0, int rows,
                                                           int R, int with_bloom, unsigned long long *__restrict__ zero_cell,
                                                           int row_begin, int row_end, uint8_t *__restrict__ u8_out) {
    if (zero_cell && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < BHR_STEP_LANES)
        zero_cell[(size_t)threadIdx.x * BHR_STEP_STRIDE] = 0ull;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // wl[3][NW]: w[|d|] at index d + OFF, zero beyond R
    const int OFF = R + 32 * T + 2, NW = 2 * R + 64 * T + 44;     // every (pair, tile) offset lands inside: no band test in the loop
    if (with_bloom) {
        const int wstride = R + 1 + WPAD;
        for (int k = threadIdx.x; k < 3 * NW; k += 256) {
            const int c = k / NW, d = k - c * NW - OFF, ad = d < 0 ? -d : d;
            lds[k] = ad <= R ? wtab[c * wstride + ad] : 0.0f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xs = (blockIdx.x * 4 + wave) * 32;                  // this wave's column strip
    const int y0 = row_begin + blockIdx.y * (32 * T);             // local row of its first output
    if (xs >= W) return;
    const int x = xs + (lane & 31), kh = lane >> 5;
    const bool x_ok = x < W;

    f32x16 acc[3][T];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;

    if (with_bloom) {
        const size_t plane = (size_t)(rows + 2 * R) * W;
        const int Rp = R + (R & 1);
        const int i_first = y0 - Rp, i_last = y0 + 32 * T - 1 + R;          // input rows (local), walked in pairs
        const int lane_off = OFF + kh - (lane & 31);
        auto load_b = [&](int i0, float (&b)[3]) {
            const int yl = i0 + kh, yg = yl + row0;
            const bool ok = x_ok && yg >= 0 && yg < H && yl >= -R && yl < rows + R;   // outside the image: a skipped tap
            const size_t o = ok ? (size_t)(yl + R) * W + x : 0;
            b[0] = ok ? hblur[o] : 0.0f;
            b[1] = ok ? hblur[plane + o] : 0.0f;
            b[2] = ok ? hblur[2 * plane + o] : 0.0f;
        };
        // PF row pairs per group: the next group's 3 PF loads are in flight under this group's 3 T PF MFMAs (64 cycles each)
        // -- an L2 round trip is several hundred ns, one pair's MFMAs cover 160
        constexpr int PF = 4;
        float b[PF][3], bn[PF][3];
#pragma unroll
        for (int q = 0; q < PF; ++q) load_b(i_first + 2 * q, b[q]);
        for (int i0 = i_first; i0 <= i_last; i0 += 2 * PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) load_b(i0 + 2 * (PF + q), bn[q]);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    // a pair outside tile t's band (or past the last input row) reads zeros from the padded table: cheaper
                    // than a branch, whose merge copies the 48 T accumulator registers
                    const int idx = lane_off + (i0 + 2 * q - y0 - 32 * t);
                    const float a0 = lds[idx], a1 = lds[NW + idx], a2 = lds[2 * NW + idx];
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[q][0], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[q][1], acc[1][t], 0, 0, 0);
                    acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b[q][2], acc[2][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) { b[q][0] = bn[q][0]; b[q][1] = bn[q][1]; b[q][2] = bn[q][2]; }
        }
    }
    if (!x_ok) return;
    // D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int yl = y0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (yl >= row_end) continue;
            const int yg = yl + row0;
            float b0 = 0, b1 = 0, b2 = 0;
            if (with_bloom) {
                b0 = acc[0][t][r] / wsum_v[yg];
                b1 = acc[1][t][r] / wsum_v[H + yg];
                b2 = acc[2][t][r] / wsum_v[2 * H + yg];
            }
            const size_t o = ((size_t)yl * W + x) * 3;
            blur_out[o + 0] = b0;
            blur_out[o + 1] = b1;
            blur_out[o + 2] = b2;
            const float f0 = fminf(fmaxf(bg[o + 0] + disk[o + 0] + b0, 0.0f), 1.0f);
            const float f1 = fminf(fmaxf(bg[o + 1] + disk[o + 1] + b1, 0.0f), 1.0f);
            const float f2 = fminf(fmaxf(bg[o + 2] + disk[o + 2] + b2, 0.0f), 1.0f);
            final_out[o + 0] = f0;
            final_out[o + 1] = f1;
            final_out[o + 2] = f2;
            if (u8_out) {
                u8_out[o + 0] = (uint8_t)(int)(f0 * 255.0f);
                u8_out[o + 1] = (uint8_t)(int)(f1 * 255.0f);
                u8_out[o + 2] = (uint8_t)(int)(f2 * 255.0f);
            }
        }
    }
}

// H pass on the matrix cores: the same product with the roles turned.  D(32 rows y x 32 outputs x) += A(32 y x 2 i) B(2 i x 32 x):
//   A  lane l holds the INPUT pixel (y0 + (l & 31), i0 + (l >> 5)) -- every lane streams along its own row of the
//      (rows, W, 3) disk layer, one 12-byte load brings the three channels of a tap;
//   B  lane l holds w[|i0 + (l >> 5) - x - (l & 31)|] from the zero-padded LDS table.
// A wave owns T adjacent 32-pixel output tiles of 32 rows; D's column index is the lane, so a store writes 32 consecutive
// floats of one planar H-blur row.  Ascending input order per output, whatever the tiling.
template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(T == 1 ? 4 : T == 2 ? 3 : 2, T == 1 ? 4 : T == 2 ? 3 : 2))) void bloom_h_mfma_kernel(
    const float *__restrict__ disk, float *__restrict__ hblur, const float *__restrict__ wtab, const float *__restrict__ wsum_h,
    int W, int rows, int R, int row_begin, int row_end) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int OFF = R + 32 * T + 2, NW = 2 * R + 64 * T + 44;
    {
        const int wstride = R + 1 + WPAD;
        for (int k = threadIdx.x; k < 3 * NW; k += 256) {
            const int c = k / NW, d = k - c * NW - OFF, ad = d < 0 ? -d : d;
            lds[k] = ad <= R ? wtab[c * wstride + ad] : 0.0f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * (32 * T);                          // first output pixel of the wave's tiles
    const int y0 = row_begin + (blockIdx.y * 4 + wave) * 32;      // its 32 rows
    if (y0 >= row_end) return;
    const int kh = lane >> 5, yl = y0 + (lane & 31);
    const bool y_ok = yl < row_end;
    const float *src = disk + (size_t)(y_ok ? yl : y0) * W * 3;

    f32x16 acc[3][T];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;

    const int Rp = R + (R & 1);
    const int i_first = x0 - Rp, i_last = x0 + 32 * T - 1 + R;
    const int lane_off = OFF + kh - (lane & 31);
    auto load_a = [&](int i0, float (&a)[3]) {
        const int i = i0 + kh;
        const bool ok = y_ok && i >= 0 && i < W;                 // outside the image: a skipped tap
        const float *q = src + (size_t)(ok ? i : 0) * 3;
        a[0] = ok ? q[0] : 0.0f;
        a[1] = ok ? q[1] : 0.0f;
        a[2] = ok ? q[2] : 0.0f;
    };
    constexpr int PF = 4;
    float a[PF][3], an[PF][3];
#pragma unroll
    for (int q = 0; q < PF; ++q) load_a(i_first + 2 * q, a[q]);
    for (int i0 = i_first; i0 <= i_last; i0 += 2 * PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) load_a(i0 + 2 * (PF + q), an[q]);
#pragma unroll
        for (int q = 0; q < PF; ++q) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int idx = lane_off + (i0 + 2 * q - x0 - 32 * t);
                const float b0 = lds[idx], b1 = lds[NW + idx], b2 = lds[2 * NW + idx];
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], b0, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], b1, acc[1][t], 0, 0, 0);
                acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], b2, acc[2][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) { a[q][0] = an[q][0]; a[q][1] = an[q][1]; a[q][2] = an[q][2]; }
    }
    // D: column = lane & 31 -> output pixel, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> row of the wave's 32
    const size_t plane = (size_t)(rows + 2 * R) * W;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int x = x0 + 32 * t + (lane & 31);
        if (x >= W) continue;
        const float s0 = wsum_h[x], s1 = wsum_h[W + x], s2 = wsum_h[2 * W + x];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (y >= row_end) continue;
            const size_t o = (size_t)(y + R) * W + x;
            hblur[o] = acc[0][t][r] / s0;
            hblur[plane + o] = acc[1][t][r] / s1;
            hblur[2 * plane + o] = acc[2][t][r] / s2;
        }
    }
}

// ---- both passes on the bf16 matrix cores, f32 operands split three ways ---------------------------------------------
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32 form.  An f32 value is EXACTLY the sum of three bf16 values
// (its 24 significand bits cut 8 + 8 + 8 by truncation: x = hi + mid + lo, every difference exact), so the product of two
// f32 values is the sum of nine bf16 products; the six of relative order >= 2^-16 (hi hi, hi mid, mid hi, hi lo, lo hi,
// mid mid) leave out 2^-23 of the result -- the rounding of ONE f32 operation, where the f32 chain rounds once per tap.
// Six MFMAs of depth 16 instead of eight of depth 2: 2.7x the f32 matrix (= vector) peak.  Products are exact in the
// MFMA, sums are kept in f32.  Not bit-identical to the f32 kernels (tools/exp_bloom.py: 2-3e-7 on layers of order 1);
// selected for the fast and hybrid arithmetic, never for strict (bhr_ctx::bloom_split).
//   pixels   : loaded as f32, cut in registers (4 VALU per value + 1.5 to pack pairs), shared by the wave's T tiles;
//   weights  : the Toeplitz operand w[|i - y|] is eight consecutive entries of the zero-padded table starting at an index
//              that depends on the lane -- the table sits in LDS as three bf16 parts x EIGHT copies shifted by 0..7
//              entries, so that every lane's window is one aligned ds_read_b128 (copies 32 B apart modulo 256: the sixteen
//              lanes of a pass hit sixteen different bank groups); built once per context in global memory
//              (bloom_wsplit_kernel), copied per block.
// K chunks are aligned to GLOBAL multiples of 16 rows / pixels, so an output's operands meet the same MFMA slots whatever
// the tiling: row blocks, chunks and whole frames give the same bits (tests/test_gpu_multidevice.py).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int SPLIT_TMAX = 2;          // the table is padded for up to 2 stacked / adjacent 32-wide tiles per wave

__host__ __device__ inline int wsplit_off(int R) { return R + 32 * SPLIT_TMAX + 24; }
__host__ __device__ inline int wsplit_nw(int R) { return (wsplit_off(R) + R + 32 * SPLIT_TMAX + 24 + 7) & ~7; }
__host__ __device__ inline int wsplit_cs(int R) {       // bytes between two shifted copies: >= 2 NW, == 32 modulo 256
    const int need = 2 * wsplit_nw(R);
    return ((need - 32 + 255) / 256) * 256 + 32;
}

__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned u = __float_as_uint(x);
    h = u & 0xffff0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));      // at most 8 significant bits left: its upper half is all of it
}

// out[((c * 3 + part) * 8 + shift) * CS / 2 + n] = part of w_c[|n + shift - OFF|] (zero beyond R), n + shift < NW
__global__ void bloom_wsplit_kernel(const float *wtab, unsigned short *out, int R) {
    const int OFF = wsplit_off(R), NW = wsplit_nw(R), CS2 = wsplit_cs(R) / 2, stride = R + 1 + WPAD;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 3 * NW) return;
    const int c = k / NW, idx = k - c * NW;
    int d = idx - OFF;
    d = d < 0 ? -d : d;
    const float w = d <= R ? wtab[c * stride + d] : 0.0f;
    unsigned part[3];
    split3(w, part[0], part[1], part[2]);
    for (int p = 0; p < 3; ++p)
        for (int sh = 0; sh < 8; ++sh) {
            const int n = idx - sh;
            if (n >= 0) out[(size_t)((c * 3 + p) * 8 + sh) * CS2 + n] = (unsigned short)(part[p] >> 16);
        }
}

// eight f32 values -> three fragments of eight bf16 (k = element index)
// `sel` = 0x07060302 held in a VGPR by the caller: v_perm_b32 with an SGPR selector issues at half rate
// (profiles/r03_valu_sgpr_operand_ubench.txt)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo, unsigned sel) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(x[j], h[j], m[j], l[j]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // dword q = elements 2q (low half) and 2q + 1 (high half): their upper 16 bits
        hi[q] = __builtin_amdgcn_perm(h[2 * q + 1], h[2 * q], sel);
        mid[q] = __builtin_amdgcn_perm(m[2 * q + 1], m[2 * q], sel);
        lo[q] = __builtin_amdgcn_perm(l[2 * q + 1], l[2 * q], sel);
    }
}

// acc += sum over the chunk's 16 k of (pixel part) x (weight part), the six products in ascending order of magnitude
#define BHR_SPLIT_MFMA6(ACC, PH, PM, PL, WH, WM, WL, PIX_IS_A)                                                                     \
    do {                                                                                                                           \
        const bf16x8 mq_ph_ = __builtin_bit_cast(bf16x8, PH), mq_pm_ = __builtin_bit_cast(bf16x8, PM), mq_pl_ = __builtin_bit_cast(bf16x8, PL); \
        const bf16x8 mq_wh_ = __builtin_bit_cast(bf16x8, WH), mq_wm_ = __builtin_bit_cast(bf16x8, WM), mq_wl_ = __builtin_bit_cast(bf16x8, WL); \
        if (PIX_IS_A) {                                                                                                            \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_pl_, mq_wh_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_ph_, mq_wl_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_pm_, mq_wm_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_pm_, mq_wh_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_ph_, mq_wm_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_ph_, mq_wh_, ACC, 0, 0, 0);                                                 \
        } else {                                                                                                                   \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_wh_, mq_pl_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_wl_, mq_ph_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_wm_, mq_pm_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_wh_, mq_pm_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_wm_, mq_ph_, ACC, 0, 0, 0);                                                 \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mq_wh_, mq_ph_, ACC, 0, 0, 0);                                                 \
        }                                                                                                                          \
    } while (0)

// copy the context's split table (9 x 8 copies of CS bytes) into LDS
__device__ __forceinline__ void stage_wsplit(unsigned char *lds, const unsigned short *__restrict__ wsplit, int R) {
    const int n16 = 72 * wsplit_cs(R) / 16;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(wsplit);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    for (int k = threadIdx.x; k < n16; k += 256) dst[k] = src[k];
}

// V pass.  A wave owns T stacked 32-row tiles of one 32-column strip; grid (ceil(W / 128), ceil(rows / 32 T)).
template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bloom_v_bf16_kernel(const float *__restrict__ hblur, const float *__restrict__ bg,
                                                           const float *__restrict__ disk, float *__restrict__ blur_out,
                                                           float *__restrict__ final_out, const unsigned short *__restrict__ wsplit,
                                                           const float *__restrict__ wsum_v, int W, int H, int row0, int rows,
                                                           int R, int with_bloom, unsigned long long *__restrict__ zero_cell,
                                                           int row_begin, int row_end, uint8_t *__restrict__ u8_out) {
    static_assert(T <= SPLIT_TMAX, "table padding");
    if (zero_cell && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < BHR_STEP_LANES)
        zero_cell[(size_t)threadIdx.x * BHR_STEP_STRIDE] = 0ull;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
    if (with_bloom) stage_wsplit(lds_b, wsplit, R);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xs = (blockIdx.x * 4 + wave) * 32;
    const int y0 = row_begin + blockIdx.y * (32 * T);
    if (xs >= W) return;
    const int n = lane & 31, kh = lane >> 5, x = xs + n;
    const bool x_ok = x < W;

    f32x16 acc[3][T];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;

    if (with_bloom) {
        const int OFF = wsplit_off(R), CS = wsplit_cs(R);
        const size_t plane = (size_t)(rows + 2 * R) * W;
        const int yg0 = y0 + row0;                                           // global row of the wave's first output
        const int ig_first = (yg0 - R) & ~15, ig_last = yg0 + 32 * T - 1 + R;   // chunk starts: global multiples of 16
        // window start of tile 0 in chunk ig: OFF + ig - yg0 + 8 kh - n; its residue modulo 8 picks the shifted copy
        const int s0 = OFF - yg0 + 8 * kh - n + ig_first;                    // >= 0 by the table's padding
        const int ph = s0 & 7;
        const unsigned char *wl = lds_b + ph * CS + (s0 - ph) * 2;          // advances 32 bytes per chunk, -64 per tile
        const int part = 8 * CS;                                             // bytes between two parts, 3 parts per channel
        // Chunks wholly outside the image are skipped taps and are not walked; the last chunk of an image whose height is
        // not a multiple of 16 runs into the halo rows below the image, which stay zero for the life of the context
        // (alloc_slot) -- skipped taps as well.  A chunk inside the image may reach up to 15 rows past what this context's
        // planes hold: rows outside the band of every output the wave stores (zero weights), read from the neighbouring
        // plane or from the BHR_HBLUR_PAD_ROWS zero rows around the allocation -- finite values, no predicate on any load.
        // Addresses: a wave-uniform row pointer (scalar registers, scalar arithmetic) + one per-lane offset.
        const unsigned lane_off = (unsigned)(8 * kh) * (unsigned)W + (x_ok ? x : W - 1);   // a column past the image: garbage nobody stores
        // first and last chunk start walked: inside the image, and not past the chunk that holds the last row of the planes (a
        // tile that overhangs the row block computes rows nobody stores from whatever the chunks it does walk contain)
        const int ig_a = max(ig_first, 0), ig_b = min(min(ig_last, (H - 1) & ~15), (row0 + rows + R - 1) & ~15);
        wl += 2 * (ig_a - ig_first);
        auto load = [&](int ig, float (&b)[3][8]) {
            const float *rowp = hblur + ((ptrdiff_t)(min(ig, ig_b) - row0 + R)) * W;     // prefetch past the last chunk: a re-read
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) b[c][j] = (rowp + c * plane + (size_t)j * W)[lane_off];
        };
        unsigned sel = 0x07060302u;
        asm volatile("" : "+v"(sel));
        auto chunk = [&](const float (&b)[3][8], const unsigned char *wq) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                u32x4 ph_, pm_, pl_;
                split8(b[c], ph_, pm_, pl_, sel);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const unsigned char *q = wq + (3 * c) * part - 64 * t;
                    const u32x4 wh = *reinterpret_cast<const u32x4 *>(q);
                    const u32x4 wm = *reinterpret_cast<const u32x4 *>(q + part);
                    const u32x4 wlo = *reinterpret_cast<const u32x4 *>(q + 2 * part);
                    BHR_SPLIT_MFMA6(acc[c][t], ph_, pm_, pl_, wh, wm, wlo, false);
                }
            }
        };
        // two chunk buffers, alternating: the loads of chunk k + 1 are issued and pinned (sched_barrier) in front of the
        // MFMAs of chunk k, which cover their latency
        float b0[3][8], b1[3][8];
        load(ig_a, b0);
        for (int ig = ig_a; ig <= ig_b;) {
            load(ig + 16, b1);
            __builtin_amdgcn_sched_barrier(0);
            chunk(b0, wl);
            ig += 16; wl += 32;
            if (ig > ig_b) break;
            load(ig + 16, b0);
            __builtin_amdgcn_sched_barrier(0);
            chunk(b1, wl);
            ig += 16; wl += 32;
        }
    }
    if (!x_ok) return;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int yl = y0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (yl >= row_end) continue;
            const int yg = yl + row0;
            float b0 = 0, b1 = 0, b2 = 0;
            if (with_bloom) {
                b0 = acc[0][t][r] / wsum_v[yg];
                b1 = acc[1][t][r] / wsum_v[H + yg];
                b2 = acc[2][t][r] / wsum_v[2 * H + yg];
            }
            const size_t o = ((size_t)yl * W + x) * 3;
            blur_out[o + 0] = b0;
            blur_out[o + 1] = b1;
            blur_out[o + 2] = b2;
            const float f0 = fminf(fmaxf(bg[o + 0] + disk[o + 0] + b0, 0.0f), 1.0f);
            const float f1 = fminf(fmaxf(bg[o + 1] + disk[o + 1] + b1, 0.0f), 1.0f);
            const float f2 = fminf(fmaxf(bg[o + 2] + disk[o + 2] + b2, 0.0f), 1.0f);
            final_out[o + 0] = f0;
            final_out[o + 1] = f1;
            final_out[o + 2] = f2;
            if (u8_out) {
                u8_out[o + 0] = (uint8_t)(int)(f0 * 255.0f);
                u8_out[o + 1] = (uint8_t)(int)(f1 * 255.0f);
                u8_out[o + 2] = (uint8_t)(int)(f2 * 255.0f);
            }
        }
    }
}

// H pass.  A wave owns T adjacent 32-pixel output tiles of 32 rows; grid (ceil(W / 32 T), ceil(rows / 128)).
template <int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bloom_h_bf16_kernel(const float *__restrict__ disk, float *__restrict__ hblur,
                                                           const unsigned short *__restrict__ wsplit, const float *__restrict__ wsum_h,
                                                           int W, int rows, int R, int row_begin, int row_end) {
    static_assert(T <= SPLIT_TMAX, "table padding");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
    stage_wsplit(lds_b, wsplit, R);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * (32 * T);
    const int y0 = row_begin + (blockIdx.y * 4 + wave) * 32;
    if (y0 >= row_end) return;
    const int n = lane & 31, kh = lane >> 5, yl = y0 + n;
    const bool y_ok = yl < row_end;

    f32x16 acc[3][T];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;

    const int OFF = wsplit_off(R), CS = wsplit_cs(R);
    const int ig_first = (x0 - R) & ~15, ig_last = x0 + 32 * T - 1 + R;
    const int s0 = OFF - x0 + 8 * kh - n + ig_first;
    const int ph = s0 & 7;
    const unsigned char *wl = lds_b + ph * CS + (s0 - ph) * 2;
    const int part = 8 * CS;
    // W is a multiple of 16 (the launcher's condition): a chunk lies whole inside the row or whole outside it, and those
    // outside are skipped taps that are not walked; rows are 16-byte aligned.  A row past the block reads row y0 and is
    // never stored.  The lane's 8 pixels x 3 channels of a chunk are 24 consecutive floats of its row.
    const int ig_a = max(ig_first, 0), ig_b = min(ig_last, W - 16);
    wl += 2 * (ig_a - ig_first);
    const unsigned lane_off = (unsigned)(y_ok ? yl : y0) * (unsigned)W * 3u + 24u * kh;   // elements; the frame is bounded by bhr_create
    auto load = [&](int ig, float (&a)[24]) {
        const float *u = disk + (size_t)min(ig, W - 16) * 3;              // wave uniform; prefetch past the end: a re-read
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const float4 f = *reinterpret_cast<const float4 *>(u + lane_off + 4 * v);
            a[4 * v] = f.x; a[4 * v + 1] = f.y; a[4 * v + 2] = f.z; a[4 * v + 3] = f.w;
        }
    };
    unsigned sel = 0x07060302u;
    asm volatile("" : "+v"(sel));
    auto chunk = [&](const float (&a)[24], const unsigned char *wq) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float xc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xc[j] = a[3 * j + c];
            u32x4 ph_, pm_, pl_;
            split8(xc, ph_, pm_, pl_, sel);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const unsigned char *q = wq + (3 * c) * part - 64 * t;
                const u32x4 wh = *reinterpret_cast<const u32x4 *>(q);
                const u32x4 wm = *reinterpret_cast<const u32x4 *>(q + part);
                const u32x4 wlo = *reinterpret_cast<const u32x4 *>(q + 2 * part);
                BHR_SPLIT_MFMA6(acc[c][t], ph_, pm_, pl_, wh, wm, wlo, true);
            }
        }
    };
    float a0[24], a1[24];                                     // two chunk buffers, alternating: see the V pass
    load(ig_a, a0);
    for (int ig = ig_a; ig <= ig_b;) {
        load(ig + 16, a1);
        __builtin_amdgcn_sched_barrier(0);
        chunk(a0, wl);
        ig += 16; wl += 32;
        if (ig > ig_b) break;
        load(ig + 16, a0);
        __builtin_amdgcn_sched_barrier(0);
        chunk(a1, wl);
        ig += 16; wl += 32;
    }
    const size_t plane = (size_t)(rows + 2 * R) * W;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int x = x0 + 32 * t + n;
        if (x >= W) continue;
        const float s0w = wsum_h[x], s1w = wsum_h[W + x], s2w = wsum_h[2 * W + x];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (y >= row_end) continue;
            const size_t o = (size_t)(y + R) * W + x;
            hblur[o] = acc[0][t][r] / s0w;
            hblur[plane + o] = acc[1][t][r] / s1w;
            hblur[2 * plane + o] = acc[2][t][r] / s2w;
        }
    }
}

// V-pass geometry of this context: columns per block, row groups per thread (rows per block = 256 / cols x 4 x groups).
// BHR_BLOOM_V="<cols>x<groups>" overrides ("32x0": the round-2 kernel with its own table below); BHR_BLOOM_H=<NG>.
struct VGeom { int cols, groups, v2; };
// Round-2 table of the 32-column kernel (v2 = 0): G = 1 (32 rows) up to fhd, 4 at 4k, 8 (256 rows) at 8k.
VGeom v_geometry(int R, int rows, bool split) {
    // default: the matrix-core kernel, one 32-row tile per wave -- 1.03 / 0.190 / 0.041 ms at 8k / 4k / fhd against 1.78 /
    // 0.247 / 0.040 of the round-2 kernel and 1.21 / 0.193 / 0.041 of the 16-column VALU kernel (tools/exp_bloom.py)
    VGeom g{128, 1, 2};
    if (split) g = VGeom{128, R >= 64 ? 2 : 1, 3};     // bf16 x 3: v2 = 3, groups = stacked tiles per wave
    if (const char *e = getenv("BHR_BLOOM_V")) {
        if (e[0] == 'b') {                         // "bf16x1" / "bf16x2"
            const int t = atoi(e + 5);
            return VGeom{128, t == 1 ? 1 : 2, 3};
        }
        if (split) return g;                       // the f32 variants below are the strict path's
    }
    if (const char *e = getenv("BHR_BLOOM_V")) {
        if (e[0] != 'm') { g = VGeom{32, 1, 0}; if (R >= 64) g.groups = (R >= 128 && rows > 128) ? 8 : 4; }
    }
    if (const char *e = getenv("BHR_BLOOM_VG")) { int v = atoi(e); if (g.v2 == 0 && (v == 1 || v == 2 || v == 4 || v == 8)) g.groups = v; }
    if (const char *e = getenv("BHR_BLOOM_V")) {
        if (e[0] == 'm') {                         // "mfma1" / "mfma2" / "mfma4": T stacked 32-row tiles per wave
            const int t = atoi(e + 4);
            g.cols = 128; g.groups = (t == 1 || t == 2 || t == 4) ? t : 2; g.v2 = 2;
            return g;
        }
        int c = 0, k = 0;
        // "16x<G>": the 16-column VALU kernel with G row groups per thread; "32x0": the round-2 kernel with its own table
        if (sscanf(e, "%dx%d", &c, &k) == 2 && c == 16 && (k == 1 || k == 2 || k == 4)) { g.cols = 16; g.groups = k; g.v2 = 1; }
    }
    return g;
}
int v_rows_per_block(const VGeom &g) { return g.v2 >= 2 ? 32 * g.groups : 256 / g.cols * 4 * g.groups; }
int v_stride(int R, const VGeom &g) {   // LDS column stride: >= tile rows, multiple of 4 with an odd quotient (bank spread)
    const int R4 = (R + 3) & ~3;
    int s = v_rows_per_block(g) + 2 * R4 + 4;
    if (((s >> 2) & 1) == 0) s += 4;
    return s;
}
bool weights_in_lds() {
    if (const char *e = getenv("BHR_BLOOM_W")) return e[0] == 'l';
    return true;
}
int h_groups(int R, bool split) {
    if (const char *e = getenv("BHR_BLOOM_H")) {
        if (e[0] == 'b') return 200 + (atoi(e + 5) == 1 ? 1 : 2);                                                // "bf16x<T>"
        if (split) return 200 + (R >= 64 ? 2 : 1);
        if (e[0] == 'm') { const int t = atoi(e + 4); return 100 + ((t == 1 || t == 2 || t == 4) ? t : 2); }   // "mfma<T>"
        int v = atoi(e); if (v == 0 || v == 1 || v == 2 || v == 4) return v;
    }
    // measured at 8k / 4k / fhd (tools/exp_bloom.py, profiles/r03_bloom_variants.md): two groups per thread with the
    // weights as VGPR operands 0.87 / 0.127 / 0.026 ms; the round-2 kernel 0.97 / 0.146 / 0.030; the MFMA form 0.87 / 0.149 / 0.048
    if (split) return 200 + (R >= 64 ? 2 : 1);
    return 2;
}

// kernels that need more than 48 KB of dynamic LDS are told so once
// the bf16 H kernel walks a row in chunks of 16 pixels that must lie whole inside or whole outside it (the V kernel's chunks
// may overhang the image: they meet zero halo rows)
bool split_geometry_ok(const bhr_ctx *ctx) { return (ctx->cfg.width & 15) == 0; }
// bloom_split: 0 = f32 kernels, 2 = both passes bf16 (BHR_BLOOM_SPLIT=1), 1 = each pass where it pays (fast / hybrid frames):
// the V pass from radius 16 (fhd, R = 38: 0.034 against 0.041 ms, +1.7 % on the two-frames-in-flight headline), the H
// pass from radius 64 (fhd: 0.031 against 0.026 -- its per-lane row streams cost more than the short convolution gains)
bool use_split_h(const bhr_ctx *ctx) {
    return split_geometry_ok(ctx) && (ctx->bloom_split == 2 || (ctx->bloom_split == 1 && ctx->bloom_R >= 64));
}
bool use_split_v(const bhr_ctx *ctx) {
    return split_geometry_ok(ctx) && (ctx->bloom_split == 2 || (ctx->bloom_split == 1 && ctx->bloom_R >= 16));
}

// (a function attribute belongs to the device it was set on: the note is kept per (device, kernel) -- row-block tiles on the
// eight devices of a node launch the same kernels from one process)
int32_t allow_lds(const void *fn, size_t bytes) {
    constexpr int CAP = 256;
    static const void *done[CAP];
    static size_t done_bytes[CAP];
    static int done_dev[CAP];
    static int n_done = 0;
    static std::mutex mu;
    if (bytes <= 48 * 1024) return BHR_OK;
    int dev = 0;
    BHR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);           // group renders submit from one host thread per tile
    for (int k = 0; k < n_done; ++k)
        if (done[k] == fn && done_dev[k] == dev && done_bytes[k] >= bytes) return BHR_OK;
    BHR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (n_done < CAP) { done[n_done] = fn; done_dev[n_done] = dev; done_bytes[n_done++] = bytes; }
    return BHR_OK;
}

}  // namespace

int32_t bhr_bloom_prepare(bhr_ctx *ctx) {
    if (ctx->bloom_ready) return BHR_OK;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const int R = ctx->bloom_R;
    const int n = R + 1 + WPAD;
    // render.py:3915: sigma_scale = (width / 640.0) ** 2 in Python floats, passed as f32
    const float sigma_scale = (float)(((double)W / 640.0) * ((double)W / 640.0));
    hipLaunchKernelGGL(bloom_weights_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_wtab, R, sigma_scale);
    {
        const int R4 = (R + 3) & ~3, nx = 2 * R4 + 8;
        hipLaunchKernelGGL(bloom_wext_kernel, dim3((nx + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_wtab, ctx->d_wext, R);
    }
    hipLaunchKernelGGL(bloom_wsum_kernel, dim3((W + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_wtab, ctx->d_wsum_h, R, W);
    hipLaunchKernelGGL(bloom_wsum_kernel, dim3((H + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_wtab, ctx->d_wsum_v, R, H);
    if (!ctx->d_wsplit) {
        const size_t bytes = (size_t)72 * wsplit_cs(R);
        BHR_HIP(hipMalloc(&ctx->d_wsplit, bytes));
        BHR_HIP(hipMemsetAsync(ctx->d_wsplit, 0, bytes, ctx->stream));
    }
    hipLaunchKernelGGL(bloom_wsplit_kernel, dim3((3 * wsplit_nw(R) + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_wtab, ctx->d_wsplit, R);
    BHR_HIP(hipGetLastError());
    ctx->bloom_ready = 1;
    return BHR_OK;
}

// H pass over the local rows [r0, r1) of the context (the pipelined row-block path blurs its halo bands first)
int32_t bhr_launch_bloom_h_rows(bhr_ctx *ctx, int32_t r0, int32_t r1) {
    const int W = ctx->cfg.width, R = ctx->bloom_R;
    int32_t rc = bhr_bloom_prepare(ctx);
    if (rc) return rc;
    if (r0 < 0 || r1 > ctx->rows || r0 > r1) return bhr_fail(BHR_ERR_INVALID, "bloom H: rows [%d,%d) of %d", r0, r1, ctx->rows);
    if (r0 == r1) return BHR_OK;
    const int R4 = (R + 3) & ~3;
    const int ng = h_groups(R, use_split_h(ctx));
    if (ng >= 200) {
        if (!split_geometry_ok(ctx)) return bhr_fail(BHR_ERR_INVALID, "bloom H: the bf16 kernels need a width that is a multiple of 16 (%d x %d)", W, ctx->cfg.height);
        const int T = ng - 200;
        dim3 mgrid((W + 32 * T - 1) / (32 * T), (r1 - r0 + 127) / 128), mblock(256);
        const size_t mlds = (size_t)72 * wsplit_cs(R);
#define BHR_HB_ARGS mgrid, mblock, mlds, ctx->stream, ctx->d_disk, ctx->d_hblur, ctx->d_wsplit, ctx->d_wsum_h, W, ctx->rows, R, r0, r1
        if (T == 1) { BHR_TRY(allow_lds((const void *)bloom_h_bf16_kernel<1>, mlds)); hipLaunchKernelGGL(bloom_h_bf16_kernel<1>, BHR_HB_ARGS); }
        else { BHR_TRY(allow_lds((const void *)bloom_h_bf16_kernel<2>, mlds)); hipLaunchKernelGGL(bloom_h_bf16_kernel<2>, BHR_HB_ARGS); }
#undef BHR_HB_ARGS
        BHR_HIP(hipGetLastError());
        return BHR_OK;
    }
    if (ng >= 100) {
        const int T = ng - 100;
        dim3 mgrid((W + 32 * T - 1) / (32 * T), (r1 - r0 + 127) / 128), mblock(256);
        const size_t mlds = (size_t)3 * (2 * R + 64 * T + 44) * sizeof(float);
#define BHR_HM_ARGS mgrid, mblock, mlds, ctx->stream, ctx->d_disk, ctx->d_hblur, ctx->d_wtab, ctx->d_wsum_h, W, ctx->rows, R, r0, r1
        if (T == 1) hipLaunchKernelGGL(bloom_h_mfma_kernel<1>, BHR_HM_ARGS);
        else if (T == 4) hipLaunchKernelGGL(bloom_h_mfma_kernel<4>, BHR_HM_ARGS);
        else hipLaunchKernelGGL(bloom_h_mfma_kernel<2>, BHR_HM_ARGS);
#undef BHR_HM_ARGS
        BHR_HIP(hipGetLastError());
        return BHR_OK;
    }
    const int pix = HB_PIX * (ng ? ng : 1);
    dim3 grid((W + pix - 1) / pix, r1 - r0), block(256);
    const bool wlds = weights_in_lds() && ng != 0;
    const size_t lds = ((size_t)3 * (pix + 2 * R4 + 4) + (wlds ? 3 * (2 * R4 + 8) : 0)) * sizeof(float);
#define BHR_H_ARGS grid, block, lds, ctx->stream, ctx->d_disk, ctx->d_hblur, ctx->d_wext, ctx->d_wsum_h, W, ctx->rows, R, r0
#define BHR_H_LAUNCH(KERNEL) do { BHR_TRY(allow_lds((const void *)KERNEL, lds)); hipLaunchKernelGGL(KERNEL, BHR_H_ARGS); } while (0)
    if (ng == 0) BHR_H_LAUNCH(bloom_h_kernel);
    else if (ng == 1) { if (wlds) BHR_H_LAUNCH((bloom_h2_kernel<1, true>)); else BHR_H_LAUNCH((bloom_h2_kernel<1, false>)); }
    else if (ng == 2) { if (wlds) BHR_H_LAUNCH((bloom_h2_kernel<2, true>)); else BHR_H_LAUNCH((bloom_h2_kernel<2, false>)); }
    else { if (wlds) BHR_H_LAUNCH((bloom_h2_kernel<4, true>)); else BHR_H_LAUNCH((bloom_h2_kernel<4, false>)); }
#undef BHR_H_LAUNCH
#undef BHR_H_ARGS
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

int32_t bhr_launch_bloom_h(bhr_ctx *ctx) { return bhr_launch_bloom_h_rows(ctx, 0, ctx->rows); }

int32_t bhr_bloom_v_tile_rows(bhr_ctx *ctx) { return v_rows_per_block(v_geometry(ctx->bloom_R, ctx->rows, use_split_v(ctx))); }

// V pass + combine over the local rows [r0, r1); u8_out != nullptr: also the quantised final rows ((rows, W, 3) u8 base)
int32_t bhr_launch_bloom_v_rows(bhr_ctx *ctx, int32_t with_bloom, int32_t r0, int32_t r1, uint8_t *u8_out) {
    const int W = ctx->cfg.width, H = ctx->cfg.height, R = ctx->bloom_R;
    int32_t rc = bhr_bloom_prepare(ctx);
    if (rc) return rc;
    if (r0 < 0 || r1 > ctx->rows || r0 > r1) return bhr_fail(BHR_ERR_INVALID, "bloom V: rows [%d,%d) of %d", r0, r1, ctx->rows);
    if (r0 == r1) return BHR_OK;
    const VGeom g = v_geometry(R, ctx->rows, use_split_v(ctx));
    if (g.v2 == 3) {
        if (!split_geometry_ok(ctx)) return bhr_fail(BHR_ERR_INVALID, "bloom V: the bf16 kernels need a width that is a multiple of 16 (%d x %d)", W, H);
        const int vb = 32 * g.groups;
        dim3 mgrid((W + 127) / 128, (r1 - r0 + vb - 1) / vb), mblock(256);
        const size_t mlds = with_bloom ? (size_t)72 * wsplit_cs(R) : 0;
#define BHR_VB_ARGS mgrid, mblock, mlds, ctx->stream, ctx->d_hblur, ctx->d_bg, ctx->d_disk, ctx->d_blur, ctx->d_final, ctx->d_wsplit, \
                    ctx->d_wsum_v, W, H, ctx->cfg.row0, ctx->rows, R, with_bloom, ctx->v_zero_cell, r0, r1, u8_out
        if (g.groups == 1) { BHR_TRY(allow_lds((const void *)bloom_v_bf16_kernel<1>, mlds)); hipLaunchKernelGGL(bloom_v_bf16_kernel<1>, BHR_VB_ARGS); }
        else { BHR_TRY(allow_lds((const void *)bloom_v_bf16_kernel<2>, mlds)); hipLaunchKernelGGL(bloom_v_bf16_kernel<2>, BHR_VB_ARGS); }
#undef BHR_VB_ARGS
        BHR_HIP(hipGetLastError());
        return BHR_OK;
    }
    if (g.v2 == 2) {
        const int vb = 32 * g.groups;
        dim3 mgrid((W + 127) / 128, (r1 - r0 + vb - 1) / vb), mblock(256);
        const size_t mlds = with_bloom ? (size_t)3 * (2 * R + 64 * g.groups + 44) * sizeof(float) : 0;
#define BHR_VM_ARGS mgrid, mblock, mlds, ctx->stream, ctx->d_hblur, ctx->d_bg, ctx->d_disk, ctx->d_blur, ctx->d_final, ctx->d_wtab, \
                    ctx->d_wsum_v, W, H, ctx->cfg.row0, ctx->rows, R, with_bloom, ctx->v_zero_cell, r0, r1, u8_out
        if (g.groups == 1) hipLaunchKernelGGL(bloom_v_mfma_kernel<1>, BHR_VM_ARGS);
        else if (g.groups == 4) hipLaunchKernelGGL(bloom_v_mfma_kernel<4>, BHR_VM_ARGS);
        else hipLaunchKernelGGL(bloom_v_mfma_kernel<2>, BHR_VM_ARGS);
#undef BHR_VM_ARGS
        BHR_HIP(hipGetLastError());
        return BHR_OK;
    }
    const int S = v_stride(R, g), vb_rows = v_rows_per_block(g);
    dim3 grid((W + g.cols - 1) / g.cols, (r1 - r0 + vb_rows - 1) / vb_rows), block(256);
    const bool wlds = weights_in_lds() && g.v2;
    const int R4v = (R + 3) & ~3;
    const size_t lds = with_bloom ? ((size_t)g.cols * S + (wlds ? 3 * (2 * R4v + 8) : 0)) * sizeof(float) : 0;
#define BHR_V_ARGS grid, block, lds, ctx->stream, ctx->d_hblur, ctx->d_bg, ctx->d_disk, ctx->d_blur, ctx->d_final, ctx->d_wext, \
                   ctx->d_wsum_v, W, H, ctx->cfg.row0, ctx->rows, R, S, with_bloom, ctx->v_zero_cell, r0, r1, u8_out
#define BHR_V_LAUNCH(KERNEL) do { BHR_TRY(allow_lds((const void *)KERNEL, lds)); hipLaunchKernelGGL(KERNEL, BHR_V_ARGS); } while (0)
    if (!g.v2) {
        if (g.groups == 8) BHR_V_LAUNCH(bloom_v_kernel<8>);
        else if (g.groups == 4) BHR_V_LAUNCH(bloom_v_kernel<4>);
        else if (g.groups == 1) BHR_V_LAUNCH(bloom_v_kernel<1>);
        else BHR_V_LAUNCH(bloom_v_kernel<2>);
    } else {
#define BHR_V_PICK(C, K) do { if (wlds) BHR_V_LAUNCH((bloom_v2_kernel<C, K, true>)); else BHR_V_LAUNCH((bloom_v2_kernel<C, K, false>)); } while (0)
        if (g.groups == 4) BHR_V_PICK(16, 4);
        else if (g.groups == 1) BHR_V_PICK(16, 1);
        else BHR_V_PICK(16, 2);
#undef BHR_V_PICK
    }
#undef BHR_V_LAUNCH
#undef BHR_V_ARGS
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

int32_t bhr_launch_bloom_v(bhr_ctx *ctx, int32_t with_bloom) { return bhr_launch_bloom_v_rows(ctx, with_bloom, 0, ctx->rows, nullptr); }
