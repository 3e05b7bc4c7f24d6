// bloom.hip -- separable RGB-dispersion bloom + final combine for gfx950.
//
// Restates _bloom_kernel (render.py:3022-3114) as the reference's render() drives it
// (render.py:3914-3918): threshold 0, radius R = int(0.02 W), weights exp(-d^2 / (sigma_c * s)) with
// sigma = {25, 80, 1600} per channel and s = (W/640)^2, out-of-image taps skipped and every channel
// divided by its own in-bounds weight sum.  With threshold 0 and a non-negative disk layer the
// "bright" copy equals the disk layer (lum > 0 fails only for all-zero pixels), so pass 1 of the
// reference is folded away.  The V pass's epilogue is render.py:3918's clip(bg + disk + blur) and, on
// request, save_image's u8 truncation (render.py:423).
//
// Two pairs of kernels (the rest of rounds 1-3 -- 27 instantiations -- lost by measurement and is gone; DESIGN 4
// keeps their numbers):
//
//   exact f32 (strict arithmetic)   H: bloom_h_f32_kernel, 4 adjacent outputs per thread from 16-byte LDS windows,
//                                   weights as VGPR operands read from LDS (an SGPR operand halves the FMA rate);
//                                   V: bloom_v_f32_kernel, the banded Toeplitz product on v_mfma_f32_32x32x2_f32 --
//                                   bit for bit a k-ordered fmaf chain.  Planar f32 intermediate (3, rows + 2R, W).
//
//   split f16 (fast / hybrid)       both passes on v_mfma_f32_32x32x16_f16 with every f32 operand cut in TWO halves:
//                                   x 2^14 = hi + lo with hi = RN16(x 2^14), lo = RN16(x 2^14 - hi) carries 22 + 2
//                                   significant bits (round to nearest gives a bit per half), so hi hi + lo hi + hi lo
//                                   leaves out 2^-24 of a product: three MFMAs per 16 taps where round 3's bf16 x 3 cut
//                                   needed six.  The scalings (pixels 2^14, weights 2^10, both exact) keep every half a
//                                   NORMAL f16 down to 4e-9, so nothing depends on how the matrix cores treat
//                                   subnormals.  What the kernels are built around is memory, not arithmetic:
//     * operands arrive cut and in MFMA fragment order.  The march kernel's epilogue writes the disk layer a second
//       time as f16 pairs laid out [channel][half][32-row block][8-pixel group][row][8 pixels] -- an 8x8 march tile
//       is exactly one 128-byte line of it -- so the H pass's A operand (lane = row, 8 consecutive pixels) is ONE
//       coalesced 16-byte load per lane and half, 1 KB contiguous per wave, where round 3 streamed 96 bytes per lane
//       along 32 different rows (16 B per cache line per load, FETCH 3.2x the layer) and cut them in registers for
//       every 64-pixel output window again.  The H pass writes its result the same way for the V pass:
//       [channel][half][8-row group][column][8 rows], two 512-byte runs per load.
//     * a wave owns up to T = 8 output tiles (32 x 32) of ONE channel and walks the 16-tap chunks of their union:
//       every loaded chunk feeds all tiles whose band it touches (wave-uniform gate: no multiply-add outside the band,
//       96 % of the issued ones inside the radius at 8k), so a chunk is fetched 2.4x at 8k instead of 5.8x.
//     * the Toeplitz operand w[|i - y|] is eight consecutive entries of a zero-padded table at a lane-dependent start:
//       the table sits in LDS as 2 halves x 8 copies shifted by 0..7 entries, placed so that the sixteen lanes of every
//       ds_read_b128 pass hit sixteen different 16-byte slots (conflict free).
//     * the three channel waves of a tile share a workgroup (one CU): their strided accesses to the interleaved
//       (rows, W, 3) layers meet in the same L1 / L2 lines.
//     * the V epilogue stores only what the caller asked for (f32 frame, blur, u8 rows -- the latter straight into
//       the frame buffer of a row-block gather, local or on a peer device), and the H epilogue mirrors the rows a
//       neighbouring row block needs straight into that block's planes (peer-mapped pointers): no copy stages.
//   Chunks are aligned to GLOBAL multiples of 16 rows / pixels and summed in ascending order, so an output's operands
//   meet the same MFMA slots whatever the tiling: row blocks, row chunks and whole frames give the same bits.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "bhr_internal.h"

namespace {

constexpr int WPAD = 8;       // zero entries behind w[R] in the f32 weight table (window overhang <= 6)
constexpr int HB_PIX = 1024;  // pixels per group of an exact H-pass block (256 threads x 4)
constexpr float PIX_SCALE = 16384.0f;            // 2^14: pixel values (<= 1) as f16 halves
constexpr float W_SCALE = 1024.0f;               // 2^10: weights (>= exp(-6.6) by construction of R and sigma)
constexpr float ACC_UNSCALE = 1.0f / 16777216.0f;   // 2^-24
constexpr float HB_RESCALE = 1.0f / 1024.0f;        // 2^-24 x 2^14: an H-pass sum back to a scaled pixel

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float weight_of(int c, int d, float sigma_scale) {
    const float dist_sq = (float)(d * d);
    return c == 0 ? expf(-dist_sq / (25.0f * sigma_scale)) : c == 1 ? expf(-dist_sq / (80.0f * sigma_scale)) : expf(-dist_sq / (1600.0f * sigma_scale));
}

// geometry of the split-f16 weight table (bhr_internal.h: bhr_split_nt): entry i of a channel's table is w[|i - OFF|] (zero
// beyond R); copy `sh` holds entry i + sh at position i.  Copy ph starts 16 * slot(ph) bytes into its CSB-byte cell,
// slot = {0, 1, 5, 9, 13, 5, 9, 13}: with the lane -> window map of the kernels below (start = OFF + 8 h - n + 16 k, copy =
// start & 7) the sixteen lanes of every ds_read_b128 group then read sixteen different 16-byte slots modulo 256 bytes
// (searched exhaustively; the plain 32-bytes-apart placement of round 3 left two-way conflicts)
__host__ __device__ inline int split_off(int NT) { return 16 * NT + 16; }
__host__ __device__ inline int split_nw(int NT) { return 32 * NT + 48; }
__host__ __device__ inline int split_slot(int ph) { return ph == 0 ? 0 : (ph <= 4 ? 4 * ph - 3 : 4 * ph - 15); }
__host__ __device__ inline int split_csb(int NT) { return ((2 * split_nw(NT) + 16 * 13 + 255) / 256) * 256; }
__host__ __device__ inline int split_table_bytes(int NT) { return 48 * split_csb(NT); }   // 3 channels x 2 halves x 8 copies

// x = hi + lo up to 2^-24 |x|.  The half that is stored and the half `lo` is formed against must be the SAME bits: left to
// itself hipcc converts twice -- v_cvt_f16_f32 for the one, v_cvt_pk_f16_f32 (on a differently contracted x) for the other --
// and the two disagree next to ties: hi one f16 ulp off against lo on 0.02 % of the values (found by decoding the planes,
// tools/dbg_bloom3.py).  The empty asm pins the converted bits in a register both uses read.
__device__ __forceinline__ void cut2(float x, _Float16 &hi, _Float16 &lo) {
    asm volatile("" : "+v"(x));
    unsigned int hb = __builtin_bit_cast(unsigned short, (_Float16)x);
    asm volatile("" : "+v"(hb));
    hi = __builtin_bit_cast(_Float16, (unsigned short)hb);
    lo = (_Float16)(x - (float)hi);
}

// One launch builds every table of the context.  blockIdx.y: 0 weights (f32 table, its unfolded form, the split-f16
// copies), 1 wsum_h, 2 wsum_v.  wsum[c][x] = sum over taps d = -R..R with 0 <= x + d < n of w_c[|d|], in tap order.
__global__ void bloom_tables_kernel(float *wtab, float *wext, unsigned short *w16, float *wsum_h, float *wsum_v, int R, int W, int H,
                                    float sigma_scale, int NT) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {
        const int n = R + 1 + WPAD;
        if (k < n)
            for (int c = 0; c < 3; ++c) wtab[c * n + k] = k <= R ? weight_of(c, k, sigma_scale) : 0.0f;
        const int R4 = (R + 3) & ~3, nx = 2 * R4 + 8;      // wext[c][i] = w_c[|i - (R4 + 3)|]: what conv4g reads
        if (k < nx) {
            int d = k - (R4 + 3);
            d = d < 0 ? -d : d;
            for (int c = 0; c < 3; ++c) wext[c * nx + k] = d <= R ? weight_of(c, d, sigma_scale) : 0.0f;
        }
        if (w16 && k < split_nw(NT)) {
            const int OFF = split_off(NT), NW = split_nw(NT), CSB = split_csb(NT);
            int d = k - OFF;
            d = d < 0 ? -d : d;
            for (int c = 0; c < 3; ++c) {
                _Float16 part[2];
                cut2(d <= R ? weight_of(c, d, sigma_scale) * W_SCALE : 0.0f, part[0], part[1]);
                for (int p = 0; p < 2; ++p)
                    for (int sh = 0; sh < 8; ++sh) {
                        const int pos = k - sh;
                        if (pos >= 0 && pos < NW)
                            w16[(size_t)(((c * 2 + p) * 8 + sh) * CSB + 16 * split_slot(sh)) / 2 + pos] = __builtin_bit_cast(unsigned short, part[p]);
                    }
            }
        }
        return;
    }
    const int n = blockIdx.y == 1 ? W : H;
    float *wsum = blockIdx.y == 1 ? wsum_h : wsum_v;
    if (k >= n) return;
    float s0 = 0, s1 = 0, s2 = 0;
    for (int d = -R; d <= R; ++d) {
        const int q = k + d;
        if (0 <= q && q < n) {
            const int ad = d < 0 ? -d : d;
            s0 += weight_of(0, ad, sigma_scale);
            s1 += weight_of(1, ad, sigma_scale);
            s2 += weight_of(2, ad, sigma_scale);
        }
    }
    wsum[0 * n + k] = s0;
    wsum[1 * n + k] = s1;
    wsum[2 * n + k] = s2;
    // the split kernels multiply: un-scaling and normalisation in one factor per output column / row
    const float un = blockIdx.y == 1 ? HB_RESCALE : ACC_UNSCALE;
    wsum[3 * n + k] = un / s0;
    wsum[4 * n + k] = un / s1;
    wsum[5 * n + k] = un / s2;
}

// ==== exact f32 kernels (strict arithmetic) ============================================================================
// NG groups of 4 adjacent outputs per thread, 1024 pixels apart, from 16-byte aligned LDS windows: acc[g][k] += sum_{m,e}
// win[4m + e] w[|4m + e - k - R4|].  `wx` is the channel's symmetric weight table unfolded to wx[i] = w[|i - (R4 + 3)|], read
// from LDS as two broadcast ds_read_b128 per tap block: VGPR operands (v_fmac with an SGPR operand issues at half rate).
template <int NG>
__device__ __forceinline__ void conv4g(const float *__restrict__ win, int gstride, int M, const float *__restrict__ wx, float (&acc)[NG][4]) {
#pragma unroll 2
    for (int m = 0; m < M; ++m) {
        const float *__restrict__ wp = wx + 4 * m;
        const float4 wa = *reinterpret_cast<const float4 *>(wp), wb = *reinterpret_cast<const float4 *>(wp + 4);
        const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
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

// H pass: (rows, W, 3) disk layer -> planar (3, rows + 2R, W) intermediate (hblur row = local row + R).  grid (ceil(W / 2048), rows).
__global__ __launch_bounds__(256) void bloom_h_f32_kernel(const float *__restrict__ disk, float *__restrict__ hblur,
                                                          const float *__restrict__ wext, const float *__restrict__ wsum_h,
                                                          int W, int rows, int R, int row_begin) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NG = 2, PIX = HB_PIX * NG;
    const int R4 = (R + 3) & ~3;
    const int span = PIX + 2 * R4 + 4;
    float *wl = lds + 3 * span;                    // the three unfolded weight tables, 2 R4 + 8 floats each
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
        conv4g<NG>(lds + c * span + 4 * tid, HB_PIX, M, wl + c * xstride, acc);
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

// what a V pass stores, (rows, W, 3) indexed by the context's LOCAL row; null = not wanted
struct VOut {
    float *final_f32;
    float *blur;
    uint8_t *u8;
};

// render.py:3912 / 3918: clip(img + disk [+ blur], 0, 1), and save_image's truncation (render.py:423) where asked
__device__ __forceinline__ void combine_store(const VOut &o, const float *__restrict__ bg, const float *__restrict__ disk, size_t at, float b) {
    if (o.blur) o.blur[at] = b;
    if (o.final_f32 || o.u8) {
        const float f = fminf(fmaxf(bg[at] + disk[at] + b, 0.0f), 1.0f);
        if (o.final_f32) o.final_f32[at] = f;
        if (o.u8) o.u8[at] = (uint8_t)(int)(f * 255.0f);
    }
}

// V pass on the f32 matrix cores.  The vertical blur of a 32-column strip is a banded Toeplitz product  out(y, x) = sum_i
// T(y, i) in(i, x),  T(y, i) = w[|i - y|], and v_mfma_f32_32x32x2_f32 takes exactly that shape: D(32 y x 32 x) += A(32 y x 2 i)
// B(2 i x 32 x), exact f32 (bit for bit a k-ordered fmaf chain) at the f32 vector peak with operands that cost almost nothing:
//   B  lane l holds in(i0 + (l >> 5), x0 + (l & 31)): two 128-byte row segments of the planar H-blur buffer;
//   A  lane l holds w[|i0 + (l >> 5) - y - (l & 31)|]: one ds_read_b32 from a zero-padded table.
// Accumulation order per output = ascending input row, whatever the tiling.  One 32-row tile per wave, 4 waves per SIMD.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void bloom_v_f32_kernel(
    const float *__restrict__ hblur, const float *__restrict__ bg, const float *__restrict__ disk, VOut out,
    const float *__restrict__ wtab, const float *__restrict__ wsum_v, int W, int H, int row0, int rows, int R, int with_bloom,
    unsigned long long *__restrict__ zero_cell, int row_begin, int row_end) {
    // housekeeping folded into the frame's last kernel: clear the ray-step counter cell of a LATER timed frame, which saves
    // a fill dispatch (and its barrier) in front of every march
    if (zero_cell && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < BHR_STEP_LANES)
        zero_cell[(size_t)threadIdx.x * BHR_STEP_STRIDE] = 0ull;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // wl[3][NW]: w[|d|] at index d + OFF, zero beyond R
    const int OFF = R + 32 + 2, NW = 2 * R + 64 + 44;             // every (pair, tile) offset lands inside: no band test in the loop
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
    const int y0 = row_begin + blockIdx.y * 32;                   // local row of its first output
    if (xs >= W) return;
    const int x = xs + (lane & 31), kh = lane >> 5;
    const bool x_ok = x < W;

    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    if (with_bloom) {
        const size_t plane = (size_t)(rows + 2 * R) * W;
        const int Rp = R + (R & 1);
        const int i_first = y0 - Rp, i_last = y0 + 31 + R;                   // input rows (local), walked in pairs
        const int lane_off = OFF + kh - (lane & 31);
        auto load_b = [&](int i0, float (&b)[3]) {
            const int yl = i0 + kh, yg = yl + row0;
            const bool ok = x_ok && yg >= 0 && yg < H && yl >= -R && yl < rows + R;   // outside the image: a skipped tap
            const size_t o = ok ? (size_t)(yl + R) * W + x : 0;
            b[0] = ok ? hblur[o] : 0.0f;
            b[1] = ok ? hblur[plane + o] : 0.0f;
            b[2] = ok ? hblur[2 * plane + o] : 0.0f;
        };
        // PF row pairs per group: the next group's 3 PF loads are in flight under this group's 3 PF MFMAs (64 cycles each)
        constexpr int PF = 4;
        float b[PF][3], bn[PF][3];
#pragma unroll
        for (int q = 0; q < PF; ++q) load_b(i_first + 2 * q, b[q]);
        for (int i0 = i_first; i0 <= i_last; i0 += 2 * PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) load_b(i0 + 2 * (PF + q), bn[q]);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                // a pair outside the tile's band (or past the last input row) reads zeros from the padded table: cheaper
                // than a branch, whose merge copies the 48 accumulator registers
                const int idx = lane_off + (i0 + 2 * q - y0);
                const float a0 = lds[idx], a1 = lds[NW + idx], a2 = lds[2 * NW + idx];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[q][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[q][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b[q][2], acc[2], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) { b[q][0] = bn[q][0]; b[q][1] = bn[q][1]; b[q][2] = bn[q][2]; }
        }
    }
    if (!x_ok) return;
    // D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int yl = y0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (yl >= row_end) continue;
        const int yg = yl + row0;
        const size_t o = ((size_t)yl * W + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) combine_store(out, bg, disk, o + c, with_bloom ? acc[c][r] / wsum_v[c * H + yg] : 0.0f);
    }
}

// ==== split-f16 kernels (fast / hybrid arithmetic) ========================================================================
// Packed operand layouts (halfs; P = 2 halves per value, hi then lo; `rows` = the context's rows, row0 its first global row):
//   pa  H-pass input, written by the march kernel (or bloom_pack_kernel)
//       [c][p][yb = y / 32][g = x / 8 + g0][y % 32][x % 8],  g0 = 2 NT - 2 zero groups in front, GP groups per row block
//   pb  H-pass output = V-pass input, rows in PLANE coordinates pr = global row - pbr, pbr = 32 floor(row0 / 32) - 16 (NT - 1)
//       [c][p][strip = x / 32][gr = pr / 8][x % 32][pr % 8],  n_tx strips (WP = 32 n_tx >= W), GR groups per strip; rows outside the
//       image stay zero.  A V-pass wave (one strip) streams contiguous memory, 1 KB per chunk; an H-pass tile (one strip, four
//       groups) stores 2 KB contiguous per half.
// In both, chunk' k (16 taps) is the pair of groups 2 k, 2 k + 1 and tile' t (32 outputs) lives on chunks [2 t, 2 t + 2 NT - 1]
// with Toeplitz offset delta = 16 (chunk' - 2 t - NT + 1).  (H: tile' = x / 32, chunk' = (x + 16 (NT - 1)) / 16.)
struct Mirror {                // a neighbouring row block's pb planes, written by this block's H pass where they hold its rows
    _Float16 *pb;
    int32_t pbr, pend, gr;     // its plane: first / past-the-last global row, groups per (channel, half)
};
constexpr int MAX_MIRRORS = 6;
#ifndef BHR_SPLIT_SUBS
#define BHR_SPLIT_SUBS 2
#endif
#ifndef BHR_SPLIT_DEPTH
#define BHR_SPLIT_DEPTH 4
#endif
constexpr int SPLIT_SUBS = BHR_SPLIT_SUBS;                     // strips (V) / 32-row blocks (H) per workgroup, three channel waves each
constexpr int SPLIT_THREADS = 192 * SPLIT_SUBS;
constexpr int SPLIT_DEPTH = BHR_SPLIT_DEPTH;                    // 16-tap chunks in flight per wave
struct HSplitArgs {
    const _Float16 *pa;
    _Float16 *pb;
    const unsigned short *w16;
    const float *wsum_h;       // (3, W)
    int32_t W, WP, rows, row0;
    int32_t YB, GP, GR, pbr;
    int32_t NT, table_bytes, n_tx;
    int32_t seg, n_seg;        // output tiles per wave (<= T), segments per row of tiles
    int32_t n_mirror;
    Mirror mirror[MAX_MIRRORS];
};
struct VSplitArgs {
    const _Float16 *pb;
    const unsigned short *w16;
    const float *wsum_v;       // (3, H)
    const float *sum;          // bg + disk of the frame: written by its march beside the packed disk layer, else by bloom_sum_kernel
    VOut out;
    unsigned long long *zero_cell;
    int32_t W, WP, H, row0;
    int32_t GR, t_first;       // t_first = floor(row0 / 32): global tile of tile' 0
    int32_t NT, table_bytes, R;
    int32_t seg_t0, seg_t1;    // tiles' of this launch
    int32_t seg, n_seg;        // output tiles per wave (<= T), segments per strip
    int32_t r_begin, r_end;    // local rows it stores
};

// the context's table (<= 60 KB) into LDS: every load of a thread issued before the first is waited for -- written as a
// loop with a run-time trip count hipcc waits for each 16-byte load before it issues the next one (8-10 trips to L2 in a
// row at the head of every workgroup: 10 us of the first version's 26 us per wave)
__device__ __forceinline__ void stage_table(unsigned char *lds, const unsigned short *__restrict__ w16, int bytes) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(w16);
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    constexpr int TRIPS = (48 * 1280 / 16 + SPLIT_THREADS - 1) / SPLIT_THREADS;      // the largest table: NT = 12
    const int n16 = bytes / 16;
    u32x4 tmp[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int k = threadIdx.x + i * SPLIT_THREADS;
        if (k < n16) tmp[i] = src[k];
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int k = threadIdx.x + i * SPLIT_THREADS;
        if (k < n16) dst[k] = tmp[i];
    }
}

// this lane's window into its channel's table for Toeplitz offset delta = 16 (k - NT + 1): byte address of half 0 at k = 0
__device__ __forceinline__ const unsigned char *weight_base(const unsigned char *lds, int ch, int NT, int n, int h) {
    const int start = split_off(NT) - 16 * (NT - 1) + 8 * h - n;      // >= 1
    const int ph = start & 7, CSB = split_csb(NT);
    return lds + ((ch * 2) * 8 + ph) * CSB + 16 * split_slot(ph) + (start - ph) * 2;
}

#define BHR_F16X8(V) __builtin_bit_cast(f16x8, V)
// acc += (data x weights) over the chunk's 16 taps: the three products that matter, smallest first
#define BHR_MFMA3_DATA_A(ACC, DH, DL, WH, WL)                                                            \
    do {                                                                                                 \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(BHR_F16X8(DL), BHR_F16X8(WH), ACC, 0, 0, 0);       \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(BHR_F16X8(DH), BHR_F16X8(WL), ACC, 0, 0, 0);       \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(BHR_F16X8(DH), BHR_F16X8(WH), ACC, 0, 0, 0);       \
    } while (0)
#define BHR_MFMA3_DATA_B(ACC, DH, DL, WH, WL)                                                            \
    do {                                                                                                 \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(BHR_F16X8(WH), BHR_F16X8(DL), ACC, 0, 0, 0);       \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(BHR_F16X8(WL), BHR_F16X8(DH), ACC, 0, 0, 0);       \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(BHR_F16X8(WH), BHR_F16X8(DH), ACC, 0, 0, 0);       \
    } while (0)

__global__ void bloom_sum_kernel(const float *__restrict__ bg, const float *__restrict__ disk, float *__restrict__ sum, long long n) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) sum[k] = __fadd_rn(bg[k], disk[k]);
}

// (rows, W, 3) f32 disk layer -> pa, for frames whose disk layer did not come from this library's march (bhr_bloom on
// written layers).  One thread per (row, 8-pixel group).
__global__ void bloom_pack_kernel(const float *__restrict__ disk, const float *__restrict__ bg, float *__restrict__ sum, _Float16 *__restrict__ pa, int W, int rows,
                                  int YB, int GP, int g0) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (g * 8 >= W) return;
    const size_t part = (size_t)YB * GP * 256;
    for (int c = 0; c < 3; ++c) {
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = g * 8 + j;
            _Float16 a, b;
            // the march's disk layer is clamp(accum, 0, 1) (render.py:3018); a caller's own layer is held to what the scaled f16
            // halves can carry (3.99) rather than turned into infinities
            cut2(x < W ? fminf(fmaxf(disk[((size_t)y * W + x) * 3 + c], 0.0f), 3.99f) * PIX_SCALE : 0.0f, a, b);
            hi[j] = a;
            lo[j] = b;
        }
        const size_t at = ((((size_t)(c * 2) * YB + (y >> 5)) * GP + g + g0) * 32 + (y & 31)) * 8;
        *reinterpret_cast<f16x8 *>(pa + at) = hi;
        *reinterpret_cast<f16x8 *>(pa + at + part) = lo;
    }
    for (int k = 0; k < 24 && (g * 8) * 3 + k < W * 3; ++k) {          // bg + disk: what the march writes beside its packed copy
        const size_t at = ((size_t)y * W + g * 8) * 3 + k;
        sum[at] = __fadd_rn(bg[at], disk[at]);
    }
}

// H pass.  Work unit = (32-row block yb, segment of `seg` <= T output tiles along x); a workgroup = 6 waves = the three
// channels of two consecutive units, sharing one copy of the weight table; grid ceil(units / 2).
template <int T>
__global__ __launch_bounds__(SPLIT_THREADS) __attribute__((amdgpu_waves_per_eu(T == 8 ? 2 : 3, T == 8 ? 2 : 3))) void bloom_h_split_kernel(HSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = wave % 3, unit = blockIdx.x * SPLIT_SUBS + wave / 3;
    if (unit >= a.n_seg * a.YB) {                    // no work: only the workgroup's table staging
        stage_table(lds_b, a.w16, a.table_bytes);
        __syncthreads();
        return;
    }
    const int yb = unit / a.n_seg, n = lane & 31, h = lane >> 5, NT = a.NT;
    const int tb = (unit - yb * a.n_seg) * a.seg, te = min(tb + a.seg, a.n_tx);
    const int part_w = 8 * split_csb(NT);
    // tile i of the wave at chunk' cp reads the window k = cp - 2 (tb + i): one address per chunk, immediate offsets per tile
    const unsigned char *wl = weight_base(lds_b, ch, NT, n, h) - 64 * (T - 1);
    const size_t part_a = (size_t)a.YB * a.GP * 256;
    const _Float16 *src = a.pa + ((((size_t)(ch * 2) * a.YB + yb) * a.GP + h) * 32 + n) * 8;      // + 512 halfs per chunk'

    f32x16 acc[T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int c0 = 2 * tb, c1 = 2 * (te - 1) + 2 * NT - 1;
    auto load = [&](int cp, u32x4 (&d)[2]) {
        const _Float16 *q = src + (size_t)min(cp, c1) * 512;          // prefetch past the end: a re-read
        d[0] = *reinterpret_cast<const u32x4 *>(q);
        d[1] = *reinterpret_cast<const u32x4 *>(q + part_a);
    };
    auto chunk = [&](int cp, const u32x4 (&d)[2]) {
        const int k0 = cp - c0;
        const unsigned char *wq = wl + 32 * k0;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int k = k0 - 2 * i;
            if (k >= 0 && k < 2 * NT && tb + i < te) {                 // wave uniform: this chunk lies in tile i's band
                const u32x4 wh = *reinterpret_cast<const u32x4 *>(wq + 64 * (T - 1 - i));
                const u32x4 wlo = *reinterpret_cast<const u32x4 *>(wq + 64 * (T - 1 - i) + part_w);
                BHR_MFMA3_DATA_A(acc[i], d[0], d[1], wh, wlo);
            }
        }
    };
    // SPLIT_DEPTH chunks in flight per wave: a chunk's MFMAs (<= 3 T x 32 cycles) are far shorter than a trip to memory.  The
    // first ones are on their way before the workgroup stages its weight table (neither waits for the other)
    u32x4 d[SPLIT_DEPTH][2];
#pragma unroll
    for (int j = 0; j < SPLIT_DEPTH; ++j) load(c0 + j, d[j]);
    stage_table(lds_b, a.w16, a.table_bytes);
    __syncthreads();
    for (int cp = c0; cp <= c1; cp += SPLIT_DEPTH) {
#pragma unroll
        for (int j = 0; j < SPLIT_DEPTH; ++j) {
            if (cp + j <= c1) chunk(cp + j, d[j]);
            load(cp + j + SPLIT_DEPTH, d[j]);
        }
    }

    // D: column = lane & 31 -> output pixel, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> row of the block: a lane's four
    // registers of quad q are rows 8 q + 4 h .. + 3 of its column -- half a pb group (8 rows = 16 bytes per half).  Lanes l and
    // l + 32 hold the two halves of the same groups: one v_permlane32_swap per register pair hands the lower lanes all of
    // group q and the upper lanes all of group q + 1, so every lane stores 16 bytes (1 KB contiguous per instruction; 8-byte
    // stores were store-issue bound: 31 of the 130 us of an 8k row block's post-pass).
    const bool aligned8 = ((a.row0 - a.pbr) & 7) == 0;
    // 2^-10 / (in-bounds weight sum) of every tile's column, all T loads in one batch: loaded tile by tile, each was a trip to
    // L2 behind an s_waitcnt vmcnt(0) that also waited for the previous tile's stores to land (stores count in vmcnt on gfx9)
    float wsv[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int x = 32 * (tb + i) + n;
        wsv[i] = (tb + i < te && x < a.W) ? a.wsum_h[(3 + ch) * a.W + x] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < T; ++i) asm volatile("" : "+v"(wsv[i]));          // all of them HERE: no load is left to wait for between the tiles' stores
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (tb + i >= te) continue;
        const int x = 32 * (tb + i) + n;
        const float ws = wsv[i];                                           // a sum back to a scaled pixel
        unsigned int ph[4][2], pl[4][2];                                   // [quad][dword]: 4 rows x f16, hi and lo halves
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            _Float16 hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cut2(acc[i][4 * q + j] * ws, hi[j], lo[j]);
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                ph[q][d] = __builtin_bit_cast(unsigned int, (f16x2){hi[2 * d], hi[2 * d + 1]});
                pl[q][d] = __builtin_bit_cast(unsigned int, (f16x2){lo[2 * d], lo[2 * d + 1]});
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q += 2)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto r = __builtin_amdgcn_permlane32_swap(ph[q][d], ph[q + 1][d], false, false);
                ph[q][d] = r[0]; ph[q + 1][d] = r[1];
                r = __builtin_amdgcn_permlane32_swap(pl[q][d], pl[q + 1][d], false, false);
                pl[q][d] = r[0]; pl[q + 1][d] = r[1];
            }
        if (x >= a.W) continue;
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
            // this lane's group: rows 8 (q + h) .. + 7 of the block, halves {ph[q], ph[q + 1]} (rows 0-3 from the lower lane, 4-7 from the upper)
            const int yq = 32 * yb + 8 * (q + h);                      // local row of the group's first row
            if (yq >= a.rows) continue;
            const u32x4 vh = {ph[q][0], ph[q][1], ph[q + 1][0], ph[q + 1][1]}, vl = {pl[q][0], pl[q][1], pl[q + 1][0], pl[q + 1][1]};
            const int gq = a.row0 + yq;                                // global row
            const bool whole = aligned8 && yq + 7 < a.rows;
            auto put = [&](_Float16 *pb, int pbr, int pend, int gr) {
                const size_t part = (size_t)a.n_tx * gr * 256;
                _Float16 *base = pb + ((((size_t)(ch * 2) * a.n_tx + (tb + i)) * gr) * 32 + n) * 8;      // + 256 halfs per group
                if (whole) {
                    if (gq < pbr || gq + 7 >= pend) return;
                    _Float16 *dst = base + (size_t)((gq - pbr) >> 3) * 256;
                    *reinterpret_cast<u32x4 *>(dst) = vh;
                    *reinterpret_cast<u32x4 *>(dst + part) = vl;
                } else {
                    const f16x8 eh = BHR_F16X8(vh), el = BHR_F16X8(vl);
                    for (int j = 0; j < 8; ++j) {
                        const int g = gq + j;
                        if (yq + j >= a.rows || g < pbr || g >= pend) continue;
                        const int pr = g - pbr;
                        _Float16 *dst = base + (size_t)(pr >> 3) * 256 + (pr & 7);
                        dst[0] = eh[j];
                        dst[part] = el[j];
                    }
                }
            };
            put(a.pb, a.pbr, a.pbr + 8 * a.GR, a.GR);
            for (int m = 0; m < a.n_mirror; ++m) put(a.mirror[m].pb, a.mirror[m].pbr, a.mirror[m].pend, a.mirror[m].gr);
        }
    }
}

// V pass + combine.  Work unit = (segment of `seg` <= T stacked output tiles, 32-column strip), strips fastest; a workgroup
// = 6 waves = the three channels of two consecutive units; grid ceil(units / 2).
template <int T>
__global__ __launch_bounds__(SPLIT_THREADS) __attribute__((amdgpu_waves_per_eu(T == 8 ? 2 : 3, T == 8 ? 2 : 3))) void bloom_v_split_kernel(VSplitArgs a) {
    if (a.zero_cell && blockIdx.x == 0 && threadIdx.x < BHR_STEP_LANES)
        a.zero_cell[(size_t)threadIdx.x * BHR_STEP_STRIDE] = 0ull;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = wave % 3, sub = wave / 3, n_strips = a.WP / 32;
    const int wg = blockIdx.x;
    int unit = wg * SPLIT_SUBS + sub;                 // strip-major: a strip's segments are neighbours (they share 2 NT - 2 chunks)
    const bool live = unit < a.n_seg * n_strips;     // a unit past the end walks unit 0 and stores nothing: every wave reaches the barriers below
    if (!live) unit = 0;
    const int strip = unit / a.n_seg, yseg = unit - strip * a.n_seg;
    const int n = lane & 31, h = lane >> 5, NT = a.NT;
    const int x = strip * 32 + n;
    const int tb = a.seg_t0 + yseg * a.seg, te = min(tb + a.seg, a.seg_t1);
    const int part_w = 8 * split_csb(NT);
    const unsigned char *wl = weight_base(lds_b, ch, NT, n, h) - 64 * (T - 1);
    const size_t part_b = (size_t)(a.WP / 32) * a.GR * 256, chunk_b = 512;                  // halfs: the strip's groups are contiguous
    const _Float16 *src = a.pb + ((((size_t)(ch * 2) * (a.WP / 32) + strip) * a.GR + h) * 32 + n) * 8;

    f32x16 acc[T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int c0 = 2 * tb, c1 = 2 * (te - 1) + 2 * NT - 1;
    auto load = [&](int cp, u32x4 (&d)[2]) {
        const _Float16 *q = src + (size_t)min(cp, c1) * chunk_b;
        d[0] = *reinterpret_cast<const u32x4 *>(q);
        d[1] = *reinterpret_cast<const u32x4 *>(q + part_b);
    };
    auto chunk = [&](int cp, const u32x4 (&d)[2]) {
        const int k0 = cp - c0;
        const unsigned char *wq = wl + 32 * k0;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int k = k0 - 2 * i;
            if (k >= 0 && k < 2 * NT && tb + i < te) {
                const u32x4 wh = *reinterpret_cast<const u32x4 *>(wq + 64 * (T - 1 - i));
                const u32x4 wlo = *reinterpret_cast<const u32x4 *>(wq + 64 * (T - 1 - i) + part_w);
                BHR_MFMA3_DATA_B(acc[i], d[0], d[1], wh, wlo);
            }
        }
    };
    u32x4 d[SPLIT_DEPTH][2];
#pragma unroll
    for (int j = 0; j < SPLIT_DEPTH; ++j) load(c0 + j, d[j]);
    stage_table(lds_b, a.w16, a.table_bytes);
    __syncthreads();
    for (int cp = c0; cp <= c1; cp += SPLIT_DEPTH) {
#pragma unroll
        for (int j = 0; j < SPLIT_DEPTH; ++j) {
            if (cp + j <= c1) chunk(cp + j, d[j]);
            load(cp + j + SPLIT_DEPTH, d[j]);
        }
    }

    const float *__restrict__ winv = a.wsum_v + (size_t)(3 + ch) * a.H;    // 2^-24 / (in-bounds weight sum) of every image row
    const float winv_mid = winv[a.H >> 1];
    // Epilogue.  The accumulators hold one channel of a tile with the column on the lane: stored from here, a wave would touch
    // 4 bytes of every 12 of the interleaved (rows, W, 3) layers, three waves one after the other, and single bytes of the u8
    // rows (measured: 80 of the 110 us of an 8k row block's V pass, 50 of fhd's 57).  Instead the three channel waves of a
    // strip put their tile into LDS as [row][column][channel] -- the order of the layers -- and share its 32 x 96 floats as
    // 768 float4s: 16-byte loads of bg and disk, 16-byte stores of the f32 frame and the blur, 4-byte stores of the u8
    // rows, every instruction on whole 384-byte row segments.  (Widths that are not multiples of 4 and the partial strip at the
    // right edge keep the per-channel path.)
    // The workgroup's units may be segments of different lengths (the last one of a strip): the barriers below are passed by
    // all six waves `n_it` times, the longest of them; a unit works in the rounds it has a tile for.
    const bool coop = (a.W & 3) == 0;                                   // uniform over the launch
    const bool whole = (strip + 1) * 32 <= a.W;                         // this unit's strip is 32 full columns
    int n_it = 0;
#pragma unroll
    for (int q = 0; q < SPLIT_SUBS; ++q) {
        const int u = wg * SPLIT_SUBS + q;
        if (u < a.n_seg * n_strips) {
            const int b0 = a.seg_t0 + (u % a.n_seg) * a.seg;
            n_it = max(n_it, min(b0 + a.seg, a.seg_t1) - b0);
        }
    }
    float *tile = reinterpret_cast<float *>(lds_b + a.table_bytes) + sub * (32 * 96);
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (i >= n_it) break;
        const bool on = live && tb + i < te;
        const int yg0 = 32 * (a.t_first + tb + i);
        if (!coop || !whole) {
            if (on && x < a.W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int yg = yg0 + (r & 3) + 8 * (r >> 2) + 4 * h, yl = yg - a.row0;
                if (yl < a.r_begin || yl >= a.r_end) continue;
                const size_t at = ((size_t)yl * a.W + x) * 3 + ch;
                const float b = acc[i][r] * winv[yg];
                if (a.out.blur) a.out.blur[at] = b;
                if (a.out.final_f32 || a.out.u8) {
                    const float f = fminf(fmaxf(__fadd_rn(a.sum[at], b), 0.0f), 1.0f);
                    if (a.out.final_f32) a.out.final_f32[at] = f;
                    if (a.out.u8) a.out.u8[at] = (uint8_t)(int)(f * 255.0f);
                }
            }
            }
            if (!coop) continue;
        }
        const bool mine = on && whole;
        // rows whose whole +-R window lies inside the image share one weight sum (the table kernel adds the same weights in
        // the same order for each): 16 loads per tile -- and the s_waitcnt vmcnt(0) in front of their use, which also waits
        // for the previous tile's stores -- only for the tiles within R of the image's top and bottom
        if (!mine) {
        } else if (yg0 >= a.R && yg0 + 31 + a.R < a.H) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                tile[(row * 32 + n) * 3 + ch] = acc[i][r] * winv_mid;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                tile[(row * 32 + n) * 3 + ch] = acc[i][r] * winv[min(yg0 + row, a.H - 1)];
            }
        }
        // this lane's four float4s of the frame's bg + disk: on their way before the barrier, not behind it (every wave of the
        // workgroup used to stop at the barrier and THEN start a trip to memory, once per tile)
        float4 sv[4];
        const bool combine = mine && (a.out.final_f32 || a.out.u8);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 192 + ch * 64 + lane;                  // float4 index into the tile: 24 per row
            const int row = idx / 24, q4 = idx - row * 24;
            const int yl = yg0 + row - a.row0;
            if (T <= 5 && combine && yl >= a.r_begin && yl < a.r_end) sv[it] = *reinterpret_cast<const float4 *>(a.sum + ((size_t)yl * a.W + strip * 32) * 3 + 4 * q4);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 192 + ch * 64 + lane;
            const int row = idx / 24, q4 = idx - row * 24;
            const int yl = yg0 + row - a.row0;
            if (!mine || yl < a.r_begin || yl >= a.r_end) continue;
            const float4 b = *reinterpret_cast<const float4 *>(tile + row * 96 + 4 * q4);
            const size_t at = ((size_t)yl * a.W + strip * 32) * 3 + 4 * q4;
            if (a.out.blur) *reinterpret_cast<float4 *>(a.out.blur + at) = b;
            if (combine) {
                const float4 s = T <= 5 ? sv[it] : *reinterpret_cast<const float4 *>(a.sum + at);   // (T = 8 has no registers to spare: 2 waves per SIMD)
                float4 f;
                f.x = fminf(fmaxf(__fadd_rn(s.x, b.x), 0.0f), 1.0f);
                f.y = fminf(fmaxf(__fadd_rn(s.y, b.y), 0.0f), 1.0f);
                f.z = fminf(fmaxf(__fadd_rn(s.z, b.z), 0.0f), 1.0f);
                f.w = fminf(fmaxf(__fadd_rn(s.w, b.w), 0.0f), 1.0f);
                if (a.out.final_f32) *reinterpret_cast<float4 *>(a.out.final_f32 + at) = f;
                if (a.out.u8)
                    *reinterpret_cast<unsigned int *>(a.out.u8 + at) = (unsigned int)(int)(f.x * 255.0f) | ((unsigned int)(int)(f.y * 255.0f) << 8) |
                                                                       ((unsigned int)(int)(f.z * 255.0f) << 16) | ((unsigned int)(int)(f.w * 255.0f) << 24);
            }
        }
        __syncthreads();                                                // the tile buffer is free for the next one
    }
}

// kernels that need more than 48 KB of dynamic LDS are told so once (a function attribute belongs to the device it was set
// on: the note is kept per (device, kernel) -- row-block tiles on the eight devices of a node launch the same kernels from
// one process)
int32_t allow_lds(const void *fn, size_t bytes) {
    constexpr int CAP = 64;
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

// How a launch is cut: `n_tiles` output tiles per row (H) / column (V), `others` such rows / columns, three channel waves per
// unit, two units per workgroup.  Two instantiations: T = 5 tiles per wave at 3 waves per SIMD (two workgroups per CU: rounds
// of 512 workgroups) and T = 8 at 2 (one per CU: rounds of 256).  A round lasts about as long as its longest wave, and a
// wave of s tiles costs ~ s tiles + (2 s + 2 NT) chunk loads + a fixed part (table staging, barriers): the (T, segment count)
// with the cheapest rounds x wave.  (Measured first with fixed tile counts: an 8k row block's 540-workgroup passes took
// THREE rounds of one-per-CU workgroups.)
struct SplitPlan { int T, n_seg, seg; };
SplitPlan plan_segments(int n_tiles, long long others, int NT, int forced_tiles) {
    if (forced_tiles > 0) {
        const int sf = min(min(forced_tiles, 8), n_tiles);
        return SplitPlan{sf > 5 ? 8 : 5, (n_tiles + sf - 1) / sf, sf};
    }
    // a launch that fits ONE round of workgroup slots even at five tiles per wave: five it is -- such a launch is not short of
    // slots, and fewer, longer waves stage the weight table and re-read the bands' overlap less often (fhd, both passes:
    // 0.0372 against 0.0406 ms with the four / three tiles per wave the round model below prefers; `bloom_tiles` sweep)
    {
        const int n_seg5 = (n_tiles + 4) / 5, sg5 = (n_tiles + n_seg5 - 1) / n_seg5;
        const long long groups5 = (others * ((n_tiles + sg5 - 1) / sg5) + SPLIT_SUBS - 1) / SPLIT_SUBS;
        if (groups5 <= 512) return SplitPlan{5, (n_tiles + sg5 - 1) / sg5, sg5};
    }
    SplitPlan best{5, n_tiles, 1};
    double best_cost = 1e300;
    long long best_rounds = 1;
    for (int n_seg = (n_tiles + 4) / 5; n_seg <= n_tiles; ++n_seg) {
        const int sg = (n_tiles + n_seg - 1) / n_seg;
        const long long groups = (others * ((n_tiles + sg - 1) / sg) + SPLIT_SUBS - 1) / SPLIT_SUBS;
        const long long rounds = (groups + 511) / 512;
        const double cost = (double)rounds * (1.0 * sg + 0.04 * (2 * sg + 2 * NT) + 1.5);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = SplitPlan{5, (n_tiles + sg - 1) / sg, sg}; best_rounds = rounds; }
    }
    // many rounds: the whole 8k frame.  Eight tiles per wave halve the chunk re-reads (measured 0.75 against 0.81 ms there;
    // the T = 8 instantiation loses everywhere else: 4k 0.21 against 0.18, an 8k row block 0.21 against 0.13).  (Five tiles
    // rather than four in the small one: an 8k row block's 17 stacked tiles cut 5 + 4 + 4 + 4 fill ONE round of workgroups.)
    if (best_rounds >= 6 && n_tiles >= 8) return SplitPlan{8, (n_tiles + 7) / 8, 8};
    return best;
}

}  // namespace

int32_t bhr_split_nt(int32_t R) { return (R + 15) / 16 + 1; }

// geometry of the split-f16 buffers of a context (bhr_internal.h: bhr_split_geom)
void bhr_split_geometry(const bhr_ctx *ctx, bhr_split_geom *g) {
    const int W = ctx->cfg.width, NT = bhr_split_nt(ctx->bloom_R);
    g->NT = NT;
    g->n_tx = (W + 31) / 32;
    g->WP = 32 * g->n_tx;
    g->YB = (ctx->rows + 31) / 32;
    g->GP = 4 * g->n_tx + 4 * NT;
    g->g0 = 2 * NT - 2;
    g->t_first = ctx->cfg.row0 / 32;
    const int t_last = (ctx->cfg.row1 - 1) / 32;
    g->n_ty = t_last - g->t_first + 1;
    g->pbr = 32 * g->t_first - 16 * (NT - 1);
    g->GR = 4 * g->n_ty + 4 * NT;
    g->pa_halfs = (size_t)6 * g->YB * g->GP * 256;
    g->pb_halfs = (size_t)6 * g->GR * g->WP * 8;
    g->table_bytes = split_table_bytes(NT);
}

int32_t bhr_bloom_prepare(bhr_ctx *ctx) {
    if (ctx->bloom_ready) return BHR_OK;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const int R = ctx->bloom_R;
    // render.py:3915: sigma_scale = (width / 640.0) ** 2 in Python floats, passed as f32
    const float sigma_scale = (float)(((double)W / 640.0) * ((double)W / 640.0));
    const int NT = bhr_split_nt(R);
    if (ctx->split_ok && !ctx->d_w16) {
        BHR_HIP(hipMalloc(&ctx->d_w16, (size_t)split_table_bytes(NT)));
        BHR_HIP(hipMemsetAsync(ctx->d_w16, 0, (size_t)split_table_bytes(NT), ctx->stream));
    }
    int n = R + 1 + WPAD;
    n = max(n, 2 * ((R + 3) & ~3) + 8);
    n = max(n, split_nw(NT));
    n = max(n, max(W, H));
    hipLaunchKernelGGL(bloom_tables_kernel, dim3((n + 255) / 256, 3), dim3(256), 0, ctx->stream, ctx->d_wtab, ctx->d_wext,
                       ctx->split_ok ? ctx->d_w16 : nullptr, ctx->d_wsum_h, ctx->d_wsum_v, R, W, H, sigma_scale, NT);
    BHR_HIP(hipGetLastError());
    ctx->bloom_ready = 1;
    return BHR_OK;
}

// (rows, W, 3) disk layer of the context -> its packed H-pass input (frames whose march did not write it)
int32_t bhr_launch_bloom_pack(bhr_ctx *ctx) {
    bhr_split_geom g;
    bhr_split_geometry(ctx, &g);
    const int groups = (ctx->cfg.width + 7) / 8;
    hipLaunchKernelGGL(bloom_pack_kernel, dim3((groups + 63) / 64, ctx->rows), dim3(64), 0, ctx->stream, ctx->d_disk,
                       ctx->d_bg, ctx->d_sum, (_Float16 *)ctx->d_pa, ctx->cfg.width, ctx->rows, g.YB, g.GP, g.g0);
    BHR_HIP(hipGetLastError());
    ctx->slots[ctx->active_slot].sum_valid = 1;
    return BHR_OK;
}

// H pass over the context's rows.  Split frames: also into the planes of the row blocks in ctx->mirrors (group.hip).
int32_t bhr_launch_bloom_h(bhr_ctx *ctx) {
    const int W = ctx->cfg.width, R = ctx->bloom_R;
    BHR_TRY(bhr_bloom_prepare(ctx));
    if (ctx->bloom_split) {
        bhr_split_geom g;
        bhr_split_geometry(ctx, &g);
        HSplitArgs a;
        a.pa = (const _Float16 *)ctx->d_pa;
        a.pb = (_Float16 *)ctx->d_pb;
        a.w16 = ctx->d_w16;
        a.wsum_h = ctx->d_wsum_h;
        a.W = W; a.WP = g.WP; a.rows = ctx->rows; a.row0 = ctx->cfg.row0;
        a.YB = g.YB; a.GP = g.GP; a.GR = g.GR; a.pbr = g.pbr;
        a.NT = g.NT; a.table_bytes = g.table_bytes; a.n_tx = g.n_tx;
        a.n_mirror = 0;
        for (int m = 0; m < ctx->n_mirrors && m < MAX_MIRRORS; ++m) {
            a.mirror[a.n_mirror].pb = (_Float16 *)ctx->mirrors[m].pb;
            a.mirror[a.n_mirror].pbr = ctx->mirrors[m].pbr;
            a.mirror[a.n_mirror].pend = ctx->mirrors[m].pbr + 8 * ctx->mirrors[m].gr;
            a.mirror[a.n_mirror].gr = ctx->mirrors[m].gr;
            ++a.n_mirror;
        }
        if (ctx->n_mirrors > MAX_MIRRORS) return bhr_fail(BHR_ERR_INVALID, "bloom H: %d mirror planes (at most %d)", ctx->n_mirrors, MAX_MIRRORS);
        const SplitPlan pl = plan_segments(g.n_tx, g.YB, g.NT, ctx->opt.bloom_tiles);
        a.seg = pl.seg;
        a.n_seg = pl.n_seg;
        dim3 grid(((long long)a.n_seg * g.YB + SPLIT_SUBS - 1) / SPLIT_SUBS), block(SPLIT_THREADS);
        if (pl.T == 8) { BHR_TRY(allow_lds((const void *)bloom_h_split_kernel<8>, g.table_bytes)); hipLaunchKernelGGL(bloom_h_split_kernel<8>, grid, block, g.table_bytes, ctx->stream, a); }
        else { BHR_TRY(allow_lds((const void *)bloom_h_split_kernel<5>, g.table_bytes)); hipLaunchKernelGGL(bloom_h_split_kernel<5>, grid, block, g.table_bytes, ctx->stream, a); }
        BHR_HIP(hipGetLastError());
        return BHR_OK;
    }
    const int R4 = (R + 3) & ~3, pix = HB_PIX * 2;
    dim3 grid((W + pix - 1) / pix, ctx->rows), block(256);
    const size_t lds = ((size_t)3 * (pix + 2 * R4 + 4) + 3 * (2 * R4 + 8)) * sizeof(float);
    BHR_TRY(allow_lds((const void *)bloom_h_f32_kernel, lds));
    hipLaunchKernelGGL(bloom_h_f32_kernel, grid, block, lds, ctx->stream, ctx->d_disk, ctx->d_hblur, ctx->d_wext, ctx->d_wsum_h, W, ctx->rows, R, 0);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

// output rows per V-pass workgroup row: the row-block schedules cut their V chunks on multiples of it
int32_t bhr_bloom_v_tile_rows(bhr_ctx *ctx) { (void)ctx; return 32; }

// V pass + combine over the local rows [r0, r1), storing what `want` names (BHR_OUT_*) into the context's buffers -- or,
// for BHR_OUT_U8 / BHR_OUT_F32 with a non-null gather base, into the (H, W, 3) frame buffer of a row-block gather.
int32_t bhr_launch_bloom_v_rows(bhr_ctx *ctx, int32_t with_bloom, int32_t r0, int32_t r1, uint32_t want, uint8_t *gather_u8, float *gather_f32) {
    const int W = ctx->cfg.width, H = ctx->cfg.height, R = ctx->bloom_R;
    BHR_TRY(bhr_bloom_prepare(ctx));
    if (r0 < 0 || r1 > ctx->rows || r0 > r1) return bhr_fail(BHR_ERR_INVALID, "bloom V: rows [%d,%d) of %d", r0, r1, ctx->rows);
    if (r0 == r1) return BHR_OK;
    VOut out;
    const size_t row0_off = (size_t)ctx->cfg.row0 * W * 3;
    out.final_f32 = (want & BHR_OUT_F32) ? (gather_f32 ? gather_f32 + row0_off : ctx->d_final) : nullptr;
    out.blur = (want & BHR_OUT_BLUR) ? ctx->d_blur : nullptr;
    out.u8 = (want & BHR_OUT_U8) ? (gather_u8 ? gather_u8 + row0_off : ctx->d_final_u8) : nullptr;
    if (ctx->bloom_split && with_bloom) {
        bhr_split_geom g;
        bhr_split_geometry(ctx, &g);
        VSplitArgs a;
        a.pb = (const _Float16 *)ctx->d_pb;
        a.w16 = ctx->d_w16;
        a.wsum_v = ctx->d_wsum_v;
        if (!ctx->slots[ctx->active_slot].sum_valid) {          // layers the caller wrote after the march (bhr_write_layer)
            const long long n = (long long)ctx->rows * W * 3;
            hipLaunchKernelGGL(bloom_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_bg, ctx->d_disk, ctx->d_sum, n);
            BHR_HIP(hipGetLastError());
            ctx->slots[ctx->active_slot].sum_valid = 1;
        }
        a.sum = ctx->d_sum;
        a.out = out;
        a.zero_cell = ctx->v_zero_cell;
        a.W = W; a.WP = g.WP; a.H = H; a.row0 = ctx->cfg.row0;
        a.GR = g.GR; a.t_first = g.t_first;
        a.NT = g.NT; a.table_bytes = g.table_bytes; a.R = R;
        a.seg_t0 = (ctx->cfg.row0 + r0) / 32 - g.t_first;
        a.seg_t1 = (ctx->cfg.row0 + r1 - 1) / 32 - g.t_first + 1;
        a.r_begin = r0; a.r_end = r1;
        const int nt = a.seg_t1 - a.seg_t0;
        const SplitPlan pl = plan_segments(nt, g.n_tx, g.NT, ctx->opt.bloom_tiles);
        a.seg = pl.seg;
        a.n_seg = pl.n_seg;
        dim3 grid(((long long)a.n_seg * g.n_tx + SPLIT_SUBS - 1) / SPLIT_SUBS), block(SPLIT_THREADS);
        const size_t lds = (size_t)g.table_bytes + SPLIT_SUBS * 32 * 96 * sizeof(float);     // table + one [32][32][3] tile per strip
        if (pl.T == 8) { BHR_TRY(allow_lds((const void *)bloom_v_split_kernel<8>, lds)); hipLaunchKernelGGL(bloom_v_split_kernel<8>, grid, block, lds, ctx->stream, a); }
        else { BHR_TRY(allow_lds((const void *)bloom_v_split_kernel<5>, lds)); hipLaunchKernelGGL(bloom_v_split_kernel<5>, grid, block, lds, ctx->stream, a); }
        BHR_HIP(hipGetLastError());
        return BHR_OK;
    }
    dim3 grid((W + 127) / 128, (r1 - r0 + 31) / 32), block(256);
    const size_t lds = with_bloom ? (size_t)3 * (2 * R + 64 + 44) * sizeof(float) : 0;
    hipLaunchKernelGGL(bloom_v_f32_kernel, grid, block, lds, ctx->stream, ctx->d_hblur, ctx->d_bg, ctx->d_disk, out, ctx->d_wtab,
                       ctx->d_wsum_v, W, H, ctx->cfg.row0, ctx->rows, R, with_bloom, ctx->v_zero_cell, r0, r1);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}
