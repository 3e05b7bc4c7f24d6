// march.hip -- fused Schwarzschild ray-march kernel for gfx950 (MI355X).
//
// One ray per lane.  Everything the reference does per pixel in
// _ray_march_kernel (render.py:2787-3018) happens inside this one kernel:
// pixel -> ray setup, adaptive-step RK4 on d2x/dl2 = -1.5 L^2 x / r^5, the
// optional variational RK4 pair for ray differentials, capture / escape tests,
// tilted-plane crossing, disk texture or mip-LOD lookup, g-factor shading
// (_apply_g_factor, render.py:2439-2516), front-to-back compositing and the
// skybox lookup of the escape direction.  There is no dense contraction, so no
// MFMA: the kernel is FP32 VALU + transcendental bound (DESIGN.md "Rooflines").
//
// Two schedules share the per-ray code:
//  * tile (default): a wave owns one 8x8 pixel tile; lanes leave the loop as their rays terminate (lane
//                efficiency ~0.95 for the default view).  Tiles are launched nearest-to-the-hole first.
//  * persistent (BHR_PERSISTENT): waves pull 8x8 tiles from a global queue; when the
//                number of live lanes drops below a threshold the dead lanes write
//                their pixel and are refilled from the next tile (wave-level
//                __ballot / popcount compaction of the *work*, not of registers).
//                Slower than the tile schedule for the BASELINE views.
// Disk sources (template parameter SRC, own kernel instantiations): 0 texture / mip stack, 1 Disk V2 mid-plane
// fields at each plane crossing, 2 Disk V2 finite-thickness emission-absorption integral (volume_segment).
// Disk crossings are parked in per-lane LDS slots and shaded wave-wide (Pending, flush_one).
//
// The fast build's arithmetic differs from a strict f32 evaluation of the reference only in rounding:
// v_rsq/v_rcp/v_sqrt instead of IEEE sqrt + divide inside the RK4 stages, FMA contraction, and the
// re-use of |new_pos| as the next step's |pos| (same value in the reference).
//
// This file is compiled twice (csrc/Makefile):
//   march.o         fast arithmetic  -- v_rsq/v_rcp/v_sqrt, FMA contraction, stage values shared
//                                       between the main and the variational right-hand sides;
//   march_strict.o  -DBHR_MARCH_STRICT=1 -ffp-contract=off -- every operation of the RK4 loop in
//                   the reference's order with IEEE sqrt and divide, so that positions, step
//                   counts and hit points are bit-identical to a strict f32 evaluation of
//                   render.py:2854-3006 (selected with bhr_config.math_mode = 1).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "bhr_internal.h"
#include "disk_v2_device.h"

#ifndef BHR_MARCH_STRICT
#define BHR_MARCH_STRICT 0
#endif
#ifndef BHR_MARCH_ILP
#define BHR_MARCH_ILP 0
#endif
#ifndef BHR_WAVE_STAMPS_BUILD
#define BHR_WAVE_STAMPS_BUILD 0
#endif

#if BHR_MARCH_STRICT && BHR_MARCH_ILP
// third compilation: the strict source scheduled with -mllvm -amdgpu-sched-strategy=max-ilp.  Its two texture kernels
// are launched (march_tile_plain_ilp, march_tile_aa_ilp, each with its own occupancy target): the plain one gains 4 %
// from the ILP-first schedule, the AA one 1-2 % once held to 4 waves per SIMD (unconstrained it took 134 VGPRs and
// lost 1.6 %); the fast build loses 3 % and keeps the default scheduler, as do the Disk V2 and persistent kernels.
#define BHR_LAUNCH_MARCH bhr_launch_march_strict_ilp
#define BHR_MARCH_RESOURCES bhr_march_resources_strict_ilp
#elif BHR_MARCH_STRICT
#define BHR_LAUNCH_MARCH bhr_launch_march_strict
#define BHR_MARCH_RESOURCES bhr_march_resources_strict
#else
#define BHR_LAUNCH_MARCH bhr_launch_march
#define BHR_MARCH_RESOURCES bhr_march_resources
#endif

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// s*a + b, component-wise
__device__ __forceinline__ V3 fma3(float s, V3 a, V3 b) {
    return mk(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z));
}
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

__device__ __forceinline__ float q_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float q_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float q_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Correctly rounded x / 6 in TWO operations: 1/6 = c_hi + c_lo up to 2^-50 (c_hi = RN(1/6), c_lo = RN(1/6 - c_hi)),
//   q = RN(x c_hi + RN(x c_lo)).
// The argument of the final rounding is within 2^-48 (relative) of x / 6, and x / 6 is never closer than 1/6 ulp to
// a rounding boundary (6 q = integer significand => the fractional position is a multiple of 1/3 of half an ulp),
// so the rounding is the correct one for every normal x.  tests/test_div6.py checks it against IEEE division on
// every f32 significand, bhr_selftest() on the device.  (Round 1 used the generic 3-operation Markstein sequence.)
__device__ __forceinline__ float div6(float x) {
    const float c_hi = 0x1.555556p-3f, c_lo = -0x1.555556p-28f;
    return fmaf(x, c_hi, x * c_lo);
}

// IEEE-754 correctly rounded sqrt, reciprocal and divide for NORMAL-range operands (no overflow or
// underflow of the result), as Newton/Markstein steps on the hardware approximations:
//   sqrt(x): y = v_rsq(x); s = x y; s' = s + (x - s s)(y/2)            5 instructions
//   1/b    : y = v_rcp(b); y' = y + (1 - b y) y                        3 instructions
//   a/b    : q = a y'; q' = q + (a - b q) y'                           6 instructions
// hipcc's own IEEE sequences take 14 / 11 / 11 (they also handle denormals and overflow, which the
// bounded quantities of the march -- r^2 in [0.5, 1e6], |L2|/r^5, 1/r -- never produce).  The
// residuals are exact thanks to FMA; that the final rounding is the correct one was established
// exhaustively on gfx950 (tools/exact_search.hip: every f32 in [2^-80, 2^80) for sqrt and 1/b, 1.7e10
// random + adversarial pairs for a/b, zero mismatches against sqrtf and operator/), and
// bhr_selftest() repeats the check on the device it runs on.  Saves 64 instructions per RK4 step.
__device__ __forceinline__ float sqrt_rn(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    float r = fmaf(-s, s, x);
    return fmaf(r, 0.5f * y, s);
}
__device__ __forceinline__ float rcp_rn(float b) {
    float y = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, y, 1.0f), y, y);
}
__device__ __forceinline__ float div_rn(float a, float b) {
    float y = rcp_rn(b);
    float q = a * y;
    return fmaf(fmaf(-b, q, a), y, q);
}
// The same sequences from a seed the caller already holds: the march issues the hardware approximations of independent
// operands back to back (a transcendental costs 8 issue cycles behind another one and ~12.7 behind a plain instruction --
// the stream changes pipes), then refines each.  Same operations on the same values as sqrt_rn / rcp_rn / div_rn.
__device__ __forceinline__ float sqrt_rn_s(float x, float y) {
    float s = x * y;
    float r = fmaf(-s, s, x);
    return fmaf(r, 0.5f * y, s);
}
__device__ __forceinline__ float rcp_rn_s(float b, float y) { return fmaf(fmaf(-b, y, 1.0f), y, y); }
__device__ __forceinline__ float div_rn_s(float a, float b, float y0) {
    float y = rcp_rn_s(b, y0);
    float q = a * y;
    return fmaf(fmaf(-b, q, a), y, q);
}
// (s_nop: a transcendental's result needs one wait state before a VALU reads it; hipcc adds it behind its own, not behind an asm)
__device__ __forceinline__ void rsq2(float x, float y, float &a, float &b) {
    asm("v_rsq_f32 %0, %2\n\tv_rsq_f32 %1, %3\n\ts_nop 0" : "=&v"(a), "=&v"(b) : "v"(x), "v"(y));
}
__device__ __forceinline__ void rcp2(float x, float y, float &a, float &b) {
    asm("v_rcp_f32 %0, %2\n\tv_rcp_f32 %1, %3\n\ts_nop 0" : "=&v"(a), "=&v"(b) : "v"(x), "v"(y));
}
__device__ __forceinline__ void rsq_rcp_rcp(float x, float y, float &rs, float &rx, float &ry) {   // rsq(x), rcp(x), rcp(y)
    asm("v_rsq_f32 %0, %3\n\tv_rcp_f32 %1, %3\n\tv_rcp_f32 %2, %4\n\ts_nop 0" : "=&v"(rs), "=&v"(rx), "=&v"(ry) : "v"(x), "v"(y));
}
// x + 0.5 y and x + 2 y: the products are exact, so one FMA rounds exactly like mul-then-add
__device__ __forceinline__ V3 add_half(V3 x, V3 y) { return mk(fmaf(0.5f, y.x, x.x), fmaf(0.5f, y.y, x.y), fmaf(0.5f, y.z, x.z)); }

// bilinear blend in the reference's evaluation order: c00 (1-fu)(1-fv) + c10 fu (1-fv) + c01 (1-fu) fv + c11 fu fv
#if BHR_MARCH_STRICT
#define BHR_BILERP(c00, c10, c01, c11) \
    ((c00) * (1 - fu) * (1 - fv) + (c10) * fu * (1 - fv) + (c01) * (1 - fu) * fv + (c11) * fu * fv)
#else
#define BHR_BILERP(c00, c10, c01, c11) ((c00) * w00 + (c10) * w10 + (c01) * w01 + (c11) * w11)
#endif

// taichi Vector.normalized(): (1/|v|) * v   -- used outside the hot loop, IEEE ops.
__device__ __forceinline__ V3 normalized(V3 v) {
    float inv = 1.0f / sqrtf(dot(v, v));
    return inv * v;
}

__device__ __forceinline__ int pymod(int a, int m) {
    int r = a % m;
    return r < 0 ? r + m : r;
}

// ---- _color_temp_to_tint (render.py:2407-2437) at DISK_COLOR_TEMPERATURE ----
// t = 60 <= 66: r = 1, g = clamp(0.390082 ln 60 - 0.631841), b = clamp(0.543207 ln 50 - 1.19625)
__device__ __forceinline__ V3 disk_tint() {
    const float t = BHR_DISK_COLOR_TEMPERATURE / 100.0f;
    float g = fminf(fmaxf(0.390082f * logf(fmaxf(t, 0.0001f)) - 0.631841f, 0.0f), 1.0f);
    float b = fminf(fmaxf(0.543207f * logf(fmaxf(t - 10.0f, 0.0001f)) - 1.19625f, 0.0f), 1.0f);
    return mk(1.0f, g, b);
}

// ---- _sample_skybox (render.py:2541-2566) ---------------------------------
__device__ __forceinline__ V3 sample_skybox(const BhrScene &sc, V3 d) {
    const int tex_w = sc.sky_w, tex_h = sc.sky_h;
    float theta = acosf(fminf(fmaxf(d.z, -1.0f), 1.0f));
    float phi = atan2f(d.y, d.x);
    if (phi < 0) phi += BHR_TWO_PI_F;
    float u = phi / BHR_TWO_PI_F * (float)tex_w;
    float v = theta / BHR_PI_F * (float)tex_h;
    int u0 = (int)floorf(u);
    int v0 = (int)floorf(v);
    float fu = u - (float)u0;
    float fv = v - (float)v0;
    int u0_w = pymod(u0, tex_w);
    int u1_w = pymod(u0 + 1, tex_w);
    int v0_h = min(max(v0, 0), tex_h - 1);
    int v1_h = min(max(v0 + 1, 0), tex_h - 1);
    const float *c00 = sc.skybox + ((size_t)v0_h * tex_w + u0_w) * 3;
    const float *c10 = sc.skybox + ((size_t)v0_h * tex_w + u1_w) * 3;
    const float *c01 = sc.skybox + ((size_t)v1_h * tex_w + u0_w) * 3;
    const float *c11 = sc.skybox + ((size_t)v1_h * tex_w + u1_w) * 3;
    float w00 = (1 - fu) * (1 - fv), w10 = fu * (1 - fv), w01 = (1 - fu) * fv, w11 = fu * fv;
    (void)w00; (void)w10; (void)w01; (void)w11;
    return mk(BHR_BILERP(c00[0], c10[0], c01[0], c11[0]), BHR_BILERP(c00[1], c10[1], c01[1], c11[1]),
              BHR_BILERP(c00[2], c10[2], c01[2], c11[2]));
}

// ---- _sample_disk / _sample_disk_mip (render.py:2568-2637) -------------------
// lod_i = 0 reproduces _sample_disk exactly (level 0 of the mip stack is the
// texture itself and n / 2^0 = n).
// `staged` (SRC == 3 kernels): the packed levels staged_from .. last of the mip stack, copied into LDS at block start
__device__ __forceinline__ float4 sample_disk_level(const BhrScene &sc, float hit_x, float hit_y, float r_inner,
                                                    float r_outer, float t_offset, int lod_i,
                                                    const float4 *staged = nullptr, int staged_from = 1 << 30) {
    float r = sqrtf(hit_x * hit_x + hit_y * hit_y);
    float phi = atan2f(hit_y, hit_x);
    float r_safe = fmaxf(r, 1e-3f);
    float omega = sqrtf(0.5f / (r_safe * r_safe * r_safe + 1e-6f));
    phi = phi + t_offset * omega;
    while (phi < 0) phi += BHR_TWO_PI_F;
    while (phi >= BHR_TWO_PI_F) phi -= BHR_TWO_PI_F;

    float scale = (float)(1 << lod_i);  // ti.pow(2.0, lod_i), exact
    float tex_w_lod = (float)sc.n_phi / scale;
    float tex_h_lod = (float)sc.n_r / scale;
    float u = phi / BHR_TWO_PI_F * tex_w_lod;
    float v = (r - r_inner) / (r_outer - r_inner) * tex_h_lod;
    int u0 = (int)floorf(u);
    int v0 = (int)floorf(v);
    float fu = u - (float)u0;
    float fv = v - (float)v0;
    int wl = (int)tex_w_lod;
    int u0_w = pymod(u0, wl);
    int u1_w = pymod(u0 + 1, wl);
    int vmax = (int)(tex_h_lod - 1.0f);
    int v0_h = min(max(v0, 0), vmax);
    int v1_h = min(max(v0 + 1, 0), vmax);
    const float4 *t = lod_i >= staged_from ? staged + (sc.mip_off[lod_i] - sc.mip_off[staged_from]) : sc.mips + sc.mip_off[lod_i];
    const int stride = sc.mip_w[lod_i];
    float4 c00 = t[(size_t)v0_h * stride + u0_w];
    float4 c10 = t[(size_t)v0_h * stride + u1_w];
    float4 c01 = t[(size_t)v1_h * stride + u0_w];
    float4 c11 = t[(size_t)v1_h * stride + u1_w];
    float w00 = (1 - fu) * (1 - fv), w10 = fu * (1 - fv), w01 = (1 - fu) * fv, w11 = fu * fv;
    (void)w00; (void)w10; (void)w01; (void)w11;
    return make_float4(BHR_BILERP(c00.x, c10.x, c01.x, c11.x), BHR_BILERP(c00.y, c10.y, c01.y, c11.y),
                       BHR_BILERP(c00.z, c10.z, c01.z, c11.z), BHR_BILERP(c00.w, c10.w, c01.w, c11.w));
}

// ---- _apply_g_factor (render.py:2439-2516) ----------------------------------
__device__ __forceinline__ V3 apply_g_factor(const BhrMarchArgs &a, V3 base_color, V3 hit_pos, float hit_r,
                                             V3 ray_dir_to_cam) {
    const float rs_f = BHR_RS;
    V3 cam_pos = ld3(a.cp);
    // |cam| is the same for every hit, so the compiler hoists it out of the march loop and keeps it in a VGPR for the
    // whole march (the strict AA kernel spilled it at 128 VGPRs).  Shading runs a handful of times per ray: recompute.
    asm volatile("" : "+v"(cam_pos.x));
    float r_obs = sqrtf(dot(cam_pos, cam_pos));
    float r_em = sqrtf(dot(hit_pos, hit_pos));
    float r_safe = fmaxf(r_em, rs_f + 1e-3f);

    float omega = sqrtf(0.5f / (r_safe * r_safe * r_safe + 1e-6f));
    float lorentz = sqrtf(fmaxf(1.0f - rs_f / r_safe, 1e-6f));
    float beta = fminf(r_safe * omega / fmaxf(lorentz, 1e-6f), 0.99f);
    float gamma = 1.0f / sqrtf(fmaxf(1.0f - beta * beta, 1e-6f));

    V3 disk_normal = mk(0.0f, -a.sin_t, a.cos_t);
    V3 r_hat = normalized(hit_pos);
    V3 v_hat = cross(r_hat, disk_normal);
    float v_norm = sqrtf(dot(v_hat, v_hat));
    if (v_norm > 1e-6f) {
        v_hat = mk(v_hat.x / v_norm, v_hat.y / v_norm, v_hat.z / v_norm);
    } else {
        v_hat = mk(0.0f, 1.0f, 0.0f);
    }
    V3 ray_hat = normalized(ray_dir_to_cam);
    float cos_theta = dot(v_hat, ray_hat);
    float denom = fmaxf(1.0f - beta * cos_theta, 1e-3f);
    float g_doppler = 1.0f / (gamma * denom);

    float grav_num = sqrtf(fmaxf(1.0f - rs_f / fmaxf(r_obs, rs_f + 1e-3f), 1e-6f));
    float grav_den = sqrtf(fmaxf(1.0f - rs_f / fmaxf(r_em, rs_f + 1e-3f), 1e-6f));
    float g_grav = grav_num / grav_den;

    float g = fminf(g_doppler * g_grav, BHR_G_FACTOR_CAP);
    float intensity = fmaxf(powf(g, BHR_G_LUMINOSITY_POWER), 0.0f);
    float brightness = BHR_G_BRIGHTNESS_GAIN * intensity / (1.0f + intensity / BHR_G_FACTOR_CAP);

    float radial_span = fmaxf(a.r_outer - a.r_inner, 1e-3f);
    float radial_t = (fmaxf(hit_r, a.r_inner) - a.r_inner) / radial_span;
    radial_t = fminf(fmaxf(radial_t, 0.0f), 1.0f);
    float radial_profile = powf(1.0f - radial_t, BHR_DISK_RADIAL_BRIGHTNESS_POWER);
    float radial_boost = BHR_DISK_RADIAL_BRIGHTNESS_MIN +
                         (BHR_DISK_RADIAL_BRIGHTNESS_MAX - BHR_DISK_RADIAL_BRIGHTNESS_MIN) * radial_profile;
    brightness *= radial_boost;

    // Wien colour shift, normalised to the green channel
    float g_safe = fmaxf(g, 0.1f);
    float wien_arg = 1.0f - 1.0f / g_safe;
    float r_scale = expf(2.21f * wien_arg);
    float g_scale = expf(2.72f * wien_arg);
    float b_scale = expf(3.13f * wien_arg);
    r_scale = fminf(r_scale / g_scale, 3.0f);
    b_scale = fminf(b_scale / g_scale, 3.0f);

    V3 tint = disk_tint();
    V3 out = mk(base_color.x * r_scale * tint.x * brightness, base_color.y * 1.0f * tint.y * brightness,
                base_color.z * b_scale * tint.z * brightness);
    out.x = fminf(fmaxf(out.x, 0.0f), 10.0f);
    out.y = fminf(fmaxf(out.y, 0.0f), 10.0f);
    out.z = fminf(fmaxf(out.z, 0.0f), 10.0f);
    return out;
}

// Analytic disk source (bhr_set_disk_source, BHR_DISK_V2): emission colour and opacity straight from
// the Disk V2 model in binary64 instead of a texture lookup -- temperature T_mid(r) F(r, phi) and
// density rho_mid(r) F(r, phi) with F = F_mode F_shear F_hotspot (disk_v2/physical_fields.py,
// structure_modulations.py), pattern advected with the model's own Omega(r).  The mapping to RGBA is
// the compose kernel's (render.py:3192-3194, 3243-3257): t = clamp(T / T_peak), T_K = T_min + t (T_max -
// T_min), rgb = blackbody(T_K) sqrt(t) with blue <= red, alpha = clamp(rho).  The reference never wired
// disk_v2 into its renderer (docs/design_ad_v2.md Phase 4), so this mapping is this build's choice.
__device__ __forceinline__ V3 disk_v2_color(float tf) {
    const float t_factor = (BHR_DISK_COLOR_TEMPERATURE - 4500.0f) / (6500.0f - 2700.0f);
    const float T_min = 2000.0f + t_factor * 1000.0f, T_max = 9000.0f + t_factor * 3000.0f;
    float tk = (T_min + tf * (T_max - T_min)) / 100.0f;
    float cr = 1.0f, cg, cb = 1.0f;   // _color_temp_to_tint (render.py:2407-2437)
    if (tk > 66.0f) cr = fminf(fmaxf(1.292936f * powf(fmaxf(tk - 60.0f, 0.0001f), -0.1332047592f), 0.0f), 1.0f);
    if (tk <= 66.0f) cg = fminf(fmaxf(0.390082f * logf(fmaxf(tk, 0.0001f)) - 0.631841f, 0.0f), 1.0f);
    else cg = fminf(fmaxf(1.129891f * powf(fmaxf(tk - 60.0f, 0.0001f), -0.0755148492f), 0.0f), 1.0f);
    if (tk < 66.0f) cb = tk <= 19.0f ? 0.0f : fminf(fmaxf(0.543207f * logf(fmaxf(tk - 10.0f, 0.0001f)) - 1.19625f, 0.0f), 1.0f);
    cb = fminf(cb, cr);
    float lum = fminf(fmaxf(sqrtf(tf), 0.0f), 1.0f);
    return mk(fminf(fmaxf(cr * lum, 0.0f), 1.0f), fminf(fmaxf(cg * lum, 0.0f), 1.0f), fminf(fmaxf(cb * lum, 0.0f), 1.0f));
}
__device__ __forceinline__ float4 disk_v2_rgba(const BhrMarchArgs &a, float hit_x, float hit_y) {
    const bhr_disk_v2_params &p = *a.dv2;
    double r = sqrt((double)hit_x * hit_x + (double)hit_y * hit_y);
    double phi = atan2((double)hit_y, (double)hit_x) + (double)a.t_offset * dv2::omega_field(r, p);
    double F = dv2::structure_total(r, phi, p, a.dv2_norm_shear, a.dv2_norm_hotspot);
    double t = fmin(fmax(dv2::t_mid(r, p) * F / a.dv2_t_peak, 0.0), 1.0);
    double rho = fmin(fmax(dv2::rho_mid(r, p) * F, 0.0), 1.0);
    V3 c = disk_v2_color((float)t);
    return make_float4(c.x, c.y, c.z, (float)rho);
}

// Shared by both builds: shade one disk crossing and composite it front to back
// (render.py:2951-3002).  hit_x/hit_y: crossing point; to_cam: -direction at the START of the
// step (render.py:2954); hdx/hdy: x,y components of the hit differentials (DIFF only).
struct Shade {
    V3 accum;
    float alpha_total;
    int unsure;   // DIFF: some crossing's LOD sat within BHR_LOD_GUARD of a truncation boundary (read by the hybrid AA kernel only)
};
// A disk crossing waiting to be shaded.  Crossings of the lanes of a wave are spread over several
// RK4 steps (measured: ~6 wave-steps per tile see a hit, each with a handful of live lanes), and
// shading is ~700 instructions, so a hit is parked and shaded together with the other lanes' hits.
// Every lane has TWO parking slots: as soon as some lane has filled both, the wave shades the older
// hit of every lane that has one (front-to-back order is kept) and the second slot moves up; the rest
// is shaded when the wave has finished marching.  A lane can therefore always park the hit it finds,
// a step never has to be repeated, and the ray state is committed unconditionally.  Results are
// unchanged -- the same operations run later.
template <bool DIFF>
struct Pending {
    float hit_x, hit_y;
    V3 to_cam;
    float dxx, dxy, dyx, dyy;   // DIFF only
};
// The two parking slots of every lane live in LDS (9 x 2 floats per lane, bank-conflict free: consecutive
// lanes, consecutive words): they are touched a handful of times per ray, and in registers they cost the AA
// kernel a wave of occupancy (128 -> 149 VGPRs).
__shared__ float g_park[2][9][256];
extern __shared__ __attribute__((aligned(16))) float4 g_mip_lds[];   // SRC == 3: the coarse mip levels of the disk texture (dynamic)
template <bool DIFF>
__device__ __forceinline__ void park_store(int slot, const Pending<DIFF> &h) {
    const int t = threadIdx.x;
    g_park[slot][0][t] = h.hit_x;
    g_park[slot][1][t] = h.hit_y;
    g_park[slot][2][t] = h.to_cam.x;
    g_park[slot][3][t] = h.to_cam.y;
    g_park[slot][4][t] = h.to_cam.z;
    if (DIFF) {
        g_park[slot][5][t] = h.dxx;
        g_park[slot][6][t] = h.dxy;
        g_park[slot][7][t] = h.dyx;
        g_park[slot][8][t] = h.dyy;
    }
}
template <bool DIFF>
__device__ __forceinline__ Pending<DIFF> park_load(int slot) {
    const int t = threadIdx.x;
    Pending<DIFF> h;
    h.hit_x = g_park[slot][0][t];
    h.hit_y = g_park[slot][1][t];
    h.to_cam = mk(g_park[slot][2][t], g_park[slot][3][t], g_park[slot][4][t]);
    if (DIFF) {
        h.dxx = g_park[slot][5][t];
        h.dxy = g_park[slot][6][t];
        h.dyx = g_park[slot][7][t];
        h.dyy = g_park[slot][8][t];
    } else {
        h.dxx = h.dxy = h.dyx = h.dyy = 0.0f;
    }
    return h;
}
template <bool DIFF, int SRC>
__device__ __forceinline__ void shade_hit(const BhrMarchArgs &a, Shade &sh, float hit_x, float hit_y, V3 to_cam,
                                          float hdx_x, float hdx_y, float hdy_x, float hdy_y) {
    float hit_r = sqrtf(hit_x * hit_x + hit_y * hit_y);
    if (!(a.r_outer >= hit_r && hit_r >= a.r_inner)) return;
    float hit_z = hit_y * a.tan_t;
    int lod_i = 0;
    if (DIFF) {
        // texture-space footprint from the ray differentials (render.py:2964-2988); same evaluation
        // order as the reference in both builds: the LOD is truncated to an integer level
        float hit_r_cyl = sqrtf(hit_x * hit_x + hit_y * hit_y + 1e-6f);
        float den = hit_r_cyl * hit_r_cyl + 1e-6f;
        float w_f = (float)a.sc.n_phi, h_f = (float)a.sc.n_r, span = a.r_outer - a.r_inner;
        float dr_dx = (hit_x * hdx_x + hit_y * hdx_y) / hit_r_cyl;
        float dphi_dx = (-hit_y * hdx_x + hit_x * hdx_y) / den;
        float dudx = dphi_dx * w_f / (2.0f * BHR_PI_F), dvdx = dr_dx * h_f / span;
        float dr_dy = (hit_x * hdy_x + hit_y * hdy_y) / hit_r_cyl;
        float dphi_dy = (-hit_y * hdy_x + hit_x * hdy_y) / den;
        float dudy = dphi_dy * w_f / (2.0f * BHR_PI_F), dvdy = dr_dy * h_f / span;
        float grad_sq = fmaxf(dudx * dudx + dvdx * dvdx, dudy * dudy + dvdy * dvdy);
        float lod = logf(fmaxf(grad_sq, 1.0f)) / logf(2.0f) * a.aa_strength;
        {
            // the level is int(clamp(lod, 0, 3)): it jumps at lod = 1, 2, 3.  A crossing whose lod lies within the guard
            // band of a jump may pick another level under a different rounding of the differentials
            const float fr = lod - floorf(lod);
            if (lod > 0.5f && lod < 3.5f && (fr < BHR_LOD_GUARD || fr > 1.0f - BHR_LOD_GUARD)) sh.unsure = 1;
        }
        lod = fminf(fmaxf(lod, 0.0f), 3.0f);
        lod_i = (int)fminf(fmaxf(lod, 0.0f), (float)(BHR_NUM_MIP_LEVELS - 1));
    }
    // SRC == 1 is a separate kernel instantiation: the binary64 model code (and its registers) never
    // touches the texture kernels
    float4 rgba = SRC == 1 ? disk_v2_rgba(a, hit_x, hit_y)
                  : SRC == 3 ? sample_disk_level(a.sc, hit_x, hit_y, a.r_inner, a.r_outer, a.t_offset, lod_i, g_mip_lds, a.mip_lds_from)
                             : sample_disk_level(a.sc, hit_x, hit_y, a.r_inner, a.r_outer, a.t_offset, lod_i);
    float base_alpha = fminf(rgba.w, 0.999f);
    float disk_alpha = 1.0f - powf(1.0f - base_alpha, BHR_DISK_ALPHA_GAIN);
    V3 col = apply_g_factor(a, mk(rgba.x, rgba.y, rgba.z), mk(hit_x, hit_y, hit_z), hit_r, to_cam);
    float front = 1.0f - sh.alpha_total;
#if BHR_MARCH_STRICT
    sh.accum = mk(sh.accum.x + col.x * disk_alpha * front, sh.accum.y + col.y * disk_alpha * front,
                  sh.accum.z + col.z * disk_alpha * front);
#else
    float wgt = disk_alpha * front;
    sh.accum = mk(fmaf(col.x, wgt, sh.accum.x), fmaf(col.y, wgt, sh.accum.y), fmaf(col.z, wgt, sh.accum.z));
#endif
    sh.alpha_total = 1.0f - front * (1.0f - disk_alpha);
}

// Finite-thickness Disk V2 (docs/design_ad_v2.md 4.2-4.3, Phase 3 -- specified there, not implemented in
// the reference): emission-absorption through the volume |zeta| <= H(r), r_in <= r <= r_out of the tilted
// disk frame.  One RK4 step = one chord p0 -> p1, cut into vol_substeps pieces sampled at their midpoints:
//   rho = rho(r, zeta) F(r, phi_adv),  T = T(r, zeta) F,  phi_adv = phi + t_offset Omega(r)   (Phase 2)
//   alpha_eff = Ca rho [1 + kg (1 - |d.n|)]                                   (grazing-angle gain, 4.3)
//   opacity of the piece a = 1 - exp(-alpha_eff ds), source colour = black body of T with the g-factor,
// composited front to back exactly like a surface crossing (render.py:3000-3002), which is the design's
// L += exp(-tau) j ds, tau += alpha ds with j = alpha S integrated exactly over each piece.
// Model in binary64 (shared with the field evaluator), compositing in f32.
__device__ __forceinline__ void volume_segment(const BhrMarchArgs &a, Shade &sh, V3 p0, V3 p1, V3 dir0, float f0, float f1,
                                               float r0, float r1) {
    const bhr_disk_v2_params &P = *a.dv2;
    const double ct = (double)a.cos_t, st = (double)a.sin_t;
    const double z0 = (double)f0 * ct, z1 = (double)f1 * ct;           // heights above the disk plane
    const bool near_plane = z0 * z1 < 0.0 || fmin(fabs(z0), fabs(z1)) <= a.vol_h_max;
    if (!(near_plane && (double)fmaxf(r0, r1) >= P.r_in && (double)fminf(r0, r1) <= a.vol_r_max)) return;
    if (sh.alpha_total >= BHR_VOLUME_OPAQUE) return;     // what lies behind contributes < 1e-4 of its colour
    const double ex = (double)p1.x - (double)p0.x, ey = (double)p1.y - (double)p0.y, ez = (double)p1.z - (double)p0.z;
    const double len = sqrt(ex * ex + ey * ey + ez * ez);
    if (!(len > 0.0)) return;
    const double mu = fabs((ez * ct - ey * st) / len);
    const double ds = len / (double)a.vol_substeps;
    const V3 to_cam = mk(-dir0.x, -dir0.y, -dir0.z);
    for (int k = 0; k < a.vol_substeps; ++k) {
        const double f = ((double)k + 0.5) / (double)a.vol_substeps;
        const double sx = (double)p0.x + f * ex, sy = (double)p0.y + f * ey, sz = (double)p0.z + f * ez;
        const double zeta = sz * ct - sy * st;
        const double yp = sy * ct + sz * st;
        const double rc = sqrt(sx * sx + yp * yp);
        if (!dv2::volume_mask(rc, zeta, P)) continue;
        const double phi = atan2(yp, sx) + (double)a.t_offset * dv2::omega_field(rc, P);
        const double F = dv2::structure_total(rc, phi, P, a.dv2_norm_shear, a.dv2_norm_hotspot);
        const double rho = fmax(dv2::rho_field(rc, zeta, P) * F, 0.0);
        const double t = fmin(fmax(dv2::t_field(rc, zeta, P) * F / a.dv2_t_peak, 0.0), 1.0);
        const double alpha_eff = a.vol_absorption * rho * (1.0 + a.vol_grazing_gain * (1.0 - mu));
        const float op = (float)(1.0 - exp(-alpha_eff * ds));
        if (!(op > 0.0f)) continue;
        V3 col = apply_g_factor(a, disk_v2_color((float)t), mk((float)sx, (float)sy, (float)sz), (float)rc, to_cam);
        const float front = 1.0f - sh.alpha_total;
        sh.accum = mk(sh.accum.x + col.x * op * front, sh.accum.y + col.y * op * front, sh.accum.z + col.z * op * front);
        sh.alpha_total = 1.0f - front * (1.0f - op);
    }
}

// render.py:3008-3018: background through the accumulated opacity + clamped disk layer.  (i, j) = column, local row.
__device__ __forceinline__ void write_pixel(const BhrMarchArgs &a, int i, int j, bool escaped, V3 esc_dir, const Shade &sh) {
    V3 bg = mk(0, 0, 0);
    if (escaped) bg = sample_skybox(a.sc, normalized(esc_dir));
    float k = 1.0f - sh.alpha_total;
    size_t o = ((size_t)j * a.width + i) * 3;
    const float bk[3] = {__fmul_rn(bg.x, k), __fmul_rn(bg.y, k), __fmul_rn(bg.z, k)};
    a.bg[o + 0] = bk[0];
    a.bg[o + 1] = bk[1];
    a.bg[o + 2] = bk[2];
    const float dk[3] = {fminf(fmaxf(sh.accum.x, 0.0f), 1.0f), fminf(fmaxf(sh.accum.y, 0.0f), 1.0f), fminf(fmaxf(sh.accum.z, 0.0f), 1.0f)};
    a.disk[o + 0] = dk[0];
    a.disk[o + 1] = dk[1];
    a.disk[o + 2] = dk[2];
    if (a.diskp) {
        // bg + disk as the V pass would form it from the two stored layers (one rounding of the product, one of the sum):
        // its combine reads 12 bytes per pixel instead of 24
        a.sum[o + 0] = __fadd_rn(bk[0], dk[0]);
        a.sum[o + 1] = __fadd_rn(bk[1], dk[1]);
        a.sum[o + 2] = __fadd_rn(bk[2], dk[2]);
        // The disk layer once more for the split-f16 bloom (bloom.hip): every value x 2^14 cut into two f16 halves (hi =
        // RN16, lo = RN16 of the rest: 24 significant bits between them), laid out [channel][half][32-row block][8-pixel
        // group][row][8 pixels] -- the H pass's MFMA operand order.  The 8x8 tile of a wave is ONE 128-byte line of it per
        // channel and half: six fully coalesced 2-byte stores per pixel instead of a 96-byte-per-lane gather and a cut in
        // the H kernel.
        const size_t part = (size_t)a.dp_yb * a.dp_gp * 256;
        _Float16 *q = a.diskp + ((((size_t)(j >> 5)) * a.dp_gp + (i >> 3) + a.dp_g0) * 32 + (j & 31)) * 8 + (i & 7);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = dk[c] * 16384.0f;
            asm volatile("" : "+v"(v));                          // one product, one conversion: the stored half and the one `lo` is
            unsigned int hb = __builtin_bit_cast(unsigned short, (_Float16)v);   // formed against are the same bits (bloom.hip: cut2)
            asm volatile("" : "+v"(hb));
            const _Float16 hi = __builtin_bit_cast(_Float16, (unsigned short)hb);
            q[(size_t)(2 * c) * part] = hi;
            q[(size_t)(2 * c + 1) * part] = (_Float16)(v - (float)hi);
        }
    }
}

// Pixel -> ray (render.py:2811-2840).  Returns the unit direction; dx1/dy1 = directions through
// the pixel one to the right / one below (differential seeds).
template <bool DIFF>
__device__ __forceinline__ V3 pixel_ray(const BhrMarchArgs &a, int i, int j_local, V3 &ddx, V3 &ddy) {
    const V3 cp = ld3(a.cp), cr = ld3(a.cr), cu = ld3(a.cu), cf = ld3(a.cf);
    V3 center = cp + 1.0f * cf;
    float half_w = a.pw * (float)a.width / 2;
    float half_h = a.ph * (float)a.height / 2;
    V3 tl = (center - half_w * cr) + half_h * cu;
    float px_f = (float)i, py_f = (float)(j_local + a.row0);
    V3 pixel_pos = (tl + ((px_f + 0.5f) * a.pw) * cr) - ((py_f + 0.5f) * a.ph) * cu;
    V3 ray_dir = normalized(pixel_pos - cp);
    if (DIFF) {
        V3 ppx1 = (tl + ((px_f + 1.5f) * a.pw) * cr) - ((py_f + 0.5f) * a.ph) * cu;
        ddx = normalized(ppx1 - cp) - ray_dir;
        V3 ppy1 = (tl + ((px_f + 0.5f) * a.pw) * cr) - ((py_f + 1.5f) * a.ph) * cu;
        ddy = normalized(ppy1 - cp) - ray_dir;
    }
    return ray_dir;
}

#if BHR_MARCH_STRICT
// =============================================================================
// strict build: render.py:2854-3006 operation by operation, 3-D state
// =============================================================================
template <bool DIFF, int SRC = 0>
struct Ray {
    V3 p, d;
    float m15L2;   // -1.5 * L2
    float r;       // |p|
    float r2p;     // |p|^2
    float f_old;   // plane function at p
    float affine;
    Shade sh;
    int n_pend;    // parked disk crossings (0..2), in the lane's LDS slots, oldest first
    int step_count;
    int pix;       // linear pixel index inside the row block, -1 = lane has no ray
    int done;      // 0 running, 2 captured or escaped (escaped() tells), 3 ran out of iterations, 4 empty lane
    bool full;     // wave-uniform: some live lane has both its parking slots occupied
    V3 dpx, ddx, dpy, ddy;   // ray differentials (DIFF only)

    __device__ __forceinline__ void init(const BhrMarchArgs &a, int i, int j_local) {
        V3 ray_dir = pixel_ray<DIFF>(a, i, j_local, ddx, ddy);
        p = ld3(a.cp);
        d = ray_dir;
        V3 Lv = cross(d, p);
        float Ln = sqrtf(dot(Lv, Lv));
        m15L2 = -1.5f * (Ln * Ln);
        r2p = dot(p, p);
        r = sqrtf(r2p);
        f_old = p.z - p.y * a.tan_t;
        affine = 0.0f;
        sh.accum = mk(0, 0, 0);
        sh.alpha_total = 0.0f;
        sh.unsure = 0;
        n_pend = 0;
        full = false;
        step_count = 0;
        done = a.max_iter <= 0 ? 3 : 0;
        pix = j_local * a.width + i;
        if (DIFF) {
            dpx = mk(0, 0, 0);
            dpy = mk(0, 0, 0);
        }
    }

    // a(s) = (-1.5 L2 / r^5) s with r = sqrt(s.s), r^5 = (r2 r2) r     (render.py:2518-2524)
    __device__ __forceinline__ float coef(float r2, float rr) const { return div_rn(m15L2, r2 * r2 * rr); }
    // factor (d_pos - 5 pos proj), proj = pos.d_pos / r2                  (render.py:2526-2539)
    __device__ __forceinline__ V3 jac(V3 s, V3 dl, float factor, float r2) const {
        float proj = div_rn(dot(s, dl), r2);
        return factor * mk(dl.x - 5.0f * s.x * proj, dl.y - 5.0f * s.y * proj, dl.z - 5.0f * s.z * proj);
    }
    __device__ __forceinline__ V3 rk_sum(V3 k1, V3 k2, V3 k3, V3 k4) const {   // (k1 + 2 k2 + 2 k3 + k4) / 6
        // 2 k is exact, so fma(2, k2, k1) == k1 + 2 k2 rounded once, as in the reference
        return mk(div6(fmaf(2.0f, k3.x, fmaf(2.0f, k2.x, k1.x)) + k4.x), div6(fmaf(2.0f, k3.y, fmaf(2.0f, k2.y, k1.y)) + k4.y),
                  div6(fmaf(2.0f, k3.z, fmaf(2.0f, k2.z, k1.z)) + k4.z));
    }

    // One iteration of the while-loop at render.py:2854-3006.  The state is committed unconditionally: a lane whose ray
    // has terminated leaves the loop and never reads it again (escaped rays read d = new_dir), and with two parking slots
    // in LDS a hit always finds room, with or without differentials.  (Until round 2 the AA kernel kept ONE slot in
    // registers and repeated the step of a lane that found it occupied; under the ILP scheduler at 4 waves per SIMD the
    // LDS scheme is 3 % faster -- 4k AA 6.44 -> 6.26 ms, same pixels -- and the redo path is gone.)
    __device__ __forceinline__ bool step(const BhrMarchArgs &a) {
        // clamps as single v_med3 / v_min instructions (no NaN can reach them: r is a finite norm); the C forms cost a
        // canonicalising v_max, and compare + select pairs
        float r_safe;                                    // max(r, r_cap + 1e-3) without the canonicalising second v_max
        asm("v_max_f32 %0, %1, %2" : "=v"(r_safe) : "v"(r), "v"(BHR_RS + 1e-3f));
        // The eleven hardware approximations of a step (five v_rsq for the exact square roots, six v_rcp for the exact
        // quotients) in six groups of independent operands, back to back (round 4; sqrt_rn_s / div_rn_s: the same
        // refinements on the same seeds, every value bit for bit what the one-at-a-time order gives).
        const float den1 = r2p * r2p * r;                // r^5 of coef(r2p, r)
        float y_s, y_q, y_1;
        rsq_rcp_rcp(r_safe, den1, y_s, y_q, y_1);
        float far_scale = __builtin_fminf(sqrt_rn_s(r_safe, y_s), 10.0f);   // sqrt(r_safe / r_cap), r_cap = 1; capped at max_fac
        float q = rcp_rn_s(r_safe, y_q);                 // r_cap / r_safe, r_cap = 1
        float near_damp = rcp_rn(fmaf(2.0f, q * q * q, 1.0f));   // 2 x is exact: one rounding, as 1 + 2 x has
        float dt_fac = __builtin_amdgcn_fmed3f(far_scale * near_damp, 0.2f, 10.0f);   // render.py:2865-2868
        float h = a.h_base * dt_fac;

        float f1 = div_rn_s(m15L2, den1, y_1);
        V3 k1p = h * d;
        V3 k1d = h * (f1 * p);
        V3 s2 = add_half(p, k1p);
        float r2_2 = dot(s2, s2);
        V3 k2p = h * add_half(d, k1d);
        V3 s3 = add_half(p, k2p);
        float r2_3 = dot(s3, s3);
        float y_2, y_3;
        rsq2(r2_2, r2_3, y_2, y_3);
        const float den2 = r2_2 * r2_2 * sqrt_rn_s(r2_2, y_2), den3 = r2_3 * r2_3 * sqrt_rn_s(r2_3, y_3);
        rcp2(den2, den3, y_2, y_3);
        float f2 = div_rn_s(m15L2, den2, y_2);
        float f3 = div_rn_s(m15L2, den3, y_3);
        V3 k2d = h * (f2 * s2);
        V3 k3p = h * add_half(d, k2d);
        V3 k3d = h * (f3 * s3);
        V3 s4 = p + k3p;
        float r2_4 = dot(s4, s4);
        V3 k4p = h * (d + k3d);
        V3 np = p + rk_sum(k1p, k2p, k3p, k4p);
        float r2n = dot(np, np);
        float y_4, y_n;
        rsq2(r2_4, r2n, y_4, y_n);
        float f4 = coef(r2_4, sqrt_rn_s(r2_4, y_4));
        float rn = sqrt_rn_s(r2n, y_n);
        V3 k4d = h * (f4 * s4);
        V3 nd = d + rk_sum(k1d, k2d, k3d, k4d);

        V3 ndpx, nddx, ndpy, nddy;
        if (DIFF) {
            {
                V3 a1p = h * ddx;
                V3 a1d = h * jac(p, dpx, f1, r2p);
                V3 a2p = h * add_half(ddx, a1d);
                V3 a2d = h * jac(s2, add_half(dpx, a1p), f2, r2_2);
                V3 a3p = h * add_half(ddx, a2d);
                V3 a3d = h * jac(s3, add_half(dpx, a2p), f3, r2_3);
                V3 a4p = h * (ddx + a3d);
                V3 a4d = h * jac(s4, dpx + a3p, f4, r2_4);
                ndpx = dpx + rk_sum(a1p, a2p, a3p, a4p);
                nddx = ddx + rk_sum(a1d, a2d, a3d, a4d);
            }
            {
                V3 a1p = h * ddy;
                V3 a1d = h * jac(p, dpy, f1, r2p);
                V3 a2p = h * add_half(ddy, a1d);
                V3 a2d = h * jac(s2, add_half(dpy, a1p), f2, r2_2);
                V3 a3p = h * add_half(ddy, a2d);
                V3 a3d = h * jac(s3, add_half(dpy, a2p), f3, r2_3);
                V3 a4p = h * (ddy + a3d);
                V3 a4d = h * jac(s4, dpy + a3p, f4, r2_4);
                ndpy = dpy + rk_sum(a1p, a2p, a3p, a4p);
                nddy = ddy + rk_sum(a1d, a2d, a3d, a4d);
            }
        }

        float aff = affine + h;
        // termination precedes the plane test (render.py:2916-2926): the ray goes on iff r_s <= |new_pos| <= r_escape and the
        // affine parameter is within its limit -- the reference's strict inequalities, the two radii as one v_med3 + one
        // compare (which of them ended the ray: escaped(), behind the loop)
        const bool ended = __builtin_amdgcn_fmed3f(rn, BHR_RS, a.r_esc) != rn || aff > a.max_affine;
        const bool alive = !ended;
        float f_new = np.z - np.y * a.tan_t;
        const bool crossing = f_old * f_new < 0;
        if (SRC == 2) {
            if (alive) volume_segment(a, sh, p, np, d, f_old, f_new, r, rn);
        } else if (__builtin_amdgcn_ballot_w64(crossing) != 0ull) {
            // a wave-uniform branch around the crossing code (a few steps per ray): `full` is a uniform value set under uniform
            // control and lives in a scalar register -- the march loop tests it instead of comparing n_pend in every step
            if (alive && crossing) {
                float t_frac = div_rn(f_old, f_old - f_new + 1e-8f);
                float hx = p.x + t_frac * (np.x - p.x);
                float hy = p.y + t_frac * (np.y - p.y);
                float hit_r = sqrt_rn(hx * hx + hy * hy);
                if (a.r_outer >= hit_r && hit_r >= a.r_inner) {   // render.py:2951
                    Pending<DIFF> h;
                    h.hit_x = hx;
                    h.hit_y = hy;
                    h.to_cam = mk(-d.x, -d.y, -d.z);              // direction at the START of the step (render.py:2954)
                    // the differentials were committed BEFORE the hit interpolation (render.py:2928-2932),
                    // hence hit_d_pos == new_d_pos in render.py:2947-2949
                    if (DIFF) { h.dxx = ndpx.x; h.dxy = ndpx.y; h.dyx = ndpy.x; h.dyy = ndpy.y; }
                    park_store<DIFF>(n_pend, h);                  // a free slot is guaranteed (march_tile_body flushes at 2)
                    n_pend += 1;
                }
            }
            full = __builtin_amdgcn_ballot_w64(n_pend == 2) != 0ull;
        }
        affine = aff;
        if (DIFF) { dpx = ndpx; ddx = nddx; dpy = ndpy; ddy = nddy; }
        p = np;
        d = nd;
        r = rn;
        r2p = r2n;
        f_old = f_new;
        step_count += 1;
        done = ended ? 2 : (step_count >= a.max_iter ? 3 : 0);
        return true;
    }

    // The loop's own termination test once more, on the state a finished lane is left with (r = |p| and the affine parameter
    // are those very values): the tile kernels call it behind the march loop instead of reading `done` back (see the fast
    // Ray's settle()).
    __device__ __forceinline__ void settle(const BhrMarchArgs &a) {
        done = (__builtin_amdgcn_fmed3f(r, BHR_RS, a.r_esc) != r || affine > a.max_affine) ? 2 : 3;
    }
    __device__ __forceinline__ bool escaped() const { return done == 2 && !(r < BHR_RS); }

    // shade the oldest parked crossing (lanes that have one), the second slot moves up
    __device__ __forceinline__ void flush_one(const BhrMarchArgs &a) {
        if (n_pend > 0) {
            const Pending<DIFF> h = park_load<DIFF>(0);
            if (n_pend == 2) park_store<DIFF>(0, park_load<DIFF>(1));
            n_pend -= 1;
            shade_hit<DIFF, SRC>(a, sh, h.hit_x, h.hit_y, h.to_cam, h.dxx, h.dxy, h.dyx, h.dyy);
        }
    }
    __device__ __forceinline__ void finish(const BhrMarchArgs &a) { write_pixel(a, pix % a.width, pix / a.width, escaped(), d, sh); }
    __device__ __forceinline__ void finish_at(const BhrMarchArgs &a, int i, int j) { write_pixel(a, i, j, escaped(), d, sh); }
};

#else
// =============================================================================
// fast build.  The force is central, so a ray never leaves the plane spanned by the camera
// position and its initial direction, and RK4 commutes with rotations: marching the 2-D state
// (U, W) in an orthonormal in-plane basis (g1, g2) visits exactly the reference's sequence of
// positions up to rounding, with a third fewer vector operations.  The basis is chosen per ray
// so that g1 is the line of nodes (orbital plane ^ disk plane): the disk-plane function
// z - y tan(tilt) = n.x then reduces to (n.g2) W, i.e. W is the scaled height above the disk and
// is SMALL where the crossing is detected -- the absolute precision of the crossing point is the
// same as with the reference's 3-D z coordinate (a basis tied to the camera direction loses a
// factor r/|z| there, measured as 2x the parity error).  Ray differentials split into an in-plane
// pair coupled through the projection term of the Jacobian and an out-of-plane component that
// sees only the isotropic term:  J d = c (d - 5 s (s.d)/r^2).
//
// The ray's own clock (round 4).  Every ray marches in an affine parameter of its own, lambda' = lambda / tau with
// tau^2 (1.5 L2) = 1: the equation of motion becomes u'' = -u / r^5 -- no coefficient to multiply in at the four radii of
// a step (c = -(1/r)^5 straight from the v_rsq) -- with velocities tau x direction and the step h_base dt_fac / tau (a
// per-lane factor in a vector register: a product with the scalar h_base issues at half rate).  RK4 is invariant
// under the rescaling, so the sequence of positions is the reference's up to rounding; tau carries a relative rounding
// error of ~1e-7 into the force constant, the size of the rounding of L2 itself.  Positions stay in r_s.
// =============================================================================
template <bool DIFF, int SRC = 0>
struct Ray {
    float u, w, du, dw;   // position / velocity (tau x direction) along (g1, g2)
    float hk;             // h_base / tau: step = dt_fac hk
    float ij;             // 1 / |(u, w)|
    float c1;             // acceleration coefficient at (u, w):  -1 / r^5
    float esc2;           // r_escape^2, in a vector register (an SGPR operand halves the v_med3's issue rate)
    float Bn;             // n . g2: the plane function z - y tan(tilt) is Bn w
    V3 g1, g2;            // in-plane orthonormal basis
    bool full;            // wave-uniform: some live lane has both its parking slots occupied
    float affine;         // in units of h_base
    Shade sh;
    int n_pend;    // parked disk crossings (0..2), in the lane's LDS slots, oldest first
    int step_count;
    int pix;
    int done;
    // differentials (DIFF only): components along (g1, g2, e3 = g1 x g2) of d_pos and d_dir
    V3 dpx, ddx, dpy, ddy;

    __device__ __forceinline__ void init(const BhrMarchArgs &a, int i, int j_local) {
        V3 gx, gy;
        V3 d0 = pixel_ray<DIFF>(a, i, j_local, gx, gy);
        const V3 p0 = ld3(a.cp);
        // L2 exactly as the reference forms it (render.py:2828)
        V3 Lv = cross(d0, p0);
        float L2 = dot(Lv, Lv);
        // tau = (1.5 L2)^(-1/2), v_rsq + one Newton step.  A radial ray (L2 -> 0: no deflection at all) marches with the
        // force of L2 ~ 1e-12: below the rounding of its velocity
        const float kap = fmaxf(1.5f * L2, 1e-12f);
        float tau = q_rsq(kap);
        tau = tau * fmaf(-0.5f * kap, tau * tau, 1.5f);
        hk = a.h_base * (kap * tau);
        // unit normal of the orbital plane; for a radial ray (L = 0) any direction orthogonal to p0
        V3 e3;
        if (L2 > 1e-20f) {
            e3 = (1.0f / sqrtf(L2)) * Lv;
        } else {
            V3 t = fabsf(p0.x) < 0.9f * a.r0 ? mk(1, 0, 0) : mk(0, 1, 0);
            V3 q = cross(p0, t);
            e3 = (1.0f / sqrtf(dot(q, q))) * q;
        }
        // g2 = in-plane part of the disk-plane normal n = (0, -tan_t, 1), g1 = g2 x e3 (line of nodes)
        const V3 n = mk(0.0f, -a.tan_t, 1.0f);
        float ne = dot(n, e3);
        V3 np_ = mk(fmaf(-ne, e3.x, n.x), fmaf(-ne, e3.y, n.y), fmaf(-ne, e3.z, n.z));
        float nn = dot(np_, np_);
        if (nn > 1e-12f) {
            g2 = (1.0f / sqrtf(nn)) * np_;
        } else {  // the ray stays inside the disk plane and never crosses it: any in-plane axis
            g2 = (1.0f / a.r0) * p0;
        }
        g1 = cross(g2, e3);
        Bn = dot(n, g2);
        u = dot(p0, g1);
        w = dot(p0, g2);
        du = tau * dot(d0, g1);
        dw = tau * dot(d0, g2);
        ij = 1.0f / a.r0;
        float i2 = ij * ij;
        c1 = -(i2 * i2 * ij);
        asm volatile("v_mov_b32 %0, %1" : "=v"(esc2) : "s"(a.r_esc2));
        full = false;
        affine = 0.0f;
        sh.accum = mk(0, 0, 0);
        sh.alpha_total = 0.0f;
        sh.unsure = 0;
        n_pend = 0;
        step_count = 0;
        done = a.max_iter <= 0 ? 3 : 0;
        pix = j_local * a.width + i;
        if (DIFF) {
            ddx = mk(tau * dot(gx, g1), tau * dot(gx, g2), tau * dot(gx, e3));
            ddy = mk(tau * dot(gy, g1), tau * dot(gy, g2), tau * dot(gy, e3));
            dpx = mk(0, 0, 0);
            dpy = mk(0, 0, 0);
        }
    }

    // coefficient c = -1 / r^5 and 1/r^2 from i1 = 1/r
    __device__ __forceinline__ float coef(float i1, float &i2) const {
        i2 = i1 * i1;
        return -(i2 * i2 * i1);
    }
    // (The radii of stages 2 and 3 are both known before either coefficient is needed, and so are stage 4's and the new
    // position's: their v_rsq go back to back -- rsq2 -- five transcendentals per step in three groups instead of five.)
    // J(s) delta with delta = (in-plane u, in-plane w, out-of-plane n)
    __device__ __forceinline__ V3 jac(float su, float sw, V3 dl, float c, float i2) const {
        float proj5 = 5.0f * fmaf(su, dl.x, sw * dl.y) * i2;
        return mk(c * fmaf(-proj5, su, dl.x), c * fmaf(-proj5, sw, dl.y), c * dl.z);
    }
    __device__ __forceinline__ void rk4_diff(V3 &dp, V3 &dd, float h, float hh, float h6, float s2u, float s2w,
                                             float s3u, float s3w, float s4u, float s4w, float c2, float c3,
                                             float c4, float i2_1, float i2_2, float i2_3, float i2_4) const {
        V3 j1 = jac(u, w, dp, c1, i2_1);
        V3 e2_ = fma3(hh, dd, dp), w2 = fma3(hh, j1, dd);
        V3 j2 = jac(s2u, s2w, e2_, c2, i2_2);
        V3 e3_ = fma3(hh, w2, dp), w3 = fma3(hh, j2, dd);
        V3 j3 = jac(s3u, s3w, e3_, c3, i2_3);
        V3 e4_ = fma3(h, w3, dp), w4 = fma3(h, j3, dd);
        V3 j4 = jac(s4u, s4w, e4_, c4, i2_4);
        V3 ndp = fma3(h6, (dd + w4) + 2.0f * (w2 + w3), dp);
        V3 ndd = fma3(h6, (j1 + j4) + 2.0f * (j2 + j3), dd);
        dp = ndp;
        dd = ndd;
    }
    __device__ __forceinline__ V3 to3d(float cu, float cw) const {
        return mk(fmaf(cu, g1.x, cw * g2.x), fmaf(cu, g1.y, cw * g2.y), fmaf(cu, g1.z, cw * g2.z));
    }

    // One iteration of the while-loop at render.py:2854-3006.  Statement order keeps every state variable
    // updated in place after its last use; the state is committed unconditionally (a terminated lane leaves
    // the loop, an escaped ray reads (du, dw) back as new_dir).
    __device__ __forceinline__ bool step(const BhrMarchArgs &a) {
        // adaptive step (render.py:2858-2869) from 1/r with ONE transcendental:  q = 1/r_safe, far_scale = min(sqrt(r_safe), 10)
        // = rsq(max(q, 0.01)), near_damp = 1 / (1 + 2 q^3), so far_scale near_damp = rsq(max(q, 0.01) (1 + 2 q^3)^2).  The
        // reference's clamp to [0.2, 10] never binds: q <= 1 / 1.001 gives far_scale >= 1 and near_damp > 1/3, and the
        // product is <= far_scale <= 10.  (Round 3: v_rsq + v_rcp, 12.7 issue cycles each inside this instruction mix.)
        float q = fminf(ij, 1.0f / (BHR_RS + 1e-3f));
        float nd = fmaf(2.0f * q, q * q, 1.0f);
        float dt_fac = q_rsq(fmaxf(q, 0.01f) * (nd * nd));
        float h = dt_fac * hk;             // in the ray's own clock
        float hh = 0.5f * h;
        float h6 = h * (1.0f / 6.0f);

        // RK4 (render.py:2872-2882) on velocities v_k = k_kp / h and accelerations a_k = k_kd / h
        float a1u = c1 * u, a1w = c1 * w;
        float s2u = fmaf(hh, du, u), s2w = fmaf(hh, dw, w);
        float v2u = fmaf(hh, a1u, du), v2w = fmaf(hh, a1w, dw);
        float i2_2, i2_3, i2_4;
        float s3u = fmaf(hh, v2u, u), s3w = fmaf(hh, v2w, w);
        float i1_2, i1_3, i1_4, i1_n;
        rsq2(fmaf(s2u, s2u, s2w * s2w), fmaf(s3u, s3u, s3w * s3w), i1_2, i1_3);
        float c2 = coef(i1_2, i2_2);
        float a2u = c2 * s2u, a2w = c2 * s2w;
        float v3u = fmaf(hh, a2u, du), v3w = fmaf(hh, a2w, dw);
        float c3 = coef(i1_3, i2_3);
        float a3u = c3 * s3u, a3w = c3 * s3w;
        float s4u = fmaf(h, v3u, u), s4w = fmaf(h, v3w, w);
        float v4u = fmaf(h, a3u, du), v4w = fmaf(h, a3w, dw);
        float nu = fmaf(h6, (du + v4u) + 2.0f * (v2u + v3u), u);
        float nw = fmaf(h6, (dw + v4w) + 2.0f * (v2w + v3w), w);
        float r2n = fmaf(nu, nu, nw * nw);
        rsq2(fmaf(s4u, s4u, s4w * s4w), r2n, i1_4, i1_n);
        float c4 = coef(i1_4, i2_4);
        float sdu = fmaf(c4, s4u, a1u) + 2.0f * (a2u + a3u);
        float sdw = fmaf(c4, s4w, a1w) + 2.0f * (a2w + a3w);

        // the affine parameter is kept in units of h_base: one plain v_add per step, compared against max_affine / h_base
        float aff = affine + dt_fac;
        // termination precedes the plane test (render.py:2916-2926); r < r_s  <=>  r^2 < r_s^2 etc.  One v_med3 and one
        // compare for the two radii (a compare costs two plain instructions' issue time): the ray goes on iff
        // r_s^2 <= r^2 <= r_esc^2, the same strict inequalities as the reference's.  Which of the two ended it is worked
        // out once, behind the loop (escaped()).
        const bool ended = __builtin_amdgcn_fmed3f(r2n, BHR_RS * BHR_RS, esc2) != r2n || aff > a.max_affine_u;
        const bool alive = !ended;
        // The plane function is Bn w: its sign changes where w's does, so the loop carries no plane function and no Bn (two
        // registers and a multiplication per step); the reference's own test, on the products, is made inside the
        // wave-uniform branch below (it also keeps a ray that lies IN the disk plane, Bn = 0, from ever crossing it).
        const bool crossing = w * nw < 0;
        const float f_old = Bn * w, f_new = Bn * nw;     // (dead in the kernels that read no guard flag)
        // Discontinuity guard (read by the hybrid kernel only): a step that crosses the disk plane registers the hit only
        // if it does not also end the ray (render.py:2916-2934) -- with a disk wider than the escape sphere that is a hit /
        // no-hit switch at |new_pos| = r_escape.  A crossing step that ends within the guard of a termination radius marks the lane.
        if (f_old * f_new < 0 && (fabsf(r2n - a.r_esc2) < BHR_R2_GUARD * a.r_esc2 || fabsf(r2n - BHR_RS * BHR_RS) < BHR_R2_GUARD)) sh.unsure = 1;
        // ... and a step that ENDS on the plane: the reference tests f_old f_new < 0, so a new_pos whose plane function
        // rounds to exactly 0 is a crossing that no step ever registers (a black pixel inside the disk: ~1e-6 of the
        // crossings, a dozen pixels of a 4k frame), and one a few ulps either side of 0 moves the hit into the next
        // step (which may be the terminating one).  Only the bit-identical arithmetic reproduces these.
        if (f_new * f_new < BHR_F_GUARD * BHR_F_GUARD * r2n) sh.unsure = 1;
        bool hit_now = false;
        if (SRC == 2) {
            if (alive) volume_segment(a, sh, to3d(u, w), to3d(nu, nw), to3d(du, dw), f_old, f_new, q_rcp(ij), r2n * q_rsq(r2n));
        } else if (__builtin_amdgcn_ballot_w64(crossing) != 0ull) {
            // (a wave-uniform branch around the crossing code: `full` is then a uniform value set under uniform control, it
            // stays in a scalar register and the march loop tests it with a scalar compare -- the per-step
            // v_cmp(n_pend == 2) + v_cmp(n_pend > 0) of round 3 cost four plain instructions' issue time.  The branch is on
            // the ONE compare's mask: combined with `alive` hipcc rebuilds the mask through v_cndmask + v_cmp)
            const float bn = fmaf(-a.tan_t, g2.y, g2.z);           // n . g2 again: Bn is not kept across the loop
            const float fo = bn * w, fn = bn * nw;
            if (alive && fo * fn < 0) {
                float t_frac = fo / (fo - fn + 1e-8f);
                float hu = fmaf(t_frac, nu - u, u), hw = fmaf(t_frac, nw - w, w);
                float hx = fmaf(hu, g1.x, hw * g2.x);
                float hy = fmaf(hu, g1.y, hw * g2.y);
                float hr2 = fmaf(hx, hx, hy * hy);
                float hit_r = hr2 * q_rsq(hr2);
                // the annulus test is the other switch: a crossing within the guard of either edge marks the lane
                if (fabsf(hit_r - a.r_outer) < BHR_EDGE_GUARD * a.r_outer || fabsf(hit_r - a.r_inner) < BHR_EDGE_GUARD * a.r_inner) sh.unsure = 1;
                if (a.r_outer >= hit_r && hit_r >= a.r_inner) {   // render.py:2951
                    V3 dir3 = to3d(du, dw);                  // direction at the START of the step (render.py:2954), x tau
                    Pending<DIFF> h;
                    h.hit_x = hx;
                    h.hit_y = hy;
                    h.to_cam = mk(-dir3.x, -dir3.y, -dir3.z);
                    if (DIFF) h.dxx = h.dxy = h.dyx = h.dyy = 0.0f;   // attached below, once the new differentials exist
                    park_store<DIFF>(n_pend, h);                      // a free slot is guaranteed (march_tile_kernel)
                    n_pend += 1;
                    hit_now = true;
                }
            }
            full = __builtin_amdgcn_ballot_w64(n_pend == 2) != 0ull;
        }
        if (DIFF && alive) {
            // variational RK4 at the same four stage positions (render.py:2888-2911); the hit reads the
            // NEW differentials (committed before the plane test, render.py:2928-2932)
            float i2_1 = ij * ij;
            rk4_diff(dpx, ddx, h, hh, h6, s2u, s2w, s3u, s3w, s4u, s4w, c2, c3, c4, i2_1, i2_2, i2_3, i2_4);
            rk4_diff(dpy, ddy, h, hh, h6, s2u, s2w, s3u, s3w, s4u, s4w, c2, c3, c4, i2_1, i2_2, i2_3, i2_4);
            if (hit_now) {                               // hit parked in THIS step: attach its footprint
                V3 e3 = cross(g1, g2);
                const float fxx = dpx.x * g1.x + dpx.y * g2.x + dpx.z * e3.x, fxy = dpx.x * g1.y + dpx.y * g2.y + dpx.z * e3.y;
                const float fyx = dpy.x * g1.x + dpy.y * g2.x + dpy.z * e3.x, fyy = dpy.x * g1.y + dpy.y * g2.y + dpy.z * e3.y;
                const int slot = n_pend - 1, t = threadIdx.x;
                g_park[slot][5][t] = fxx;
                g_park[slot][6][t] = fxy;
                g_park[slot][7][t] = fyx;
                g_park[slot][8][t] = fyy;
            }
        }
        affine = aff;
        du = fmaf(h6, sdu, du);                          // escaped rays read these back as the escape
        dw = fmaf(h6, sdw, dw);                          // direction = new_dir (render.py:2921)
        u = nu;
        w = nw;
        ij = i1_n;
        float i2 = ij * ij;
        c1 = -(i2 * i2 * ij);
        step_count += 1;
        done = ended ? 2 : (step_count >= a.max_iter ? 3 : 0);      // 2 = captured or escaped: escaped() tells
        return true;     // two parking slots: a step never has to be repeated
    }

    // The loop's own termination test once more, on the state a finished lane is left with (the same expressions on the same
    // values): the tile kernels call it behind the march loop instead of reading `done` back -- a value written inside a
    // loop that lanes leave at different trips and read behind it costs three scalar mask instructions per trip and per
    // bit to keep (the exit mask alone is the loop's own).
    __device__ __forceinline__ void settle(const BhrMarchArgs &a) {
        const float r2 = fmaf(u, u, w * w);
        done = (__builtin_amdgcn_fmed3f(r2, BHR_RS * BHR_RS, esc2) != r2 || affine > a.max_affine_u) ? 2 : 3;
    }
    // which of the two radii (or the affine limit) ended the ray: captured = inside r_s, as the loop's own test has it
    __device__ __forceinline__ bool escaped() const { return done == 2 && !(fmaf(u, u, w * w) < BHR_RS * BHR_RS); }

    // shade the oldest parked crossing (lanes that have one), the second slot moves up
    __device__ __forceinline__ void flush_one(const BhrMarchArgs &a) {
        if (n_pend > 0) {
            const Pending<DIFF> h = park_load<DIFF>(0);
            if (n_pend == 2) park_store<DIFF>(0, park_load<DIFF>(1));
            n_pend -= 1;
            shade_hit<DIFF, SRC>(a, sh, h.hit_x, h.hit_y, h.to_cam, h.dxx, h.dxy, h.dyx, h.dyy);
        }
    }
    __device__ __forceinline__ void finish(const BhrMarchArgs &a) { write_pixel(a, pix % a.width, pix / a.width, escaped(), to3d(du, dw), sh); }
    __device__ __forceinline__ void finish_at(const BhrMarchArgs &a, int i, int j) { write_pixel(a, i, j, escaped(), to3d(du, dw), sh); }
};
#endif  // BHR_MARCH_STRICT

__device__ __forceinline__ unsigned long long wave_sum_u32(unsigned int v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, BHR_WAVE);
    return s;
}

// ---------------------------------------------------------------------------
// tile schedule: block = 4 waves = 4 horizontally adjacent 8x8 tiles (32x8 px).
// blockIdx maps to tiles in plain row-major order: the dispatcher deals consecutive blocks
// round-robin over the 8 XCDs, which spreads the expensive rows (those through the photon
// ring) evenly.  An XCD-banded remap was measured and rejected (-13 %: the kernel is VALU
// bound, texture traffic is negligible, and bands of rows differ in cost).
// ---------------------------------------------------------------------------
// The texture kernels are held to 128 VGPRs (4 waves per SIMD): the strict arithmetic is a chain of dependent
// exact-rounding sequences and needs the waves to cover its latency (measured at 4k with AA: 141 VGPRs / 3 waves
// 7.6 ms, 128 / 4 waves 6.7 ms).  The binary64 Disk V2 instantiations take what they need.
// GUARD (the fast list of a hybrid march, fast object only): a lane that came within a guard band of one of the
// algorithm's switches (Shade.unsure) does not write its pixel; it appends it to the context's fix list, which
// march_fix_kernel (strict objects) marches again with the strict Ray.
template <bool DIFF, int SRC = 0, bool GUARD = false, bool COSTS = true>
__device__ __forceinline__ void march_tile_body(const BhrMarchArgs &a, const int slot) {
    const int lane = threadIdx.x & 63;
    // one 8x8 tile per wave; `slot` is its position in the launch order
    // tiles are handed out longest first (tile_order: by distance from the image of the hole, where rays take the
    // most steps), so that the launch does not end on a few late, long waves
    // n_list = launch slots of THIS launch: all tiles of the row block, or the sub-list a hybrid launch hands this kernel
    const int tile = slot < a.n_list ? (a.tile_order ? a.tile_order[slot] : slot) : a.n_tiles;
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int i = tx * 8 + (lane & 7);
    const int j = ty * 8 + (lane >> 3);
    const bool valid = tile < a.n_tiles && i < a.width && j < a.rows;
    // (slot, tile, tx, ty are wave-uniform: march_tile_kernel / march_tiles_of_wave hand over a readfirstlane'd slot)

#if BHR_WAVE_STAMPS_BUILD
    const unsigned long long t_start = a.wave_stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
    Ray<DIFF, SRC> ray;
    ray.init(a, valid ? i : 0, valid ? j : 0);
    if (!valid) ray.done = 4;
    // Divergent loop: a lane leaves when its ray terminates, the wave leaves when its EXEC mask is
    // empty (the hardware form of "loop while __ballot(alive)").  Written without an inner `if` because
    // hipcc otherwise shuttles the whole ray state through v_mov at every iteration (24 moves/step).
    unsigned int flushes = 0;     // wave-uniform
    // Values that are uniform over the live lanes but read behind the divergent loop (the step count, the
    // number of shading passes) are kept in scalar registers by hipcc and copied into a vector register in EVERY
    // iteration for the lanes that leave (v_mov from an SGPR: 4 issue cycles each).  The lane's own count in a vector
    // register costs one plain v_add.  The shading passes inside the loop are counted only by the instantiations that
    // fill the row-cost profile (COSTS: BHR_ROW_COSTS launches): the plain kernel sits exactly at 80 registers = 6 waves
    // per SIMD, and one more value alive across the loop costs it a wave of occupancy.
    int cnt = 0, passes = 0;
    asm volatile("" : "+v"(cnt));
    if (COSTS) asm volatile("" : "+v"(passes));
    // The loop body twice per trip: the state a step leaves in fresh registers (new position, new plane function) is the
    // next step's input where it stands -- rolled once, hipcc closed every iteration with three v_mov to bring it back to
    // the registers the loop head expects.
#define BHR_FAST_STEP()                                                                                                     \
    ray.step(a);                                                                                                            \
    cnt += 1;                                                                                                               \
    if (ray.full) { /* wave-uniform, a scalar register (Ray::step): some live lane has filled both its parking slots */     \
        asm volatile("" : "+v"(ray.n_pend)); /* the lanes' own n_pend > 0 test stays inside this branch */                  \
        ray.flush_one(a);                                                                                                   \
        ray.full = false;                                                                                                   \
        if (COSTS) passes += 1;                                                                                             \
    }
    while (ray.done == 0) {
        BHR_FAST_STEP()
        if (ray.done != 0) break;
        BHR_FAST_STEP()
    }
#undef BHR_FAST_STEP
    ray.step_count = cnt;
    if (cnt > 0) ray.settle(a);
    else ray.done = 3;               // no step taken (max_iter <= 0, or no ray: those lanes store nothing)
    if (COSTS) {   // the lanes that were alive at the wave's last pass have seen them all
        int m = passes;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, BHR_WAVE));
        flushes = (unsigned int)m;
    }
    if (__ballot(ray.n_pend > 0)) { ray.flush_one(a); flushes += 1u; }
    if (__ballot(ray.n_pend > 0)) { ray.flush_one(a); flushes += 1u; }
    {
        // The lane's pixel, worked out AGAIN from the thread index behind an optimisation barrier: nothing that is
        // only needed here (pixel index, validity, store addresses) stays in a register across the march loop -- the
        // strict AA kernel spilled five such values at 128 VGPRs (round 2: 6 spills, 28 B of scratch).
        int t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        const int i2 = tx * 8 + (t2 & 7), j2 = ty * 8 + ((t2 & 63) >> 3);
        const bool valid2 = tile < a.n_tiles && i2 < a.width && j2 < a.rows;
        bool again = false;
        if (GUARD) {
            again = valid2 && ray.sh.unsure != 0;
            const unsigned long long m = __ballot(again);
            if (m) {                                   // wave-aggregated append
                const int lane2 = t2 & 63, first = __ffsll((long long)m) - 1;
                unsigned int base = 0;
                if (lane2 == first) base = atomicAdd(a.fix_count, (unsigned int)__popcll(m));
                base = __shfl(base, first, BHR_WAVE);
                const unsigned int at = base + (unsigned int)__popcll(m & ((1ull << lane2) - 1ull));
                if (again && at < (unsigned int)a.fix_cap) a.fix_list[at] = j2 * a.width + i2;
                else again = false;                    // list full: the fast pixel stands
            }
            if (again) ray.step_count = 0;             // its steps are counted by the strict re-march
        }
        if (valid2 && !again) ray.finish_at(a, i2, j2);
    }
    // a lane executes one step per loop iteration: its step count is the number of steps it executed (0: no ray)
    unsigned long long tot = wave_sum_u32((unsigned int)ray.step_count);
    if (lane == 0) {
        atomicAdd(a.ray_steps + (size_t)(blockIdx.x & (BHR_STEP_LANES - 1)) * BHR_STEP_STRIDE, tot);
        // BHR_ROW_COSTS: cost profile over tile rows = ray-steps + the wave's shading passes, each priced as
        // BHR_FLUSH_COST wave-steps (a pass is ~1000 instructions, a strict step ~220)
        if (a.row_steps && tile < a.n_tiles) atomicAdd(a.row_steps + ty, tot + (unsigned long long)flushes * (64u * BHR_FLUSH_COST));
#if BHR_WAVE_STAMPS_BUILD
        if (a.wave_stamps && slot < a.n_tiles) {          // diagnostic: when this wave lived (100 MHz ticks) and what it did
            unsigned long long *w = a.wave_stamps + (size_t)slot * 4;
            w[0] = t_start;
            w[1] = __builtin_amdgcn_s_memrealtime();
            w[2] = tot | ((unsigned long long)flushes << 40);
            w[3] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        }
#endif
    }
}

template <bool DIFF, int SRC = 0, bool COSTS = false>
__global__ __launch_bounds__(256) void march_tile_kernel(BhrMarchArgs a) {
    march_tile_body<DIFF, SRC, false, COSTS>(a, blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

#if !BHR_MARCH_STRICT
// BASELINE.json's north star asks for "mipmap levels staged through LDS".  Opt-in (BHR_MIP_LDS=1, fast arithmetic,
// anti-aliased views): every block copies the coarse levels of the packed mip stack -- as many of levels 3, 2, 1 as fit
// 44 KB -- into LDS before it marches, and _sample_disk_mip reads those levels from there.  Not the default, by
// measurement: the texture gathers are cache resident (FETCH_SIZE 0.36x the algorithmic bytes) and the kernel
// is issue bound, while 40 KB of LDS beside the parking slots leave two blocks per CU; and the BASELINE textures' level 3 (4.8 MB at 4k) does not
// fit any LDS -- the launcher falls back to the plain kernel when nothing fits.
__global__ __launch_bounds__(256) void march_tile_mipstaged_kernel(BhrMarchArgs a) {
    const int from = a.mip_lds_from;
    const int n = a.sc.mip_off[3] + a.sc.mip_h[3] * a.sc.mip_w[3] - a.sc.mip_off[from];     // levels from .. 3: the LOD is clamped to 3
    const float4 *src = a.sc.mips + a.sc.mip_off[from];
    for (int k = threadIdx.x; k < n; k += 256) g_mip_lds[k] = src[k];
    __syncthreads();
    march_tile_body<true, 3, false, false>(a, blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// The plain fast march (no differentials, texture source) as an entry of its own, so that its occupancy target can be set
// without touching the other instantiations of the template: BHR_FAST_WAVES waves per SIMD (512 / waves registers per lane).
#ifndef BHR_FAST_WAVES
#define BHR_FAST_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BHR_FAST_WAVES, BHR_FAST_WAVES))) void march_tile_plain_fast(BhrMarchArgs a) {
    march_tile_body<false, 0, false, false>(a, blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

template <bool DIFF, bool COSTS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DIFF ? 4 : 5, DIFF ? 4 : 5))) void march_tile_guard_kernel(BhrMarchArgs a) {
    march_tile_body<DIFF, 0, true, COSTS>(a, blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}
#endif

#if BHR_MARCH_STRICT && BHR_MARCH_ILP
// Second half of the hybrid march's fast list: the pixels march_tile_guard_kernel put on the fix list, 64 per wave whatever
// tile they came from, marched with the strict Ray -- bit-identical to math_mode 1.  Launched with a grid for the list's
// capacity; waves beyond the count the device holds exit at once.
template <bool DIFF>
__global__ __launch_bounds__(256) void march_fix_kernel(BhrMarchArgs a) {
    const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int n = *a.fix_count;
    if (n > (unsigned int)a.fix_cap) n = (unsigned int)a.fix_cap;
    if ((unsigned int)wave * 64u >= n) return;
    const int lane = threadIdx.x & 63;
    const unsigned int k = (unsigned int)wave * 64u + (unsigned int)lane;
    const bool valid = k < n;
    const int pix = valid ? a.fix_list[k] : 0;
    Ray<DIFF, 0> ray;
    ray.init(a, pix % a.width, pix / a.width);
    if (!valid) ray.done = 4;
    while (ray.done == 0) {
        ray.step(a);
        if (ray.full) { ray.flush_one(a); ray.full = false; }
    }
    if (__ballot(ray.n_pend > 0)) ray.flush_one(a);
    if (__ballot(ray.n_pend > 0)) ray.flush_one(a);
    if (valid) ray.finish_at(a, pix % a.width, pix / a.width);
    // the row-cost profile (BHR_ROW_COSTS): the guard kernel left these pixels' steps out, they are strict steps of their row band
    if (a.row_steps && valid) atomicAdd(a.row_steps + (pix / a.width) / 8, (unsigned long long)ray.step_count);
    const unsigned long long tot = wave_sum_u32((unsigned int)ray.step_count);
    if (lane == 0) atomicAdd(a.ray_steps + (size_t)(blockIdx.x & (BHR_STEP_LANES - 1)) * BHR_STEP_STRIDE, tot);
}

// The ILP-scheduled object launches two kernels, each with the occupancy its register allocation should aim for
// (A/B on fhd / 4k, isolated launches): plain texture march at 5 waves per SIMD (96 VGPRs, no spills; 0.697 -> 0.692 ms,
// 6 waves: 0.695), AA march at 4 (128 VGPRs; 6.50 -> 6.39 ms at 4k against the default scheduler).
// BHR_TPW tiles per wave, one after the other (a fixed, uniform trip count; wave w takes tiles w, w + W, w + 2W ... of
// the launch order, W = waves in the launch).  Measured for the default: the per-wave timeline (tools/wave_timeline.py)
// shows 88-92 % slot occupancy in the body of an fhd launch and a ~90 us ragged end; blocks of 64 threads (4x the
// workgroups) take 0.89 ms instead of 0.69, so the workgroup dispatcher matters -- but 2 / 3 / 4 tiles per wave do not
// buy it back (0.696 / 0.717 / 0.726 ms against 0.679 at one; 1501 fps with two frames in flight at 2, 1498 at 1).
// One tile per wave stays.  A dynamic tile queue (resident waves popping tiles from a counter) was tried twice: with a
// data-dependent exit it compiled into a non-terminating loop, with a fixed trip count it ran correctly at 1.08-1.23 ms
// whatever the grid (each wave is latency-bound at ~15 cycles per instruction, so fewer, longer-lived waves only
// lengthen the critical path); both removed.
#ifndef BHR_TPW
#define BHR_TPW 1
#endif
template <bool DIFF>
__device__ __forceinline__ void march_tiles_of_wave(const BhrMarchArgs &a) {
    const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
#pragma unroll 1
    for (int t = 0; t < BHR_TPW; ++t) {
        const int slot = wave + t * n_waves;
        if (slot < a.n_list) march_tile_body<DIFF, 0>(a, slot);
    }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void march_tile_plain_ilp(BhrMarchArgs a) { march_tiles_of_wave<false>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void march_tile_aa_ilp(BhrMarchArgs a) { march_tiles_of_wave<true>(a); }

#endif

// ---------------------------------------------------------------------------
// persistent schedule: waves pull pixels from a queue (8x8-tile-major order, so refilled lanes
// stay spatially coherent) and refill dead lanes when fewer than `refill_below` are alive:
// __ballot gives the live mask, popcount of the lower lanes the slot of each lane that wants work.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool tile_pixel(const BhrMarchArgs &a, unsigned int w, int &i, int &j) {
    unsigned int tile = w >> 6, in = w & 63u;
    if ((int)tile >= a.n_tiles) return false;
    int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    i = tx * 8 + (in & 7);
    j = ty * 8 + (in >> 3);
    return i < a.width && j < a.rows;
}

template <bool DIFF>
__global__ __launch_bounds__(256) void march_persistent_kernel(BhrMarchArgs a, int refill_below) {
    const int lane = threadIdx.x & 63;
    const unsigned int total = (unsigned int)a.n_tiles * 64u;
    Ray<DIFF> ray;
    ray.done = 4;  // empty lane: no pixel, nothing parked, nothing accumulated
    ray.pix = -1;
    ray.n_pend = 0;
    ray.sh.accum = mk(0, 0, 0);
    ray.sh.alpha_total = 0.0f;
    unsigned int executed = 0;
    bool queue_empty = false;

    for (;;) {
        unsigned long long live = __ballot(ray.done == 0);
        int n_live = __popcll(live);
        if (!queue_empty && n_live < refill_below) {
            // retire finished lanes, then hand every non-running lane a new pixel
            for (int k = 0; k < 2; ++k)
                if (__ballot(ray.n_pend > 0 && ray.done != 0)) {
                    if (ray.done != 0) ray.flush_one(a);
                }
            if (ray.done >= 1 && ray.done <= 3) ray.finish(a);
            unsigned long long want = ~live;
            int n_want = 64 - n_live;
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(a.queue, (unsigned int)n_want);
            base = __shfl(base, 0, BHR_WAVE);
            unsigned long long below = want & ((1ull << lane) - 1ull);
            unsigned int w = base + (unsigned int)__popcll(below);
            if ((want >> lane) & 1ull) {   // running lanes keep their ray
                ray.done = 4;
                ray.pix = -1;
                ray.n_pend = 0;       // a lane that gets no pixel (queue exhausted) must not look as if it had a hit parked
                int i, j;
                if (w < total && tile_pixel(a, w, i, j)) ray.init(a, i, j);
            }
            if (base + (unsigned int)n_want >= total) queue_empty = true;
            live = __ballot(ray.done == 0);
            if (!live && queue_empty) break;
            continue;
        }
        if (!live) {
            for (int k = 0; k < 2; ++k)
                if (__ballot(ray.n_pend > 0)) ray.flush_one(a);
            if (ray.done >= 1 && ray.done <= 3) ray.finish(a);
            break;
        }
        bool blocked = false;
        if (ray.done == 0) {
            blocked = !ray.step(a);
            executed += blocked ? 0u : 1u;
        }
        if (__ballot(blocked || ray.n_pend == 2)) ray.flush_one(a);
    }
    unsigned long long tot = wave_sum_u32(executed);
    if (lane == 0) atomicAdd(a.ray_steps + (size_t)(blockIdx.x & (BHR_STEP_LANES - 1)) * BHR_STEP_STRIDE, tot);
}

#if BHR_MARCH_STRICT
// ---- self-test of the hand-written exact arithmetic against hipcc's IEEE sequences ----------
__device__ __forceinline__ unsigned int lcg(unsigned int &s) { s = s * 1664525u + 1013904223u; return s; }
__global__ void selftest_kernel(unsigned long long *out, unsigned int div_rounds) {
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    unsigned long long bad_sqrt = 0, bad_div = 0, bad_div6 = 0, n = 0;
    // every f32 in [2^-80, 2^80): exponent field 47..206, all significands
    for (unsigned long long k = tid; k < 160ull << 23; k += nthreads) {
        float x = __uint_as_float((unsigned int)(k + (47ull << 23)));
        bad_sqrt += sqrt_rn(x) != sqrtf(x);
        bad_div += rcp_rn(x) != 1.0f / x;
        bad_div += rcp_rn(-x) != 1.0f / -x;
        bad_div6 += div6(x) != x / 6.0f;
        bad_div6 += div6(-x) != -x / 6.0f;
        n += 5;
    }
    // random pairs: a, b with exponents in [2^-24, 2^24), random significands and signs
    unsigned int st = tid * 2654435761u + 12345u;
    for (unsigned int k = 0; k < div_rounds; ++k) {
        unsigned int ra = lcg(st), rb = lcg(st), re = lcg(st);
        unsigned int ea = 103u + (re & 0xffffu) % 48u, eb = 103u + (re >> 16) % 48u;
        float a = __uint_as_float((ra & 0x807fffffu) | (ea << 23));
        float b = __uint_as_float((rb & 0x807fffffu) | (eb << 23));
        bad_div += div_rn(a, b) != a / b;
        bad_div += div_rn(1.0f, b) != 1.0f / b;
        n += 2;
    }
    atomicAdd(out + 0, bad_sqrt);
    atomicAdd(out + 1, bad_div);
    atomicAdd(out + 2, bad_div6);
    atomicAdd(out + 3, n);
}
#endif

}  // namespace

#if BHR_MARCH_STRICT && !BHR_MARCH_ILP
int32_t bhr_selftest_strict(bhr_ctx *ctx, unsigned long long *d_out4) {
    BHR_HIP(hipMemsetAsync(d_out4, 0, 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(selftest_kernel, dim3(2048), dim3(256), 0, ctx->stream, d_out4, 2048u);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}
#endif

// Tile order of this context: 8x8 tiles sorted by the distance of their centre from the centre of the FULL
// image (the camera looks at the hole, build_camera), nearest first.  Built once per context.
static int32_t ensure_tile_order(bhr_ctx *ctx, int tiles_x, int n_tiles) {
    if (ctx->d_tile_order && ctx->tile_order_n == n_tiles) return BHR_OK;
    if (ctx->d_tile_order) (void)hipFree(ctx->d_tile_order);
    ctx->d_tile_order = nullptr;
    std::vector<std::pair<float, int>> key((size_t)n_tiles);
    const float cx = 0.5f * (float)ctx->cfg.width, cy = 0.5f * (float)ctx->cfg.height;
    for (int t = 0; t < n_tiles; ++t) {
        const float x = (float)((t % tiles_x) * 8 + 4) - cx, y = (float)(ctx->cfg.row0 + (t / tiles_x) * 8 + 4) - cy;
        key[(size_t)t] = {x * x + y * y, t};
    }
    std::stable_sort(key.begin(), key.end(),
                     [](const std::pair<float, int> &l, const std::pair<float, int> &r) { return l.first < r.first; });
    // the host keeps a copy: hybrid and pipelined launches partition this order into sub-lists (api.hip, hybrid.hip)
    free(ctx->h_tile_order);
    ctx->h_tile_order = (int32_t *)malloc((size_t)n_tiles * sizeof(int32_t));
    if (!ctx->h_tile_order) return bhr_fail(BHR_ERR_NOMEM, "tile order: out of host memory");
    for (int t = 0; t < n_tiles; ++t) ctx->h_tile_order[t] = key[(size_t)t].second;
    BHR_HIP(hipMalloc((void **)&ctx->d_tile_order, (size_t)n_tiles * sizeof(int32_t)));
    BHR_HIP(hipMemcpy(ctx->d_tile_order, ctx->h_tile_order, (size_t)n_tiles * sizeof(int32_t), hipMemcpyHostToDevice));
    ctx->tile_order_n = n_tiles;
    return BHR_OK;
}

#if !BHR_MARCH_STRICT
int32_t bhr_ensure_tile_order(bhr_ctx *ctx) {
    const int tiles_x = (ctx->cfg.width + 7) / 8;
    return ensure_tile_order(ctx, tiles_x, tiles_x * ((ctx->rows + 7) / 8));
}
#endif

int32_t BHR_MARCH_RESOURCES(int32_t *vgprs, int32_t *lds, int32_t diff) {
    hipFuncAttributes at;
#if BHR_MARCH_STRICT && BHR_MARCH_ILP
    const void *f = diff ? (const void *)march_tile_aa_ilp : (const void *)march_tile_plain_ilp;
#elif !BHR_MARCH_STRICT
    const void *f = diff ? (const void *)march_tile_kernel<true, 0> : (const void *)march_tile_plain_fast;
#else
    const void *f = diff ? (const void *)march_tile_kernel<true, 0> : (const void *)march_tile_kernel<false, 0>;
#endif
    BHR_HIP(hipFuncGetAttributes(&at, f));
    *vgprs = at.numRegs;
    *lds = (int32_t)at.sharedSizeBytes;
    return BHR_OK;
}

int32_t BHR_LAUNCH_MARCH(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    const bhr_config &c = ctx->cfg;
#if !BHR_MARCH_STRICT
    if (!(ctx->part.active && (ctx->part.math_resolved || ctx->part.n <= 0))) {   // an empty part only records the bracket events
        int mode = c.math_mode;
        if (flags & BHR_FORCE_FAST) mode = BHR_MATH_FAST;
        if (flags & BHR_FORCE_STRICT) mode = BHR_MATH_STRICT;
        if (flags & BHR_FORCE_HYBRID) mode = BHR_MATH_HYBRID;
        // hybrid = two launches over complementary tile lists (hybrid.hip); schedules and disk sources that have no
        // list form run strict
        if (mode == BHR_MATH_HYBRID && (ctx->disk_source != BHR_DISK_TEXTURE || (flags & BHR_PERSISTENT))) mode = BHR_MATH_STRICT;
        // (the frame's post-pass kernels were chosen by bhr_frame_begin from the same decision: api.hip)
        if (mode == BHR_MATH_HYBRID) return bhr_launch_march_hybrid(ctx, cam, flags);
        if (mode == BHR_MATH_STRICT) return bhr_launch_march_strict(ctx, cam, flags);
    }
#endif
#if BHR_MARCH_STRICT && !BHR_MARCH_ILP
    // the texture kernels (plain and AA) live in the ILP-scheduled object (see the top of this file)
    if (ctx->disk_source == BHR_DISK_TEXTURE && !(flags & BHR_PERSISTENT))
        return bhr_launch_march_strict_ilp(ctx, cam, flags);
#endif
    if (!ctx->d_skybox) return bhr_fail(BHR_ERR_STATE, "bhr_render: no skybox set (bhr_set_skybox)");
    if (!ctx->d_mips) return bhr_fail(BHR_ERR_STATE, "bhr_render: no disk texture set (bhr_set_disk_texture)");

    BhrMarchArgs a;
    for (int k = 0; k < 3; ++k) {
        a.cp[k] = cam->pos[k];
        a.cr[k] = cam->right[k];
        a.cu[k] = cam->up[k];
        a.cf[k] = cam->forward[k];
    }
    a.pw = cam->pixel_width;
    a.ph = cam->pixel_height;
    a.r_esc = cam->r_escape;
    a.r_esc2 = a.r_esc * a.r_esc;
    a.h_base = c.step_size;
    a.r_inner = c.r_disk_inner;
    a.r_outer = c.r_disk_outer;
    a.t_offset = cam->t_offset;
    // render.py:2808: tilt_rad = disk_tilt * pi / 180 in f32
    a.tilt_rad = c.disk_tilt_deg * BHR_PI_F / 180.0f;
    a.tan_t = tanf(a.tilt_rad);
    a.sin_t = sinf(a.tilt_rad);
    a.cos_t = cosf(a.tilt_rad);
    a.aa_strength = c.aa_strength;
    // orbital-plane basis shared by all rays (fast build): e1 = cam / |cam|
    a.r0 = sqrtf(a.cp[0] * a.cp[0] + a.cp[1] * a.cp[1] + a.cp[2] * a.cp[2]);
    for (int k = 0; k < 3; ++k) a.e1[k] = a.cp[k] / a.r0;
    a.A = a.e1[2] - a.e1[1] * a.tan_t;
    // render.py:2817-2818
    a.max_iter = (int32_t)(a.r_esc * 40.0f / a.h_base);
    a.max_affine = a.r_esc * 40.0f;
    a.max_affine_u = a.max_affine / a.h_base;      // fast build: the affine parameter in units of h_base
    a.width = c.width;
    a.height = c.height;
    a.row0 = c.row0;
    a.rows = ctx->rows;
    a.sc.skybox = ctx->d_skybox;
    a.sc.sky_h = ctx->sky_h;
    a.sc.sky_w = ctx->sky_w;
    a.sc.mips = ctx->d_mips;
    for (int l = 0; l < BHR_NUM_MIP_LEVELS; ++l) {
        a.sc.mip_off[l] = ctx->mip_off[l];
        a.sc.mip_h[l] = ctx->mip_h[l];
        a.sc.mip_w[l] = ctx->mip_w[l];
    }
    a.sc.n_r = ctx->n_r;
    a.sc.n_phi = ctx->n_phi;
    a.bg = ctx->d_bg;
    a.disk = ctx->d_disk;
    a.diskp = nullptr;
    a.dp_yb = a.dp_gp = a.dp_g0 = 0;
    a.sum = nullptr;
    if (ctx->bloom_split && ctx->d_pa && ctx->d_sum && !(flags & BHR_SKIP_BLOOM)) {      // split-f16 post-pass: the march feeds its H pass directly
        bhr_split_geom g;
        bhr_split_geometry(ctx, &g);
        a.diskp = (_Float16 *)ctx->d_pa;
        a.dp_yb = g.YB;
        a.dp_gp = g.GP;
        a.dp_g0 = g.g0;
        a.sum = ctx->d_sum;
        ctx->slots[ctx->active_slot].sum_valid = 1;
    }
    // timed launches (bhr_render) count into their ring slot; group launches into the scalar
    const int slot = ctx->cur_slot;
    a.ray_steps = slot >= 0 ? ctx->d_steps_ring + (size_t)slot * BHR_STEP_CELL : ctx->d_ray_steps;
    a.queue = ctx->d_queue;
    a.dv2 = ctx->disk_source != BHR_DISK_TEXTURE ? ctx->d_dv2_params : nullptr;
    a.vol_absorption = ctx->vol_opts[0];
    a.vol_grazing_gain = ctx->vol_opts[1];
    a.vol_h_max = ctx->vol_opts[2];
    a.vol_r_max = ctx->vol_opts[3];
    a.vol_substeps = ctx->vol_substeps;
    a.dv2_norm_shear = ctx->dv2_norm[0];
    a.dv2_norm_hotspot = ctx->dv2_norm[1];
    a.dv2_t_peak = ctx->dv2_norm[2];
    a.tiles_x = (c.width + 7) / 8;
    a.n_tiles = a.tiles_x * ((ctx->rows + 7) / 8);
    a.n_list = a.n_tiles;
    a.fix_count = ctx->fix_count;
    a.mip_lds_from = -1;
    a.fix_list = ctx->fix_list;
    a.fix_cap = ctx->fix_cap;
    // a partial launch (ctx->part: hybrid arithmetic, pipelined row bands) marches the tiles of a caller-made list; the
    // first part records the start event and clears an untimed counter, the last part records the end event
    const bhr_march_part part = ctx->part;
    const bool first_part = !part.active || part.first, last_part = !part.active || part.last;
    a.row_steps = nullptr;
    if (flags & BHR_ROW_COSTS) {
        // two profiles side by side: [0, n) the steps taken by the fast arithmetic, [n, 2n) by the strict one (a hybrid frame
        // fills both, from its two tile lists); cleared by the frame's first part, on the stream every other part follows
        const size_t n = (size_t)((ctx->rows + 7) / 8);
        if (!ctx->d_row_steps) BHR_HIP(hipMalloc((void **)&ctx->d_row_steps, 2 * n * sizeof(unsigned long long)));
        if (first_part) BHR_HIP(hipMemsetAsync(ctx->d_row_steps, 0, 2 * n * sizeof(unsigned long long), ctx->stream));
        a.row_steps = ctx->d_row_steps + (BHR_MARCH_STRICT ? n : 0);
    }
    a.wave_stamps = nullptr;
    // diagnostic (builds with -DBHR_WAVE_STAMPS_BUILD=1 only: the stamps cost the plain kernel three spilled registers):
    // BHR_WAVE_STAMPS=<file> dumps per-wave start / end times of THIS launch (tools/wave_timeline.py)
#if BHR_WAVE_STAMPS_BUILD
    const char *stamp_path = getenv("BHR_WAVE_STAMPS");
#else
    const char *stamp_path = nullptr;                          // the shipped library reads no environment on the render path
#endif
    unsigned long long *d_stamps = nullptr;
    if (stamp_path && stamp_path[0]) {
        BHR_HIP(hipMalloc((void **)&d_stamps, (size_t)a.n_tiles * 4 * sizeof(unsigned long long)));
        BHR_HIP(hipMemsetAsync(d_stamps, 0, (size_t)a.n_tiles * 4 * sizeof(unsigned long long), ctx->stream));
        a.wave_stamps = d_stamps;
    }
    a.tile_order = nullptr;
    if (part.active) {
        a.tile_order = part.d_list;
        a.n_list = part.n;
    } else {
        if (!ctx->opt.tile_order_rows) {                    // BHR_TILE_ORDER: "centre" (default) | "row": row-major, for A/B runs
            BHR_TRY(ensure_tile_order(ctx, a.tiles_x, a.n_tiles));
            a.tile_order = ctx->d_tile_order;
        }
    }

    // anti_alias "disabled": the reference still integrates the differentials (skip_diff = 0 on
    // the CLI path) but never reads them (render.py:2957-2959) => skipping them is pixel-identical.
    const bool want_diff = c.anti_alias != 0 && !(flags & BHR_SKIP_DIFFERENTIALS);

    // ring cells are cleared ahead of time (at reset, then by the previous frame's last kernel)
    if (slot < 0 && first_part) BHR_HIP(hipMemsetAsync(a.ray_steps, 0, sizeof(unsigned long long) * BHR_STEP_CELL, ctx->stream));
    if (flags & BHR_PERSISTENT) BHR_HIP(hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned int), ctx->stream));
    // timed launches (bhr_render) use their ring slot's events, the others the context's scalar ones
    if (first_part) BHR_HIP(hipEventRecord(slot >= 0 ? ctx->ring_ev[slot * 3 + 0] : ctx->ev[0], ctx->stream));
    if (part.active && part.n <= 0 && part.repair != 2) {
        // empty list: nothing to launch
    } else if (!(flags & BHR_PERSISTENT) || a.dv2 || a.row_steps || part.active) {   // the persistent schedule has no Disk V2 / row-cost variant
        // waves per block: a block keeps its CU slot until its slowest wave has finished, so small
        // blocks shorten the tail; BHR_TILE_BLOCK overrides for experiments
        const int bt = ctx->opt.tile_block;               // 256; BHR_TILE_BLOCK = 64 / 128 for experiments
        const int wpb = bt / 64;
        dim3 grid((a.n_list + wpb - 1) / wpb), block(bt);
        if (ctx->disk_source == BHR_DISK_V2_VOLUME) {   // finite-thickness Disk V2: no texture footprint to track
            hipLaunchKernelGGL((march_tile_kernel<false, 2>), grid, block, 0, ctx->stream, a);
        } else if (a.dv2) {   // analytic Disk V2 source: its own instantiations
            if (want_diff)
                hipLaunchKernelGGL((march_tile_kernel<true, 1>), grid, block, 0, ctx->stream, a);
            else
                hipLaunchKernelGGL((march_tile_kernel<false, 1>), grid, block, 0, ctx->stream, a);
#if BHR_MARCH_STRICT && BHR_MARCH_ILP
        } else {
            const int waves = (a.n_list + BHR_TPW - 1) / BHR_TPW;          // BHR_TPW tiles per wave
            const dim3 g((waves + wpb - 1) / wpb);
            if (part.active && part.repair == 2) {                          // the fix list of a hybrid march
                const dim3 gf((a.fix_cap / 64 + wpb - 1) / wpb);
                if (want_diff) hipLaunchKernelGGL(march_fix_kernel<true>, gf, block, 0, ctx->stream, a);
                else hipLaunchKernelGGL(march_fix_kernel<false>, gf, block, 0, ctx->stream, a);
            } else if (want_diff)
                hipLaunchKernelGGL(march_tile_aa_ilp, g, block, 0, ctx->stream, a);
            else
                hipLaunchKernelGGL(march_tile_plain_ilp, g, block, 0, ctx->stream, a);
        }
#else
#if !BHR_MARCH_STRICT
        } else if (part.active && part.repair == 1) {           // fast list of a hybrid march: guards + fix list
            if (a.row_steps) {                                  // the row-cost probe: the instantiations that count shading passes
                if (want_diff) hipLaunchKernelGGL((march_tile_guard_kernel<true, true>), grid, block, 0, ctx->stream, a);
                else hipLaunchKernelGGL((march_tile_guard_kernel<false, true>), grid, block, 0, ctx->stream, a);
            } else if (want_diff) hipLaunchKernelGGL((march_tile_guard_kernel<true>), grid, block, 0, ctx->stream, a);
            else hipLaunchKernelGGL((march_tile_guard_kernel<false>), grid, block, 0, ctx->stream, a);
        } else if (a.row_steps) {
            if (want_diff) hipLaunchKernelGGL((march_tile_kernel<true, 0, true>), grid, block, 0, ctx->stream, a);
            else hipLaunchKernelGGL((march_tile_kernel<false, 0, true>), grid, block, 0, ctx->stream, a);
#endif
        } else if (want_diff) {
#if !BHR_MARCH_STRICT
            // BHR_MIP_LDS=1: the coarse mip levels through LDS where any of them fits 44 KB (see the kernel)
            size_t staged_bytes = 0;
            {
                if (ctx->opt.mip_lds && !part.active && bt == 256) {
                    const int last = 3;                                     // int(clamp(lod, 0, 3)): the coarsest level ever sampled
                    for (int l = last; l >= 1; --l) {
                        if (a.sc.mip_h[last] <= 0 || a.sc.mip_w[last] <= 0) break;                  // a texture too small to have it
                        const size_t bytes = ((size_t)a.sc.mip_off[last] + (size_t)a.sc.mip_h[last] * a.sc.mip_w[last] - (size_t)a.sc.mip_off[l]) * sizeof(float4);
                        if (bytes > 44 * 1024) break;                       // 64 KB per block less the 18 KB of parking slots
                        a.mip_lds_from = l;
                        staged_bytes = bytes;
                    }
                }
            }
            if (a.mip_lds_from >= 0) {
                ctx->mip_lds_from = a.mip_lds_from;
                hipLaunchKernelGGL(march_tile_mipstaged_kernel, grid, block, staged_bytes, ctx->stream, a);
            } else {
                ctx->mip_lds_from = -1;
                hipLaunchKernelGGL((march_tile_kernel<true, 0>), grid, block, 0, ctx->stream, a);
            }
#else
            hipLaunchKernelGGL((march_tile_kernel<true, 0>), grid, block, 0, ctx->stream, a);
#endif
        } else {
#if !BHR_MARCH_STRICT
            hipLaunchKernelGGL(march_tile_plain_fast, grid, block, 0, ctx->stream, a);
#else
            hipLaunchKernelGGL((march_tile_kernel<false, 0>), grid, block, 0, ctx->stream, a);
#endif
        }
#endif
    } else {
        // enough resident waves to fill the chip; every wave drains the queue and exits
        int blocks = (a.n_tiles + 3) / 4;
        const int max_blocks = 256 * 8;
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks < 1) blocks = 1;
        dim3 grid(blocks), block(256);
        const int refill_below = 40;
        if (want_diff)
            hipLaunchKernelGGL(march_persistent_kernel<true>, grid, block, 0, ctx->stream, a, refill_below);
        else
            hipLaunchKernelGGL(march_persistent_kernel<false>, grid, block, 0, ctx->stream, a, refill_below);
    }
    BHR_HIP(hipGetLastError());
    // group / tile renders (slot < 0) record the march's end only on request: the event is a ~5 us bubble between the march and
    // the H pass of a tile whose whole tail is ~0.12 ms
    if (last_part && (slot >= 0 || ctx->group_time_march)) BHR_HIP(hipEventRecord(slot >= 0 ? ctx->ring_ev[slot * 3 + 1] : ctx->ev[1], ctx->stream));
    if (last_part) ctx->march_end_recorded = slot >= 0 || ctx->group_time_march;
    if (d_stamps) {
        std::vector<unsigned long long> h((size_t)a.n_tiles * 4);
        BHR_HIP(hipMemcpyAsync(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(d_stamps);
        if (FILE *f = fopen(stamp_path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    ctx->last_steps_ptr = a.ray_steps;
    ctx->counters.rays = (uint64_t)c.width * ctx->rows;
    return BHR_OK;
}
