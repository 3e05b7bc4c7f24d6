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
//  * tile      : a wave owns one 8x8 pixel tile and loops until __ballot says
//                no lane is alive (lane efficiency ~0.95 for the default view);
//  * persistent: waves pull 8x8 tiles from a global queue; when the number of
//                live lanes drops below a threshold the dead lanes write their
//                pixel and are refilled from the next tile (wave-level
//                __ballot / mbcnt compaction of the *work*, not of registers).
//
// Arithmetic differs from a strict f32 evaluation of the reference only in rounding: v_rsq/v_rcp/v_sqrt
// instead of IEEE sqrt + divide inside the RK4 stages, FMA contraction, and the
// re-use of |new_pos| as the next step's |pos| (same value in the reference).
#include "bhr_internal.h"

#ifndef BHR_PRECISE_MARCH
#define BHR_PRECISE_MARCH 0
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

#if BHR_PRECISE_MARCH
__device__ __forceinline__ float q_rsq(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ float q_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float q_sqrt(float x) { return sqrtf(x); }
#else
__device__ __forceinline__ float q_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float q_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float q_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
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
    return mk(c00[0] * w00 + c10[0] * w10 + c01[0] * w01 + c11[0] * w11,
              c00[1] * w00 + c10[1] * w10 + c01[1] * w01 + c11[1] * w11,
              c00[2] * w00 + c10[2] * w10 + c01[2] * w01 + c11[2] * w11);
}

// ---- _sample_disk / _sample_disk_mip (render.py:2568-2637) -------------------
// lod_i = 0 reproduces _sample_disk exactly (level 0 of the mip stack is the
// texture itself and n / 2^0 = n).
__device__ __forceinline__ float4 sample_disk_level(const BhrScene &sc, float hit_x, float hit_y, float r_inner,
                                                    float r_outer, float t_offset, int lod_i) {
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
    const float4 *t = sc.mips + sc.mip_off[lod_i];
    const int stride = sc.mip_w[lod_i];
    float4 c00 = t[(size_t)v0_h * stride + u0_w];
    float4 c10 = t[(size_t)v0_h * stride + u1_w];
    float4 c01 = t[(size_t)v1_h * stride + u0_w];
    float4 c11 = t[(size_t)v1_h * stride + u1_w];
    float w00 = (1 - fu) * (1 - fv), w10 = fu * (1 - fv), w01 = (1 - fu) * fv, w11 = fu * fv;
    return make_float4(c00.x * w00 + c10.x * w10 + c01.x * w01 + c11.x * w11,
                       c00.y * w00 + c10.y * w10 + c01.y * w01 + c11.y * w11,
                       c00.z * w00 + c10.z * w10 + c01.z * w01 + c11.z * w11,
                       c00.w * w00 + c10.w * w10 + c01.w * w01 + c11.w * w11);
}

// ---- _apply_g_factor (render.py:2439-2516) ----------------------------------
__device__ __forceinline__ V3 apply_g_factor(const BhrMarchArgs &a, V3 base_color, V3 hit_pos, float hit_r,
                                             V3 ray_dir_to_cam) {
    const float rs_f = BHR_RS;
    V3 cam_pos = ld3(a.cp);
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

// Acceleration coefficient c(s) with a(s) = c * s:  -1.5 L2 / r^5  (render.py:2518-2524);
// also hands back 1/r^2 for the Jacobian (render.py:2526-2539).
__device__ __forceinline__ float accel_coef(V3 s, float m15L2, float &inv_r2) {
    float r2 = dot(s, s);
    float ir = q_rsq(r2);
    inv_r2 = ir * ir;
    return m15L2 * (inv_r2 * inv_r2 * ir);
}
// J(s) applied to delta: c * (delta - 5 s (s.delta)/r^2)
__device__ __forceinline__ V3 jac(V3 s, V3 delta, float c, float inv_r2) {
    float proj5 = 5.0f * dot(s, delta) * inv_r2;
    return c * mk(fmaf(-proj5, s.x, delta.x), fmaf(-proj5, s.y, delta.y), fmaf(-proj5, s.z, delta.z));
}

template <bool DIFF>
struct Ray {
    V3 p, d;
    float m15L2;   // -1.5 * L2
    float r;       // |p|
    float c1;      // accel coefficient at p
    float ir2;     // 1/|p|^2
    float f_old;   // plane function at p
    float affine;
    V3 accum;
    float alpha_total;
    int step_count;
    int pix;       // linear pixel index inside the row block, -1 = lane has no ray
    int done;      // 0 running, 1 captured, 2 escaped, 3 ran out of iterations
    V3 esc;
    // ray differentials (DIFF only)
    V3 dpx, ddx, dpy, ddy;

    __device__ __forceinline__ void init(const BhrMarchArgs &a, int i, int j_local) {
        const V3 cp = ld3(a.cp), cr = ld3(a.cr), cu = ld3(a.cu), cf = ld3(a.cf);
        // render.py:2811-2812, 2820-2828
        V3 center = cp + 1.0f * cf;
        float half_w = a.pw * (float)a.width / 2;
        float half_h = a.ph * (float)a.height / 2;
        V3 tl = (center - half_w * cr) + half_h * cu;
        float px_f = (float)i, py_f = (float)(j_local + a.row0);
        V3 pixel_pos = (tl + ((px_f + 0.5f) * a.pw) * cr) - ((py_f + 0.5f) * a.ph) * cu;
        V3 ray_dir = normalized(pixel_pos - cp);
        p = cp;
        d = ray_dir;
        V3 Lv = cross(d, p);
        float Ln = sqrtf(dot(Lv, Lv));
        m15L2 = -1.5f * (Ln * Ln);
        float r2 = dot(p, p);
        r = sqrtf(r2);
        c1 = accel_coef(p, m15L2, ir2);
        f_old = p.z - p.y * a.tan_t;
        affine = 0.0f;
        accum = mk(0, 0, 0);
        alpha_total = 0.0f;
        step_count = 0;
        done = 0;
        esc = mk(0, 0, 0);
        pix = j_local * a.width + i;
        if (DIFF) {
            V3 ppx1 = (tl + ((px_f + 1.5f) * a.pw) * cr) - ((py_f + 0.5f) * a.ph) * cu;
            ddx = normalized(ppx1 - cp) - ray_dir;
            V3 ppy1 = (tl + ((px_f + 0.5f) * a.pw) * cr) - ((py_f + 1.5f) * a.ph) * cu;
            ddy = normalized(ppy1 - cp) - ray_dir;
            dpx = mk(0, 0, 0);
            dpy = mk(0, 0, 0);
        }
        if (a.max_iter <= 0) done = 3;
    }

    // Disk hit at fraction t of the segment old -> new (render.py:2941-3002).
    __device__ __forceinline__ void shade_hit(const BhrMarchArgs &a, V3 np, float f_new, V3 hdx, V3 hdy,
                                              uint32_t skip_diff) {
        float t_frac = f_old / (f_old - f_new + 1e-8f);
        float hit_x = p.x + t_frac * (np.x - p.x);
        float hit_y = p.y + t_frac * (np.y - p.y);
        float hit_r = sqrtf(hit_x * hit_x + hit_y * hit_y);
        if (!(a.r_outer >= hit_r && hit_r >= a.r_inner)) return;
        float hit_z = hit_y * a.tan_t;
        int lod_i = 0;
        if (DIFF && !skip_diff) {
            // texture-space footprint from the ray differentials (render.py:2964-2988)
            float hit_r_cyl = sqrtf(hit_x * hit_x + hit_y * hit_y + 1e-6f);
            float inv_den = 1.0f / (hit_r_cyl * hit_r_cyl + 1e-6f);
            float ku = (float)a.sc.n_phi / (2.0f * BHR_PI_F);
            float kv = (float)a.sc.n_r / (a.r_outer - a.r_inner);
            float dr_dx = (hit_x * hdx.x + hit_y * hdx.y) / hit_r_cyl;
            float dphi_dx = (-hit_y * hdx.x + hit_x * hdx.y) * inv_den;
            float dudx = dphi_dx * ku, dvdx = dr_dx * kv;
            float dr_dy = (hit_x * hdy.x + hit_y * hdy.y) / hit_r_cyl;
            float dphi_dy = (-hit_y * hdy.x + hit_x * hdy.y) * inv_den;
            float dudy = dphi_dy * ku, dvdy = dr_dy * kv;
            float grad_sq = fmaxf(dudx * dudx + dvdx * dvdx, dudy * dudy + dvdy * dvdy);
            float lod = logf(fmaxf(grad_sq, 1.0f)) / logf(2.0f) * a.aa_strength;
            lod = fminf(fmaxf(lod, 0.0f), 3.0f);
            lod_i = (int)fminf(fmaxf(lod, 0.0f), (float)(BHR_NUM_MIP_LEVELS - 1));
        }
        float4 rgba = sample_disk_level(a.sc, hit_x, hit_y, a.r_inner, a.r_outer, a.t_offset, lod_i);
        float base_alpha = fminf(rgba.w, 0.999f);
        float disk_alpha = 1.0f - powf(1.0f - base_alpha, BHR_DISK_ALPHA_GAIN);
        V3 col = apply_g_factor(a, mk(rgba.x, rgba.y, rgba.z), mk(hit_x, hit_y, hit_z), hit_r, mk(-d.x, -d.y, -d.z));
        float front = 1.0f - alpha_total;
        float wgt = disk_alpha * front;
        accum = mk(fmaf(col.x, wgt, accum.x), fmaf(col.y, wgt, accum.y), fmaf(col.z, wgt, accum.z));
        alpha_total = 1.0f - front * (1.0f - disk_alpha);
    }

    // One iteration of the while-loop at render.py:2854-3006.
    __device__ __forceinline__ void step(const BhrMarchArgs &a, uint32_t skip_diff) {
        // adaptive step (render.py:2858-2869); r_cap = RS = 1
        float r_safe = fmaxf(r, BHR_RS + 1e-3f);
        float far_scale = fminf(q_sqrt(r_safe), 10.0f);
        float q = q_rcp(r_safe);
        float near_damp = q_rcp(fmaf(2.0f * q, q * q, 1.0f));
        float dt_fac = fminf(fmaxf(far_scale * near_damp, 0.2f), 10.0f);
        float h = a.h_base * dt_fac;
        float hh = 0.5f * h;
        float h6 = h * (1.0f / 6.0f);

        // main RK4 (render.py:2872-2882), written on velocities v_k = k_kp / h and
        // accelerations a_k = k_kd / h
        V3 a1 = c1 * p;
        V3 s2 = fma3(hh, d, p);
        V3 v2 = fma3(hh, a1, d);
        float i2, i3, i4;
        float c2 = accel_coef(s2, m15L2, i2);
        V3 a2 = c2 * s2;
        V3 s3 = fma3(hh, v2, p);
        V3 v3 = fma3(hh, a2, d);
        float c3 = accel_coef(s3, m15L2, i3);
        V3 a3 = c3 * s3;
        V3 s4 = fma3(h, v3, p);
        V3 v4 = fma3(h, a3, d);
        float c4 = accel_coef(s4, m15L2, i4);
        V3 a4 = c4 * s4;
        V3 np = fma3(h6, (d + v4) + 2.0f * (v2 + v3), p);
        V3 nd = fma3(h6, (a1 + a4) + 2.0f * (a2 + a3), d);

        V3 ndpx, nddx, ndpy, nddy;
        if (DIFF) {
            if (!skip_diff) {
                // variational RK4 at the same four stage positions (render.py:2888-2911)
                {
                    V3 j1 = jac(p, dpx, c1, ir2);
                    V3 e2 = fma3(hh, ddx, dpx), w2 = fma3(hh, j1, ddx);
                    V3 j2 = jac(s2, e2, c2, i2);
                    V3 e3 = fma3(hh, w2, dpx), w3 = fma3(hh, j2, ddx);
                    V3 j3 = jac(s3, e3, c3, i3);
                    V3 e4 = fma3(h, w3, dpx), w4 = fma3(h, j3, ddx);
                    V3 j4 = jac(s4, e4, c4, i4);
                    ndpx = fma3(h6, (ddx + w4) + 2.0f * (w2 + w3), dpx);
                    nddx = fma3(h6, (j1 + j4) + 2.0f * (j2 + j3), ddx);
                }
                {
                    V3 j1 = jac(p, dpy, c1, ir2);
                    V3 e2 = fma3(hh, ddy, dpy), w2 = fma3(hh, j1, ddy);
                    V3 j2 = jac(s2, e2, c2, i2);
                    V3 e3 = fma3(hh, w2, dpy), w3 = fma3(hh, j2, ddy);
                    V3 j3 = jac(s3, e3, c3, i3);
                    V3 e4 = fma3(h, w3, dpy), w4 = fma3(h, j3, ddy);
                    V3 j4 = jac(s4, e4, c4, i4);
                    ndpy = fma3(h6, (ddy + w4) + 2.0f * (w2 + w3), dpy);
                    nddy = fma3(h6, (j1 + j4) + 2.0f * (j2 + j3), ddy);
                }
            } else {
                ndpx = dpx; nddx = ddx; ndpy = dpy; nddy = ddy;
            }
        }

        float r2n = dot(np, np);
        float irn = q_rsq(r2n);
        float rn = r2n * irn;
        affine += h;

        // termination precedes the plane test (render.py:2916-2926)
        if (rn < BHR_RS) {
            done = 1;
            return;
        }
        if (rn > a.r_esc || affine > a.max_affine) {
            done = 2;
            esc = nd;
            return;
        }
        if (DIFF) {
            // committed BEFORE the hit interpolation (render.py:2928-2932), hence
            // hit_d_pos == new_d_pos in render.py:2947-2949
            dpx = ndpx; ddx = nddx; dpy = ndpy; ddy = nddy;
        }
        float f_new = np.z - np.y * a.tan_t;
        if (f_old * f_new < 0) {
            if (DIFF)
                shade_hit(a, np, f_new, dpx, dpy, skip_diff);
            else
                shade_hit(a, np, f_new, mk(0, 0, 0), mk(0, 0, 0), 1u);
        }
        p = np;
        d = nd;
        r = rn;
        ir2 = irn * irn;
        c1 = m15L2 * (ir2 * ir2 * irn);
        f_old = f_new;
        step_count += 1;
        if (step_count >= a.max_iter) done = 3;
    }

    // render.py:3008-3018
    __device__ __forceinline__ void finish(const BhrMarchArgs &a) {
        V3 bg = mk(0, 0, 0);
        if (done == 2) bg = sample_skybox(a.sc, normalized(esc));
        float k = 1.0f - alpha_total;
        size_t o = (size_t)pix * 3;
        a.bg[o + 0] = bg.x * k;
        a.bg[o + 1] = bg.y * k;
        a.bg[o + 2] = bg.z * k;
        a.disk[o + 0] = fminf(fmaxf(accum.x, 0.0f), 1.0f);
        a.disk[o + 1] = fminf(fmaxf(accum.y, 0.0f), 1.0f);
        a.disk[o + 2] = fminf(fmaxf(accum.z, 0.0f), 1.0f);
    }
};

__device__ __forceinline__ unsigned long long wave_sum_u32(unsigned int v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, BHR_WAVE);
    return s;
}

// Blocks are dealt round-robin over the 8 XCDs (each with its own L2).  Remap the
// linear block id so that one XCD renders a contiguous band of tiles and its L2
// keeps a coherent slice of the disk texture / skybox.  Speed only.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int X = 8;
    int per = n / X, rem = n % X;
    int x = b % X, k = b / X;
    // XCD x owns per (+1 if x < rem) consecutive logical blocks
    int start = x * per + min(x, rem);
    int len = per + (x < rem ? 1 : 0);
    return (k < len) ? start + k : b;  // k >= len cannot happen for a round-robin deal
}

// ---------------------------------------------------------------------------
// tile schedule: block = 4 waves = 4 horizontally adjacent 8x8 tiles (32x8 px)
// ---------------------------------------------------------------------------
template <bool DIFF>
__global__ __launch_bounds__(256) void march_tile_kernel(BhrMarchArgs a, uint32_t skip_diff) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nblk = gridDim.x;
    const int b = xcd_remap(blockIdx.x, nblk);
    const int bx_n = (a.width + 31) / 32;
    const int bx = b % bx_n, by = b / bx_n;
    const int i = bx * 32 + wave * 8 + (lane & 7);
    const int j = by * 8 + (lane >> 3);
    const bool valid = i < a.width && j < a.rows;

    Ray<DIFF> ray;
    ray.init(a, valid ? i : 0, valid ? j : 0);
    if (!valid) ray.done = 4;
    unsigned int executed = 0;
    while (__ballot(ray.done == 0)) {
        if (ray.done == 0) {
            ray.step(a, skip_diff);
            executed++;
        }
    }
    if (valid) ray.finish(a);
    unsigned long long tot = wave_sum_u32(executed);
    if (lane == 0) atomicAdd(a.ray_steps, tot);
}

// ---------------------------------------------------------------------------
// persistent schedule: waves pull 8x8 tiles from a queue and refill dead lanes.
// Work unit = one pixel; the queue hands out pixels in 8x8-tile-major order so
// that refilled lanes stay spatially coherent with their neighbours.
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
__global__ __launch_bounds__(256) void march_persistent_kernel(BhrMarchArgs a, uint32_t skip_diff,
                                                               int refill_below) {
    const int lane = threadIdx.x & 63;
    const unsigned int total = (unsigned int)a.n_tiles * 64u;
    Ray<DIFF> ray;
    ray.done = 4;  // empty lane
    ray.pix = -1;
    unsigned int executed = 0;
    bool queue_empty = false;

    for (;;) {
        unsigned long long live = __ballot(ray.done == 0);
        int n_live = __popcll(live);
        if (!queue_empty && n_live < refill_below) {
            // retire finished lanes, then hand every non-running lane a new pixel
            if (ray.done >= 1 && ray.done <= 3) ray.finish(a);
            unsigned long long want = ~live;
            int n_want = 64 - n_live;
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(a.queue, (unsigned int)n_want);
            base = __shfl(base, 0, BHR_WAVE);
            // rank of this lane among the lanes that want work
            unsigned long long below = want & ((1ull << lane) - 1ull);
            unsigned int w = base + (unsigned int)__popcll(below);
            ray.done = 4;
            ray.pix = -1;
            if ((want >> lane) & 1ull) {
                int i, j;
                if (w < total && tile_pixel(a, w, i, j)) ray.init(a, i, j);
            }
            if (base + (unsigned int)n_want >= total) queue_empty = true;
            live = __ballot(ray.done == 0);
            if (!live && queue_empty) break;
            continue;
        }
        if (!live) {
            if (ray.done >= 1 && ray.done <= 3) ray.finish(a);
            break;
        }
        if (ray.done == 0) {
            ray.step(a, skip_diff);
            executed++;
        }
    }
    unsigned long long tot = wave_sum_u32(executed);
    if (lane == 0) atomicAdd(a.ray_steps, tot);
}

}  // namespace

int32_t bhr_march_resources(int32_t *vgprs, int32_t *lds, int32_t diff) {
    hipFuncAttributes at;
    const void *f = diff ? (const void *)march_persistent_kernel<true> : (const void *)march_persistent_kernel<false>;
    BHR_HIP(hipFuncGetAttributes(&at, f));
    *vgprs = at.numRegs;
    *lds = (int32_t)at.sharedSizeBytes;
    return BHR_OK;
}

int32_t bhr_launch_march(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    const bhr_config &c = ctx->cfg;
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
    // render.py:2817-2818
    a.max_iter = (int32_t)(a.r_esc * 40.0f / a.h_base);
    a.max_affine = a.r_esc * 40.0f;
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
    // timed launches (bhr_render) count into their ring slot; group launches into the scalar
    const int slot = ctx->cur_slot;
    a.ray_steps = slot >= 0 ? ctx->d_steps_ring + slot : ctx->d_ray_steps;
    a.queue = ctx->d_queue;
    a.tiles_x = (c.width + 7) / 8;
    a.n_tiles = a.tiles_x * ((ctx->rows + 7) / 8);

    // anti_alias "disabled": the reference still integrates the differentials (skip_diff = 0
    // on the CLI path) but never reads them (render.py:2957-2959) => skipping them is pixel-identical.
    const bool want_diff = c.anti_alias != 0 && !(flags & BHR_SKIP_DIFFERENTIALS);
    const uint32_t skip_diff = want_diff ? 0u : 1u;

    BHR_HIP(hipMemsetAsync(a.ray_steps, 0, sizeof(unsigned long long), ctx->stream));
    BHR_HIP(hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned int), ctx->stream));
    BHR_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    if (slot >= 0) BHR_HIP(hipEventRecord(ctx->ring_ev[slot * 3 + 0], ctx->stream));
    if (flags & BHR_NO_COMPACTION) {
        const int bx_n = (c.width + 31) / 32, by_n = (ctx->rows + 7) / 8;
        dim3 grid(bx_n * by_n), block(256);
        if (want_diff)
            hipLaunchKernelGGL(march_tile_kernel<true>, grid, block, 0, ctx->stream, a, skip_diff);
        else
            hipLaunchKernelGGL(march_tile_kernel<false>, grid, block, 0, ctx->stream, a, skip_diff);
    } else {
        // enough resident waves to fill the chip; every wave drains the queue and exits
        int waves_total = a.n_tiles;
        int blocks = (waves_total + 3) / 4;
        const int max_blocks = 256 * 8;
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks < 1) blocks = 1;
        dim3 grid(blocks), block(256);
        const int refill_below = 40;
        if (want_diff)
            hipLaunchKernelGGL(march_persistent_kernel<true>, grid, block, 0, ctx->stream, a, skip_diff, refill_below);
        else
            hipLaunchKernelGGL(march_persistent_kernel<false>, grid, block, 0, ctx->stream, a, skip_diff, refill_below);
    }
    BHR_HIP(hipGetLastError());
    BHR_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    if (slot >= 0) BHR_HIP(hipEventRecord(ctx->ring_ev[slot * 3 + 1], ctx->stream));
    ctx->last_steps_ptr = a.ray_steps;
    ctx->counters.rays = (uint64_t)c.width * ctx->rows;
    return BHR_OK;
}
