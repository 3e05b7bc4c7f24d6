/*
 * bhr_oracle.c -- CPU restatement of the hwuu/black-hole-renderer device kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path
 * in black-hole-renderer_amd/csrc.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.  The product path
 * (libbhr_hip.so + the bhr_amd Python package) never links or imports it.
 *
 * What it restates: the reference's Taichi @ti.kernel / @ti.func bodies in
 * /root/reference/render.py, statement by statement, in f32 with the
 * reference's operation order (Taichi default_fp = f32, default_ip = i32,
 * integer '%' is floor-mod, float->int casts truncate):
 *
 *   _color_temp_to_tint          render.py:2407-2437
 *   _apply_g_factor              render.py:2439-2516
 *   _compute_acceleration        render.py:2518-2524
 *   _compute_acc_jacobian        render.py:2526-2539
 *   _sample_skybox               render.py:2541-2566
 *   _sample_disk                 render.py:2568-2598
 *   _sample_disk_mip             render.py:2600-2637
 *   _grad3_dot/_simplex/_fbm     render.py:2642-2785
 *   _ray_march_kernel            render.py:2787-3018
 *   _bloom_kernel                render.py:3022-3114
 *   _compose_disk_texture_kernel render.py:3169-3257
 *   _mipmap_*_kernel             render.py:3261-3283
 *   _noise_eval_kernel           render.py:3305-3326
 *   _generate_background_kernel  render.py:3332-3451
 *
 * Parity pinning (see DESIGN.md "Oracle"): the Taichi kernels cannot be
 * executed in the build container (taichi is not installed, no network), and
 * the reference's only pin for the march/bloom is an MD5 of float bytes from
 * the author's LLVM fast-math build (tests/e2e_baseline.txt) which carries no
 * values.  => march/bloom/background: PARITY UNPINNED by reference vectors;
 * pinned by physics known-answer tests and the reference's own property tests
 * restated in tests/.  compose + mipmaps ARE pinned: tests/golden holds
 * outputs of the reference's importable NumPy twin
 * (_generate_disk_texture_rotating_from_state, generate_disk_mipmaps), the
 * same comparison the reference makes in tests/unit/test_gpu_texture_compose.py.
 *
 * Build: see oracle/Makefile.  Strict variant: -O2 -ffp-contract=off
 * (checker).  Fast variant: -O3 -ffast-math -fopenmp (cpu_baseline timing;
 * mirrors Taichi's fast_math=True default).
 *
 * Array layouts follow the reference's Taichi fields:
 *   image/disk_layer/bright/blur : (W, H, 3)  index [i][j][c], i = screen x
 *   skybox                       : (tex_h, tex_w, 3)
 *   disk_tex                     : (n_r, n_phi, 4)
 *   disk_mips                    : (levels, n_r, n_phi, 4) padded to level 0
 *   comp                         : (13, n_r, n_phi)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* Buffers crossing the API are always IEEE binary32.  With -DORACLE_F64 every intermediate of
 * the SAME statements is evaluated in binary64 (constants keep their f32 values): the "exact"
 * value of the reference algorithm, against which the rounding noise of any f32 evaluation
 * order -- this file's strict build, Taichi's fast-math build, the HIP kernels -- is measured. */
typedef float f32;
#ifdef ORACLE_F64
#define float double
#define sqrtf sqrt
#define fminf fmin
#define fmaxf fmax
#define floorf floor
#define powf pow
#define expf exp
#define logf log
#define sinf sin
#define cosf cos
#define tanf tan
#define acosf acos
#define atan2f atan2
#endif

/* ---- constants: render.py:37-59 ---------------------------------------- */
#define RS_F 1.0f
#define G_FACTOR_CAP 1.5f
#define G_LUMINOSITY_POWER 1.5f
#define G_BRIGHTNESS_GAIN 0.38f
#define DISK_COLOR_TEMPERATURE 6000.0f
#define DISK_ALPHA_GAIN 6.0f
#define DISK_RADIAL_BRIGHTNESS_POWER 1.2f
#define DISK_RADIAL_BRIGHTNESS_MIN 0.2f
#define DISK_RADIAL_BRIGHTNESS_MAX 8.0f

static const float PI_F = (f32)3.141592653589793;         /* ti.math.pi -> f32 (also in the f64 build) */
static const float TWO_PI_F = (f32)(2 * 3.141592653589793); /* 2 * ti.math.pi folded in Python */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline v3 v3_divs(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v3_norm(v3 a) { return sqrtf(v3_dot(a, a)); }
/* taichi Vector.normalized(eps=0): invlen = 1/(norm+eps); invlen * v */
static inline v3 v3_normalized(v3 a) { float inv = 1.0f / (v3_norm(a) + 0.0f); return v3_scale(inv, a); }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
/* Python-style floor modulo for i32 (taichi '%'), m > 0 */
static inline int32_t pymod(int32_t a, int32_t m) { int32_t r = a % m; if (r < 0) r += m; return r; }
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* ---- render.py:2407-2437 ------------------------------------------------ */
static v3 color_temp_to_tint(float temp)
{
    float t = temp / 100.0f;
    float r = 1.0f;
    if (t > 66.0f)
        r = fminf(fmaxf(1.292936f * powf(fmaxf(t - 60.0f, 0.0001f), -0.1332047592f), 0.0f), 1.0f);
    float g = 0.0f;
    if (t <= 66.0f)
        g = fminf(fmaxf(0.390082f * logf(fmaxf(t, 0.0001f)) - 0.631841f, 0.0f), 1.0f);
    else
        g = fminf(fmaxf(1.129891f * powf(fmaxf(t - 60.0f, 0.0001f), -0.0755148492f), 0.0f), 1.0f);
    float b = 1.0f;
    if (t < 66.0f) {
        if (t <= 19.0f)
            b = 0.0f;
        else
            b = fminf(fmaxf(0.543207f * logf(fmaxf(t - 10.0f, 0.0001f)) - 1.19625f, 0.0f), 1.0f);
    }
    return v3_make(r, g, b);
}

/* ---- render.py:2439-2516 ------------------------------------------------ */
static v3 apply_g_factor(v3 base_color, v3 hit_pos, float hit_r, v3 ray_dir_to_cam, v3 cam_pos,
                         float r_inner, float r_outer, float tilt_rad)
{
    const float rs_f = RS_F;
    float r_obs = v3_norm(cam_pos);
    float r_em = v3_norm(hit_pos);
    float r_safe = fmaxf(r_em, rs_f + 1e-3f);

    float omega = sqrtf(0.5f / (r_safe * r_safe * r_safe + 1e-6f));
    float lorentz = sqrtf(fmaxf(1.0f - rs_f / r_safe, 1e-6f));
    float beta = fminf(r_safe * omega / fmaxf(lorentz, 1e-6f), 0.99f);
    float gamma = 1.0f / sqrtf(fmaxf(1.0f - beta * beta, 1e-6f));

    float sin_t = sinf(tilt_rad);
    float cos_t = cosf(tilt_rad);
    v3 disk_normal = v3_make(0.0f, -sin_t, cos_t);
    v3 r_hat = v3_normalized(hit_pos);
    v3 v_hat = v3_cross(r_hat, disk_normal);
    float v_norm = v3_norm(v_hat);
    if (v_norm > 1e-6f)
        v_hat = v3_divs(v_hat, v_norm);
    else
        v_hat = v3_make(0.0f, 1.0f, 0.0f);

    v3 ray_hat = v3_normalized(ray_dir_to_cam);
    float cos_theta = v3_dot(v_hat, ray_hat);
    float denom = fmaxf(1.0f - beta * cos_theta, 1e-3f);
    float g_doppler = 1.0f / (gamma * denom);

    float grav_num = sqrtf(fmaxf(1.0f - rs_f / fmaxf(r_obs, rs_f + 1e-3f), 1e-6f));
    float grav_den = sqrtf(fmaxf(1.0f - rs_f / fmaxf(r_em, rs_f + 1e-3f), 1e-6f));
    float g_grav = grav_num / grav_den;

    float g = fminf(g_doppler * g_grav, G_FACTOR_CAP);
    float intensity = fmaxf(powf(g, G_LUMINOSITY_POWER), 0.0f);
    float brightness = G_BRIGHTNESS_GAIN * intensity / (1.0f + intensity / G_FACTOR_CAP);

    float radial_span = fmaxf(r_outer - r_inner, 1e-3f);
    float radial_t = (fmaxf(hit_r, r_inner) - r_inner) / radial_span;
    radial_t = fminf(fmaxf(radial_t, 0.0f), 1.0f);
    float radial_profile = powf(1.0f - radial_t, DISK_RADIAL_BRIGHTNESS_POWER);
    float min_boost = DISK_RADIAL_BRIGHTNESS_MIN;
    float max_boost = DISK_RADIAL_BRIGHTNESS_MAX;
    float radial_boost = min_boost + (max_boost - min_boost) * radial_profile;
    brightness *= radial_boost;

    float g_safe = fmaxf(g, 0.1f);
    float wien_arg = 1.0f - 1.0f / g_safe;
    float r_scale = expf(2.21f * wien_arg);
    float g_scale = expf(2.72f * wien_arg);
    float b_scale = expf(3.13f * wien_arg);
    float norm = g_scale;
    r_scale = fminf(r_scale / norm, 3.0f);
    g_scale = 1.0f;
    b_scale = fminf(b_scale / norm, 3.0f);

    v3 shifted = v3_make(base_color.x * r_scale, base_color.y * g_scale, base_color.z * b_scale);
    v3 tint = color_temp_to_tint(DISK_COLOR_TEMPERATURE);
    v3 out = v3_make(shifted.x * tint.x * brightness, shifted.y * tint.y * brightness,
                     shifted.z * tint.z * brightness);
    out.x = clampf(out.x, 0.0f, 10.0f);
    out.y = clampf(out.y, 0.0f, 10.0f);
    out.z = clampf(out.z, 0.0f, 10.0f);
    return out;
}

/* ---- render.py:2518-2524 ------------------------------------------------ */
static inline v3 compute_acceleration(v3 pos, float L2)
{
    float r2 = v3_dot(pos, pos);
    float r = sqrtf(r2);
    float r5 = r2 * r2 * r;
    return v3_scale(-1.5f * L2 / r5, pos);
}

/* ---- render.py:2526-2539 ------------------------------------------------ */
static inline v3 compute_acc_jacobian(v3 pos, v3 d_pos, float L2)
{
    float r2 = v3_dot(pos, pos);
    float r = sqrtf(r2);
    float r5 = r2 * r2 * r;
    float factor = -1.5f * L2 / r5;
    float proj = v3_dot(pos, d_pos) / r2;
    /* d_pos - 5.0 * pos * proj : (5.0*pos)*proj, left to right */
    v3 t = v3_make(5.0f * pos.x * proj, 5.0f * pos.y * proj, 5.0f * pos.z * proj);
    return v3_scale(factor, v3_sub(d_pos, t));
}

typedef struct {
    const f32 *skybox; int32_t tex_h, tex_w;
    const f32 *disk_tex; int32_t dtex_h, dtex_w;
    const f32 *disk_mips; int32_t num_mip_levels; /* padded (levels, dtex_h, dtex_w, 4) */
} scene_t;

/* ---- render.py:2541-2566 ------------------------------------------------ */
static v3 sample_skybox(const scene_t *s, v3 d)
{
    const int32_t tex_w = s->tex_w, tex_h = s->tex_h;
    float x = d.x, y = d.y, z = d.z;
    float theta = acosf(fminf(fmaxf(z, -1.0f), 1.0f));
    float phi = atan2f(y, x);
    if (phi < 0) phi += TWO_PI_F;
    float u = phi / TWO_PI_F * (float)tex_w;
    float v = theta / PI_F * (float)tex_h;
    int32_t u0 = (int32_t)floorf(u);
    int32_t v0 = (int32_t)floorf(v);
    float fu = u - (float)u0;
    float fv = v - (float)v0;
    int32_t u0_w = pymod(u0, tex_w);
    int32_t u1_w = pymod(u0 + 1, tex_w);
    int32_t v0_h = imin(imax(v0, 0), tex_h - 1);
    int32_t v1_h = imin(imax(v0 + 1, 0), tex_h - 1);
    const f32 *c00 = s->skybox + ((size_t)v0_h * tex_w + u0_w) * 3;
    const f32 *c10 = s->skybox + ((size_t)v0_h * tex_w + u1_w) * 3;
    const f32 *c01 = s->skybox + ((size_t)v1_h * tex_w + u0_w) * 3;
    const f32 *c11 = s->skybox + ((size_t)v1_h * tex_w + u1_w) * 3;
    float o[3];
    for (int c = 0; c < 3; ++c)
        o[c] = c00[c] * (1 - fu) * (1 - fv) + c10[c] * fu * (1 - fv) + c01[c] * (1 - fu) * fv + c11[c] * fu * fv;
    return v3_make(o[0], o[1], o[2]);
}

static inline float wrap_phi(float phi)
{
    while (phi < 0) phi += TWO_PI_F;
    while (phi >= TWO_PI_F) phi -= TWO_PI_F;
    return phi;
}

static inline v4 bilerp4(const f32 *c00, const f32 *c10, const f32 *c01, const f32 *c11, float fu, float fv)
{
    float o[4];
    for (int c = 0; c < 4; ++c)
        o[c] = c00[c] * (1 - fu) * (1 - fv) + c10[c] * fu * (1 - fv) + c01[c] * (1 - fu) * fv + c11[c] * fu * fv;
    v4 r = {o[0], o[1], o[2], o[3]};
    return r;
}

/* ---- render.py:2568-2598 ------------------------------------------------ */
static v4 sample_disk(const scene_t *s, float hit_x, float hit_y, float r_inner, float r_outer, float t_offset)
{
    const int32_t dtex_w = s->dtex_w, dtex_h = s->dtex_h;
    float r = sqrtf(hit_x * hit_x + hit_y * hit_y);
    float phi = atan2f(hit_y, hit_x);
    float r_safe = fmaxf(r, 1e-3f);
    float omega = sqrtf(0.5f / (r_safe * r_safe * r_safe + 1e-6f));
    phi = phi + t_offset * omega;
    phi = wrap_phi(phi);
    float u = phi / TWO_PI_F * (float)dtex_w;
    float v = (r - r_inner) / (r_outer - r_inner) * (float)dtex_h;
    int32_t u0 = (int32_t)floorf(u);
    int32_t v0 = (int32_t)floorf(v);
    float fu = u - (float)u0;
    float fv = v - (float)v0;
    int32_t u0_w = pymod(u0, dtex_w);
    int32_t u1_w = pymod(u0 + 1, dtex_w);
    int32_t v0_h = imin(imax(v0, 0), dtex_h - 1);
    int32_t v1_h = imin(imax(v0 + 1, 0), dtex_h - 1);
    const f32 *t = s->disk_tex;
    return bilerp4(t + ((size_t)v0_h * dtex_w + u0_w) * 4, t + ((size_t)v0_h * dtex_w + u1_w) * 4,
                   t + ((size_t)v1_h * dtex_w + u0_w) * 4, t + ((size_t)v1_h * dtex_w + u1_w) * 4, fu, fv);
}

/* ---- render.py:2600-2637 ------------------------------------------------ */
static v4 sample_disk_mip(const scene_t *s, float hit_x, float hit_y, float r_inner, float r_outer,
                          float t_offset, float lod)
{
    const int32_t dtex_w = s->dtex_w, dtex_h = s->dtex_h;
    float r = sqrtf(hit_x * hit_x + hit_y * hit_y);
    float phi = atan2f(hit_y, hit_x);
    float r_safe = fmaxf(r, 1e-3f);
    float omega = sqrtf(0.5f / (r_safe * r_safe * r_safe + 1e-6f));
    phi = phi + t_offset * omega;
    phi = wrap_phi(phi);

    int32_t lod_i = (int32_t)fminf(fmaxf(lod, 0.0f), (float)(s->num_mip_levels - 1));
    float tex_w_lod = (float)dtex_w / powf(2.0f, (float)lod_i);
    float tex_h_lod = (float)dtex_h / powf(2.0f, (float)lod_i);

    float u = phi / TWO_PI_F * tex_w_lod;
    float v = (r - r_inner) / (r_outer - r_inner) * tex_h_lod;
    int32_t u0 = (int32_t)floorf(u);
    int32_t v0 = (int32_t)floorf(v);
    float fu = u - (float)u0;
    float fv = v - (float)v0;
    int32_t wl = (int32_t)tex_w_lod;
    int32_t u0_w = pymod(u0, wl);
    int32_t u1_w = pymod(u0 + 1, wl);
    int32_t v0_h = imin(imax(v0, 0), (int32_t)(tex_h_lod - 1));
    int32_t v1_h = imin(imax(v0 + 1, 0), (int32_t)(tex_h_lod - 1));
    const f32 *t = s->disk_mips + (size_t)lod_i * dtex_h * dtex_w * 4;
    return bilerp4(t + ((size_t)v0_h * dtex_w + u0_w) * 4, t + ((size_t)v0_h * dtex_w + u1_w) * 4,
                   t + ((size_t)v1_h * dtex_w + u0_w) * 4, t + ((size_t)v1_h * dtex_w + u1_w) * 4, fu, fv);
}

/* ---- unit probes of the device functions (tests/test_reference_kernels.py): the shading, the LOD
 * sampler and the tint on recorded argument lists of the reference's own @ti.func calls.  binary64
 * across the API so that the f64 build is probed at full precision. ------------------------------- */
ORACLE_API void oracle_probe_g_factor(const double *in16, int64_t n, double *out3)
{   /* rows: base_color[3], hit_pos[3], hit_r, ray_dir_to_cam[3], cam_pos[3], r_inner, r_outer, tilt_rad */
    for (int64_t k = 0; k < n; ++k) {
        const double *a = in16 + 16 * k;
        v3 o = apply_g_factor(v3_make((float)a[0], (float)a[1], (float)a[2]), v3_make((float)a[3], (float)a[4], (float)a[5]),
                              (float)a[6], v3_make((float)a[7], (float)a[8], (float)a[9]),
                              v3_make((float)a[10], (float)a[11], (float)a[12]), (float)a[13], (float)a[14], (float)a[15]);
        out3[3 * k] = o.x; out3[3 * k + 1] = o.y; out3[3 * k + 2] = o.z;
    }
}

ORACLE_API void oracle_probe_tint(const double *temp, int64_t n, double *out3)
{
    for (int64_t k = 0; k < n; ++k) {
        v3 o = color_temp_to_tint((float)temp[k]);
        out3[3 * k] = o.x; out3[3 * k + 1] = o.y; out3[3 * k + 2] = o.z;
    }
}

ORACLE_API void oracle_probe_disk_mip(const f32 *disk_mips, int32_t levels, int32_t dtex_h, int32_t dtex_w,
                                      const double *in6, int64_t n, double *out4)
{   /* rows: hit_x, hit_y, r_inner, r_outer, t_offset, lod */
    scene_t sc = {0, 0, 0, 0, dtex_h, dtex_w, disk_mips, levels};
    for (int64_t k = 0; k < n; ++k) {
        const double *a = in6 + 6 * k;
        v4 o = sample_disk_mip(&sc, (float)a[0], (float)a[1], (float)a[2], (float)a[3], (float)a[4], (float)a[5]);
        out4[4 * k] = o.x; out4[4 * k + 1] = o.y; out4[4 * k + 2] = o.z; out4[4 * k + 3] = o.w;
    }
}

ORACLE_API void oracle_probe_skybox(const f32 *skybox, int32_t tex_h, int32_t tex_w, const double *dir3, int64_t n, double *out3)
{
    scene_t sc = {skybox, tex_h, tex_w, 0, 0, 0, 0, 0};
    for (int64_t k = 0; k < n; ++k) {
        v3 o = sample_skybox(&sc, v3_make((float)dir3[3 * k], (float)dir3[3 * k + 1], (float)dir3[3 * k + 2]));
        out3[3 * k] = o.x; out3[3 * k + 1] = o.y; out3[3 * k + 2] = o.z;
    }
}

/* ---- simplex / fbm: render.py:2269-2288 (perm), 2642-2785 --------------- */
static const int32_t PERM256[256] = {
    151,160,137,91,90,15,131,13,201,95,96,53,194,233,7,225,
    140,36,103,30,69,142,8,99,37,240,21,10,23,190,6,148,
    247,120,234,75,0,26,197,62,94,252,219,203,117,35,11,32,
    57,177,33,88,237,149,56,87,174,20,125,136,171,168,68,175,
    74,165,71,134,139,48,27,166,77,146,158,231,83,111,229,122,
    60,211,133,230,220,105,92,41,55,46,245,40,244,102,143,54,
    65,25,63,161,1,216,80,73,209,76,132,187,208,89,18,169,
    200,196,135,130,116,188,159,86,164,100,109,198,173,186,3,64,
    52,217,226,250,124,123,5,202,38,147,118,126,255,82,85,212,
    207,206,59,227,47,16,58,17,182,189,28,42,223,183,170,213,
    119,248,152,2,44,154,163,70,221,153,101,155,167,43,172,9,
    129,22,39,253,19,98,108,110,79,113,224,232,178,185,112,104,
    218,246,97,228,251,34,242,193,238,210,144,12,191,179,162,241,
    81,51,145,235,249,14,239,107,49,192,214,31,181,199,106,157,
    184,84,204,176,115,121,50,45,127,4,150,254,138,236,205,93,
    222,114,67,29,24,72,243,141,128,195,78,66,215,61,156,180,
};
static inline int32_t perm(int32_t i) { return PERM256[i & 255]; } /* perm_field = _perm + _perm, idx < 512 */

static inline float grad3_dot(int32_t hash_val, float x, float y, float z)
{
    int32_t h = pymod(hash_val, 12);
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    float r1 = (h & 1) == 0 ? u : -u;
    float r2 = (h & 2) == 0 ? v : -v;
    return r1 + r2;
}

static float simplex_noise_3d(float x, float y, float z)
{
    const float F3 = 1.0f / 3.0f;
    const float G3 = 1.0f / 6.0f;
    float s = (x + y + z) * F3;
    int32_t i = (int32_t)floorf(x + s);
    int32_t j = (int32_t)floorf(y + s);
    int32_t k = (int32_t)floorf(z + s);
    float t = (float)(i + j + k) * G3;
    float x0 = x - ((float)i - t);
    float y0 = y - ((float)j - t);
    float z0 = z - ((float)k - t);
    int32_t i1 = 0, j1 = 0, k1 = 0, i2 = 0, j2 = 0, k2 = 0;
    if (x0 >= y0) {
        if (y0 >= z0)      { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
        else if (x0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 0; k2 = 1; }
        else               { i1 = 0; j1 = 0; k1 = 1; i2 = 1; j2 = 0; k2 = 1; }
    } else {
        if (y0 < z0)       { i1 = 0; j1 = 0; k1 = 1; i2 = 0; j2 = 1; k2 = 1; }
        else if (x0 < z0)  { i1 = 0; j1 = 1; k1 = 0; i2 = 0; j2 = 1; k2 = 1; }
        else               { i1 = 0; j1 = 1; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
    }
    float x1 = x0 - (float)i1 + G3, y1 = y0 - (float)j1 + G3, z1 = z0 - (float)k1 + G3;
    float x2 = x0 - (float)i2 + 2.0f * G3, y2 = y0 - (float)j2 + 2.0f * G3, z2 = z0 - (float)k2 + 2.0f * G3;
    float x3 = x0 - 1.0f + 3.0f * G3, y3 = y0 - 1.0f + 3.0f * G3, z3 = z0 - 1.0f + 3.0f * G3;
    int32_t ii = i & 255, jj = j & 255, kk = k & 255;
    int32_t gi0 = perm(ii + perm(jj + perm(kk)));
    int32_t gi1 = perm(ii + i1 + perm(jj + j1 + perm(kk + k1)));
    int32_t gi2 = perm(ii + i2 + perm(jj + j2 + perm(kk + k2)));
    int32_t gi3 = perm(ii + 1 + perm(jj + 1 + perm(kk + 1)));
    float n = 0.0f;
    float t0 = 0.6f - x0 * x0 - y0 * y0 - z0 * z0;
    if (t0 >= 0.0f) { t0 = t0 * t0; n += t0 * t0 * grad3_dot(gi0, x0, y0, z0); }
    float t1 = 0.6f - x1 * x1 - y1 * y1 - z1 * z1;
    if (t1 >= 0.0f) { t1 = t1 * t1; n += t1 * t1 * grad3_dot(gi1, x1, y1, z1); }
    float t2 = 0.6f - x2 * x2 - y2 * y2 - z2 * z2;
    if (t2 >= 0.0f) { t2 = t2 * t2; n += t2 * t2 * grad3_dot(gi2, x2, y2, z2); }
    float t3 = 0.6f - x3 * x3 - y3 * y3 - z3 * z3;
    if (t3 >= 0.0f) { t3 = t3 * t3; n += t3 * t3 * grad3_dot(gi3, x3, y3, z3); }
    return 32.0f * n;
}

static float fbm_3d(float x, float y, float z, int32_t octaves, float persistence, float lacunarity)
{
    float value = 0.0f, amplitude = 1.0f, freq = 1.0f;
    for (int32_t o = 0; o < octaves; ++o) {
        value += amplitude * simplex_noise_3d(x * freq, y * freq, z * freq);
        amplitude *= persistence;
        freq *= lacunarity;
    }
    return value;
}

/* ---- _noise_eval_kernel: render.py:3305-3326 ---------------------------- */
ORACLE_API void oracle_eval_noise(const f32 *coords, int64_t n, int32_t mode, int32_t octaves,
                                  f32 persistence, f32 lacunarity, f32 *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float cx = coords[i * 3 + 0], cy = coords[i * 3 + 1], cz = coords[i * 3 + 2];
        out[i] = mode == 0 ? simplex_noise_3d(cx, cy, cz) : fbm_3d(cx, cy, cz, octaves, persistence, lacunarity);
    }
}

/* ---- Disk V2 (reference: disk_v2/geometry.py, physical_fields.py, structure_modulations.py) ------
 * Scalar binary64 restatement, pinned by tests/golden/disk_v2.npz (made from the reference package).
 * The random tables of shear_modulation / hotspot_modulation are drawn by the caller. */
#define DV2_MAX_TERMS 32
typedef struct {
    double r_in, r_out, h0, beta_h, rho_power, temp_scale, omega_scale, edge_softness;
    double mode1_strength, mode2_strength, shear_strength, hotspot_strength;
    double hotspot_phi_sigma, hotspot_logr_sigma, hotspot_inner_bias;
    int32_t shear_components, hotspot_count;
    int32_t shear_phi_freq[DV2_MAX_TERMS], shear_logr_freq[DV2_MAX_TERMS];
    double shear_phase[DV2_MAX_TERMS];
    double hotspot_phase[DV2_MAX_TERMS], hotspot_log_r[DV2_MAX_TERMS], hotspot_weight[DV2_MAX_TERMS];
} oracle_dv2_params;

#define DV2_EPS 2.220446049250313e-16
static double dv2_smoothstep(double e0, double e1, double x)       /* geometry.py:15-47 */
{
    double t = fmin(fmax((x - e0) / (e1 - e0), 0.0), 1.0);
    return t * t * (3.0 - 2.0 * t);
}
static double dv2_H(double r, const oracle_dv2_params *p)          /* geometry.py:50-77 */
{
    double safe_r = fmax(r, p->r_in);
    return p->h0 * safe_r * pow(safe_r / p->r_in, p->beta_h);
}
static int dv2_mask_r(double r, const oracle_dv2_params *p) { return r >= p->r_in && r <= p->r_out; }   /* 80-113 */
static double dv2_W_r(double r, const oracle_dv2_params *p)        /* geometry.py:116-185 */
{
    double span = p->r_out - p->r_in;
    double soft = fmax(span * p->edge_softness, DV2_EPS);
    double inner = dv2_smoothstep(p->r_in, p->r_in + soft, r);
    double outer = 1.0 - dv2_smoothstep(p->r_out - soft, p->r_out, r);
    return (r <= p->r_in || r >= p->r_out) ? 0.0 : inner * outer;
}
static double dv2_W_z(double r, double z, const oracle_dv2_params *p)   /* geometry.py:188-235 */
{
    double th = fmax(dv2_H(r, p), DV2_EPS);
    double w = 1.0 - dv2_smoothstep(0.0, 1.0, fabs(z) / th);
    return dv2_mask_r(r, p) ? w : 0.0;
}
static int dv2_mask_vol(double r, double z, const oracle_dv2_params *p) { return dv2_mask_r(r, p) && fabs(z) <= dv2_H(r, p); }
static double dv2_omega(double r, const oracle_dv2_params *p)      /* physical_fields.py:21-49 */
{
    return p->omega_scale * pow(fmax(r, p->r_in) / p->r_in, -1.5);
}
static double dv2_rho_mid(double r, const oracle_dv2_params *p)    /* physical_fields.py:52-79 */
{
    return pow(fmax(r, p->r_in) / p->r_in, -p->rho_power) * dv2_W_r(r, p);
}
static double dv2_t_mid(double r, const oracle_dv2_params *p)      /* physical_fields.py:82-116 */
{
    double safe_r = fmax(r, p->r_in);
    double inner = fmax(1.0 - sqrt(p->r_in / safe_r), 0.0);
    double t = p->temp_scale * pow(safe_r / p->r_in, -0.75) * pow(inner, 0.25) * dv2_W_r(r, p);
    return r <= p->r_in ? 0.0 : t;
}
static double dv2_rho(double r, double z, const oracle_dv2_params *p)   /* physical_fields.py:119-160 */
{
    double th = fmax(dv2_H(r, p), DV2_EPS), q = z / th;
    double v = dv2_rho_mid(r, p) * exp(-0.5 * (q * q)) * dv2_W_z(r, z, p);
    return dv2_mask_vol(r, z, p) ? v : 0.0;
}
static double dv2_T(double r, double z, const oracle_dv2_params *p)     /* physical_fields.py:163-205 */
{
    double th = fmax(dv2_H(r, p), DV2_EPS);
    double vf = fmin(fmax(1.0 - 0.25 * fabs(z) / th, 0.0), 1.0);
    double v = dv2_t_mid(r, p) * vf * dv2_W_z(r, z, p);
    return dv2_mask_vol(r, z, p) ? v : 0.0;
}
static double dv2_logr(double r, const oracle_dv2_params *p) { return log(fmax(r, p->r_in) / p->r_in); }
static double dv2_raw_shear(double r, double phi, const oracle_dv2_params *p)   /* structure_modulations.py:145-207 */
{
    double lr = dv2_logr(r, p), s = 0.0, amp = 1.0;
    for (int k = 0; k < p->shear_components; ++k) {
        double pf = (double)p->shear_phi_freq[k], lf = (double)p->shear_logr_freq[k], ph = p->shear_phase[k];
        s += amp * cos(pf * phi + lf * lr + ph);
        s += 0.6 * amp * sin((pf + 1.0) * phi - (lf + 0.5) * lr + 0.7 * ph);
        amp *= 0.5;
    }
    return s;
}
static double dv2_raw_hotspot(double r, double phi, const oracle_dv2_params *p) /* structure_modulations.py:210-289 */
{
    double lr = dv2_logr(r, p), s = 0.0;
    for (int k = 0; k < p->hotspot_count; ++k) {
        double d = phi - p->hotspot_phase[k];
        double dphi = atan2(sin(d), cos(d));
        double a = dphi / p->hotspot_phi_sigma, dl = (lr - p->hotspot_log_r[k]) / p->hotspot_logr_sigma;
        double core = exp(-0.5 * (a * a) - 0.5 * (dl * dl));
        double b = dphi / (1.8 * p->hotspot_phi_sigma), c = (lr - p->hotspot_log_r[k]) / (1.8 * p->hotspot_logr_sigma);
        double halo = exp(-0.5 * (b * b) - 0.5 * (c * c));
        s += p->hotspot_weight[k] * (core - 0.6 * halo);
    }
    return s;
}
static double dv2_mode(double r, double phi, const oracle_dv2_params *p)        /* structure_modulations.py:95-142 */
{
    double lr = dv2_logr(r, p);
    double raw = p->mode1_strength * cos(phi + 0.35 * lr) + p->mode2_strength * cos(2.0 * phi - 0.65 * lr);
    return dv2_W_r(r, p) > 0.0 ? 1.0 + raw : 1.0;
}
/* structure_modulation (292-334) with the normalisation maxima supplied by the caller */
static double dv2_structure(double r, double phi, const oracle_dv2_params *p, double norm_shear, double norm_hotspot)
{
    if (!(dv2_W_r(r, p) > 0.0)) return 1.0;
    double sh = 1.0 + p->shear_strength * (norm_shear <= DV2_EPS ? 0.0 : dv2_raw_shear(r, phi, p) / norm_shear);
    double hs = 1.0 + p->hotspot_strength * (norm_hotspot <= DV2_EPS ? 0.0 : dv2_raw_hotspot(r, phi, p) / norm_hotspot);
    return dv2_mode(r, phi, p) * sh * hs;
}

/* field ids follow include/bhr_disk_v2.h: 0 H, 1 mask_r, 2 W_r, 3 W_z, 4 mask_vol, 5 Omega, 6 rho_mid,
 * 7 T_mid, 8 rho, 9 T, 10 F_mode, 11 raw shear sum, 12 raw hotspot sum, 13 F_total (given norms) */
ORACLE_API void oracle_dv2_eval(const oracle_dv2_params *p, int32_t field, const double *r, const double *z, const double *phi,
                                int64_t n, double norm_shear, double norm_hotspot, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        double ri = r[i], zi = z ? z[i] : 0.0, ph = phi ? phi[i] : 0.0, v = 0.0;
        switch (field) {
            case 0: v = dv2_H(ri, p); break;
            case 1: v = dv2_mask_r(ri, p); break;
            case 2: v = dv2_W_r(ri, p); break;
            case 3: v = dv2_W_z(ri, zi, p); break;
            case 4: v = dv2_mask_vol(ri, zi, p); break;
            case 5: v = dv2_omega(ri, p); break;
            case 6: v = dv2_rho_mid(ri, p); break;
            case 7: v = dv2_t_mid(ri, p); break;
            case 8: v = dv2_rho(ri, zi, p); break;
            case 9: v = dv2_T(ri, zi, p); break;
            case 10: v = dv2_mode(ri, ph, p); break;
            case 11: v = dv2_raw_shear(ri, ph, p); break;
            case 12: v = dv2_raw_hotspot(ri, ph, p); break;
            default: v = dv2_structure(ri, ph, p, norm_shear, norm_hotspot);
        }
        out[i] = v;
    }
}

/* ---- finite-thickness Disk V2 source (docs/design_ad_v2.md 4.2-4.3; include/bhr_disk_v2.h:
 * BHR_DISK_V2_VOLUME).  oracle_set_volume(NULL, ...) switches back to the textured thin disk. */
typedef struct {
    oracle_dv2_params p;
    double norm_shear, norm_hotspot, t_peak, absorption, grazing_gain;
    int32_t substeps, on;
} volume_t;
static volume_t g_vol;

ORACLE_API void oracle_set_volume(const oracle_dv2_params *p, double norm_shear, double norm_hotspot, double t_peak,
                                  double absorption, double grazing_gain, int32_t substeps)
{
    g_vol.on = p != NULL;
    if (!p) return;
    g_vol.p = *p;
    g_vol.norm_shear = norm_shear; g_vol.norm_hotspot = norm_hotspot; g_vol.t_peak = t_peak;
    g_vol.absorption = absorption; g_vol.grazing_gain = grazing_gain; g_vol.substeps = substeps;
}

/* black-body colour of the normalised temperature t in [0, 1]: the compose kernel's mapping
 * (render.py:3243-3257) as csrc/march.hip disk_v2_color applies it */
static v3 dv2_color(float tf)
{
    const float t_factor = (DISK_COLOR_TEMPERATURE - 4500.0f) / (6500.0f - 2700.0f);
    const float T_min = 2000.0f + t_factor * 1000.0f, T_max = 9000.0f + t_factor * 3000.0f;
    v3 c = color_temp_to_tint(T_min + tf * (T_max - T_min));
    c.z = fminf(c.z, c.x);
    float lum = fminf(fmaxf(sqrtf(tf), 0.0f), 1.0f);
    return v3_make(clampf(c.x * lum, 0.0f, 1.0f), clampf(c.y * lum, 0.0f, 1.0f), clampf(c.z * lum, 0.0f, 1.0f));
}

/* one RK4 step's chord through the volume; *accum / *alpha_total are composited front to back */
static void volume_segment(v3 p0, v3 p1, v3 dir0, float tilt_rad, float t_offset, v3 cam_pos, float r_inner, float r_outer,
                           v3 *accum, float *alpha_total)
{
    const oracle_dv2_params *P = &g_vol.p;
    if (*alpha_total >= 0.9999f) return;   /* BHR_VOLUME_OPAQUE: the ray has stopped sampling */
    const double ct = (double)cosf(tilt_rad), st = (double)sinf(tilt_rad);
    const double ex = (double)p1.x - (double)p0.x, ey = (double)p1.y - (double)p0.y, ez = (double)p1.z - (double)p0.z;
    const double len = sqrt(ex * ex + ey * ey + ez * ez);
    if (!(len > 0.0)) return;
    const double mu = fabs((ez * ct - ey * st) / len);
    const double ds = len / (double)g_vol.substeps;
    const v3 to_cam = v3_make(-dir0.x, -dir0.y, -dir0.z);
    for (int k = 0; k < g_vol.substeps; ++k) {
        const double f = ((double)k + 0.5) / (double)g_vol.substeps;
        const double sx = (double)p0.x + f * ex, sy = (double)p0.y + f * ey, sz = (double)p0.z + f * ez;
        const double zeta = sz * ct - sy * st;
        const double yp = sy * ct + sz * st;
        const double rc = sqrt(sx * sx + yp * yp);
        if (!dv2_mask_vol(rc, zeta, P)) continue;
        const double phi = atan2(yp, sx) + (double)t_offset * dv2_omega(rc, P);
        const double F = dv2_structure(rc, phi, P, g_vol.norm_shear, g_vol.norm_hotspot);
        const double rho = fmax(dv2_rho(rc, zeta, P) * F, 0.0);
        const double t = fmin(fmax(dv2_T(rc, zeta, P) * F / g_vol.t_peak, 0.0), 1.0);
        const double alpha_eff = g_vol.absorption * rho * (1.0 + g_vol.grazing_gain * (1.0 - mu));
        const float op = (float)(1.0 - exp(-alpha_eff * ds));
        if (!(op > 0.0f)) continue;
        v3 col = apply_g_factor(dv2_color((float)t), v3_make((float)sx, (float)sy, (float)sz), (float)rc, to_cam, cam_pos,
                                r_inner, r_outer, tilt_rad);
        const float front = 1.0f - *alpha_total;
        accum->x += col.x * op * front;
        accum->y += col.y * op * front;
        accum->z += col.z * op * front;
        *alpha_total = 1.0f - front * (1.0f - op);
    }
}

/* Test hook: when set, every escaped ray also stores its escape direction here ((W, H, 3) doubles; captured
 * rays store zeros).  Used by the convergence test against an independent geodesic integrator. */
static double *g_escape_out;
ORACLE_API void oracle_set_escape_out(double *buf) { g_escape_out = buf; }

/* ---- camera uniforms as uploaded at render.py:3886-3892 ----------------- */
typedef struct {
    f32 cam_pos[3], cam_right[3], cam_up[3], cam_forward[3];
    f32 pixel_width, pixel_height, r_escape;
} oracle_camera;

typedef struct {
    int32_t width, height;
    f32 h_base, r_inner, r_outer, t_offset, disk_tilt;
    int32_t skip_diff;        /* kernel argument skip_diff */
    int32_t anti_alias_mode;  /* 0 disabled, 1 lod_radius (compile-time in the reference) */
    f32 aa_strength;
} oracle_march_params;

/* ---- _ray_march_kernel: render.py:2787-3018 ------------------------------
 * image_out / disk_out: (W, H, 3).  steps_out (optional, may be NULL): (W, H)
 * i32 = number of while-loop iterations executed, counting the terminating
 * one (SURVEY 8d definition of a ray-step).  Returns total ray-steps.
 * i_lo..i_hi / j_lo..j_hi restrict the pixel range (for sampling/timing). */
ORACLE_API int64_t oracle_ray_march(const oracle_camera *cam, const oracle_march_params *p,
                                    const f32 *skybox, int32_t tex_h, int32_t tex_w,
                                    const f32 *disk_tex, int32_t dtex_h, int32_t dtex_w,
                                    const f32 *disk_mips, int32_t num_mip_levels,
                                    f32 *image_out, f32 *disk_out, int32_t *steps_out,
                                    int32_t j_lo, int32_t j_hi)
{
    scene_t sc = {skybox, tex_h, tex_w, disk_tex, dtex_h, dtex_w, disk_mips, num_mip_levels};
    const int32_t width = p->width, height = p->height;
    const v3 cp = v3_make(cam->cam_pos[0], cam->cam_pos[1], cam->cam_pos[2]);
    const v3 cr = v3_make(cam->cam_right[0], cam->cam_right[1], cam->cam_right[2]);
    const v3 cu = v3_make(cam->cam_up[0], cam->cam_up[1], cam->cam_up[2]);
    const v3 cf = v3_make(cam->cam_forward[0], cam->cam_forward[1], cam->cam_forward[2]);
    const float pw = cam->pixel_width, ph = cam->pixel_height;
    const float h_base = p->h_base, r_inner = p->r_inner, r_outer = p->r_outer, t_offset = p->t_offset;
    const int32_t skip_diff = p->skip_diff;

    const float tilt_rad = p->disk_tilt * PI_F / 180.0f;
    const float min_fac = 0.2f;
    const v3 center = v3_add(cp, v3_scale(1.0f, cf));
    const v3 tl = v3_add(v3_sub(center, v3_scale(pw * (float)width / 2, cr)), v3_scale(ph * (float)height / 2, cu));
    const float max_fac = 10.0f;
    const float r_cap = RS_F;
    const float r_esc = cam->r_escape;
    const int32_t max_iter = (int32_t)(r_esc * 40 / h_base);
    const float max_affine = r_esc * 40.0f;
    int64_t total_steps = 0;

    if (j_lo < 0) j_lo = 0;
    if (j_hi > height) j_hi = height;

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total_steps)
#endif
    for (int32_t j = j_lo; j < j_hi; ++j) {
        for (int32_t i = 0; i < width; ++i) {
            float px_f = (float)i, py_f = (float)j;
            v3 pixel_pos = v3_sub(v3_add(tl, v3_scale((px_f + 0.5f) * pw, cr)), v3_scale((py_f + 0.5f) * ph, cu));
            v3 ray_dir = v3_normalized(v3_sub(pixel_pos, cp));

            v3 pos = cp;
            v3 dir_ = ray_dir;
            float L2n = v3_norm(v3_cross(dir_, pos));
            float L2_val = L2n * L2n;

            v3 d_pos_dx = v3_make(0, 0, 0), d_dir_dx = v3_make(0, 0, 0);
            v3 d_pos_dy = v3_make(0, 0, 0), d_dir_dy = v3_make(0, 0, 0);
            if (skip_diff == 0) {
                v3 ppx1 = v3_sub(v3_add(tl, v3_scale((px_f + 1.5f) * pw, cr)), v3_scale((py_f + 0.5f) * ph, cu));
                d_dir_dx = v3_sub(v3_normalized(v3_sub(ppx1, cp)), ray_dir);
                v3 ppy1 = v3_sub(v3_add(tl, v3_scale((px_f + 0.5f) * pw, cr)), v3_scale((py_f + 1.5f) * ph, cu));
                d_dir_dy = v3_sub(v3_normalized(v3_sub(ppy1, cp)), ray_dir);
            }

            int escaped = 0, event_horizon_hit = 0;
            v3 escape_dir = v3_make(0, 0, 0);
            v3 accum_disk = v3_make(0, 0, 0);
            float disk_alpha_total = 0.0f;
            int32_t step_count = 0;
            int32_t executed = 0;
            float affine = 0.0f;
            v3 hit_d_pos_dx = v3_make(0, 0, 0), hit_d_pos_dy = v3_make(0, 0, 0);
            const float tan_t = tanf(tilt_rad);

            while (step_count < max_iter) {
                executed++;
                v3 old_pos = pos;
                float old_z = pos.z, old_y = pos.y;
                float r_cur = v3_norm(pos);
                float r_safe = fmaxf(r_cur, r_cap + 1e-3f);
                float far_scale = sqrtf(r_safe / r_cap);
                if (far_scale > max_fac) far_scale = max_fac;
                float q = r_cap / r_safe;
                float near_damp = 1.0f / (1.0f + 2.0f * (q * q * q));
                float dt_fac = far_scale * near_damp;
                if (dt_fac < min_fac) dt_fac = min_fac;
                if (dt_fac > max_fac) dt_fac = max_fac;
                float h = h_base * dt_fac;

                v3 k1p = v3_scale(h, dir_);
                v3 k1d = v3_scale(h, compute_acceleration(pos, L2_val));
                v3 k2p = v3_scale(h, v3_add(dir_, v3_scale(0.5f, k1d)));
                v3 k2d = v3_scale(h, compute_acceleration(v3_add(pos, v3_scale(0.5f, k1p)), L2_val));
                v3 k3p = v3_scale(h, v3_add(dir_, v3_scale(0.5f, k2d)));
                v3 k3d = v3_scale(h, compute_acceleration(v3_add(pos, v3_scale(0.5f, k2p)), L2_val));
                v3 k4p = v3_scale(h, v3_add(dir_, k3d));
                v3 k4d = v3_scale(h, compute_acceleration(v3_add(pos, k3p), L2_val));

                v3 new_pos = v3_add(pos, v3_divs(v3_add(v3_add(v3_add(k1p, v3_scale(2, k2p)), v3_scale(2, k3p)), k4p), 6));
                v3 new_dir = v3_add(dir_, v3_divs(v3_add(v3_add(v3_add(k1d, v3_scale(2, k2d)), v3_scale(2, k3d)), k4d), 6));

                v3 new_d_pos_dx = d_pos_dx, new_d_dir_dx = d_dir_dx;
                v3 new_d_pos_dy = d_pos_dy, new_d_dir_dy = d_dir_dy;
                if (skip_diff == 0) {
                    v3 s2 = v3_add(pos, v3_scale(0.5f, k1p));
                    v3 s3 = v3_add(pos, v3_scale(0.5f, k2p));
                    v3 s4 = v3_add(pos, k3p);
                    {
                        v3 a1p = v3_scale(h, d_dir_dx);
                        v3 a1d = v3_scale(h, compute_acc_jacobian(pos, d_pos_dx, L2_val));
                        v3 a2p = v3_scale(h, v3_add(d_dir_dx, v3_scale(0.5f, a1d)));
                        v3 a2d = v3_scale(h, compute_acc_jacobian(s2, v3_add(d_pos_dx, v3_scale(0.5f, a1p)), L2_val));
                        v3 a3p = v3_scale(h, v3_add(d_dir_dx, v3_scale(0.5f, a2d)));
                        v3 a3d = v3_scale(h, compute_acc_jacobian(s3, v3_add(d_pos_dx, v3_scale(0.5f, a2p)), L2_val));
                        v3 a4p = v3_scale(h, v3_add(d_dir_dx, a3d));
                        v3 a4d = v3_scale(h, compute_acc_jacobian(s4, v3_add(d_pos_dx, a3p), L2_val));
                        new_d_pos_dx = v3_add(d_pos_dx, v3_divs(v3_add(v3_add(v3_add(a1p, v3_scale(2, a2p)), v3_scale(2, a3p)), a4p), 6));
                        new_d_dir_dx = v3_add(d_dir_dx, v3_divs(v3_add(v3_add(v3_add(a1d, v3_scale(2, a2d)), v3_scale(2, a3d)), a4d), 6));
                    }
                    {
                        v3 a1p = v3_scale(h, d_dir_dy);
                        v3 a1d = v3_scale(h, compute_acc_jacobian(pos, d_pos_dy, L2_val));
                        v3 a2p = v3_scale(h, v3_add(d_dir_dy, v3_scale(0.5f, a1d)));
                        v3 a2d = v3_scale(h, compute_acc_jacobian(s2, v3_add(d_pos_dy, v3_scale(0.5f, a1p)), L2_val));
                        v3 a3p = v3_scale(h, v3_add(d_dir_dy, v3_scale(0.5f, a2d)));
                        v3 a3d = v3_scale(h, compute_acc_jacobian(s3, v3_add(d_pos_dy, v3_scale(0.5f, a2p)), L2_val));
                        v3 a4p = v3_scale(h, v3_add(d_dir_dy, a3d));
                        v3 a4d = v3_scale(h, compute_acc_jacobian(s4, v3_add(d_pos_dy, a3p), L2_val));
                        new_d_pos_dy = v3_add(d_pos_dy, v3_divs(v3_add(v3_add(v3_add(a1p, v3_scale(2, a2p)), v3_scale(2, a3p)), a4p), 6));
                        new_d_dir_dy = v3_add(d_dir_dy, v3_divs(v3_add(v3_add(v3_add(a1d, v3_scale(2, a2d)), v3_scale(2, a3d)), a4d), 6));
                    }
                }

                float r = v3_norm(new_pos);
                affine += h;

                if (r < r_cap) { event_horizon_hit = 1; break; }
                else if (r > r_esc) { escaped = 1; escape_dir = v3_normalized(new_dir); break; }
                else if (affine > max_affine) { escaped = 1; escape_dir = v3_normalized(new_dir); break; }

                /* render.py:2928-2932: the differential state is committed HERE, before the
                 * plane test that interpolates it (2947-2949). */
                if (skip_diff == 0) {
                    d_pos_dx = new_d_pos_dx; d_dir_dx = new_d_dir_dx;
                    d_pos_dy = new_d_pos_dy; d_dir_dy = new_d_dir_dy;
                }

                float new_z = new_pos.z, new_y = new_pos.y;
                float f_old = old_z - old_y * tan_t;
                float f_new = new_z - new_y * tan_t;
                if (g_vol.on) {
                    volume_segment(old_pos, new_pos, dir_, tilt_rad, t_offset, cp, r_inner, r_outer, &accum_disk, &disk_alpha_total);
                } else if (f_old * f_new < 0) {
                    float t_frac = f_old / (f_old - f_new + 1e-8f);
                    float hit_x = old_pos.x + t_frac * (new_pos.x - old_pos.x);
                    float hit_y = old_pos.y + t_frac * (new_pos.y - old_pos.y);
                    float hit_r = sqrtf(hit_x * hit_x + hit_y * hit_y);

                    if (skip_diff == 0) {
                        /* NB (render.py:2928-2932 precede 2947-2949): d_pos_dx was ALREADY
                         * overwritten with new_d_pos_dx, so the interpolation degenerates to
                         * d_pos + t*(new - d_pos) with d_pos == new  ==> hit_d_pos = new_d_pos. */
                        hit_d_pos_dx = v3_add(d_pos_dx, v3_scale(t_frac, v3_sub(new_d_pos_dx, d_pos_dx)));
                        hit_d_pos_dy = v3_add(d_pos_dy, v3_scale(t_frac, v3_sub(new_d_pos_dy, d_pos_dy)));
                    }

                    if (r_outer >= hit_r && hit_r >= r_inner) {
                        float hit_z = hit_y * tan_t;
                        v3 hit_pos_vec = v3_make(hit_x, hit_y, hit_z);
                        v3 ray_to_cam = v3_make(-dir_.x, -dir_.y, -dir_.z);
                        v4 disk_rgba;
                        if (p->anti_alias_mode == 0 || skip_diff == 1) {
                            disk_rgba = sample_disk(&sc, hit_x, hit_y, r_inner, r_outer, t_offset);
                        } else {
                            float hit_r_cyl = sqrtf(hit_x * hit_x + hit_y * hit_y + 1e-6f);
                            float dr_dx = (hit_x * hit_d_pos_dx.x + hit_y * hit_d_pos_dx.y) / hit_r_cyl;
                            float dphi_dx = (-hit_y * hit_d_pos_dx.x + hit_x * hit_d_pos_dx.y) / (hit_r_cyl * hit_r_cyl + 1e-6f);
                            float dudx = dphi_dx * (float)dtex_w / (2.0f * PI_F);
                            float dvdx = dr_dx * (float)dtex_h / (r_outer - r_inner);
                            float dr_dy = (hit_x * hit_d_pos_dy.x + hit_y * hit_d_pos_dy.y) / hit_r_cyl;
                            float dphi_dy = (-hit_y * hit_d_pos_dy.x + hit_x * hit_d_pos_dy.y) / (hit_r_cyl * hit_r_cyl + 1e-6f);
                            float dudy = dphi_dy * (float)dtex_w / (2.0f * PI_F);
                            float dvdy = dr_dy * (float)dtex_h / (r_outer - r_inner);
                            float grad_sq_x = dudx * dudx + dvdx * dvdx;
                            float grad_sq_y = dudy * dudy + dvdy * dvdy;
                            float grad_sq = fmaxf(grad_sq_x, grad_sq_y);
                            float lod_diff = logf(fmaxf(grad_sq, 1.0f)) / logf(2.0f) * p->aa_strength;
                            lod_diff = fminf(fmaxf(lod_diff, 0.0f), 3.0f);
                            disk_rgba = sample_disk_mip(&sc, hit_x, hit_y, r_inner, r_outer, t_offset, lod_diff);
                        }
                        v3 disk_col = v3_make(disk_rgba.x, disk_rgba.y, disk_rgba.z);
                        float base_alpha = fminf(disk_rgba.w, 0.999f);
                        float disk_alpha = 1.0f - powf(1.0f - base_alpha, DISK_ALPHA_GAIN);
                        v3 col_shifted = apply_g_factor(disk_col, hit_pos_vec, hit_r, ray_to_cam, cp, r_inner, r_outer, tilt_rad);
                        float front_factor = 1.0f - disk_alpha_total;
                        accum_disk.x += col_shifted.x * disk_alpha * front_factor;
                        accum_disk.y += col_shifted.y * disk_alpha * front_factor;
                        accum_disk.z += col_shifted.z * disk_alpha * front_factor;
                        disk_alpha_total = 1.0f - front_factor * (1.0f - disk_alpha);
                    }
                }
                pos = new_pos;
                dir_ = new_dir;
                step_count += 1;
            }

            v3 bg_color = v3_make(0, 0, 0);
            if (event_horizon_hit) bg_color = v3_make(0, 0, 0);
            else if (escaped) bg_color = sample_skybox(&sc, escape_dir);
            bg_color = v3_scale(1.0f - disk_alpha_total, bg_color);

            size_t o = ((size_t)i * height + j) * 3;
            if (g_escape_out) {
                g_escape_out[o + 0] = escaped ? (double)escape_dir.x : 0.0;
                g_escape_out[o + 1] = escaped ? (double)escape_dir.y : 0.0;
                g_escape_out[o + 2] = escaped ? (double)escape_dir.z : 0.0;
            }
            image_out[o + 0] = bg_color.x; image_out[o + 1] = bg_color.y; image_out[o + 2] = bg_color.z;
            disk_out[o + 0] = clampf(accum_disk.x, 0.0f, 1.0f);
            disk_out[o + 1] = clampf(accum_disk.y, 0.0f, 1.0f);
            disk_out[o + 2] = clampf(accum_disk.z, 0.0f, 1.0f);
            if (steps_out) steps_out[(size_t)i * height + j] = executed;
            total_steps += executed;
        }
    }
    return total_steps;
}

/* ---- _bloom_kernel: render.py:3022-3114 ----------------------------------
 * image (W,H,3) is modified in place by the last loop exactly as the
 * reference does; bright and blur are (W,H,3) scratch/outputs. */
ORACLE_API void oracle_bloom(f32 *image, f32 *bright, f32 *blur, int32_t w, int32_t h,
                             f32 threshold, f32 intensity, int32_t kernel_radius, f32 sigma_scale)
{
#define AT(buf, i, j) ((buf) + ((size_t)(i) * h + (j)) * 3)
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int32_t i = 0; i < w; ++i)
        for (int32_t j = 0; j < h; ++j) {
            const f32 *col = AT(image, i, j);
            float lum = col[0] * 0.2126f + col[1] * 0.7152f + col[2] * 0.0722f;
            f32 *b = AT(bright, i, j);
            if (lum > threshold) { b[0] = col[0]; b[1] = col[1]; b[2] = col[2]; }
            else { b[0] = 0.0f; b[1] = 0.0f; b[2] = 0.0f; }
        }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int32_t i = 0; i < w; ++i)
        for (int32_t j = 0; j < h; ++j) {
            float sum_r = 0, sum_g = 0, sum_b = 0, weight_r = 0, weight_g = 0, weight_b = 0;
            for (int32_t dx = -kernel_radius; dx <= kernel_radius; ++dx) {
                int32_t ni = i + dx;
                if (0 <= ni && ni < w) {
                    float dist_sq = (float)(dx * dx);
                    const f32 *col = AT(bright, ni, j);
                    float w_r = expf(-dist_sq / (25.0f * sigma_scale));
                    float w_g = expf(-dist_sq / (80.0f * sigma_scale));
                    float w_b = expf(-dist_sq / (1600.0f * sigma_scale));
                    sum_r += col[0] * w_r; sum_g += col[1] * w_g; sum_b += col[2] * w_b;
                    weight_r += w_r; weight_g += w_g; weight_b += w_b;
                }
            }
            f32 *o = AT(blur, i, j);
            if (weight_r > 0.0f) { o[0] = sum_r / weight_r; o[1] = sum_g / weight_g; o[2] = sum_b / weight_b; }
            else { o[0] = 0; o[1] = 0; o[2] = 0; }
        }
    memcpy(bright, blur, (size_t)w * h * 3 * sizeof(f32));
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int32_t i = 0; i < w; ++i)
        for (int32_t j = 0; j < h; ++j) {
            float sum_r = 0, sum_g = 0, sum_b = 0, weight_r = 0, weight_g = 0, weight_b = 0;
            for (int32_t dy = -kernel_radius; dy <= kernel_radius; ++dy) {
                int32_t nj = j + dy;
                if (0 <= nj && nj < h) {
                    float dist_sq = (float)(dy * dy);
                    const f32 *col = AT(bright, i, nj);
                    float w_r = expf(-dist_sq / (25.0f * sigma_scale));
                    float w_g = expf(-dist_sq / (80.0f * sigma_scale));
                    float w_b = expf(-dist_sq / (1600.0f * sigma_scale));
                    sum_r += col[0] * w_r; sum_g += col[1] * w_g; sum_b += col[2] * w_b;
                    weight_r += w_r; weight_g += w_g; weight_b += w_b;
                }
            }
            f32 *o = AT(blur, i, j);
            if (weight_r > 0.0f) { o[0] = sum_r / weight_r; o[1] = sum_g / weight_g; o[2] = sum_b / weight_b; }
            else { o[0] = 0; o[1] = 0; o[2] = 0; }
        }
    for (int32_t i = 0; i < w; ++i)
        for (int32_t j = 0; j < h; ++j) {
            f32 *c = AT(image, i, j);
            const f32 *b = AT(blur, i, j);
            for (int k = 0; k < 3; ++k) c[k] = clampf(c[k] + b[k] * intensity, 0.0f, 1.0f);
        }
#undef AT
}

/* ---- _compose_disk_texture_kernel: render.py:3169-3257 ------------------- */
ORACLE_API void oracle_compose_disk_texture(f32 *disk_tex, const f32 *comp, const f32 *omega,
                                            const f32 *edge, const f32 *stats, const f32 *row_stats,
                                            int32_t n_r, int32_t n_phi, f32 t_offset, int32_t enable_rt,
                                            f32 color_temp_val)
{
    float density_p98 = stats[0];
    float struct_scale = stats[1];
    float t_factor = (color_temp_val - 4500.0f) / (6500.0f - 2700.0f);
    float T_min = 2000.0f + t_factor * 1000.0f;
    float T_max = 9000.0f + t_factor * 3000.0f;
    float rt_w = 0.20f;
    if (enable_rt == 0) rt_w = 0.0f;
    const size_t plane = (size_t)n_r * n_phi;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int32_t ri = 0; ri < n_r; ++ri)
        for (int32_t phi_i = 0; phi_i < n_phi; ++phi_i) {
            float omega_val = omega[ri];
            int32_t shift = (int32_t)(t_offset * omega_val / (2.0f * PI_F) * (float)n_phi);
            int32_t src = pymod(phi_i + shift, n_phi);
            if (src < 0) src += n_phi;
            size_t q = (size_t)ri * n_phi + src;
            float tb = comp[0 * plane + q], sp = comp[1 * plane + q], sp_t = comp[2 * plane + q];
            float turb = comp[3 * plane + q], turb_t = comp[4 * plane + q];
            float arc = comp[5 * plane + q], arc_t = comp[6 * plane + q];
            float rt = comp[7 * plane + q], rt_t = comp[8 * plane + q];
            float hs = comp[9 * plane + q], hs_t = comp[10 * plane + q];
            float az = comp[11 * plane + q], dm = comp[12 * plane + q];

            float density = (0.15f + 0.10f * sp + 0.30f * turb + 0.20f * hs + 0.30f * arc + rt_w * rt) * dm * edge[ri];
            density = fminf(fmaxf(density / (density_p98 + 1e-6f), 0.0f), 1.0f);

            float temp_struct = (sp_t + turb_t + arc_t + rt_t + hs_t) * dm;
            float ts_scaled = fminf(fmaxf(temp_struct / (struct_scale + 1e-6f) * 0.8f, 0.0f), 1.2f);

            float max_r = row_stats[ri * 2 + 0];
            float p70_r = row_stats[ri * 2 + 1];
            float ceiling = fmaxf(p70_r, 0.05f);
            float tb_clamped = fminf(tb, ceiling);
            tb_clamped = fminf(tb_clamped, max_r);

            float temperature = fminf(fmaxf(fmaxf(tb_clamped, ts_scaled), 0.0f), 1.0f);
            float temp_aniso = fminf(fmaxf(temperature * (0.9f + 0.25f * az), 0.0f), 1.0f);
            float T_K = T_min + temp_aniso * (T_max - T_min);
            v3 bb = color_temp_to_tint(T_K);
            float bb_b = fminf(bb.z, bb.x);
            float lum = fminf(fmaxf(sqrtf(temp_aniso), 0.0f), 1.0f);

            f32 *o = disk_tex + ((size_t)ri * n_phi + phi_i) * 4;
            o[0] = fminf(fmaxf(bb.x * lum, 0.0f), 1.0f);
            o[1] = fminf(fmaxf(bb.y * lum, 0.0f), 1.0f);
            o[2] = fminf(fmaxf(bb_b * lum, 0.0f), 1.0f);
            o[3] = density;
        }
}

/* ---- mip kernels: render.py:3261-3283, driven as at render.py:3761-3767 ---
 * mips: (levels, n_r, n_phi, 4) padded storage, must be zero-initialised by the
 * caller (Taichi fields start at zero); builds levels 0..levels-1. */
ORACLE_API void oracle_build_mips(f32 *mips, const f32 *base, int32_t n_r, int32_t n_phi, int32_t levels)
{
    const size_t lvl = (size_t)n_r * n_phi * 4;
    memcpy(mips, base, lvl * sizeof(f32));
    int32_t h = n_r, w = n_phi;
    for (int32_t level = 1; level < levels; ++level) {
        int32_t dst_h = h / 2, dst_w = w / 2;
        const f32 *src = mips + (size_t)(level - 1) * lvl;
        f32 *dst = mips + (size_t)level * lvl;
        for (int32_t ri = 0; ri < dst_h; ++ri)
            for (int32_t pi = 0; pi < dst_w; ++pi)
                for (int c = 0; c < 4; ++c) {
                    float a = src[((size_t)(ri * 2) * n_phi + pi * 2) * 4 + c];
                    float b = src[((size_t)(ri * 2) * n_phi + pi * 2 + 1) * 4 + c];
                    float cc = src[((size_t)(ri * 2 + 1) * n_phi + pi * 2) * 4 + c];
                    float d = src[((size_t)(ri * 2 + 1) * n_phi + pi * 2 + 1) * 4 + c];
                    dst[((size_t)ri * n_phi + pi) * 4 + c] = (a + b + cc + d) / 4.0f;
                }
        h /= 2; w /= 2;
    }
}

/* ---- _generate_background_kernel: render.py:3332-3451 -------------------- */
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

ORACLE_API void oracle_generate_background(f32 *comp, int32_t n_r, int32_t n_phi, int32_t az_freq,
                                           f32 az_shear, f32 r_inner, f32 r_outer, f32 t)
{
    const float pi2 = 2.0f * PI_F;
    const size_t plane = (size_t)n_r * n_phi;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int32_t ri = 0; ri < n_r; ++ri)
        for (int32_t phi_i = 0; phi_i < n_phi; ++phi_i) {
            size_t q = (size_t)ri * n_phi + phi_i;
            float r = (float)ri / (float)n_r;
            float phi = (float)phi_i / (float)n_phi * pi2;
            float r_phys = r_inner + (r_outer - r_inner) * r;
            float omega = sqrtf(0.5f / (r_phys * r_phys * r_phys + 1e-6f));
            float phi_rot = phi + omega * t;
            float cx = cosf(phi_rot);
            float cy = sinf(phi_rot);

            float decay = powf(fmaxf(1.0f - r, 0.0f), 1.3f);
            float tb_noise = clamp01(0.5f + 0.5f * fbm_3d(cx * 8.0f, cy * 8.0f, r * 8.0f + t * 0.05f, 4, 0.6f, 2.0f));
            comp[0 * plane + q] = decay * (0.85f + 0.15f * tb_noise) * 0.25f;
            comp[1 * plane + q] = 0.0f;
            comp[2 * plane + q] = 0.0f;

            float t_coarse = clamp01(0.5f + 0.5f * fbm_3d(cx * 8.0f, cy * 8.0f, r * 4.0f + t * 0.06f, 3, 0.45f, 2.0f)) * 0.08f;
            float t_mid = clamp01(0.5f + 0.5f * fbm_3d(cx * 24.0f, cy * 24.0f, r * 12.0f + t * 0.08f, 4, 0.45f, 2.0f)) * 0.15f;
            float t_fine = clamp01(0.5f + 0.5f * fbm_3d(cx * 80.0f, cy * 80.0f, r * 40.0f + t * 0.1f, 5, 0.45f, 2.0f)) * 0.25f;
            float t_extra = clamp01(0.5f + 0.5f * fbm_3d(cx * 200.0f, cy * 200.0f, r * 100.0f + t * 0.12f, 4, 0.4f, 2.0f)) * 0.22f;
            float t_ultra = clamp01(0.5f + 0.5f * fbm_3d(cx * 400.0f, cy * 400.0f, r * 200.0f + t * 0.15f, 3, 0.35f, 2.0f)) * 0.18f;
            float t_pixel = clamp01(simplex_noise_3d(cx * 800.0f, cy * 800.0f, r * 400.0f + t * 0.2f)) * 0.12f;
            float turb = clamp01(t_coarse + t_mid + t_fine + t_extra + t_ultra + t_pixel);
            comp[3 * plane + q] = turb;
            comp[4 * plane + q] = 0.05f * turb;

            float shear = powf(r, 1.2f) * az_shear;
            float az_wave = 0.5f + 0.5f * sinf((phi_rot + shear) * (float)az_freq);
            float az_n = clamp01(0.5f + 0.5f * fbm_3d(cx * 3.0f, cy * 3.0f, r * 3.0f + t * 0.04f, 3, 0.5f, 2.0f));
            comp[11 * plane + q] = az_wave * az_n;

            float d_coarse = clamp01(0.5f + 0.5f * fbm_3d(cx * 8.0f, cy * 8.0f, r * 4.0f + t * 0.003f, 3, 0.5f, 2.0f)) * 0.05f;
            float d_mid = clamp01(0.5f + 0.5f * fbm_3d(cx * 32.0f, cy * 32.0f, r * 16.0f + t * 0.005f, 3, 0.5f, 2.0f)) * 0.15f;
            float d_fine = clamp01(0.5f + 0.5f * fbm_3d(cx * 100.0f, cy * 100.0f, r * 50.0f + t * 0.006f, 4, 0.45f, 2.0f)) * 0.30f;
            float d_extra = clamp01(0.5f + 0.5f * fbm_3d(cx * 250.0f, cy * 250.0f, r * 125.0f + t * 0.008f, 4, 0.4f, 2.0f)) * 0.30f;
            float d_pixel = clamp01(simplex_noise_3d(cx * 500.0f, cy * 500.0f, r * 250.0f + t * 0.01f)) * 0.20f;
            float disturb_raw = (d_coarse + d_mid + d_fine + d_extra + d_pixel) * 1.4f;
            disturb_raw = fminf(fmaxf(disturb_raw, 0.05f), 1.0f);
            float radial_preserve = 0.6f + 0.4f * r;
            comp[12 * plane + q] = fminf(fmaxf(disturb_raw * radial_preserve, 0.1f), 1.0f);
        }
}

ORACLE_API int32_t oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int32_t n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
