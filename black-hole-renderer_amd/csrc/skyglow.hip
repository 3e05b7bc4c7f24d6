// skyglow.hip -- the procedural skybox on the device (reference: generate_skybox, render.py:153-341, host NumPy there).
//
// What depends on NumPy's random stream stays on the host and arrives as small tables (skybox.sky_tables): the
// 1/16-resolution nebula noise already quantised to u8, and per star its centre, colour and the 81 values of its
// 9 x 9 Gaussian blob (host f32 exp, 1.9 MB for 6000 stars).  Everything per texel happens here, bit-identical to
// the NumPy / Pillow result:
//   * nebula: Pillow's BILINEAR resize of the u8 noise, reproduced exactly -- two fixed-point passes (22-bit
//     coefficients computed by the host with Pillow's formulae, u8 rounding after each pass: Resample.c), then
//     sky = f32(f64(0.003f) + u8 / 255.0 * 0.04) as `sky += resized / 255.0 * 0.04` evaluates;
//   * stars: np.add.at(sky, (py, px), colour * value) adds in star order, patch order within a star; a texel
//     gathers its contributions in that very order (scatter-add as an ordered gather, no atomics), with NumPy's
//     index arithmetic: f32 centre + offset, truncation toward zero, floored modulo in x, rows outside dropped;
//   * Milky-Way glow: a closed-form function of the galactic coordinates of each texel, evaluated in binary64
//     exactly as the NumPy expressions are written, added and clipped:
//       sky = clip(f32(f64(sky) + glow * (1, 0.95, 0.85)), 0, 1)
//     (differences to the host version: the last bit of the f64 transcendentals, <= 6e-8 after the f32 rounding).
#include "bhr_internal.h"

namespace {

__global__ __launch_bounds__(256) void sky_glow_kernel(float *__restrict__ sky, int tex_h, int tex_w, double v_step, double u_step) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long long)tex_h * tex_w) return;
    const int row = (int)(p / tex_w), col = (int)(p - (long long)row * tex_w);
    const double PI = 3.141592653589793;
    // np.linspace(0, pi, tex_h)[row], np.linspace(0, 2 pi, tex_w)[col]: i * step, the last sample is the stop value
    const double vv = row == tex_h - 1 ? PI : (double)row * v_step;
    const double uu = col == tex_w - 1 ? 2 * PI : (double)col * u_step;
    const double INCL = 62.87 * (PI / 180.0), RA0 = 266.4 * (PI / 180.0);       // np.radians(x) = x * (pi / 180)
    const double dec = PI / 2 - vv;
    const double sd = sin(dec), cd = cos(dec), ci = cos(INCL), si = sin(INCL);
    const double sr = sin(uu - RA0), cr = cos(uu - RA0);
    const double sinb = sd * ci - cd * si * sr;
    const double b = asin(fmin(fmax(sinb, -1.0), 1.0));
    const double sin_l_cos_b = cd * ci * sr + sd * si;
    const double cos_l_cos_b = cd * cr;
    const double lon = atan2(sin_l_cos_b, cos_l_cos_b);
    const double r6 = 6.0 * (PI / 180.0), r8 = 8.0 * (PI / 180.0), r15 = 15.0 * (PI / 180.0), r30 = 30.0 * (PI / 180.0);
    double q = b / r6;
    double glow = 0.10 * exp(-0.5 * (q * q));                                   // MILKY_WAY_GLOW
    glow += 0.08 * exp(-0.5 * (lon * lon + b * b) / (r15 * r15));               // GALACTIC_CENTER_GLOW
    const double arms = 0.4 + 0.6 * (0.5 + 0.5 * cos(4 * lon + r30));
    q = b / r8;
    const double near_plane = exp(-0.5 * (q * q));
    glow *= (1.0 - near_plane) + near_plane * arms;
    float *t = sky + p * 3;
    const double tint[3] = {1.0, 0.95, 0.85};
    for (int c = 0; c < 3; ++c) {
        const float s = (float)((double)t[c] + glow * tint[c]);
        t[c] = fminf(fmaxf(s, 0.0f), 1.0f);
    }
}

// Pillow ImagingResampleHorizontal_8bpc on an (h, w, 3) u8 image -> (h, out_w, 3) u8
__global__ void nebula_h_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int *__restrict__ k,
                                const int *__restrict__ bounds, int ksize, int h, int w, int out_w) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xx >= out_w) return;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    for (int c = 0; c < 3; ++c) {
        int ss = 1 << 21;
        for (int x = 0; x < n; ++x) ss += (int)src[((size_t)y * w + xmin + x) * 3 + c] * k[xx * ksize + x];
        ss >>= 22;
        dst[((size_t)y * out_w + xx) * 3 + c] = (uint8_t)(ss < 0 ? 0 : ss > 255 ? 255 : ss);
    }
}

__device__ __forceinline__ int floor_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// vertical resize pass + base level + the stars that touch this texel, in NumPy's accumulation order
__global__ __launch_bounds__(256) void sky_build_kernel(float *__restrict__ sky, const uint8_t *__restrict__ tmp, const int *__restrict__ kv,
                                                       const int *__restrict__ bv, int ksize_v, int tex_h, int tex_w, int n_stars,
                                                       const float *__restrict__ cx, const float *__restrict__ cy,
                                                       const float *__restrict__ colors, const float *__restrict__ vals, int R) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long long)tex_h * tex_w) return;
    const int row = (int)(p / tex_w), col = (int)(p - (long long)row * tex_w);
    float t[3];
    const int ymin = bv[2 * row], ny = bv[2 * row + 1];
    for (int c = 0; c < 3; ++c) {
        int ss = 1 << 21;
        for (int y = 0; y < ny; ++y) ss += (int)tmp[((size_t)(ymin + y) * tex_w + col) * 3 + c] * kv[row * ksize_v + y];
        ss >>= 22;
        const int u8 = ss < 0 ? 0 : ss > 255 ? 255 : ss;
        t[c] = (float)((double)0.003f + (double)u8 / 255.0 * 0.04);      // f32 sky += f64 array: added in f64, stored as f32
    }
    const int side = 2 * R + 1;
    const float reach = (float)(R + 1);
    for (int s = 0; s < n_stars; ++s) {
        const float sx = cx[s], sy = cy[s];
        if (fabsf((float)row - sy) > reach) continue;
        float ddx = fabsf((float)col - sx);
        ddx = fminf(ddx, (float)tex_w - ddx);                            // the blob wraps in x
        if (ddx > reach) continue;
        const float *v = vals + (size_t)s * side * side;
        for (int iy = 0; iy < side; ++iy) {
            const int py = (int)(sy + (float)(iy - R));                  // f32 add, truncation toward zero (astype(int))
            if (py != row) continue;                                     // rows outside [0, tex_h) never match
            for (int ix = 0; ix < side; ++ix) {
                const int px = floor_mod((int)(sx + (float)(ix - R)), tex_w);
                if (px != col) continue;
                const float w = v[iy * side + ix];
                t[0] += colors[3 * s + 0] * w;
                t[1] += colors[3 * s + 1] * w;
                t[2] += colors[3 * s + 2] * w;
            }
        }
    }
    float *o = sky + p * 3;
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}

}  // namespace

extern "C" int32_t bhr_skybox_build(bhr_ctx *ctx, int32_t tex_h, int32_t tex_w, const uint8_t *coarse_rgb, int32_t coarse_h,
                                    int32_t coarse_w, const int32_t *kh, const int32_t *bounds_h, int32_t ksize_h,
                                    const int32_t *kv, const int32_t *bounds_v, int32_t ksize_v, int32_t n_stars,
                                    const float *cx, const float *cy, const float *colors, const float *vals, int32_t patch_r) {
    if (!ctx || !coarse_rgb || !kh || !bounds_h || !kv || !bounds_v || tex_h < 2 || tex_w < 2 || coarse_h < 1 || coarse_w < 1 ||
        ksize_h < 1 || ksize_v < 1 || n_stars < 0 || patch_r < 0 || (n_stars > 0 && (!cx || !cy || !colors || !vals)))
        return bhr_fail(BHR_ERR_INVALID, "bhr_skybox_build: bad argument");
    BHR_TRY(bhr_enter(ctx));
    if (!ctx->d_skybox || ctx->sky_h != tex_h || ctx->sky_w != tex_w)
        return bhr_fail(BHR_ERR_STATE, "bhr_skybox_build: set a (%d, %d, 3) skybox first (bhr_set_skybox allocates it)", tex_h, tex_w);
    const size_t side = (size_t)(2 * patch_r + 1);
    uint8_t *d_coarse = nullptr, *d_tmp = nullptr;
    int *d_kh = nullptr, *d_bh = nullptr, *d_kv = nullptr, *d_bv = nullptr;
    float *d_star = nullptr;
    auto cleanup = [&]() {
        void *b[] = {d_coarse, d_tmp, d_kh, d_bh, d_kv, d_bv, d_star};
        for (void *q : b) if (q) (void)hipFree(q);
    };
    auto up = [&](void **d, const void *h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 1);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, ctx->stream);
        return e;
    };
    hipError_t e = up((void **)&d_coarse, coarse_rgb, (size_t)coarse_h * coarse_w * 3);
    if (e == hipSuccess) e = hipMalloc((void **)&d_tmp, (size_t)coarse_h * tex_w * 3);
    if (e == hipSuccess) e = up((void **)&d_kh, kh, (size_t)tex_w * ksize_h * sizeof(int));
    if (e == hipSuccess) e = up((void **)&d_bh, bounds_h, (size_t)tex_w * 2 * sizeof(int));
    if (e == hipSuccess) e = up((void **)&d_kv, kv, (size_t)tex_h * ksize_v * sizeof(int));
    if (e == hipSuccess) e = up((void **)&d_bv, bounds_v, (size_t)tex_h * 2 * sizeof(int));
    // one allocation for the star tables: cx | cy | colours | values
    const size_t n = (size_t)n_stars, star_floats = n * (2 + 3 + side * side);
    if (e == hipSuccess) e = hipMalloc((void **)&d_star, (star_floats ? star_floats : 1) * sizeof(float));
    if (e == hipSuccess && n) {
        e = hipMemcpyAsync(d_star, cx, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_star + n, cy, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_star + 2 * n, colors, 3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_star + 5 * n, vals, n * side * side * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        // bounds were validated by the host that computed them; the kernels index src[xmin .. xmin + n) only
        hipLaunchKernelGGL(nebula_h_kernel, dim3((tex_w + 255) / 256, coarse_h), dim3(256), 0, ctx->stream, d_coarse, d_tmp, d_kh, d_bh,
                           ksize_h, coarse_h, coarse_w, tex_w);
        const long long px = (long long)tex_h * tex_w;
        hipLaunchKernelGGL(sky_build_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_skybox, d_tmp, d_kv, d_bv,
                           ksize_v, tex_h, tex_w, n_stars, d_star, d_star + n, d_star + 2 * n, d_star + 5 * n, patch_r);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the host tables may go away after the call
    cleanup();
    if (e != hipSuccess) return bhr_fail(BHR_ERR_HIP, "bhr_skybox_build: %s", hipGetErrorString(e));
    return BHR_OK;
}

extern "C" int32_t bhr_skybox_add_glow(bhr_ctx *ctx) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "bhr_skybox_add_glow: null ctx");
    if (!ctx->d_skybox) return bhr_fail(BHR_ERR_STATE, "bhr_skybox_add_glow: no skybox set (bhr_set_skybox)");
    if (ctx->sky_h < 2 || ctx->sky_w < 2) return bhr_fail(BHR_ERR_INVALID, "bhr_skybox_add_glow: skybox %dx%d too small", ctx->sky_h, ctx->sky_w);
    BHR_TRY(bhr_enter(ctx));
    const double PI = 3.141592653589793;
    const long long n = (long long)ctx->sky_h * ctx->sky_w;
    hipLaunchKernelGGL(sky_glow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_skybox, ctx->sky_h,
                       ctx->sky_w, PI / (double)(ctx->sky_h - 1), 2 * PI / (double)(ctx->sky_w - 1));
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

extern "C" int32_t bhr_get_skybox(bhr_ctx *ctx, float *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_get_skybox: bad argument");
    if (!ctx->d_skybox) return bhr_fail(BHR_ERR_STATE, "bhr_get_skybox: no skybox set");
    BHR_TRY(bhr_enter(ctx));
    BHR_HIP(hipMemcpyAsync(out, ctx->d_skybox, (size_t)ctx->sky_h * ctx->sky_w * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    return BHR_OK;
}
