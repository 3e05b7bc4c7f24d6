// skyglow.hip -- the Milky-Way glow of the procedural skybox on the device (reference: generate_skybox,
// render.py:296-341, host NumPy there; 0.4 s of its 0.5 s at 2048 x 1024).
//
// The random parts of the sky -- nebula noise and star splats, whose result depends on NumPy's random stream
// and on the order of np.add.at -- stay on the host (skybox.py, bit-identical to the reference).  The glow is a
// closed-form function of the galactic coordinates of each texel, evaluated here in binary64 exactly as the
// NumPy expressions are written, added to the uploaded sky and clipped:
//   sky = clip(f32(f64(sky) + glow * (1, 0.95, 0.85)), 0, 1)
// Differences to the host version come from the last bit of the f64 transcendentals only (<= 6e-8 after the
// f32 rounding).
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

}  // namespace

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
