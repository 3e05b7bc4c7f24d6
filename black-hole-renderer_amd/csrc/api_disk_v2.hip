// api_disk_v2.hip -- extern "C" entry point of include/bhr_disk_v2.h
#include "bhr_internal.h"

namespace {
struct DevBuf {
    double *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int32_t alloc(size_t n) {
        hipError_t e = hipMalloc((void **)&p, n * sizeof(double));
        if (e != hipSuccess) { p = nullptr; return bhr_fail(BHR_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
        return BHR_OK;
    }
};
}  // namespace

extern "C" int32_t bhr_disk_v2_eval(bhr_ctx *ctx, const bhr_disk_v2_params *p, int32_t field, const double *r,
                                    const double *z, const double *phi, int64_t n, double norm_shear,
                                    double norm_hotspot, double *out, double *max_out) {
    if (!ctx || !p || !r || !out || n < 0) return bhr_fail(BHR_ERR_INVALID, "bhr_disk_v2_eval: bad argument");
    if (field < BHR_DV2_H || field > BHR_DV2_F_TOTAL) return bhr_fail(BHR_ERR_INVALID, "bhr_disk_v2_eval: field %d", field);
    if (p->shear_components < 0 || p->shear_components > BHR_DV2_MAX_TERMS || p->hotspot_count < 0 ||
        p->hotspot_count > BHR_DV2_MAX_TERMS)
        return bhr_fail(BHR_ERR_INVALID, "bhr_disk_v2_eval: at most %d shear components / hotspots", BHR_DV2_MAX_TERMS);
    const bool needs_z = field == BHR_DV2_W_Z || field == BHR_DV2_MASK_VOL || field == BHR_DV2_RHO || field == BHR_DV2_T;
    const bool needs_phi = field >= BHR_DV2_F_MODE;
    if ((needs_z && !z) || (needs_phi && !phi)) return bhr_fail(BHR_ERR_INVALID, "bhr_disk_v2_eval: field %d needs %s", field, needs_z ? "z" : "phi");
    if (n == 0) return BHR_OK;
    BHR_TRY(bhr_enter(ctx));
    DevBuf dr, dz, dphi, dout, daux, dmax;
    const size_t bytes = (size_t)n * sizeof(double);
    int32_t rc;
    if ((rc = dr.alloc(n)) || (rc = dout.alloc(n)) || (rc = dmax.alloc(2))) return rc;
    BHR_HIP(hipMemcpyAsync(dr.p, r, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (needs_z) {
        if ((rc = dz.alloc(n))) return rc;
        BHR_HIP(hipMemcpyAsync(dz.p, z, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (needs_phi) {
        if ((rc = dphi.alloc(n))) return rc;
        BHR_HIP(hipMemcpyAsync(dphi.p, phi, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (field == BHR_DV2_F_TOTAL && (rc = daux.alloc(n))) return rc;
    if ((rc = bhr_launch_disk_v2(ctx, p, dr.p, dz.p, dphi.p, n, field, dout.p, daux.p, dmax.p, norm_shear, norm_hotspot))) return rc;
    BHR_HIP(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    double mx[2] = {0, 0};
    BHR_HIP(hipMemcpyAsync(mx, dmax.p, sizeof(mx), hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    if (max_out) { max_out[0] = mx[0]; max_out[1] = mx[1]; }
    return BHR_OK;
}

extern "C" int32_t bhr_set_disk_source(bhr_ctx *ctx, int32_t source, const bhr_disk_v2_params *p, double norm_shear,
                                       double norm_hotspot, double t_peak) {
    if (!ctx || (source != BHR_DISK_TEXTURE && source != BHR_DISK_V2 && source != BHR_DISK_V2_VOLUME))
        return bhr_fail(BHR_ERR_INVALID, "bhr_set_disk_source: bad argument");
    BHR_TRY(bhr_enter(ctx));
    if (source != BHR_DISK_TEXTURE) {
        if (!p || !(norm_shear > 0.0) || !(norm_hotspot > 0.0) || !(t_peak > 0.0))
            return bhr_fail(BHR_ERR_INVALID, "bhr_set_disk_source: Disk V2 needs parameters and positive normalisation constants");
        if (!(p->r_out > p->r_in) || !(p->r_in > 0.0) || !(p->h0 > 0.0) || p->shear_components < 0 ||
            p->shear_components > BHR_DV2_MAX_TERMS || p->hotspot_count < 0 || p->hotspot_count > BHR_DV2_MAX_TERMS)
            return bhr_fail(BHR_ERR_INVALID, "bhr_set_disk_source: inconsistent Disk V2 parameters");
        if (!ctx->d_dv2_params) BHR_HIP(hipMalloc((void **)&ctx->d_dv2_params, sizeof(bhr_disk_v2_params)));
        BHR_HIP(hipMemcpyAsync(ctx->d_dv2_params, p, sizeof(*p), hipMemcpyHostToDevice, ctx->stream));
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        ctx->dv2_norm[0] = norm_shear;
        ctx->dv2_norm[1] = norm_hotspot;
        ctx->dv2_norm[2] = t_peak;
        // bounding slab of the volume: H(r) = h0 r (r / r_in)^beta is monotonic, its maximum sits at an end
        const double h_in = p->h0 * p->r_in, h_out = p->h0 * p->r_out * pow(p->r_out / p->r_in, p->beta_h);
        ctx->vol_opts[2] = h_in > h_out ? h_in : h_out;
        ctx->vol_opts[3] = sqrt(p->r_out * p->r_out + ctx->vol_opts[2] * ctx->vol_opts[2]);
        if (ctx->vol_substeps == 0) {   // options never set: defaults
            ctx->vol_opts[0] = 4.0;
            ctx->vol_opts[1] = 1.0;
            ctx->vol_substeps = 2;
        }
    }
    ctx->disk_source = source;
    return BHR_OK;
}

extern "C" int32_t bhr_set_disk_volume_options(bhr_ctx *ctx, double absorption, double grazing_gain, int32_t substeps) {
    if (!ctx || !(absorption >= 0.0) || !(grazing_gain >= 0.0) || substeps < 1 || substeps > 16)
        return bhr_fail(BHR_ERR_INVALID, "bhr_set_disk_volume_options: absorption/grazing_gain must be >= 0, substeps in 1..16");
    ctx->vol_opts[0] = absorption;
    ctx->vol_opts[1] = grazing_gain;
    ctx->vol_substeps = substeps;
    return BHR_OK;
}
