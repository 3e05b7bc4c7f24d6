// disk_v2.hip -- the reference's analytic "Disk V2" model evaluated on the device in binary64
// (the reference is float64 NumPy and is not wired into its renderer: docs/design_ad_v2.md Phase 4).
//
//   geometry.py            smoothstep 15-47, H(r) 50-77, radial mask/weight 80-185,
//                          vertical weight 188-235, volume mask 238-280
//   physical_fields.py     Omega 21-49, rho_mid 52-79, T_mid 82-116, rho(r,z) 119-160, T(r,z) 163-205
//   structure_modulations  F_mode 95-142, F_shear 145-207, F_hotspot 210-289, product 292-334
//
// The random draws of shear / hotspot (numpy default_rng(seed) / default_rng(seed + 1)) are made by
// the host binding in the reference's order and arrive here as coefficient tables.  _normalize_signed
// divides by the maximum |raw| over the evaluated array: the kernel returns the raw signed sums and a
// device-side max reduction, and a second pass normalises -- or uses a fixed normalisation constant
// when one is supplied (per-ray shading needs a constant that does not depend on which rays hit).
#include "disk_v2_device.h"

using namespace dv2;

namespace {

__device__ __forceinline__ void atomic_max_f64(double *addr, double v) {   // v >= 0
    unsigned long long *a = (unsigned long long *)addr;
    atomicMax(a, (unsigned long long)__double_as_longlong(v));               // order-preserving for v >= 0
}

// pass 1: plain fields, or the raw signed sums of shear / hotspot + their max |.|
__global__ __launch_bounds__(256) void disk_v2_kernel(bhr_disk_v2_params p, const double *__restrict__ r,
                                                      const double *__restrict__ z, const double *__restrict__ phi,
                                                      long long n, int field, double *__restrict__ out,
                                                      double *__restrict__ aux, double *__restrict__ maxabs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double m0 = 0.0, m1 = 0.0;
    if (i < n) {
        double ri = r[i], zi = z ? z[i] : 0.0, ph = phi ? phi[i] : 0.0, v = 0.0;
        switch (field) {
            case BHR_DV2_H: v = half_thickness(ri, p); break;
            case BHR_DV2_MASK_R: v = radial_mask(ri, p) ? 1.0 : 0.0; break;
            case BHR_DV2_W_R: v = radial_weight(ri, p); break;
            case BHR_DV2_W_Z: v = vertical_weight(ri, zi, p); break;
            case BHR_DV2_MASK_VOL: v = volume_mask(ri, zi, p) ? 1.0 : 0.0; break;
            case BHR_DV2_OMEGA: v = omega_field(ri, p); break;
            case BHR_DV2_RHO_MID: v = rho_mid(ri, p); break;
            case BHR_DV2_T_MID: v = t_mid(ri, p); break;
            case BHR_DV2_RHO: v = rho_field(ri, zi, p); break;
            case BHR_DV2_T: v = t_field(ri, zi, p); break;
            case BHR_DV2_F_MODE: v = mode_factor(ri, ph, p); break;
            case BHR_DV2_F_SHEAR: v = raw_shear(ri, ph, p); m0 = fabs(v); break;
            case BHR_DV2_F_HOTSPOT: v = raw_hotspot(ri, ph, p); m0 = fabs(v); break;
            case BHR_DV2_F_TOTAL:
                v = raw_shear(ri, ph, p);
                m0 = fabs(v);
                aux[i] = raw_hotspot(ri, ph, p);
                m1 = fabs(aux[i]);
                break;
        }
        out[i] = v;
    }
    if (field >= BHR_DV2_F_SHEAR) {
        for (int off = 32; off > 0; off >>= 1) {
            m0 = fmax(m0, __shfl_down(m0, off, 64));
            m1 = fmax(m1, __shfl_down(m1, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomic_max_f64(maxabs + 0, m0);
            if (field == BHR_DV2_F_TOTAL) atomic_max_f64(maxabs + 1, m1);
        }
    }
}

// pass 2: 1 + strength * raw / max|raw| inside the disk, 1 outside (structure_modulations.py:27-44, 204-206)
__global__ __launch_bounds__(256) void disk_v2_normalize_kernel(bhr_disk_v2_params p, const double *__restrict__ r,
                                                                const double *__restrict__ phi, long long n, int field,
                                                                double *__restrict__ out, const double *__restrict__ aux,
                                                                const double *__restrict__ maxabs, double norm0,
                                                                double norm1) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double m0 = norm0 > 0.0 ? norm0 : maxabs[0], m1 = norm1 > 0.0 ? norm1 : maxabs[1];
    bool inside = radial_weight(r[i], p) > 0.0;
    auto signed_of = [](double raw, double m) { return m <= DBL_EPSILON ? 0.0 : raw / m; };
    if (field == BHR_DV2_F_SHEAR) {
        out[i] = inside ? 1.0 + p.shear_strength * signed_of(out[i], m0) : 1.0;
    } else if (field == BHR_DV2_F_HOTSPOT) {
        out[i] = inside ? 1.0 + p.hotspot_strength * signed_of(out[i], m0) : 1.0;
    } else {  // F_TOTAL: out = raw shear, aux = raw hotspot (drawn with seed + 1 on the host)
        double sh = inside ? 1.0 + p.shear_strength * signed_of(out[i], m0) : 1.0;
        double hs = inside ? 1.0 + p.hotspot_strength * signed_of(aux[i], m1) : 1.0;
        double v = mode_factor(r[i], phi[i], p) * sh * hs;
        out[i] = inside ? v : 1.0;
    }
}

}  // namespace

int32_t bhr_launch_disk_v2(bhr_ctx *ctx, const bhr_disk_v2_params *p, const double *d_r, const double *d_z,
                           const double *d_phi, int64_t n, int32_t field, double *d_out, double *d_aux,
                           double *d_maxabs, double norm0, double norm1) {
    BHR_HIP(hipMemsetAsync(d_maxabs, 0, 2 * sizeof(double), ctx->stream));
    int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(disk_v2_kernel, dim3(blocks), dim3(256), 0, ctx->stream, *p, d_r, d_z, d_phi, (long long)n,
                       field, d_out, d_aux, d_maxabs);
    if (field >= BHR_DV2_F_SHEAR)
        hipLaunchKernelGGL(disk_v2_normalize_kernel, dim3(blocks), dim3(256), 0, ctx->stream, *p, d_r, d_phi,
                           (long long)n, field, d_out, d_aux, d_maxabs, norm0, norm1);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}
