// disk_v2_device.h -- binary64 device functions of the Disk V2 model, shared by the field evaluator
// (disk_v2.hip) and the march kernel's analytic disk source (march.hip).  See disk_v2.hip for the
// reference citations.
#pragma once
#include <float.h>

#include "bhr_internal.h"

namespace dv2 {

__device__ __forceinline__ double smoothstep(double e0, double e1, double x) {
    double t = fmin(fmax((x - e0) / (e1 - e0), 0.0), 1.0);
    return t * t * (3.0 - 2.0 * t);
}
__device__ __forceinline__ double half_thickness(double r, const bhr_disk_v2_params &p) {
    double safe_r = fmax(r, p.r_in);
    return p.h0 * safe_r * pow(safe_r / p.r_in, p.beta_h);
}
__device__ __forceinline__ bool radial_mask(double r, const bhr_disk_v2_params &p) { return r >= p.r_in && r <= p.r_out; }
__device__ __forceinline__ double radial_weight(double r, const bhr_disk_v2_params &p) {
    double span = p.r_out - p.r_in;
    double soft = fmax(span * p.edge_softness, DBL_EPSILON);
    double inner = smoothstep(p.r_in, p.r_in + soft, r);
    double outer = 1.0 - smoothstep(p.r_out - soft, p.r_out, r);
    double w = inner * outer;
    return (r <= p.r_in || r >= p.r_out) ? 0.0 : w;
}
__device__ __forceinline__ double vertical_weight(double r, double z, const bhr_disk_v2_params &p) {
    double th = fmax(half_thickness(r, p), DBL_EPSILON);
    double xi = fabs(z) / th;
    double w = 1.0 - smoothstep(0.0, 1.0, xi);
    return radial_mask(r, p) ? w : 0.0;
}
__device__ __forceinline__ bool volume_mask(double r, double z, const bhr_disk_v2_params &p) {
    return radial_mask(r, p) && fabs(z) <= half_thickness(r, p);
}
__device__ __forceinline__ double omega_field(double r, const bhr_disk_v2_params &p) {
    return p.omega_scale * pow(fmax(r, p.r_in) / p.r_in, -1.5);
}
__device__ __forceinline__ double rho_mid(double r, const bhr_disk_v2_params &p) {
    return pow(fmax(r, p.r_in) / p.r_in, -p.rho_power) * radial_weight(r, p);
}
__device__ __forceinline__ double t_mid(double r, const bhr_disk_v2_params &p) {
    double safe_r = fmax(r, p.r_in);
    double inner = fmax(1.0 - sqrt(p.r_in / safe_r), 0.0);
    double t = p.temp_scale * pow(safe_r / p.r_in, -0.75) * pow(inner, 0.25) * radial_weight(r, p);
    return r <= p.r_in ? 0.0 : t;
}
// rho(r, z) = rho_mid exp(-z^2 / 2H^2) W_z inside the volume, 0 outside   (physical_fields.py:119-160)
__device__ __forceinline__ double rho_field(double r, double z, const bhr_disk_v2_params &p) {
    double th = fmax(half_thickness(r, p), DBL_EPSILON), q = z / th;
    double v = rho_mid(r, p) * exp(-0.5 * (q * q)) * vertical_weight(r, z, p);
    return volume_mask(r, z, p) ? v : 0.0;
}
// T(r, z) = T_mid clip(1 - |z| / 4H, 0, 1) W_z inside the volume, 0 outside   (physical_fields.py:163-205)
__device__ __forceinline__ double t_field(double r, double z, const bhr_disk_v2_params &p) {
    double th = fmax(half_thickness(r, p), DBL_EPSILON);
    double vf = fmin(fmax(1.0 - 0.25 * fabs(z) / th, 0.0), 1.0);
    double v = t_mid(r, p) * vf * vertical_weight(r, z, p);
    return volume_mask(r, z, p) ? v : 0.0;
}
__device__ __forceinline__ double log_radius(double r, const bhr_disk_v2_params &p) { return log(fmax(r, p.r_in) / p.r_in); }
__device__ __forceinline__ double wrapped_dphi(double phi, double c) { return atan2(sin(phi - c), cos(phi - c)); }

__device__ __forceinline__ double raw_shear(double r, double phi, const bhr_disk_v2_params &p) {
    double lr = log_radius(r, p), s = 0.0, amp = 1.0;
    for (int k = 0; k < p.shear_components; ++k) {
        double pf = (double)p.shear_phi_freq[k], lf = (double)p.shear_logr_freq[k], ph = p.shear_phase[k];
        s += amp * cos(pf * phi + lf * lr + ph);
        s += 0.6 * amp * sin((pf + 1.0) * phi - (lf + 0.5) * lr + 0.7 * ph);
        amp *= 0.5;   // 0.5 ** component_idx
    }
    return s;
}
__device__ __forceinline__ double raw_hotspot(double r, double phi, const bhr_disk_v2_params &p) {
    double lr = log_radius(r, p), s = 0.0;
    const double halo_phi = 1.8, halo_logr = 1.8, halo_w = 0.6;
    for (int k = 0; k < p.hotspot_count; ++k) {
        double dphi = wrapped_dphi(phi, p.hotspot_phase[k]);
        double dl = (lr - p.hotspot_log_r[k]) / p.hotspot_logr_sigma;
        double a = dphi / p.hotspot_phi_sigma;
        double core = exp(-0.5 * (a * a) - 0.5 * (dl * dl));
        double b = dphi / (halo_phi * p.hotspot_phi_sigma);
        double c = (lr - p.hotspot_log_r[k]) / (halo_logr * p.hotspot_logr_sigma);
        double halo = exp(-0.5 * (b * b) - 0.5 * (c * c));
        s += p.hotspot_weight[k] * (core - halo_w * halo);
    }
    return s;
}
__device__ __forceinline__ double mode_factor(double r, double phi, const bhr_disk_v2_params &p) {
    double lr = log_radius(r, p);
    double raw = p.mode1_strength * cos(phi + 0.35 * lr) + p.mode2_strength * cos(2.0 * phi - 0.65 * lr);
    return radial_weight(r, p) > 0.0 ? 1.0 + raw : 1.0;
}


// structure_modulation with fixed normalisation constants (per-ray use): F_mode * F_shear * F_hotspot
__device__ __forceinline__ double structure_total(double r, double phi, const bhr_disk_v2_params &p, double norm_shear,
                                                  double norm_hotspot) {
    if (!(radial_weight(r, p) > 0.0)) return 1.0;
    double sh = 1.0 + p.shear_strength * (norm_shear <= DBL_EPSILON ? 0.0 : raw_shear(r, phi, p) / norm_shear);
    double hs = 1.0 + p.hotspot_strength * (norm_hotspot <= DBL_EPSILON ? 0.0 : raw_hotspot(r, phi, p) / norm_hotspot);
    return mode_factor(r, phi, p) * sh * hs;
}

}  // namespace dv2
