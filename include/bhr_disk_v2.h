/*
 * bhr_disk_v2.h -- C ABI for the reference's analytic "Disk V2" model on the device (binary64).
 *
 * Replaces the NumPy package disk_v2/ of the reference (geometry.py, physical_fields.py,
 * structure_modulations.py, params.py).  The Python binding black-hole-renderer_amd/disk_v2.py
 * exposes the same function names and argument meaning.
 */
#ifndef BHR_DISK_V2_H
#define BHR_DISK_V2_H

#include "bhr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BHR_DV2_MAX_TERMS 32

/* DiskV2Params (params.py:12-68) + DiskV2StructureParams (params.py:70-144) + the random tables the
 * reference draws inside shear_modulation (structure_modulations.py:170-176, default_rng(seed)) and
 * hotspot_modulation (247-250, default_rng(seed)); the host binding draws them in that order. */
typedef struct {
    double r_in, r_out, h0, beta_h, rho_power, temp_scale, omega_scale, edge_softness;
    double mode1_strength, mode2_strength, shear_strength, hotspot_strength;
    double hotspot_phi_sigma, hotspot_logr_sigma, hotspot_inner_bias;
    int32_t shear_components, hotspot_count;
    int32_t shear_phi_freq[BHR_DV2_MAX_TERMS], shear_logr_freq[BHR_DV2_MAX_TERMS];
    double shear_phase[BHR_DV2_MAX_TERMS];
    double hotspot_phase[BHR_DV2_MAX_TERMS], hotspot_log_r[BHR_DV2_MAX_TERMS], hotspot_weight[BHR_DV2_MAX_TERMS];
} bhr_disk_v2_params;

typedef enum {
    BHR_DV2_H = 0,         /* disk_half_thickness        geometry.py:50-77   */
    BHR_DV2_MASK_R = 1,    /* disk_radial_mask           geometry.py:80-113  (1.0 / 0.0) */
    BHR_DV2_W_R = 2,       /* disk_radial_weight         geometry.py:116-185 */
    BHR_DV2_W_Z = 3,       /* disk_vertical_weight       geometry.py:188-235 */
    BHR_DV2_MASK_VOL = 4,  /* disk_volume_mask           geometry.py:238-280 (1.0 / 0.0) */
    BHR_DV2_OMEGA = 5,     /* angular_velocity_field     physical_fields.py:21-49  */
    BHR_DV2_RHO_MID = 6,   /* midplane_density_field     physical_fields.py:52-79  */
    BHR_DV2_T_MID = 7,     /* midplane_temperature_field physical_fields.py:82-116 */
    BHR_DV2_RHO = 8,       /* density_field              physical_fields.py:119-160 */
    BHR_DV2_T = 9,         /* temperature_field          physical_fields.py:163-205 */
    BHR_DV2_F_MODE = 10,   /* weak_mode_modulation       structure_modulations.py:95-142 */
    BHR_DV2_F_SHEAR = 11,  /* shear_modulation           structure_modulations.py:145-207 */
    BHR_DV2_F_HOTSPOT = 12,/* hotspot_modulation         structure_modulations.py:210-289 */
    BHR_DV2_F_TOTAL = 13   /* structure_modulation       structure_modulations.py:292-334; the hotspot
                              tables must hold the draws of seed + 1 */
} bhr_disk_v2_field;

/* Evaluates one field at n points (already broadcast by the caller).  z / phi may be NULL for fields
 * that do not use them.  For F_SHEAR / F_HOTSPOT / F_TOTAL the signed sums are normalised by their
 * maximum |.| over THESE n points, as the reference does; norm_shear / norm_hotspot > 0 replace that
 * maximum by a fixed constant (per-ray use).  max_out (may be NULL) receives the two maxima found. */
BHR_API int32_t bhr_disk_v2_eval(bhr_ctx *ctx, const bhr_disk_v2_params *params, int32_t field, const double *r,
                                 const double *z, const double *phi, int64_t n, double norm_shear, double norm_hotspot,
                                 double *out, double *max_out);

/* Disk source of the march kernel: BHR_DISK_TEXTURE (default) samples the disk texture / mip stack;
 * BHR_DISK_V2 evaluates the Disk V2 model per hit in binary64 (temperature -> black-body colour,
 * density -> opacity; csrc/march.hip: disk_v2_rgba).  norm_shear / norm_hotspot are the maxima of the raw
 * structure sums on a reference grid (bhr_disk_v2_eval(..., max_out)), t_peak the maximum of T_mid(r). */
#define BHR_DISK_TEXTURE 0
#define BHR_DISK_V2 1
/* BHR_DISK_V2_VOLUME: the finite-thickness disk of docs/design_ad_v2.md 4.2-4.3 (Phase 2 advection + Phase 3
 * emission-absorption integral; specified in the reference's design, not implemented there).  Every RK4 step
 * that passes through the volume |zeta| <= H(r), r_in <= r <= r_out of the tilted disk frame is cut into
 * `substeps` pieces; a piece at density rho, with direction cosine mu to the disk normal, has opacity
 * 1 - exp(-absorption * rho * (1 + grazing_gain * (1 - mu)) * ds) and the black-body colour of T(r, zeta) F
 * under the g-factor; pieces composite front to back like surface crossings (csrc/march.hip:
 * volume_segment); a ray stops sampling once its accumulated opacity reaches 0.9999.
 * Options are set with bhr_set_disk_volume_options before bhr_set_disk_source. */
#define BHR_DISK_V2_VOLUME 2
BHR_API int32_t bhr_set_disk_source(bhr_ctx *ctx, int32_t source, const bhr_disk_v2_params *params, double norm_shear,
                                    double norm_hotspot, double t_peak);

/* Defaults: absorption 4.0 per unit length at rho = 1, grazing_gain 1.0, substeps 2 (1..16). */
BHR_API int32_t bhr_set_disk_volume_options(bhr_ctx *ctx, double absorption, double grazing_gain, int32_t substeps);

#ifdef __cplusplus
}
#endif
#endif
