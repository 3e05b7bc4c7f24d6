/*
 * bhr_lifecycle.h -- C ABI for the entity layer and the compose statistics on the device.
 *
 * Replaces the NumPy halves of TaichiRenderer.accumulate_entity_layer (render.py:3564-3653) and
 * TaichiRenderer.recompute_interactive_stats (render.py:3655-3712): the per-texel rasterisation of
 * ~245 entities and the percentile statistics no longer run on host cores (152 ms + 45 ms per fhd
 * frame measured) nor cross PCIe (6 planes up, 13 planes down per call).
 */
#ifndef BHR_LIFECYCLE_H
#define BHR_LIFECYCLE_H

#include "bhr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One (filament, row) pair, render.py:3629-3638.  The per-row scalars are evaluated by the host
 * binding with the reference's expressions (they depend on NumPy's scalar promotion rules); the
 * device evaluates the azimuthal Gaussian in binary64 and accumulates into f32 like
 * `staging[row] += profile * (scale * r_w)`. */
typedef struct {
    double center;        /* (source_phi - omega[row] * age) % 2pi          */
    double inv_2s_phi;    /* 0.5 / sigma_phi_t^2                            */
    double coef_d;        /* scale_d * r_w                                  */
    double coef_t;        /* scale_t * r_w                                  */
} bhr_filament_row;

/* One (hotspot | RT spike, row) pair, render.py:3643-3649: a pre-rasterised row of the entity's
 * profile (uploaded once with bhr_entity_profile_upload) rolled by `shift` texels and scaled. */
typedef struct {
    int64_t offset;       /* first texel of this row's density profile inside the pool */
    int32_t shift;        /* np.roll(row, -shift)                            */
    int32_t plane;        /* 2 = RT spikes, 4 = hotspots (density plane; temperature = plane + 1) */
    float alpha;          /* fade factor                                     */
    int32_t pool_stride_; /* n_rows * n_phi of the entity: temperature row = offset + pool_stride_ */
} bhr_rolled_row;

/* Stores the (n_rows, n_phi) density and temperature profiles of one entity on the device and
 * returns the pool offset of its first density texel in *offset_out; temp rows live at
 * offset + n_rows * n_phi.  Profiles are immutable after spawn (render.py:1778-1793, 1848-1866). */
BHR_API int32_t bhr_entity_profile_upload(bhr_ctx *ctx, const float *density, const float *temp, int32_t n_rows,
                                          int64_t *offset_out);
/* Releases every profile (the pool is a bump allocator; call when the population is rebuilt). */
BHR_API int32_t bhr_entity_profile_reset(bhr_ctx *ctx);

/* Rasterises all pairs into comp[5..10] (zeroed first).  Pairs are grouped by texture row in CSR
 * form: row r owns fil[fil_ptr[r] .. fil_ptr[r+1]) and rolled[rol_ptr[r] .. rol_ptr[r+1]), each
 * group in the order the reference visits the entities (that order fixes the f32 rounding).
 * phi[n_phi] = linspace(0, 2pi, n_phi, endpoint=False) in binary64. */
BHR_API int32_t bhr_accumulate_entities(bhr_ctx *ctx, const bhr_filament_row *fil, const int32_t *fil_ptr,
                                        const bhr_rolled_row *rolled, const int32_t *rol_ptr, const double *phi);

/* The same rasterisation from ENTITIES instead of pair tables: what the producer (lifecycle.EntityFactory, the
 * reference's EntityFactory render.py:624-792) holds per structure, as Python floats.  The library evaluates
 * filament_strength / the fade trapezoid and the per-(entity, row) scalars itself -- the reference's expressions with
 * the reference's roundings (binary64 libm exp like math.exp; binary32 centre and roll arithmetic like NumPy's f32
 * scalars, np.remainder semantics) -- groups the pairs by row and launches without waiting for the stream.
 * Entities are visited in array order (filaments; then `rolled` in the order given: RT spikes before hotspots,
 * render.py:3640-3649).  Rows of an entity are contiguous: [row_lo, row_lo + n_rows). */
typedef struct {
    double birth_time, source_phi, sigma_phi0, shear_rate, peak_density, peak_temp, cooling_time;
    int32_t row_lo, n_rows;
} bhr_filament_entity;
typedef struct {
    double birth_time, lifetime, ramp_in, ramp_out;   /* fade trapezoid: ramp in, plateau, ramp out */
    int64_t offset;                                     /* bhr_entity_profile_upload's offset of its first density row */
    int32_t row_lo, n_rows;
    int32_t plane, pad_;                                /* 2 = RT spike, 4 = hotspot */
} bhr_rolled_entity;
/* radial_weights: for every filament in turn its n_rows binary64 radial weights exp(-(r_norm[row] - base_r)^2 / (2
 * sigma_r^2)) (static over its life; computed by the producer).  omega_rows: (n_r) f32 Keplerian angular velocities. */
BHR_API int32_t bhr_accumulate_population(bhr_ctx *ctx, double now, const bhr_filament_entity *filaments, int32_t n_filaments,
                                          const double *radial_weights, const bhr_rolled_entity *rolled, int32_t n_rolled,
                                          const float *omega_rows);

/* recompute_interactive_stats on the device, in three steps so that the binding can apply NumPy's
 * own index and interpolation arithmetic (which depends on the NumPy version: 2.x evaluates the
 * virtual index (n - 1) q and the lerp in the array's dtype, f32):
 *  1. bhr_stats_prepare builds the edge-weighted density and the structural temperature from comp
 *     (render.py:3677-3688) and returns how many structural temperatures are positive;
 *  2. bhr_stats_select returns the k-th smallest (0-based) value of the density (which = 0) or of the
 *     positive structural temperatures (which = 1), exactly, by 11+11+10-bit radix selection;
 *  3. bhr_stats_row_statistics sorts every row of clip(temp_struct / div * 0.8, 0, 1.2) in LDS and
 *     returns (n_r, 4) = {row max, sorted[lo], sorted[hi], max of the temp_base row}
 *     (render.py:3694-3706). */
BHR_API int32_t bhr_stats_prepare(bhr_ctx *ctx, int32_t enable_rt, uint64_t *n_positive_out);
BHR_API int32_t bhr_stats_select(bhr_ctx *ctx, int32_t which, uint64_t rank, float *value_out);
BHR_API int32_t bhr_stats_row_statistics(bhr_ctx *ctx, float div, int32_t lo, int32_t hi, float *rows_out);

#ifdef __cplusplus
}
#endif
#endif
