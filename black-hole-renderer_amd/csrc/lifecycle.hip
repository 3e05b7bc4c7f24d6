// lifecycle.hip -- entity layer rasterisation and compose statistics on the device
// (include/bhr_lifecycle.h; replaces the NumPy halves of render.py:3564-3653 and 3655-3712).
// Compiled without FMA contraction: the f32 expressions must round exactly like NumPy's.
#include <math.h>
#include <string.h>

#include <vector>

#include "bhr_internal.h"

namespace {

// ---- accumulate_entity_layer: one thread per texel, pairs of its row in reference order ----------
__global__ __launch_bounds__(256) void entity_kernel(float *__restrict__ comp, int n_r, int n_phi,
                                                     const bhr_filament_row *__restrict__ fil,
                                                     const int *__restrict__ fil_ptr,
                                                     const bhr_rolled_row *__restrict__ rol,
                                                     const int *__restrict__ rol_ptr, const float *__restrict__ pool,
                                                     const double *__restrict__ phi) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int ri = blockIdx.y;
    if (j >= n_phi) return;
    const double two_pi = 2.0 * 3.141592653589793;
    float acc[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const double ph = phi[j];
    // filaments (render.py:3633-3638): d_phi wrapped with round-half-even, Gaussian in binary64,
    // `staging[row] += profile * (scale * r_w)` = f32(f64(acc) + contribution)
    for (int k = fil_ptr[ri]; k < fil_ptr[ri + 1]; ++k) {
        const bhr_filament_row f = fil[k];
        double d = ph - f.center;
        d = d - two_pi * rint(d / two_pi);
        double prof = exp(-d * d * f.inv_2s_phi);
        acc[0] = (float)((double)acc[0] + prof * f.coef_d);
        acc[1] = (float)((double)acc[1] + prof * f.coef_t);
    }
    // hotspots / RT spikes (render.py:3645-3649): np.roll(row, -shift) * alpha in f32, f32 accumulate
    for (int k = rol_ptr[ri]; k < rol_ptr[ri + 1]; ++k) {
        const bhr_rolled_row r = rol[k];
        int src = (j + r.shift) % n_phi;
        if (src < 0) src += n_phi;
        const float dv = pool[r.offset + src];
        const float tv = pool[r.offset + r.pool_stride_ + src];
        acc[r.plane] = acc[r.plane] + dv * r.alpha;
        acc[r.plane + 1] = acc[r.plane + 1] + tv * r.alpha;
    }
    const size_t plane = (size_t)n_r * n_phi, q = (size_t)ri * n_phi + j;
#pragma unroll
    for (int c = 0; c < 6; ++c) comp[(5 + c) * plane + q] = acc[c];
}

// ---- recompute_interactive_stats ------------------------------------------------------------------
// density = (0.15 + 0.10 sp + 0.30 turb + 0.20 hs + 0.30 arc + rt_w rt) * dm * edge     (render.py:3677-3679)
// temp_struct = (sp_t + turb_t + arc_t + rt_t + hs_t) * dm                                   (render.py:3688)
__global__ __launch_bounds__(256) void stats_fields_kernel(const float *__restrict__ comp,
                                                           const float *__restrict__ edge, int n_r, int n_phi,
                                                           float rt_w, float *__restrict__ density,
                                                           float *__restrict__ temp_struct) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int ri = blockIdx.y;
    if (j >= n_phi) return;
    const size_t plane = (size_t)n_r * n_phi, q = (size_t)ri * n_phi + j;
    const float sp = comp[1 * plane + q], sp_t = comp[2 * plane + q], turb = comp[3 * plane + q];
    const float turb_t = comp[4 * plane + q], arc = comp[5 * plane + q], arc_t = comp[6 * plane + q];
    const float rt = comp[7 * plane + q], rt_t = comp[8 * plane + q], hs = comp[9 * plane + q];
    const float hs_t = comp[10 * plane + q], dm = comp[12 * plane + q];
    float d = (0.15f + 0.10f * sp + 0.30f * turb + 0.20f * hs + 0.30f * arc + rt_w * rt) * dm;
    density[q] = d * edge[ri];
    temp_struct[q] = (sp_t + turb_t + arc_t + rt_t + hs_t) * dm;
}

__device__ __forceinline__ unsigned int order_key(float v) {   // monotone map f32 -> u32
    unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// histogram of `nbits` key bits at `shift` over the elements whose higher bits equal `prefix`
__global__ __launch_bounds__(256) void select_hist_kernel(const float *__restrict__ v, long long n,
                                                          unsigned int prefix, int prefix_bits, int shift, int nbits,
                                                          int only_positive, unsigned int *__restrict__ hist) {
    __shared__ unsigned int lh[2048];
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    const unsigned int mask = (1u << nbits) - 1u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = v[i];
        if (only_positive && !(x > 0.0f)) continue;
        const unsigned int key = order_key(x);
        if (prefix_bits > 0 && (key >> (32 - prefix_bits)) != prefix) continue;
        atomicAdd(&lh[(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += blockDim.x)
        if (lh[k]) atomicAdd(&hist[k], lh[k]);
}

// One block per texture row: scaled = clip(ts / div * 0.8, 0, 1.2) sorted in LDS (bitonic), then
// {max, sorted[lo], sorted[hi], max(temp_base row)}  (render.py:3694-3706)
__global__ __launch_bounds__(1024) void row_stats_kernel(const float *__restrict__ temp_struct,
                                                         const float *__restrict__ temp_base, int n_phi, int n_pad,
                                                         float div, int lo, int hi, float *__restrict__ out) {
    extern __shared__ float s[];
    const int ri = blockIdx.x;
    const float *row = temp_struct + (size_t)ri * n_phi;
    const float *tb = temp_base + (size_t)ri * n_phi;
    float tbm = -INFINITY;
    for (int k = threadIdx.x; k < n_pad; k += blockDim.x) {
        float x = INFINITY;
        if (k < n_phi) {
            x = fminf(fmaxf(row[k] / div * 0.8f, 0.0f), 1.2f);
            tbm = fmaxf(tbm, tb[k]);
        }
        s[k] = x;
    }
    __syncthreads();
    for (int size = 2; size <= n_pad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = threadIdx.x; k < (n_pad >> 1); k += blockDim.x) {
                int i0 = 2 * k - (k & (stride - 1));   // lower index of the pair
                int i1 = i0 + stride;
                bool up = (i0 & size) == 0;
                float a = s[i0], b = s[i1];
                if ((a > b) == up) { s[i0] = b; s[i1] = a; }
            }
            __syncthreads();
        }
    }
    // block max of temp_base
    __shared__ float red[32];
    for (int off = 32; off > 0; off >>= 1) tbm = fmaxf(tbm, __shfl_down(tbm, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tbm;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        out[ri * 4 + 0] = s[n_phi - 1];
        out[ri * 4 + 1] = s[lo];
        out[ri * 4 + 2] = s[hi];
        out[ri * 4 + 3] = m;
    }
}

struct Scratch {
    float *density = nullptr, *temp_struct = nullptr;
    unsigned int *hist = nullptr;
};

int32_t ensure_scratch(bhr_ctx *ctx) {
    const size_t plane = (size_t)ctx->bg_n_r * ctx->bg_n_phi;
    if (ctx->d_stats_scratch && ctx->stats_scratch_elems >= plane) return BHR_OK;
    if (ctx->d_stats_scratch) (void)hipFree(ctx->d_stats_scratch);
    ctx->d_stats_scratch = nullptr;
    BHR_HIP(hipMalloc((void **)&ctx->d_stats_scratch, (2 * plane + 2048 + (size_t)ctx->bg_n_r * 4) * sizeof(float)));
    ctx->stats_scratch_elems = plane;
    return BHR_OK;
}

// k-th smallest (0-based) of the selected elements by 11 + 11 + 10 bit radix passes
int32_t radix_select(bhr_ctx *ctx, const float *d_v, long long n, unsigned long long k, int only_positive,
                     unsigned int *d_hist, float *value_out, unsigned long long *count_out) {
    unsigned int prefix = 0;
    int prefix_bits = 0;
    const int bits[3] = {11, 11, 10};
    std::vector<unsigned int> h(2048);
    for (int pass = 0; pass < 3; ++pass) {
        const int nb = bits[pass], shift = 32 - prefix_bits - nb;
        BHR_HIP(hipMemsetAsync(d_hist, 0, 2048 * sizeof(unsigned int), ctx->stream));
        hipLaunchKernelGGL(select_hist_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_v, n, prefix, prefix_bits, shift, nb,
                           only_positive, d_hist);
        BHR_HIP(hipGetLastError());
        BHR_HIP(hipMemcpyAsync(h.data(), d_hist, 2048 * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        unsigned long long total = 0;
        for (int b = 0; b < (1 << nb); ++b) total += h[b];
        if (pass == 0 && count_out) *count_out = total;
        if (total == 0) return bhr_fail(BHR_ERR_STATE, "radix_select: empty selection");
        if (k >= total) k = total - 1;
        unsigned long long cum = 0;
        int b = 0;
        for (; b < (1 << nb); ++b) {
            if (cum + h[b] > k) break;
            cum += h[b];
        }
        k -= cum;
        prefix = (prefix << nb) | (unsigned int)b;
        prefix_bits += nb;
    }
    unsigned int u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;   // invert order_key
    float f;
    memcpy(&f, &u, 4);
    *value_out = f;
    return BHR_OK;
}

}  // namespace

extern "C" {

int32_t bhr_entity_profile_reset(bhr_ctx *ctx) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "null ctx");
    BHR_TRY(bhr_enter_components(ctx));
    BHR_HIP(hipStreamSynchronize(ctx->stream));     // an entity pass in flight may still read the profiles about to be overwritten
    ctx->pool_used = 0;
    return BHR_OK;
}

int32_t bhr_entity_profile_upload(bhr_ctx *ctx, const float *density, const float *temp, int32_t n_rows,
                                  int64_t *offset_out) {
    if (!ctx || !density || !temp || n_rows <= 0 || !offset_out) return bhr_fail(BHR_ERR_INVALID, "bhr_entity_profile_upload: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call init_background_layer() first");
    BHR_TRY(bhr_enter_components(ctx));
    const int64_t need = 2ll * n_rows * ctx->bg_n_phi;
    if (ctx->pool_used + need > ctx->pool_cap) {   // grow geometrically, keep the contents
        int64_t cap = ctx->pool_cap ? ctx->pool_cap : (int64_t)64 * ctx->bg_n_phi * 64;
        while (cap < ctx->pool_used + need) cap *= 2;
        float *p = nullptr;
        BHR_HIP(hipMalloc((void **)&p, (size_t)cap * sizeof(float)));
        if (ctx->d_pool) {
            BHR_HIP(hipStreamSynchronize(ctx->stream));
            BHR_HIP(hipMemcpy(p, ctx->d_pool, (size_t)ctx->pool_used * sizeof(float), hipMemcpyDeviceToDevice));
            (void)hipFree(ctx->d_pool);
        }
        ctx->d_pool = p;
        ctx->pool_cap = cap;
    }
    // Blocking copies into a region no launch has been given yet (the pool only grows between resets, and a reset waits
    // for the stream): they need not queue behind the background pass on the scene stream, and the caller's arrays
    // are free on return.
    const size_t half = (size_t)n_rows * ctx->bg_n_phi;
    BHR_HIP(hipMemcpy(ctx->d_pool + ctx->pool_used, density, half * sizeof(float), hipMemcpyHostToDevice));
    BHR_HIP(hipMemcpy(ctx->d_pool + ctx->pool_used + half, temp, half * sizeof(float), hipMemcpyHostToDevice));
    *offset_out = ctx->pool_used;
    ctx->pool_used += need;
    return BHR_OK;
}

int32_t bhr_accumulate_entities(bhr_ctx *ctx, const bhr_filament_row *fil, const int32_t *fil_ptr,
                                const bhr_rolled_row *rolled, const int32_t *rol_ptr, const double *phi) {
    if (!ctx || !fil_ptr || !rol_ptr || !phi) return bhr_fail(BHR_ERR_INVALID, "bhr_accumulate_entities: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call init_background_layer() first");
    BHR_TRY(bhr_enter_components(ctx));
    const int n_r = ctx->bg_n_r, n_phi = ctx->bg_n_phi;
    const int n_fil = fil_ptr[n_r], n_rol = rol_ptr[n_r];
    if ((n_fil > 0 && !fil) || (n_rol > 0 && !rolled)) return bhr_fail(BHR_ERR_INVALID, "bhr_accumulate_entities: missing pair table");
    if (n_rol > 0 && !ctx->d_pool) return bhr_fail(BHR_ERR_STATE, "bhr_accumulate_entities: no profiles uploaded");
    // one staging allocation: [fil | rol | fil_ptr | rol_ptr | phi]
    const size_t b_fil = (size_t)n_fil * sizeof(bhr_filament_row), b_rol = (size_t)n_rol * sizeof(bhr_rolled_row);
    const size_t b_ptr = (size_t)(n_r + 1) * sizeof(int32_t), b_phi = (size_t)n_phi * sizeof(double);
    auto up8 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t total = up8(b_fil) + up8(b_rol) + 2 * up8(b_ptr) + up8(b_phi);
    if (ctx->pairs_cap < total) {
        if (ctx->d_pairs) { BHR_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_pairs); }
        ctx->d_pairs = nullptr;
        BHR_HIP(hipMalloc((void **)&ctx->d_pairs, total * 2));
        ctx->pairs_cap = total * 2;
    }
    char *base = (char *)ctx->d_pairs;
    char *d_fil = base, *d_rol = d_fil + up8(b_fil), *d_fp = d_rol + up8(b_rol), *d_rp = d_fp + up8(b_ptr), *d_phi = d_rp + up8(b_ptr);
    if (b_fil) BHR_HIP(hipMemcpyAsync(d_fil, fil, b_fil, hipMemcpyHostToDevice, ctx->stream));
    if (b_rol) BHR_HIP(hipMemcpyAsync(d_rol, rolled, b_rol, hipMemcpyHostToDevice, ctx->stream));
    BHR_HIP(hipMemcpyAsync(d_fp, fil_ptr, b_ptr, hipMemcpyHostToDevice, ctx->stream));
    BHR_HIP(hipMemcpyAsync(d_rp, rol_ptr, b_ptr, hipMemcpyHostToDevice, ctx->stream));
    BHR_HIP(hipMemcpyAsync(d_phi, phi, b_phi, hipMemcpyHostToDevice, ctx->stream));
    dim3 grid((n_phi + 255) / 256, n_r), block(256);
    hipLaunchKernelGGL(entity_kernel, grid, block, 0, ctx->stream, ctx->d_comp, n_r, n_phi, (const bhr_filament_row *)d_fil,
                       (const int *)d_fp, (const bhr_rolled_row *)d_rol, (const int *)d_rp, ctx->d_pool, (const double *)d_phi);
    BHR_HIP(hipGetLastError());
    BHR_HIP(hipStreamSynchronize(ctx->stream));   // the host tables may be reused by the caller
    return BHR_OK;
}

}  // extern "C"

// ---- the same rasterisation fed with ENTITIES: the (entity, row) tables are built here, on the host side of the
// library, from the records the producer keeps -- the per-frame Python work of the binding (1.4 ms at fhd) becomes
// ~30 us of C++, and nothing waits for the stream: tables go through a double-buffered pinned staging area.
namespace {

struct PopStage {
    char *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool busy = false;
};
struct PopHost {
    PopStage stage[2];
    int next = 0;
    double *d_phi = nullptr;         // linspace(0, 2 pi, n_phi, endpoint=False), binary64
    int32_t phi_n = 0;
    std::vector<int32_t> fil_cnt, rol_cnt;
    std::vector<bhr_filament_row> fil;
    std::vector<bhr_rolled_row> rol;
};

// np.remainder for f32 operands (numpy/_core/src/npymath: npy_remainderf), divisor > 0
inline float numpy_remainder_f32(float a, float b) {
    float mod = fmodf(a, b);
    if (mod != 0.0f) {
        if (mod < 0.0f) mod += b;
    } else {
        mod = 0.0f;
    }
    return mod;
}

// filament_strength / envelope of the producer (black-hole-renderer_amd/lifecycle.py; render.py:504-560, 3606-3649)
inline double cooling(const bhr_filament_entity &e, double age) { return e.cooling_time > 0 ? exp(-age / e.cooling_time) : 1.0; }

inline double trapezoid(const bhr_rolled_entity &e, double now) {
    double t = now - e.birth_time;
    if (t < 0) return 0.0;
    if (t < e.ramp_in) return e.ramp_in > 0 ? t / e.ramp_in : 1.0;
    t -= e.ramp_in;
    if (t < e.lifetime) return 1.0;
    t -= e.lifetime;
    if (t < e.ramp_out) return e.ramp_out > 0 ? 1.0 - t / e.ramp_out : 0.0;
    return 0.0;
}

}  // namespace

void bhr_population_free(bhr_ctx *ctx) {
    PopHost *p = (PopHost *)ctx->pop_host;
    if (!p) return;
    for (auto &st : p->stage) {
        if (st.ev) (void)hipEventDestroy(st.ev);
        if (st.host) (void)hipHostFree(st.host);
    }
    if (p->d_phi) (void)hipFree(p->d_phi);
    delete p;
    ctx->pop_host = nullptr;
}

extern "C" {

int32_t bhr_accumulate_population(bhr_ctx *ctx, double now, const bhr_filament_entity *fil, int32_t n_fil,
                                  const double *radial_weights, const bhr_rolled_entity *rolled, int32_t n_rolled,
                                  const float *omega_rows) {
    if (!ctx || !omega_rows || n_fil < 0 || n_rolled < 0 || (n_fil > 0 && (!fil || !radial_weights)) || (n_rolled > 0 && !rolled))
        return bhr_fail(BHR_ERR_INVALID, "bhr_accumulate_population: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call init_background_layer() first");
    BHR_TRY(bhr_enter_components(ctx));
    const int n_r = ctx->bg_n_r, n_phi = ctx->bg_n_phi;
    if (!ctx->pop_host) ctx->pop_host = new PopHost();
    PopHost *P = (PopHost *)ctx->pop_host;
    const double two_pi = 2.0 * 3.141592653589793;
    if (P->phi_n != n_phi) {
        std::vector<double> phi((size_t)n_phi);
        const double step = two_pi / (double)n_phi;                 // np.linspace(0, 2 pi, n_phi, endpoint=False)
        for (int j = 0; j < n_phi; ++j) phi[(size_t)j] = (double)j * step;
        if (P->d_phi) { BHR_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(P->d_phi); P->d_phi = nullptr; }
        BHR_HIP(hipMalloc((void **)&P->d_phi, sizeof(double) * (size_t)n_phi));
        BHR_HIP(hipMemcpyAsync(P->d_phi, phi.data(), sizeof(double) * (size_t)n_phi, hipMemcpyHostToDevice, ctx->stream));
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        P->phi_n = n_phi;
    }
    // pass 1: who contributes, with which per-entity scalars; rows per texture row
    struct FilScalars { double inv2s, sc_d, sc_t; float src32, age32; int64_t rw_at; };
    std::vector<FilScalars> fs((size_t)n_fil);
    std::vector<char> fil_on((size_t)n_fil, 0), rol_on((size_t)n_rolled, 0);
    std::vector<float> rol_alpha((size_t)n_rolled), rol_age((size_t)n_rolled);
    P->fil_cnt.assign((size_t)n_r + 1, 0);
    P->rol_cnt.assign((size_t)n_r + 1, 0);
    int64_t rw_at = 0;
    for (int k = 0; k < n_fil; ++k) {
        const bhr_filament_entity &e = fil[k];
        if (e.n_rows < 0 || e.row_lo < 0 || (int64_t)e.row_lo + e.n_rows > n_r)
            return bhr_fail(BHR_ERR_INVALID, "bhr_accumulate_population: filament %d covers rows %d..%d of %d", k, e.row_lo, e.row_lo + e.n_rows, n_r);
        fs[(size_t)k].rw_at = rw_at;
        rw_at += e.n_rows;
        const double age = now - e.birth_time;
        const double s0 = e.sigma_phi0 > 1e-6 ? e.sigma_phi0 : 1e-6;
        const double sigma_phi = s0 + e.shear_rate * age;
        const double cool = cooling(e, age);
        if (s0 / sigma_phi * cool < 0.008 || e.n_rows == 0) continue;      // FILAMENT_DEATH_THRESHOLD (render.py:495)
        const double amp_d = e.peak_density * s0 / sigma_phi, amp_t = e.peak_temp * s0 / sigma_phi;
        const double ramp = age / 5.0;                                      // FILAMENT_BIRTH_FADE_DUR (render.py:497)
        const double born = ramp < 1.0 ? ramp : 1.0;
        fs[(size_t)k].inv2s = 0.5 / (sigma_phi * sigma_phi);
        fs[(size_t)k].sc_d = amp_d * born * cool;
        fs[(size_t)k].sc_t = amp_t * born * cool;
        fs[(size_t)k].src32 = (float)e.source_phi;
        fs[(size_t)k].age32 = (float)age;
        fil_on[(size_t)k] = 1;
        for (int r = e.row_lo; r < e.row_lo + e.n_rows; ++r) P->fil_cnt[(size_t)r + 1] += 1;
    }
    for (int k = 0; k < n_rolled; ++k) {
        const bhr_rolled_entity &e = rolled[k];
        if (e.n_rows < 0 || e.row_lo < 0 || (int64_t)e.row_lo + e.n_rows > n_r || (e.plane != 2 && e.plane != 4))
            return bhr_fail(BHR_ERR_INVALID, "bhr_accumulate_population: rolled entity %d (rows %d..%d of %d, plane %d)", k, e.row_lo,
                            e.row_lo + e.n_rows, n_r, e.plane);
        const double alpha = trapezoid(e, now);
        if (alpha <= 0 || e.n_rows == 0) continue;
        if (!ctx->d_pool || e.offset < 0 || e.offset + 2ll * e.n_rows * n_phi > ctx->pool_used)
            return bhr_fail(BHR_ERR_STATE, "bhr_accumulate_population: rolled entity %d has no uploaded profile", k);
        rol_on[(size_t)k] = 1;
        rol_alpha[(size_t)k] = (float)alpha;
        rol_age[(size_t)k] = (float)(now - e.birth_time);
        for (int r = e.row_lo; r < e.row_lo + e.n_rows; ++r) P->rol_cnt[(size_t)r + 1] += 1;
    }
    for (int r = 0; r < n_r; ++r) {
        P->fil_cnt[(size_t)r + 1] += P->fil_cnt[(size_t)r];
        P->rol_cnt[(size_t)r + 1] += P->rol_cnt[(size_t)r];
    }
    const int n_fp = P->fil_cnt[(size_t)n_r], n_rp = P->rol_cnt[(size_t)n_r];
    // staging: [fil | rol | fil_ptr | rol_ptr] in pinned memory, two areas in turn
    const size_t b_fil = (size_t)n_fp * sizeof(bhr_filament_row), b_rol = (size_t)n_rp * sizeof(bhr_rolled_row);
    const size_t b_ptr = (size_t)(n_r + 1) * sizeof(int32_t);
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t total = up16(b_fil) + up16(b_rol) + 2 * up16(b_ptr);
    PopStage &st = P->stage[P->next];
    P->next ^= 1;
    if (!st.ev) BHR_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    if (st.busy) { BHR_HIP(hipEventSynchronize(st.ev)); st.busy = false; }   // the launch two frames back has read it
    if (st.cap < total) {
        if (st.host) (void)hipHostFree(st.host);
        st.host = nullptr;
        st.cap = 0;
        BHR_HIP(hipHostMalloc((void **)&st.host, total * 2, hipHostMallocDefault));
        st.cap = total * 2;
    }
    bhr_filament_row *h_fil = (bhr_filament_row *)st.host;
    bhr_rolled_row *h_rol = (bhr_rolled_row *)(st.host + up16(b_fil));
    int32_t *h_fp = (int32_t *)(st.host + up16(b_fil) + up16(b_rol)), *h_rp = (int32_t *)((char *)h_fp + up16(b_ptr));
    memcpy(h_fp, P->fil_cnt.data(), b_ptr);
    memcpy(h_rp, P->rol_cnt.data(), b_ptr);
    // pass 2: fill, entities in the caller's order inside every row (the order fixes the f32 rounding)
    std::vector<int32_t> &fcur = P->fil_cnt, &rcur = P->rol_cnt;             // running insert positions
    const float two_pi_f = (float)two_pi, n_phi_f = (float)n_phi;
    for (int k = 0; k < n_fil; ++k) {
        if (!fil_on[(size_t)k]) continue;
        const bhr_filament_entity &e = fil[k];
        const FilScalars &f = fs[(size_t)k];
        for (int q = 0; q < e.n_rows; ++q) {
            const int r = e.row_lo + q;
            const float turned = omega_rows[r] * f.age32;                    // f32 product, then f32 difference, as NumPy
            const float center = numpy_remainder_f32(f.src32 - turned, two_pi_f);
            const double rw = radial_weights[f.rw_at + q];
            bhr_filament_row &o = h_fil[fcur[(size_t)r]++];
            o.center = (double)center;
            o.inv_2s_phi = f.inv2s;
            o.coef_d = f.sc_d * rw;
            o.coef_t = f.sc_t * rw;
        }
    }
    for (int k = 0; k < n_rolled; ++k) {
        if (!rol_on[(size_t)k]) continue;
        const bhr_rolled_entity &e = rolled[k];
        for (int q = 0; q < e.n_rows; ++q) {
            const int r = e.row_lo + q;
            const float a = rol_age[(size_t)k] * omega_rows[r];
            const float b = a / two_pi_f;
            const float c = b * n_phi_f;
            bhr_rolled_row &o = h_rol[rcur[(size_t)r]++];
            o.offset = e.offset + (int64_t)q * n_phi;
            o.shift = (int32_t)(int64_t)c;                                    // .astype(np.int64): toward zero
            o.plane = e.plane;
            o.alpha = rol_alpha[(size_t)k];
            o.pool_stride_ = e.n_rows * n_phi;
        }
    }
    if (ctx->pairs_cap < total) {
        if (ctx->d_pairs) { BHR_HIP(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_pairs); }
        ctx->d_pairs = nullptr;
        ctx->pairs_cap = 0;
        BHR_HIP(hipMalloc((void **)&ctx->d_pairs, total * 2));
        ctx->pairs_cap = total * 2;
    }
    char *base = (char *)ctx->d_pairs;
    BHR_HIP(hipMemcpyAsync(base, st.host, total, hipMemcpyHostToDevice, ctx->stream));
    char *d_fil = base, *d_rol = d_fil + up16(b_fil), *d_fp = d_rol + up16(b_rol), *d_rp = d_fp + up16(b_ptr);
    dim3 grid((n_phi + 255) / 256, n_r), block(256);
    hipLaunchKernelGGL(entity_kernel, grid, block, 0, ctx->stream, ctx->d_comp, n_r, n_phi, (const bhr_filament_row *)d_fil,
                       (const int *)d_fp, (const bhr_rolled_row *)d_rol, (const int *)d_rp, ctx->d_pool, (const double *)P->d_phi);
    BHR_HIP(hipGetLastError());
    BHR_HIP(hipEventRecord(st.ev, ctx->stream));
    st.busy = true;
    return BHR_OK;
}

int32_t bhr_stats_prepare(bhr_ctx *ctx, int32_t enable_rt, uint64_t *n_positive_out) {
    if (!ctx || !n_positive_out) return bhr_fail(BHR_ERR_INVALID, "bhr_stats_prepare: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call init_background_layer() first");
    BHR_TRY(bhr_enter(ctx));
    int32_t rc = ensure_scratch(ctx);
    if (rc) return rc;
    const int n_r = ctx->bg_n_r, n_phi = ctx->bg_n_phi;
    const long long n = (long long)n_r * n_phi;
    float *d_density = ctx->d_stats_scratch, *d_ts = d_density + n;
    unsigned int *d_hist = (unsigned int *)(d_ts + n);
    dim3 grid((n_phi + 255) / 256, n_r), block(256);
    hipLaunchKernelGGL(stats_fields_kernel, grid, block, 0, ctx->stream, ctx->d_comp, ctx->d_edge, n_r, n_phi,
                       enable_rt ? 0.20f : 0.0f, d_density, d_ts);
    BHR_HIP(hipGetLastError());
    std::vector<unsigned int> h(2048);
    BHR_HIP(hipMemsetAsync(d_hist, 0, 2048 * sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(select_hist_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_ts, n, 0u, 0, 21, 11, 1, d_hist);
    BHR_HIP(hipGetLastError());
    BHR_HIP(hipMemcpyAsync(h.data(), d_hist, 2048 * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long cnt = 0;
    for (unsigned int v : h) cnt += v;
    *n_positive_out = cnt;
    ctx->stats_prepared = 1;
    return BHR_OK;
}

int32_t bhr_stats_select(bhr_ctx *ctx, int32_t which, uint64_t rank, float *value_out) {
    if (!ctx || !value_out || (which != 0 && which != 1)) return bhr_fail(BHR_ERR_INVALID, "bhr_stats_select: bad argument");
    if (!ctx->bg_ready || !ctx->stats_prepared) return bhr_fail(BHR_ERR_STATE, "bhr_stats_select: call bhr_stats_prepare first");
    BHR_TRY(bhr_enter(ctx));
    const long long n = (long long)ctx->bg_n_r * ctx->bg_n_phi;
    float *d_density = ctx->d_stats_scratch, *d_ts = d_density + n;
    unsigned int *d_hist = (unsigned int *)(d_ts + n);
    return radix_select(ctx, which == 0 ? d_density : d_ts, n, rank, which, d_hist, value_out, nullptr);
}

int32_t bhr_stats_row_statistics(bhr_ctx *ctx, float div, int32_t lo, int32_t hi, float *rows_out) {
    if (!ctx || !rows_out) return bhr_fail(BHR_ERR_INVALID, "bhr_stats_row_statistics: bad argument");
    if (!ctx->bg_ready || !ctx->stats_prepared) return bhr_fail(BHR_ERR_STATE, "bhr_stats_row_statistics: call bhr_stats_prepare first");
    BHR_TRY(bhr_enter(ctx));
    const int n_r = ctx->bg_n_r, n_phi = ctx->bg_n_phi;
    if (lo < 0 || hi < lo || hi >= n_phi) return bhr_fail(BHR_ERR_INVALID, "bhr_stats_row_statistics: bad indices %d %d", lo, hi);
    int n_pad = 1;
    while (n_pad < n_phi) n_pad <<= 1;
    if ((size_t)n_pad * sizeof(float) > 150 * 1024) return bhr_fail(BHR_ERR_INVALID, "bhr_stats_row_statistics: n_phi %d too large for the LDS sort", n_phi);
    const long long n = (long long)n_r * n_phi;
    float *d_ts = ctx->d_stats_scratch + n;
    float *d_rows = d_ts + n + 2048;
    const size_t lds_bytes = (size_t)n_pad * sizeof(float);
    if (lds_bytes > 48 * 1024)   // above the default dynamic-LDS limit: opt in to what this launch needs
        BHR_HIP(hipFuncSetAttribute((const void *)row_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(row_stats_kernel, dim3(n_r), dim3(1024), lds_bytes, ctx->stream, d_ts, ctx->d_comp,
                       n_phi, n_pad, div, lo, hi, d_rows);
    BHR_HIP(hipGetLastError());
    BHR_HIP(hipMemcpyAsync(rows_out, d_rows, (size_t)n_r * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    return BHR_OK;
}

}  // extern "C"
