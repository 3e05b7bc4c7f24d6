// flare.hip -- lens-flare post effect on the device (reference: TaichiRenderer._apply_lens_flare,
// render.py:3925-4028, host NumPy there).
//
// The effect depends on three whole-frame sums of the disk layer -- S0 = sum glow (f32),
// S1 = sum x glow, S2 = sum y glow (f64), glow = max(r, g, b) -- and the flare geometry is sensitive
// to their last bits (one f32 ulp of S0 moves a 2.5-pixel ghost's alpha by 1e-6).  They are therefore
// summed in NumPy's own order: np.sum over the reference's C-contiguous (W, H) array walks it in
// chunks of 8192 elements (the ufunc buffer size), sums each chunk with the pairwise routine
// (128-element leaves with 8 strided accumulators, halves split at a multiple of 8) and accumulates
// the chunk sums sequentially.  A full chunk is a perfect tree of 64 leaves = one wavefront, one
// leaf per lane, combined with shuffles; the ragged last chunk runs a tree program built on the host.
//
//   flare_glow_kernel       disk (rows, W, 3) -> glow (rows, W)
//   flare_transpose_kernel  glow (H, W) -> (W, H), the reference's memory order
//   flare_chunk_kernel      one wave per full chunk -> 3 chunk sums
//   flare_fold_kernel       last chunk + sequential fold -> S0, S1, S2
//   flare_apply_kernel      one pixel per thread: eight ghosts, three rings, the hexagonal ring and
//                           four streaks added to the final layer, clipped to [0, 1]
//
// Row blocks (bhr_group_render): every tile writes its glow rows, tile 0 receives them over xGMI,
// sums, and the three totals go to every tile's apply launch.
//
// Pixel arithmetic follows NumPy's promotion in the reference line by line: geometry in f64, the ghost
// alphas rounded to f32 (stored in an f32 array, 3958-3959), intensity and the streak gain f32 (f32
// scalar x Python float), every `flare[..., c] += alpha * tint` an f64 add rounded to f32.
#include "bhr_internal.h"

#include <vector>

namespace {

constexpr int CHUNK = 8192;   // np.getbufsize()
constexpr int LEAF = 128;     // PW_BLOCKSIZE
constexpr int MAX_TAIL_LEAVES = 128;

__global__ __launch_bounds__(256) void flare_glow_kernel(const float *__restrict__ disk, float *__restrict__ glow, long long n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float *q = disk + p * 3;
    glow[p] = fmaxf(fmaxf(q[0], q[1]), q[2]);
}

__global__ __launch_bounds__(256) void flare_transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int y = by + k, x = bx + tx;
        if (y < H && x < W) tile[k][tx] = in[(size_t)y * W + x];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int x = bx + k, y = by + tx;
        if (y < H && x < W) out[(size_t)x * H + y] = tile[tx][k];
    }
}

struct Tri { float s0; double s1, s2; };

__device__ __forceinline__ Tri tri_add(Tri a, Tri b) { return Tri{a.s0 + b.s0, a.s1 + b.s1, a.s2 + b.s2}; }

// One pairwise leaf (n <= 128) of the three sums over elements [e0, e0 + n) of the (W, H) order.
__device__ __forceinline__ Tri leaf_sum(const float *__restrict__ g, long long e0, int n, int H) {
    int x = (int)(e0 / H), y = (int)(e0 - (long long)x * H);
    auto next = [&](long long e) {
        const float v = g[e];
        Tri t{v, (double)x * (double)v, (double)y * (double)v};
        if (++y == H) { y = 0; ++x; }
        return t;
    };
    if (n < 8) {
        Tri r{0.0f, 0.0, 0.0};
        for (int i = 0; i < n; ++i) r = tri_add(r, next(e0 + i));
        return r;
    }
    Tri r[8];
    for (int j = 0; j < 8; ++j) r[j] = next(e0 + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] = tri_add(r[j], next(e0 + i + j));
    Tri res = tri_add(tri_add(tri_add(r[0], r[1]), tri_add(r[2], r[3])), tri_add(tri_add(r[4], r[5]), tri_add(r[6], r[7])));
    for (; i < n; ++i) res = tri_add(res, next(e0 + i));
    return res;
}

__global__ __launch_bounds__(64) void flare_chunk_kernel(const float *__restrict__ g, int H, float *__restrict__ c0,
                                                         double *__restrict__ c1, double *__restrict__ c2) {
    const long long e0 = (long long)blockIdx.x * CHUNK + (long long)threadIdx.x * LEAF;
    Tri t = leaf_sum(g, e0, LEAF, H);
    for (int off = 1; off < 64; off <<= 1) {   // adjacent pairs first: the recursion's tree on 64 equal leaves
        Tri o{__shfl_down(t.s0, off, 64), __shfl_down(t.s1, off, 64), __shfl_down(t.s2, off, 64)};
        t = tri_add(t, o);
    }
    if (threadIdx.x == 0) { c0[blockIdx.x] = t.s0; c1[blockIdx.x] = t.s1; c2[blockIdx.x] = t.s2; }
}

// prog: [n_leaves, n_ops, (off, len) * n_leaves, (dst, src) * n_ops]; value[dst] += value[src] in order.
__global__ __launch_bounds__(256) void flare_fold_kernel(const float *__restrict__ g, int H, long long tail_e0,
                                                         const int *__restrict__ prog, int n_chunks,
                                                         const float *__restrict__ c0, const double *__restrict__ c1,
                                                         const double *__restrict__ c2, double *__restrict__ sums) {
    __shared__ float l0[MAX_TAIL_LEAVES];
    __shared__ double l1[MAX_TAIL_LEAVES], l2[MAX_TAIL_LEAVES];
    __shared__ float b0[256];
    __shared__ double b1[256], b2[256];
    const int n_leaves = prog[0], n_ops = prog[1];
    if ((int)threadIdx.x < n_leaves) {
        const Tri t = leaf_sum(g, tail_e0 + prog[2 + 2 * threadIdx.x], prog[3 + 2 * threadIdx.x], H);
        l0[threadIdx.x] = t.s0; l1[threadIdx.x] = t.s1; l2[threadIdx.x] = t.s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int *ops = prog + 2 + 2 * n_leaves;
        for (int k = 0; k < n_ops; ++k) {
            const int d = ops[2 * k], s = ops[2 * k + 1];
            l0[d] += l0[s]; l1[d] += l1[s]; l2[d] += l2[s];
        }
    }
    Tri r{0.0f, 0.0, 0.0};
    for (int base = 0; base < n_chunks; base += 256) {
        __syncthreads();
        const int c = base + threadIdx.x;
        if (c < n_chunks) { b0[threadIdx.x] = c0[c]; b1[threadIdx.x] = c1[c]; b2[threadIdx.x] = c2[c]; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = n_chunks - base < 256 ? n_chunks - base : 256;
            for (int k = 0; k < m; ++k) r = tri_add(r, Tri{b0[k], b1[k], b2[k]});
        }
    }
    if (threadIdx.x == 0) {
        if (n_leaves > 0) r = tri_add(r, Tri{l0[0], l1[0], l2[0]});
        sums[0] = (double)r.s0;
        sums[1] = r.s1;
        sums[2] = r.s2;
    }
}

struct FlareSums { double s0, s1, s2; };

__device__ __forceinline__ void add_tint(float (&fl)[3], double alpha, double t0, double t1, double t2) {
    fl[0] = (float)((double)fl[0] + alpha * t0);
    fl[1] = (float)((double)fl[1] + alpha * t1);
    fl[2] = (float)((double)fl[2] + alpha * t2);
}

__device__ __forceinline__ double clip01(double v) { return fmin(fmax(v, 0.0), 1.0); }
// np.mod for a positive divisor: result in [0, m)
__device__ __forceinline__ double np_mod(double a, double m) {
    double r = fmod(a, m);
    if (r < 0.0) r += m;
    return r;
}

__global__ __launch_bounds__(256) void flare_apply_kernel(float *__restrict__ fin, int W, int H, int row0, int rows,
                                                          const double *__restrict__ d_sums, FlareSums host_sums, int use_host) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long long)rows * W) return;
    FlareSums S = host_sums;
    if (!use_host) { S.s0 = d_sums[0]; S.s1 = d_sums[1]; S.s2 = d_sums[2]; }
    const float total = (float)S.s0;                 // np.sum of an f32 array is f32
    if (total < 0.01f) return;
    const int yl = (int)(p / W), xi = (int)(p - (long long)yl * W);
    const double x = (double)xi, y = (double)(row0 + yl);
    const double w = (double)W, h = (double)H;
    const double scale = fmin(w, h) / 360.0;
    const double src_x = S.s1 / (double)total, src_y = S.s2 / (double)total;
    const double mid_x = w / 2, mid_y = h / 2;
    // f32 scalar / Python float stays f32 under NumPy 2 promotion; Python's min() hands back the
    // float 1.0 when the ratio exceeds it, and intensity / gain are then plain doubles (3942, 4011)
    const float ratio = total / (float)(w * h * 0.3);
    const bool saturated = 1.0f < ratio;
    const float strength_f = ratio * 1.5f;
    const double strength = saturated ? 1.5 : (double)strength_f;
    const double gain = saturated ? 1.5 * 0.3 : (double)(strength_f * 0.3f);

    float fl[3] = {0.0f, 0.0f, 0.0f};

    // Every shape is zero outside a disc, an annulus or a narrow wedge.  A conservative test on the squared
    // distance (slack 1e-9, far above the rounding of the exact expressions) skips the square roots, atan2 and
    // exp where the contribution is exactly 0 -- and adding 0 to the f32 accumulator would not change it.
    const double SLACK = 1.0 + 1e-9;
    for (int g = 0; g < 8; ++g) {
        const double t = (g + 1) * 0.15;
        const double gx = src_x + (mid_x - src_x) * t, gy = src_y + (mid_y - src_y) * t;
        const double radius = (25 + g * 30) * scale;
        const double d2 = (x - gx) * (x - gx) + (y - gy) * (y - gy);
        if (d2 > radius * radius * SLACK) continue;
        const double dist = sqrt(d2);
        float alpha = 0.0f;
        if (dist < radius) {
            const double u = 1 - dist / radius;
            alpha = (float)(u * u * (1 - g * 0.08) * strength);
        }
        add_tint(fl, (double)alpha, 1.0, 0.9, 0.7);
    }
    for (int k = 0; k < 3; ++k) {
        const double t = 0.35 + k * 0.15;
        const double rx = src_x + (mid_x - src_x) * t, ry = src_y + (mid_y - src_y) * t;
        const double ring_r = (60 + k * 40) * scale, ring_w = (6 + k * 3) * scale;
        const double d2 = (x - rx) * (x - rx) + (y - ry) * (y - ry);
        const double lo = fmax(ring_r - ring_w, 0.0), hi = ring_r + ring_w;
        if (d2 > hi * hi * SLACK || d2 * SLACK < lo * lo) continue;
        const double dist = sqrt(d2);
        const double c = clip01(1 - fabs(dist - ring_r) / ring_w);
        const double alpha = c * c * 0.5 * strength * (1 - k * 0.25);
        if (k == 0) add_tint(fl, alpha, 0.3, 0.4, 1.0);
        else if (k == 1) add_tint(fl, alpha, 0.5, 0.5, 0.9);
        else add_tint(fl, alpha, 0.7, 0.5, 0.8);
    }
    const double PI = 3.141592653589793;
    {
        const double hx = src_x + (mid_x - src_x) * 0.5, hy = src_y + (mid_y - src_y) * 0.5;
        const double dx = x - hx, dy = y - hy;
        const double d2 = dx * dx + dy * dy;
        const double lo = 85 * scale, hi = 115 * scale;           // |dist - 100 scale| < 15 scale
        if (!(d2 > hi * hi * SLACK || d2 * SLACK < lo * lo)) {
            const double angle = atan2(dy, dx);
            const double dist = sqrt(d2);
            const double edge = fabs(np_mod(angle, PI / 3) - PI / 6);
            const double facet = clip01(1 - edge / 0.2);
            const double off = fabs(dist - 100 * scale);
            const double c = clip01(1 - off / (15 * scale));
            const double alpha = c * c * facet * 0.3 * strength;
            add_tint(fl, alpha, 0.6, 0.7, 1.0);
        }
    }
    {
        const double reach = fmin(w, h) * 0.4;
        const double dx = x - src_x, dy = y - src_y;
        // within 0.05 rad of one of the four axes  <=>  min(|dx|, |dy|) <= tan(0.05) max(|dx|, |dy|)
        const double ax = fabs(dx), ay = fabs(dy);
        if (fmin(ax, ay) <= 0.05005 * fmax(ax, ay)) {                 // tan(0.05) = 0.050042
            const double dist = sqrt(dx * dx + dy * dy);
            const double angle = atan2(dy, dx);
            const double a = exp(-dist / reach) * gain;     // falloff * streak_alpha, then * colour (4026)
            const double axes[4] = {0.0, PI / 2, PI, 3 * PI / 2};
            for (int s = 0; s < 4; ++s) {
                const double delta = fabs(np_mod(angle - axes[s] + PI, 2 * PI) - PI);
                const bool on = delta < 0.05;
                fl[0] = (float)((double)fl[0] + (on ? a * 1.0 : 0.0));
                fl[1] = (float)((double)fl[1] + (on ? a * 0.95 : 0.0));
                fl[2] = (float)((double)fl[2] + (on ? a * 0.9 : 0.0));
            }
        }
    }
    float *q = fin + p * 3;
    for (int c = 0; c < 3; ++c) q[c] = fminf(fmaxf(q[c] + fl[c], 0.0f), 1.0f);
}

// The pairwise recursion of one chunk of n elements as leaves + an in-order list of adds; the value of
// a subtree ends up in the slot of its leftmost leaf.  Returns that slot.
int build_tree(int off, int n, std::vector<int> &leaves, std::vector<int> &ops) {
    if (n <= LEAF) {
        leaves.push_back(off);
        leaves.push_back(n);
        return (int)leaves.size() / 2 - 1;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const int a = build_tree(off, n2, leaves, ops);
    const int b = build_tree(off + n2, n - n2, leaves, ops);
    ops.push_back(a);
    ops.push_back(b);
    return a;
}

// ctx->d_glow_* / d_flare_c* / d_flare_sums are the ACTIVE frame slot's (api.hip: activate_slot); what is allocated
// here is stored back into the slot.  The tail program is read-only and shared.
int32_t ensure_buffers_(bhr_ctx *ctx, bool whole_frame) {
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const size_t want_rows = whole_frame ? (size_t)H : (size_t)ctx->rows;
    if (ctx->flare_glow_rows < (int64_t)want_rows) {
        if (ctx->d_glow_hw) (void)hipFree(ctx->d_glow_hw);
        ctx->d_glow_hw = nullptr;
        BHR_HIP(hipMalloc((void **)&ctx->d_glow_hw, want_rows * W * sizeof(float)));
        ctx->flare_glow_rows = (int64_t)want_rows;
    }
    if (!ctx->d_flare_sums) BHR_HIP(hipMalloc((void **)&ctx->d_flare_sums, 3 * sizeof(double)));
    if (whole_frame && !ctx->d_glow_wh) {
        const long long n = (long long)W * H;
        const int n_chunks = (int)(n / CHUNK);
        BHR_HIP(hipMalloc((void **)&ctx->d_glow_wh, (size_t)n * sizeof(float)));
        BHR_HIP(hipMalloc((void **)&ctx->d_flare_c0, (size_t)(n_chunks + 1) * sizeof(float)));
        BHR_HIP(hipMalloc((void **)&ctx->d_flare_c12, (size_t)(n_chunks + 1) * 2 * sizeof(double)));
    }
    if (whole_frame && !ctx->d_flare_prog) {
        const long long n = (long long)W * H;
        const int tail = (int)(n % CHUNK);
        std::vector<int> leaves, ops;
        if (tail > 0) build_tree(0, tail, leaves, ops);
        if ((int)leaves.size() / 2 > MAX_TAIL_LEAVES) return bhr_fail(BHR_ERR_INVALID, "flare: tail tree has %d leaves", (int)leaves.size() / 2);
        std::vector<int> prog{(int)leaves.size() / 2, (int)ops.size() / 2};
        prog.insert(prog.end(), leaves.begin(), leaves.end());
        prog.insert(prog.end(), ops.begin(), ops.end());
        BHR_HIP(hipMalloc((void **)&ctx->d_flare_prog, prog.size() * sizeof(int)));
        BHR_HIP(hipMemcpy(ctx->d_flare_prog, prog.data(), prog.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return BHR_OK;
}

int32_t ensure_buffers(bhr_ctx *ctx, bool whole_frame) {
    const int32_t rc = ensure_buffers_(ctx, whole_frame);
    bhr_frame_slot &f = ctx->slots[ctx->active_slot];
    f.d_glow_hw = ctx->d_glow_hw;
    f.d_glow_wh = ctx->d_glow_wh;
    f.d_flare_c0 = ctx->d_flare_c0;
    f.d_flare_c12 = ctx->d_flare_c12;
    f.d_flare_sums = ctx->d_flare_sums;
    f.flare_glow_rows = ctx->flare_glow_rows;
    return rc;
}

}  // namespace

// glow of this context's rows -> its d_glow_hw (whole-frame context: at the rows' place in the frame)
int32_t bhr_launch_flare_glow(bhr_ctx *ctx, bool whole_frame) {
    if (int32_t rc = ensure_buffers(ctx, whole_frame)) return rc;
    const long long n = (long long)ctx->rows * ctx->cfg.width;
    float *dst = ctx->d_glow_hw + (whole_frame ? (size_t)ctx->cfg.row0 * ctx->cfg.width : 0);
    hipLaunchKernelGGL(flare_glow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_disk, dst, n);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

// d_glow_hw holds the WHOLE frame's glow: NumPy-ordered sums -> d_flare_sums
int32_t bhr_launch_flare_sums(bhr_ctx *ctx) {
    if (int32_t rc = ensure_buffers(ctx, true)) return rc;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const long long n = (long long)W * H;
    const int n_chunks = (int)(n / CHUNK);
    hipLaunchKernelGGL(flare_transpose_kernel, dim3((W + 31) / 32, (H + 31) / 32), dim3(256), 0, ctx->stream, ctx->d_glow_hw,
                       ctx->d_glow_wh, H, W);
    double *c1 = ctx->d_flare_c12, *c2 = ctx->d_flare_c12 + n_chunks + 1;
    if (n_chunks > 0)
        hipLaunchKernelGGL(flare_chunk_kernel, dim3(n_chunks), dim3(64), 0, ctx->stream, ctx->d_glow_wh, H, ctx->d_flare_c0, c1, c2);
    hipLaunchKernelGGL(flare_fold_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_glow_wh, H, (long long)n_chunks * CHUNK,
                       ctx->d_flare_prog, n_chunks, ctx->d_flare_c0, c1, c2, ctx->d_flare_sums);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

// sums == nullptr: use this context's own device-resident sums (whole-frame context, no host sync).
int32_t bhr_launch_flare_apply(bhr_ctx *ctx, const double *sums) {
    if (int32_t rc = ensure_buffers(ctx, false)) return rc;
    FlareSums hs{0.0, 0.0, 0.0};
    if (sums) hs = FlareSums{sums[0], sums[1], sums[2]};
    const long long n = (long long)ctx->rows * ctx->cfg.width;
    hipLaunchKernelGGL(flare_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_final, ctx->cfg.width,
                       ctx->cfg.height, ctx->cfg.row0, ctx->rows, ctx->d_flare_sums, hs, sums ? 1 : 0);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}
