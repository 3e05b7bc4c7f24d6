// hybrid.hip -- math_mode BHR_MATH_HYBRID: the strict march where the geodesic is unstable, the fast march elsewhere.
//
// Why a static, per-tile choice is enough.  The reference's equation of motion (render.py:2518-2524) is the
// Schwarzschild null geodesic in Binet form; it is integrable, and the only rays that amplify rounding are those whose
// impact parameter b = |pos x dir| (= sqrt(L2) of render.py:2828, conserved) lies next to the critical value
// b_c = (3 sqrt 3 / 2) r_s of the photon sphere: they wind around r = 1.5 r_s and their deflection grows like
// -ln|b / b_c - 1|, so an error eps made anywhere on the inbound leg leaves as eps b / |b - b_c|.  Everything else --
// direct disk hits, weakly bent sky rays, rays that fall straight in -- carries the ~1e-7 per-step rounding of the fast
// arithmetic through unamplified.  A ray is unstable for its whole life or not at all (b is fixed when it is
// launched), hence no switching in flight: an 8x8-pixel tile whose rays have b inside [b_c - lo, b_c + hi] is marched
// by the STRICT kernel (bit-identical paths, as math_mode 1), every other tile by the FAST kernel.  The two kernels are
// the ones the other two modes launch; hybrid is host code: the classification (b at the tile corners in binary64
// from the camera uniforms, padded by one tile's span), a stable partition of the context's longest-first tile order
// into two device lists, and two launches bracketed as one march.  The lists are cached per frame slot and reused
// while the view's geometry (|cam|, cam . {forward, right, up}, pixel pitch) is unchanged -- an orbit at constant
// radius keeps them for the whole video.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bhr_internal.h"

namespace {

constexpr double B_CRIT = 2.598076211353316;   // 3 sqrt(3) / 2 r_s, r_s = 1
constexpr double PLANE_SIN = 0.02;             // orbital planes within 1.1 degrees of the disk plane march strict (classify)

constexpr int HYBRID_LISTS = 3;   // base lists a march can be launched over: whole block, halo bands, the rest (bhr_march_part.id)

struct FixList {           // per frame slot: pixels the fast list's guard kernel leaves to the strict fix kernel
    unsigned int *d_count;
    int32_t *d_list;
    int32_t cap;
};

constexpr int LIST_RING = 4;
struct SlotLists {
    int32_t *d_list;       // the base list partitioned: strict tiles first (launch order kept), then the fast ones
    int32_t *h_pinned;
    hipEvent_t copied;     // the last upload from h_pinned
    double key[12];
    int32_t n_strict, base_n, valid, pending;
    // device classification: a partition is written on the classification stream while the slot's earlier frames may still
    // march over the previous one -- four buffers in turn, each guarded by the event of the last march that read it
    int32_t *d_ring[LIST_RING];
    hipEvent_t used[LIST_RING];
    int32_t used_set[LIST_RING];
    int32_t cur;           // ring entry the slot marches over, -1 none yet
    int32_t on_device;     // which of the two paths made the list in use
    const int32_t *d_active;
};

struct Hybrid {
    SlotLists slot[BHR_MAX_FRAME_SLOTS][HYBRID_LISTS];
    FixList fix[BHR_MAX_FRAME_SLOTS];
    // last classification on the host
    std::vector<uint8_t> strict;   // per tile of the row block: marched strict
    double key[12];
    int32_t n_strict, valid, n_tiles, on_device;
    double lo, hi;                 // band below / above b_c (BHR_HYBRID_BAND="lo,hi")
    double eff_lo, eff_hi;         // ... as the last march used it (widened with the step size beyond 0.1)
    int32_t last_fix_slot;         // frame slot whose fix list the last march used, -1: it ran without guards
    // device classification
    hipStream_t cls_stream;
    uint8_t *d_flags;
    int32_t *d_counts, *d_total, *h_total;
    int32_t n_tiles_alloc;
};

void view_key(const bhr_camera *cam, double lo, double hi, double pad, double tilt_deg, double key[12]) {
    double p[3], r2 = 0, pf = 0, pr = 0, pu = 0;
    for (int k = 0; k < 3; ++k) {
        p[k] = cam->pos[k];
        r2 += p[k] * p[k];
        pf += p[k] * (double)cam->forward[k];
        pr += p[k] * (double)cam->right[k];
        pu += p[k] * (double)cam->up[k];
    }
    key[0] = sqrt(r2); key[1] = pf; key[2] = pr; key[3] = pu;
    key[4] = cam->pixel_width; key[5] = cam->pixel_height; key[6] = lo; key[7] = hi + 16.0 * pad;      // (the padding factor rides on the band's upper width: a key, not a quantity)
    // the in-plane family (classify) only exists for a camera within PLANE_SIN of the disk plane as seen from the hole: the
    // view's orientation against the disk normal enters the key there and nowhere else (an orbit about a tilted disk keeps
    // its cached lists for all the frames in which it is clear of the plane)
    const double tilt = tilt_deg * 3.14159265358979323846 / 180.0;
    const double nrm[3] = {0.0, -sin(tilt), cos(tilt)};
    double pn = 0, fn = 0, rn = 0, un = 0;
    for (int k = 0; k < 3; ++k) {
        pn += p[k] * nrm[k];
        fn += (double)cam->forward[k] * nrm[k];
        rn += (double)cam->right[k] * nrm[k];
        un += (double)cam->up[k] * nrm[k];
    }
    const bool near_plane = fabs(pn) <= (PLANE_SIN * 1.05) * sqrt(r2);
    key[8] = near_plane ? pn : 1e30; key[9] = near_plane ? fn : 0; key[10] = near_plane ? rn : 0; key[11] = near_plane ? un : 0;
}

// the classification only depends on the view through the key; geometry within 1e-5 (absolute, in r_s) of the cached
// one moves b by less than the 1e-3 the band is padded with
bool same_view(const double a[12], const double b[12]) {
    for (int k = 0; k < 4; ++k)
        if (fabs(a[k] - b[k]) > 1e-5) return false;
    for (int k = 4; k < 12; ++k)
        if (a[k] != b[k]) return false;
    return true;
}

// Padding of a tile's range of b in the band test.  A tile whose span is a large part of the band (low resolutions, far cameras:
// >= 0.2 r_s against a band of 0.445) is padded by its whole span, as in round 3 -- halving it there raised the share of fuzzed
// 192 x 128 views beyond 6e-5 from 1.3 to 1.8 % (`profiles/r04d_fuzz_hybrid_*`); one whose span is <= 0.1 r_s (the ring tiles of
// an fhd frame from 6 r_s: 0.08) by the share pad_f of it (option hybrid_pad, default 0.5: the sagitta that the padding is for
// is 0.2 % of the span there), linearly in between.
__host__ __device__ inline double tile_pad(double span, double pad_f) {
#pragma clang fp contract(off)
    const double t = span <= 0.1 ? 0.0 : (span >= 0.2 ? 1.0 : (span - 0.1) / 0.1);
    return (pad_f + (1.0 - pad_f) * t) * span + 1e-3;
}

// strict[t] = 1 for the tiles of this row block whose rays may have b in [b_c - lo, b_c + hi]
void classify(const bhr_ctx *ctx, const bhr_camera *cam, double lo, double hi, double pad_f, std::vector<uint8_t> &strict) {
    const int W = ctx->cfg.width, H = ctx->cfg.height, row0 = ctx->cfg.row0, rows = ctx->rows;
    const int tiles_x = (W + 7) / 8, tiles_y = (rows + 7) / 8;
    double cp[3], cr[3], cu[3], cf[3], tl[3];
    for (int k = 0; k < 3; ++k) { cp[k] = cam->pos[k]; cr[k] = cam->right[k]; cu[k] = cam->up[k]; cf[k] = cam->forward[k]; }
    const double pw = cam->pixel_width, ph = cam->pixel_height;
    const double half_w = pw * W / 2, half_h = ph * H / 2;
    for (int k = 0; k < 3; ++k) tl[k] = cp[k] + cf[k] - half_w * cr[k] + half_h * cu[k];   // render.py:2811-2816
    const double r0sq = cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2];
    // b and the radial sense at the tile-boundary grid: x = 8 gx - 0.5, y = row0 + 8 gy - 0.5 in pixel-centre units
    const int gx_n = tiles_x + 1, gy_n = tiles_y + 1;
    std::vector<float> bgrid((size_t)gx_n * gy_n);
    std::vector<uint8_t> outgoing((size_t)gx_n * gy_n);
    // A second unstable family: rays whose orbital plane all but coincides with the disk plane (a camera within a degree of
    // the disk plane sees them as a line through the hole's image).  The plane function is ~0 all along such a ray -- where it
    // "crosses" is decided by rounding, in the reference's arithmetic as in any other -- and the fast kernel's basis (g1 =
    // line of nodes of the two planes) is ill defined.  sgrid = sin of the angle between the planes, up = side of the disk
    // plane the ray leaves the camera on; tiles that may hold a ray with sgrid < PLANE_SIN go to the strict list.
    const double tilt = (double)ctx->cfg.disk_tilt_deg * 3.14159265358979323846 / 180.0;
    const double nrm[3] = {0.0, -sin(tilt), cos(tilt)};                      // z cos(tilt) - y sin(tilt) = 0
    const double cpn = cp[0] * nrm[0] + cp[1] * nrm[1] + cp[2] * nrm[2];
    std::vector<float> sgrid((size_t)gx_n * gy_n), blgrid((size_t)gx_n * gy_n);
    std::vector<uint8_t> up((size_t)gx_n * gy_n);
    for (int gy = 0; gy < gy_n; ++gy) {
        const double y = (double)row0 + (double)(gy * 8 < rows ? gy * 8 : rows) - 0.5;
        for (int gx = 0; gx < gx_n; ++gx) {
            const double x = (double)(gx * 8 < W ? gx * 8 : W) - 0.5;
            double d[3], dn = 0, pd = 0;
            for (int k = 0; k < 3; ++k) {
                d[k] = tl[k] + (x + 0.5) * pw * cr[k] - (y + 0.5) * ph * cu[k] - cp[k];
                dn += d[k] * d[k];
                pd += cp[k] * d[k];
            }
            pd /= sqrt(dn);
            {
                const double inv_d = 1.0 / sqrt(dn);
                double dnn = 0, v2 = 0;
                for (int k = 0; k < 3; ++k) dnn += d[k] * inv_d * nrm[k];
                for (int k = 0; k < 3; ++k) { const double v = d[k] * inv_d * cpn - cp[k] * dnn; v2 += v * v; }   // (cp x d) x n
                const double bl = sqrt(r0sq - pd * pd > 1e-18 ? r0sq - pd * pd : 1e-18);
                sgrid[(size_t)gy * gx_n + gx] = (float)(sqrt(v2) / bl);
                blgrid[(size_t)gy * gx_n + gx] = (float)bl;
                up[(size_t)gy * gx_n + gx] = dnn > 0;
            }
            // The orbit is fixed by the first integral of the path equation the reference integrates (u'' + u = 3/2 u^2,
            // u = 1 / r; render.py:2928-2934 is its Cartesian form):  u'^2 + u^2 - u^3 = 1 / b_l^2 - 1 / r0^3  with the LOCAL
            // moment b_l = |pos x dir| -- not by b_l itself.  The ray whirls at the photon sphere when that integral is
            // 4 / 27 = 1 / b_c^2; expressed as a length, b = (1 / b_l^2 - 1 / r0^3)^(-1/2), which is b_l for a far camera, 1.6 %
            // more at the default pov (6 r_s) and 17 % more at 2.6 r_s (found by the fuzzed views of tests/test_gpu_fuzz.py:
            // a band on b_l missed every near-critical ray of cameras inside 3 r_s).
            const double bl2 = r0sq - pd * pd;
            const double inv = (bl2 > 1e-12 ? 1.0 / bl2 : 1e12) - 1.0 / (r0sq * sqrt(r0sq));
            bgrid[(size_t)gy * gx_n + gx] = (float)(inv > 1e-6 ? 1.0 / sqrt(inv) : 1e3);
            outgoing[(size_t)gy * gx_n + gx] = pd > 0;
        }
    }
    const bool far_cam = r0sq > 9.0;     // outside 3 r_s an outgoing ray never comes near the photon sphere
    strict.assign((size_t)tiles_x * tiles_y, 0);
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            const size_t g = (size_t)ty * gx_n + tx;
            const float c[4] = {bgrid[g], bgrid[g + 1], bgrid[g + gx_n], bgrid[g + gx_n + 1]};
            float bmin = c[0], bmax = c[0];
            for (int k = 1; k < 4; ++k) { bmin = c[k] < bmin ? c[k] : bmin; bmax = c[k] > bmax ? c[k] : bmax; }
            {
                // sin^2 of the angle between the planes ~ (cam . n / b_l)^2 + (|cam| (d . n) / b_l)^2: the family is a thin wedge
                // around the line d . n = 0, present only where |cam . n| / b_l is small.  A tile belongs to it when that line
                // runs through it (its corners leave the camera on both sides of the disk plane) or a corner lies in the wedge.
                const size_t q[4] = {g, g + 1, g + gx_n, g + gx_n + 1};
                float smin = sgrid[q[0]], blmax = blgrid[q[0]];
                int ups = 0;
                for (int k = 0; k < 4; ++k) {
                    smin = sgrid[q[k]] < smin ? sgrid[q[k]] : smin;
                    blmax = blgrid[q[k]] > blmax ? blgrid[q[k]] : blmax;
                    ups += up[q[k]];
                }
                if (fabs(cpn) < PLANE_SIN * (double)blmax && ((ups != 0 && ups != 4) || (double)smin < 1.5 * PLANE_SIN)) {
                    strict[(size_t)ty * tiles_x + tx] = 1;
                    continue;
                }
            }
            if (far_cam && outgoing[g] && outgoing[g + 1] && outgoing[g + gx_n] && outgoing[g + gx_n + 1]) continue;
            // b grows with the distance from the hole's image in a convex sense (the field of view stays under 180 degrees): its
            // maximum over the tile is at a corner, its minimum may lie on an edge -- pad by the tile's own span; by a share of it
            // (pad_f) where the span is small against the band (tile_pad)
            const double pad = tile_pad((double)(bmax - bmin), pad_f);
            if (bmax + pad >= B_CRIT - lo && bmin - pad <= B_CRIT + hi) strict[(size_t)ty * tiles_x + tx] = 1;
        }
}


// ---- the same classification on the device (round 4) -------------------------------------------------------------------
// `classify` above is O(tiles) of binary64 work on the submitting thread -- 32 400 tiles at fhd, 518 400 at 8k (50 ms) -- and
// the partition of the launch order behind it another pass over every tile: fine for an orbit at constant radius (cached),
// a stall in front of every frame of a camera path that changes its distance.  Here every tile evaluates the same rule, the
// same operations in the same order (no FMA contraction in this file), in a kernel; the stable partition of the launch order
// is three small kernels (per-block counts, scan of the counts, scatter); all of it runs on a stream of its own, the host
// waits for the strict count alone (it sizes the two march launches): ~40 us per view change whatever the resolution.
struct ClassifyArgs {
    double cp[3], cr[3], cu[3], tl[3], nrm[3];
    double pw, ph, r0sq, cpn, lo, hi, pad_f;
    int32_t W, rows, row0, tiles_x, tiles_y, far_cam;
};

struct Corner { float b, s, bl; int32_t bits; };   // bits: 1 leaves the camera above the disk plane, 2 outgoing

// one corner of the tile grid: the body of classify's first loop, operation for operation
__device__ __forceinline__ Corner classify_corner(const ClassifyArgs &a, int gx, int gy) {
#pragma clang fp contract(off)
    const double y = (double)a.row0 + (double)(gy * 8 < a.rows ? gy * 8 : a.rows) - 0.5;
    const double x = (double)(gx * 8 < a.W ? gx * 8 : a.W) - 0.5;
    double d[3], dn = 0, pd = 0;
    for (int k = 0; k < 3; ++k) {
        d[k] = a.tl[k] + (x + 0.5) * a.pw * a.cr[k] - (y + 0.5) * a.ph * a.cu[k] - a.cp[k];
        dn += d[k] * d[k];
        pd += a.cp[k] * d[k];
    }
    pd /= sqrt(dn);
    const double inv_d = 1.0 / sqrt(dn);
    double dnn = 0, v2 = 0;
    for (int k = 0; k < 3; ++k) dnn += d[k] * inv_d * a.nrm[k];
    for (int k = 0; k < 3; ++k) { const double v = d[k] * inv_d * a.cpn - a.cp[k] * dnn; v2 += v * v; }
    const double bl = sqrt(a.r0sq - pd * pd > 1e-18 ? a.r0sq - pd * pd : 1e-18);
    Corner c;
    c.s = (float)(sqrt(v2) / bl);
    c.bl = (float)bl;
    const double bl2 = a.r0sq - pd * pd;
    const double inv = (bl2 > 1e-12 ? 1.0 / bl2 : 1e12) - 1.0 / (a.r0sq * sqrt(a.r0sq));
    c.b = (float)(inv > 1e-6 ? 1.0 / sqrt(inv) : 1e3);
    c.bits = (dnn > 0 ? 1 : 0) | (pd > 0 ? 2 : 0);
    return c;
}

constexpr int CLS_EDGE = 16;     // a workgroup classifies 16 x 16 tiles from their 17 x 17 corners (each evaluated once, through LDS)
__global__ __launch_bounds__(CLS_EDGE * CLS_EDGE) void hybrid_classify_kernel(ClassifyArgs a, uint8_t *__restrict__ flags, int32_t *__restrict__ total_all) {
#pragma clang fp contract(off)
    __shared__ Corner cs[(CLS_EDGE + 1) * (CLS_EDGE + 1)];
    const int bx = blockIdx.x * CLS_EDGE, by = blockIdx.y * CLS_EDGE;
    for (int k = threadIdx.x; k < (CLS_EDGE + 1) * (CLS_EDGE + 1); k += CLS_EDGE * CLS_EDGE) {
        const int gx = bx + k % (CLS_EDGE + 1), gy = by + k / (CLS_EDGE + 1);
        if (gx <= a.tiles_x && gy <= a.tiles_y) cs[k] = classify_corner(a, gx, gy);
    }
    __syncthreads();
    const int lx = threadIdx.x % CLS_EDGE, ly = threadIdx.x / CLS_EDGE;
    const int tx = bx + lx, ty = by + ly;
    int f = 0;
    const bool live = tx < a.tiles_x && ty < a.tiles_y;
    if (live) {
        const Corner c[4] = {cs[ly * (CLS_EDGE + 1) + lx], cs[ly * (CLS_EDGE + 1) + lx + 1], cs[(ly + 1) * (CLS_EDGE + 1) + lx], cs[(ly + 1) * (CLS_EDGE + 1) + lx + 1]};
        float bmin = c[0].b, bmax = c[0].b, smin = c[0].s, blmax = c[0].bl;
        int ups = 0, outs = 0;
        for (int k = 0; k < 4; ++k) {
            bmin = c[k].b < bmin ? c[k].b : bmin;
            bmax = c[k].b > bmax ? c[k].b : bmax;
            smin = c[k].s < smin ? c[k].s : smin;
            blmax = c[k].bl > blmax ? c[k].bl : blmax;
            ups += c[k].bits & 1;
            outs += (c[k].bits >> 1) & 1;
        }
        if (fabs(a.cpn) < PLANE_SIN * (double)blmax && ((ups != 0 && ups != 4) || (double)smin < 1.5 * PLANE_SIN)) f = 1;
        else if (a.far_cam && outs == 4) f = 0;
        else {
            const double pad = tile_pad((double)(bmax - bmin), a.pad_f);
            f = (bmax + pad >= B_CRIT - a.lo && bmin - pad <= B_CRIT + a.hi) ? 1 : 0;
        }
        flags[(size_t)ty * a.tiles_x + tx] = (uint8_t)f;
    }
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(total_all, __popcll(m));
}

constexpr int PART_BLOCK = 256;
// counts[b] = strict tiles among launch-order positions [256 b, 256 b + 256)
__global__ __launch_bounds__(PART_BLOCK) void hybrid_count_kernel(const int32_t *__restrict__ order, int n, const uint8_t *__restrict__ flags,
                                                                   int32_t *__restrict__ counts) {
    __shared__ int wave_n[PART_BLOCK / 64];
    const int pos = blockIdx.x * PART_BLOCK + threadIdx.x;
    const int f = pos < n ? flags[order[pos]] : 0;
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
}
// exclusive scan of counts[0 .. nb) in place (one workgroup); the list's strict count -> *total and, with the frame's, -> pinned host memory
__global__ __launch_bounds__(1024) void hybrid_scan_kernel(int32_t *__restrict__ counts, int nb, int32_t *__restrict__ total, const int32_t *__restrict__ total_all,
                                                            int32_t *__restrict__ total_host) {
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int k = base + threadIdx.x;
        const int v = k < nb ? counts[k] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {              // Hillis-Steele: 10 steps per 1024 counts
            const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (k < nb) counts[k] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) { *total = carry; total_host[0] = carry; total_host[1] = *total_all; }
}
// stable partition of the launch order: strict tiles first, then the fast ones, both in launch order
__global__ __launch_bounds__(PART_BLOCK) void hybrid_scatter_kernel(const int32_t *__restrict__ order, int n, const uint8_t *__restrict__ flags,
                                                                     const int32_t *__restrict__ offsets, const int32_t *__restrict__ total,
                                                                     int32_t *__restrict__ out) {
    __shared__ int wave_n[PART_BLOCK / 64];
    const int pos = blockIdx.x * PART_BLOCK + threadIdx.x;
    int tile = 0, f = 0;
    if (pos < n) { tile = order[pos]; f = flags[tile]; }
    const unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_n[w] = __popcll(m);
    __syncthreads();
    int before = offsets[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) before += wave_n[q];
    if (pos < n) out[f ? before : *total + (pos - before)] = tile;
}

}  // namespace

void bhr_hybrid_free(bhr_ctx *ctx) {
    Hybrid *h = (Hybrid *)ctx->hybrid;
    if (!h) return;
    for (auto &row : h->slot)
        for (auto &s : row) {
            if (s.d_list) (void)hipFree(s.d_list);
            if (s.h_pinned) (void)hipHostFree(s.h_pinned);
            if (s.copied) (void)hipEventDestroy(s.copied);
        }
    for (auto &f : h->fix) {
        if (f.d_count) (void)hipFree(f.d_count);
        if (f.d_list) (void)hipFree(f.d_list);
    }
    for (auto &row : h->slot)
        for (auto &sl : row)
            for (int k = 0; k < LIST_RING; ++k) {
                if (sl.d_ring[k]) (void)hipFree(sl.d_ring[k]);
                if (sl.used[k]) (void)hipEventDestroy(sl.used[k]);
            }
    if (h->cls_stream) { (void)hipStreamSynchronize(h->cls_stream); (void)hipStreamDestroy(h->cls_stream); }
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->d_counts) (void)hipFree(h->d_counts);
    if (h->d_total) (void)hipFree(h->d_total);
    if (h->h_total) (void)hipHostFree(h->h_total);
    delete h;
    ctx->hybrid = nullptr;
}

int32_t bhr_hybrid_active_list(bhr_ctx *ctx, const int32_t **list, int32_t *n) {
    Hybrid *h = (Hybrid *)ctx->hybrid;
    const int k = ctx->active_slot >= 0 && ctx->active_slot < BHR_MAX_FRAME_SLOTS ? ctx->active_slot : 0;
    if (!h || !h->slot[k][0].valid || !h->slot[k][0].d_active) return bhr_fail(BHR_ERR_STATE, "no hybrid march has run on this context's active slot");
    *list = h->slot[k][0].d_active;
    *n = h->slot[k][0].base_n;
    return BHR_OK;
}

extern "C" int32_t bhr_hybrid_info(bhr_ctx *ctx, int32_t out_tiles[2], double out_band[2]) {
    if (!ctx || !out_tiles || !out_band) return bhr_fail(BHR_ERR_INVALID, "bhr_hybrid_info: bad argument");
    Hybrid *h = (Hybrid *)ctx->hybrid;
    if (!h || !h->valid) return bhr_fail(BHR_ERR_STATE, "bhr_hybrid_info: no hybrid march has run on this context");
    out_tiles[0] = h->n_strict;
    out_tiles[1] = h->n_tiles;
    out_band[0] = h->eff_lo;
    out_band[1] = h->eff_hi;
    return BHR_OK;
}

// {pixels the guard kernel of the last hybrid frame put on its fix list (it keeps counting past the capacity: the pixels beyond
// it keep their fast value), capacity of the list}; {0, 0} when that frame ran without guards.  Synchronises.
extern "C" int32_t bhr_hybrid_repairs(bhr_ctx *ctx, int32_t out[2]) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_hybrid_repairs: bad argument");
    Hybrid *h = (Hybrid *)ctx->hybrid;
    if (!h || !h->valid) return bhr_fail(BHR_ERR_STATE, "bhr_hybrid_repairs: no hybrid march has run on this context");
    out[0] = out[1] = 0;
    if (h->last_fix_slot < 0) return BHR_OK;
    FixList &fx = h->fix[h->last_fix_slot];
    if (!fx.d_count) return BHR_OK;
    BHR_TRY(bhr_enter(ctx));
    unsigned int n = 0;
    BHR_HIP(hipMemcpyAsync(&n, fx.d_count, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    out[0] = (int32_t)(n > 0x7fffffffu ? 0x7fffffffu : n);
    out[1] = fx.cap;
    return BHR_OK;
}

int32_t bhr_launch_march_hybrid(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    BHR_TRY(bhr_ensure_tile_order(ctx));
    Hybrid *h = (Hybrid *)ctx->hybrid;
    if (!h) {
        h = new Hybrid();
        memset(h->slot, 0, sizeof(h->slot));
        for (auto &row : h->slot)
            for (auto &sl : row) sl.cur = -1;
        h->n_tiles = 0;
        h->on_device = -1;
        memset(h->fix, 0, sizeof(h->fix));
        h->cls_stream = nullptr;
        h->d_flags = nullptr;
        h->d_counts = h->d_total = h->h_total = nullptr;
        h->n_tiles_alloc = 0;
        h->valid = 0;
        h->n_strict = 0;
        h->last_fix_slot = -1;
        // band around b_c, in r_s: measured on the fixtures and the fhd / 4k / e2e frames (tools/hybrid_sweep.py)
        ctx->hybrid = h;
    }
    h->lo = 0.085;           // in the orbit's own b (classify): the band certified in round 3 on |pos x dir| at the 6 r_s pov,
    h->hi = 0.36;            // [2.478, 2.898], is [b_c - 0.084, b_c + 0.357] there
    if (ctx->opt.hybrid_band_set) { h->lo = ctx->opt.hybrid_band[0]; h->hi = ctx->opt.hybrid_band[1]; }   // BHR_HYBRID_BAND="lo,hi" at bhr_create, or bhr_set_option
    // the list to split: the whole row block, or the sub-list of a pipelined launch (halo bands / the rest)
    const bhr_march_part base = ctx->part;
    const int32_t *base_list = base.active ? base.h_list : ctx->h_tile_order;
    const int base_n = base.active ? base.n : ctx->tile_order_n;
    const int id = base.active ? base.id : 0;
    if (id < 0 || id >= HYBRID_LISTS || !base_list) return bhr_fail(BHR_ERR_INVALID, "hybrid march: bad base list %d", id);
    double key[12];
    // the band was certified at step sizes up to 0.1; a coarser march amplifies more per step around the ring (a 0.3 march
    // lost a faint crossing at b_c + 0.59 that the binary64 evaluation keeps): the band widens with the step
    const double widen = ctx->cfg.step_size > 0.1f ? (double)ctx->cfg.step_size / 0.1 : 1.0;
    const double lo = h->lo * widen, hi = h->hi * widen;
    h->eff_lo = lo;
    h->eff_hi = hi;
    const double pad_f = ctx->opt.hybrid_pad;
    view_key(cam, lo, hi, pad_f, (double)ctx->cfg.disk_tilt_deg, key);
    const int n_tiles = ctx->tile_order_n;
    const bool on_device = ctx->opt.hybrid_classify != 0;
    SlotLists &s = h->slot[ctx->active_slot >= 0 && ctx->active_slot < BHR_MAX_FRAME_SLOTS ? ctx->active_slot : 0][id];
    const bool new_view = !h->valid || h->on_device != (int32_t)on_device || !same_view(h->key, key);
    if (on_device) {
        const int32_t *d_base = base.active ? base.d_list : ctx->d_tile_order;
        if (!d_base) return bhr_fail(BHR_ERR_INVALID, "hybrid march: the base list has no device copy");
        if (!h->cls_stream) {
            // highest priority: the host waits for these four kernels; and HIP keeps a pool of hardware queues per priority, so this
            // stream does not move the frame streams' places in the normal-priority pool (api.hip: pad_streams)
            int lo_p = 0, hi_p = 0;
            BHR_HIP(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
            BHR_HIP(hipStreamCreateWithPriority(&h->cls_stream, hipStreamNonBlocking, hi_p));
            BHR_HIP(hipMalloc((void **)&h->d_flags, (size_t)n_tiles));
            BHR_HIP(hipMalloc((void **)&h->d_counts, (size_t)((n_tiles + PART_BLOCK - 1) / PART_BLOCK + 1) * sizeof(int32_t)));
            BHR_HIP(hipMalloc((void **)&h->d_total, 2 * sizeof(int32_t)));
            BHR_HIP(hipHostMalloc((void **)&h->h_total, 2 * sizeof(int32_t), hipHostMallocDefault));
            h->n_tiles_alloc = n_tiles;
        }
        if (h->n_tiles_alloc != n_tiles) return bhr_fail(BHR_ERR_STATE, "hybrid march: the tile grid changed under a live context");
        const bool new_list = new_view || !s.valid || s.base_n != base_n || s.on_device != 1 || memcmp(s.key, h->key, sizeof(s.key)) != 0;
        if (new_list) {
            // every march that reads the list this slot used so far has been submitted: its buffer is free once the slot's
            // stream has passed this point.  The next buffer of the ring was released that way three view changes ago.
            if (s.cur >= 0) {
                BHR_HIP(hipEventRecord(s.used[s.cur], ctx->stream));
                s.used_set[s.cur] = 1;
            }
            const int nxt = (s.cur + 1) % LIST_RING;
            if (!s.d_ring[nxt]) {
                BHR_HIP(hipMalloc((void **)&s.d_ring[nxt], (size_t)n_tiles * sizeof(int32_t)));
                BHR_HIP(hipEventCreateWithFlags(&s.used[nxt], hipEventDisableTiming));
            }
            if (s.used_set[nxt]) BHR_HIP(hipStreamWaitEvent(h->cls_stream, s.used[nxt], 0));
            if (new_view) {
                ClassifyArgs ca;
                const int W = ctx->cfg.width, H = ctx->cfg.height;
                double cf[3];
                for (int k = 0; k < 3; ++k) { ca.cp[k] = cam->pos[k]; ca.cr[k] = cam->right[k]; ca.cu[k] = cam->up[k]; cf[k] = cam->forward[k]; }
                ca.pw = cam->pixel_width; ca.ph = cam->pixel_height;
                const double half_w = ca.pw * W / 2, half_h = ca.ph * H / 2;
                for (int k = 0; k < 3; ++k) ca.tl[k] = ca.cp[k] + cf[k] - half_w * ca.cr[k] + half_h * ca.cu[k];
                ca.r0sq = ca.cp[0] * ca.cp[0] + ca.cp[1] * ca.cp[1] + ca.cp[2] * ca.cp[2];
                const double tilt = (double)ctx->cfg.disk_tilt_deg * 3.14159265358979323846 / 180.0;
                ca.nrm[0] = 0.0; ca.nrm[1] = -sin(tilt); ca.nrm[2] = cos(tilt);
                ca.cpn = ca.cp[0] * ca.nrm[0] + ca.cp[1] * ca.nrm[1] + ca.cp[2] * ca.nrm[2];
                ca.lo = lo; ca.hi = hi; ca.pad_f = pad_f;
                ca.W = W; ca.rows = ctx->rows; ca.row0 = ctx->cfg.row0;
                ca.tiles_x = (W + 7) / 8; ca.tiles_y = (ctx->rows + 7) / 8;
                ca.far_cam = ca.r0sq > 9.0;
                if ((long long)ca.tiles_x * ca.tiles_y != n_tiles) return bhr_fail(BHR_ERR_STATE, "hybrid march: %d x %d tiles, launch order of %d", ca.tiles_x, ca.tiles_y, n_tiles);
                BHR_HIP(hipMemsetAsync(h->d_total + 1, 0, sizeof(int32_t), h->cls_stream));
                hipLaunchKernelGGL(hybrid_classify_kernel, dim3((ca.tiles_x + CLS_EDGE - 1) / CLS_EDGE, (ca.tiles_y + CLS_EDGE - 1) / CLS_EDGE), dim3(CLS_EDGE * CLS_EDGE), 0,
                                   h->cls_stream, ca, h->d_flags, h->d_total + 1);
            }
            const int nb = (base_n + PART_BLOCK - 1) / PART_BLOCK;
            if (nb > 0) {
                hipLaunchKernelGGL(hybrid_count_kernel, dim3(nb), dim3(PART_BLOCK), 0, h->cls_stream, d_base, base_n, h->d_flags, h->d_counts);
                hipLaunchKernelGGL(hybrid_scan_kernel, dim3(1), dim3(1024), 0, h->cls_stream, h->d_counts, nb, h->d_total, h->d_total + 1, h->h_total);
                hipLaunchKernelGGL(hybrid_scatter_kernel, dim3(nb), dim3(PART_BLOCK), 0, h->cls_stream, d_base, base_n, h->d_flags, h->d_counts, h->d_total, s.d_ring[nxt]);
                BHR_HIP(hipGetLastError());
                // the one host wait of a view change: the strict count sizes the two launches (the lists themselves stay on the device)
                BHR_HIP(hipStreamSynchronize(h->cls_stream));
                s.n_strict = h->h_total[0];
                if (new_view) h->n_strict = h->h_total[1];
            } else {
                s.n_strict = 0;
            }
            if (s.n_strict < 0 || s.n_strict > base_n) return bhr_fail(BHR_ERR_HIP, "hybrid march: the partition counted %d strict tiles of %d", s.n_strict, base_n);
            s.cur = nxt;
            s.d_active = s.d_ring[nxt];
            memcpy(h->key, key, sizeof(key));
            h->valid = 1;
            h->on_device = 1;
            h->n_tiles = n_tiles;
            memcpy(s.key, h->key, sizeof(s.key));
            s.base_n = base_n;
            s.valid = 1;
            s.on_device = 1;
        }
    } else {
        if (new_view) {
            classify(ctx, cam, lo, hi, pad_f, h->strict);
            int n = 0;
            for (int k = 0; k < ctx->tile_order_n; ++k) n += h->strict[(size_t)k];
            h->n_strict = n;
            h->n_tiles = n_tiles;
            memcpy(h->key, key, sizeof(key));
            h->valid = 1;
            h->on_device = 0;
        }
        if (!s.d_list) {
            BHR_HIP(hipMalloc((void **)&s.d_list, (size_t)ctx->tile_order_n * sizeof(int32_t)));
            BHR_HIP(hipHostMalloc((void **)&s.h_pinned, (size_t)ctx->tile_order_n * sizeof(int32_t), hipHostMallocDefault));
            BHR_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        }
        if (!s.valid || s.base_n != base_n || s.on_device != 0 || memcmp(s.key, h->key, sizeof(s.key)) != 0) {
            // the slot's stream is in order: the upload lands behind the slot's previous march; the pinned source is free
            // once its previous upload has completed
            if (s.pending) BHR_HIP(hipEventSynchronize(s.copied));
            int n = 0;                                   // stable partition: the launch order is kept inside both halves
            for (int k = 0; k < base_n; ++k)
                if (h->strict[(size_t)base_list[k]]) s.h_pinned[n++] = base_list[k];
            s.n_strict = n;
            for (int k = 0; k < base_n; ++k)
                if (!h->strict[(size_t)base_list[k]]) s.h_pinned[n++] = base_list[k];
            BHR_HIP(hipMemcpyAsync(s.d_list, s.h_pinned, (size_t)base_n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            BHR_HIP(hipEventRecord(s.copied, ctx->stream));
            s.pending = 1;
            memcpy(s.key, h->key, sizeof(s.key));
            s.base_n = base_n;
            s.valid = 1;
            s.on_device = 0;
            s.d_active = s.d_list;
        }
    }
    const uint32_t f = flags & ~(BHR_FORCE_FAST | BHR_FORCE_STRICT | BHR_FORCE_HYBRID);
    bhr_march_part p;
    p.h_list = nullptr;
    p.id = id;
    p.active = 1;
    p.math_resolved = 1;
    p.repair = 0;
    // Two launches.  On ONE stream the fast list waits for the last strict wave (the chip drains in between); on TWO the
    // fast tiles run on the context's low-priority second stream beside the strict ones and fill the slots they leave.
    // The bracket (start event, counter clear / end event) is an empty first / last part on the frame's own stream.
    // one stream where two frame slots keep frames in flight (the other frame's kernels fill this one's gaps, and every further
    // stream is one more place in HIP's queue lottery: DESIGN 7), two where a frame runs alone (row blocks, one slot)
    int streams = ctx->opt.hybrid_streams > 0 ? ctx->opt.hybrid_streams : (ctx->n_slots > 1 && ctx->cur_slot >= 0 ? 1 : 2);
    if (base.active) streams = 1;                    // a pipelined row block already runs its two halves on two streams
    if (s.n_strict == 0) streams = 1;                // nothing for a second stream to do (row blocks away from the hole's image)
    int32_t rc = BHR_OK;
    // The fast list's kernel carries guards: a lane that comes within a guard band of one of the algorithm's switches -- the
    // truncated mip level, a disk crossing in or next to the terminating step, a step that ends on the disk plane, the
    // disk's edges -- appends its pixel to a fix list instead of writing it, and a third launch marches the listed pixels
    // with the strict arithmetic (march.hip: march_tile_guard_kernel / march_fix_kernel).  ~0.1 % of the pixels.
    // Default: on for anti-aliased views -- the mip-level switch alone flips ~300 pixels of a 4k frame (5e-4 RMSE) -- and for
    // TILTED disks: the plane function z - y tan(tilt) of a point on the plane is then a difference of O(1) numbers, it
    // rounds to exactly 0 for ~1e-6 of the crossings, and the reference's `f_old f_new < 0` never registers those (12 pixels
    // of a 4k tilt-25 frame by a whole disk colour: 6.5e-4 RMSE).  With tilt 0 the function is z itself, which has full
    // relative precision at the plane: no exact zeros, and the third, dependent launch would cost 17 % of the fhd frame
    // rate (guard kernel +25 us, fix kernel 55 us: one strict wave's lifetime that nothing overlaps) -- off.
    // BHR_HYBRID_REPAIR=1 / 0 (environment of bhr_create) forces it on / off.
    const bool aa = ctx->cfg.anti_alias != 0 && !(flags & BHR_SKIP_DIFFERENTIALS);
    bool repair = aa || ctx->cfg.disk_tilt_deg != 0.0f;
    if (ctx->opt.hybrid_repair >= 0) repair = ctx->opt.hybrid_repair != 0;
    const int slot_k = ctx->active_slot >= 0 && ctx->active_slot < BHR_MAX_FRAME_SLOTS ? ctx->active_slot : 0;
    FixList &fx = h->fix[slot_k];
    if (repair && !fx.d_list) {
        const long long px = (long long)ctx->cfg.width * ctx->rows;
        // an eighth of the block's pixels (measured shares: 0.1-0.4 % on the BASELINE views, up to 3 % on fuzzed anti-aliased
        // ones with the 1e-2 level guard); the guard kernel counts past it and bhr_hybrid_repairs tells
        fx.cap = (int32_t)(px / 8 < 4096 ? 4096 : (px / 8 > (1 << 23) ? (1 << 23) : px / 8));
        fx.cap = (fx.cap + 255) / 256 * 256;
        BHR_HIP(hipMalloc((void **)&fx.d_count, 64));
        BHR_HIP(hipMalloc((void **)&fx.d_list, (size_t)fx.cap * sizeof(int32_t)));
    }
    h->last_fix_slot = repair ? slot_k : -1;
    ctx->fix_count = repair ? fx.d_count : nullptr;
    ctx->fix_list = repair ? fx.d_list : nullptr;
    ctx->fix_cap = repair ? fx.cap : 0;
    auto launch = [&](const int32_t *list, int n, int first, int last, int kind) -> int32_t {   // kind 0 strict list, 1 fast list, 2 fix list
        p.d_list = list; p.n = n; p.first = first; p.last = last;
        p.repair = kind == 0 ? 0 : (repair ? kind : 0);
        ctx->part = p;
        return (kind == 0 || kind == 2) ? bhr_launch_march_strict(ctx, cam, f) : bhr_launch_march(ctx, cam, f);
    };
    const int first0 = base.active ? base.first : 1, last0 = base.active ? base.last : 1;
    if (streams == 1) {
        // longest rays first: the strict tiles are the ones around the photon ring
        rc = launch(s.d_active, s.n_strict, first0, 0, 0);
        if (rc == BHR_OK && repair) rc = hipMemsetAsync(fx.d_count, 0, sizeof(unsigned int), ctx->stream) == hipSuccess ? BHR_OK : bhr_fail(BHR_ERR_HIP, "hipMemsetAsync failed");
        if (rc == BHR_OK) rc = launch(s.d_active + s.n_strict, base_n - s.n_strict, 0, repair ? 0 : last0, 1);
        if (rc == BHR_OK && repair) rc = launch(nullptr, 0, 0, last0, 2);
    } else {
        hipStream_t main_stream = ctx->stream;
        rc = launch(nullptr, 0, 1, 0, 0);                                          // prologue on the frame's stream
        if (rc == BHR_OK) rc = bhr_aux_fork(ctx);
        // which list rides the frame's own stream: the one that ends last, so that the post-pass follows it on the same
        // hardware queue (a wait on another queue's event costs ~10 us after that queue's kernel has ended; on a finished one,
        // nothing).  That is the fast list -- ten times the tiles of the strict one -- except under option "hybrid_swap" 0.
        const bool fast_on_main = ctx->opt.hybrid_swap != 0;
        if (fast_on_main) ctx->stream = ctx->aux_stream;
        if (rc == BHR_OK) rc = launch(s.d_active, s.n_strict, 0, 0, 0);
        if (rc == BHR_OK) {
            ctx->stream = fast_on_main ? main_stream : ctx->aux_stream;
            if (repair && hipMemsetAsync(fx.d_count, 0, sizeof(unsigned int), ctx->stream) != hipSuccess) rc = bhr_fail(BHR_ERR_HIP, "hipMemsetAsync failed");
            if (rc == BHR_OK) rc = launch(s.d_active + s.n_strict, base_n - s.n_strict, 0, 0, 1);
            if (rc == BHR_OK && repair) rc = launch(nullptr, 0, 0, 0, 2);
        }
        ctx->stream = main_stream;
        if (rc == BHR_OK) rc = bhr_aux_join(ctx);
        if (rc == BHR_OK) rc = launch(nullptr, 0, 0, 1, 0);                        // epilogue: the end event
    }
    ctx->part = base;
    ctx->fix_count = nullptr;
    ctx->fix_list = nullptr;
    ctx->fix_cap = 0;
    return rc;
}
