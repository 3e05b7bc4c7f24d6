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

struct SlotLists {
    int32_t *d_list;       // the base list partitioned: strict tiles first (launch order kept), then the fast ones
    int32_t *h_pinned;
    hipEvent_t copied;     // the last upload from h_pinned
    double key[12];
    int32_t n_strict, base_n, valid, pending;
};

struct Hybrid {
    SlotLists slot[BHR_MAX_FRAME_SLOTS][HYBRID_LISTS];
    FixList fix[BHR_MAX_FRAME_SLOTS];
    // last classification on the host
    std::vector<uint8_t> strict;   // per tile of the row block: marched strict
    double key[12];
    int32_t n_strict, valid;
    double lo, hi;                 // band below / above b_c (BHR_HYBRID_BAND="lo,hi")
    double eff_lo, eff_hi;         // ... as the last march used it (widened with the step size beyond 0.1)
    int32_t last_fix_slot;         // frame slot whose fix list the last march used, -1: it ran without guards
};

void view_key(const bhr_camera *cam, double lo, double hi, double tilt_deg, double key[12]) {
    double p[3], r2 = 0, pf = 0, pr = 0, pu = 0;
    for (int k = 0; k < 3; ++k) {
        p[k] = cam->pos[k];
        r2 += p[k] * p[k];
        pf += p[k] * (double)cam->forward[k];
        pr += p[k] * (double)cam->right[k];
        pu += p[k] * (double)cam->up[k];
    }
    key[0] = sqrt(r2); key[1] = pf; key[2] = pr; key[3] = pu;
    key[4] = cam->pixel_width; key[5] = cam->pixel_height; key[6] = lo; key[7] = hi;
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

// strict[t] = 1 for the tiles of this row block whose rays may have b in [b_c - lo, b_c + hi]
void classify(const bhr_ctx *ctx, const bhr_camera *cam, double lo, double hi, std::vector<uint8_t> &strict) {
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
            // maximum over the tile is at a corner, its minimum may lie on an edge -- pad by the tile's own span
            const double pad = (double)(bmax - bmin) + 1e-3;
            if (bmax + pad >= B_CRIT - lo && bmin - pad <= B_CRIT + hi) strict[(size_t)ty * tiles_x + tx] = 1;
        }
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
    delete h;
    ctx->hybrid = nullptr;
}

extern "C" int32_t bhr_hybrid_info(bhr_ctx *ctx, int32_t out_tiles[2], double out_band[2]) {
    if (!ctx || !out_tiles || !out_band) return bhr_fail(BHR_ERR_INVALID, "bhr_hybrid_info: bad argument");
    Hybrid *h = (Hybrid *)ctx->hybrid;
    if (!h || !h->valid) return bhr_fail(BHR_ERR_STATE, "bhr_hybrid_info: no hybrid march has run on this context");
    out_tiles[0] = h->n_strict;
    out_tiles[1] = (int32_t)h->strict.size();
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
        memset(h->fix, 0, sizeof(h->fix));
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
    view_key(cam, lo, hi, (double)ctx->cfg.disk_tilt_deg, key);
    if (!h->valid || !same_view(h->key, key)) {
        classify(ctx, cam, lo, hi, h->strict);
        int n = 0;
        for (int k = 0; k < ctx->tile_order_n; ++k) n += h->strict[(size_t)k];
        h->n_strict = n;
        memcpy(h->key, key, sizeof(key));
        h->valid = 1;
    }
    SlotLists &s = h->slot[ctx->active_slot >= 0 && ctx->active_slot < BHR_MAX_FRAME_SLOTS ? ctx->active_slot : 0][id];
    if (!s.d_list) {
        BHR_HIP(hipMalloc((void **)&s.d_list, (size_t)ctx->tile_order_n * sizeof(int32_t)));
        BHR_HIP(hipHostMalloc((void **)&s.h_pinned, (size_t)ctx->tile_order_n * sizeof(int32_t), hipHostMallocDefault));
        BHR_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    }
    if (!s.valid || s.base_n != base_n || memcmp(s.key, h->key, sizeof(s.key)) != 0) {
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
    int streams = ctx->opt.hybrid_streams;           // 2; BHR_HYBRID_STREAMS=1: both lists on the frame's stream
    if (base.active) streams = 1;                    // a pipelined row block already runs its two halves on two streams
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
        rc = launch(s.d_list, s.n_strict, first0, 0, 0);
        if (rc == BHR_OK && repair) rc = hipMemsetAsync(fx.d_count, 0, sizeof(unsigned int), ctx->stream) == hipSuccess ? BHR_OK : bhr_fail(BHR_ERR_HIP, "hipMemsetAsync failed");
        if (rc == BHR_OK) rc = launch(s.d_list + s.n_strict, base_n - s.n_strict, 0, repair ? 0 : last0, 1);
        if (rc == BHR_OK && repair) rc = launch(nullptr, 0, 0, last0, 2);
    } else {
        hipStream_t main_stream = ctx->stream;
        rc = launch(nullptr, 0, 1, 0, 0);                                          // prologue on the frame's stream
        if (rc == BHR_OK) rc = bhr_aux_fork(ctx);
        if (rc == BHR_OK) rc = launch(s.d_list, s.n_strict, 0, 0, 0);
        if (rc == BHR_OK) {
            ctx->stream = ctx->aux_stream;
            if (repair && hipMemsetAsync(fx.d_count, 0, sizeof(unsigned int), ctx->stream) != hipSuccess) rc = bhr_fail(BHR_ERR_HIP, "hipMemsetAsync failed");
            if (rc == BHR_OK) rc = launch(s.d_list + s.n_strict, base_n - s.n_strict, 0, 0, 1);
            if (rc == BHR_OK && repair) rc = launch(nullptr, 0, 0, 0, 2);
            ctx->stream = main_stream;
        }
        if (rc == BHR_OK) rc = bhr_aux_join(ctx);
        if (rc == BHR_OK) rc = launch(nullptr, 0, 0, 1, 0);                        // epilogue: the end event
    }
    ctx->part = base;
    ctx->fix_count = nullptr;
    ctx->fix_list = nullptr;
    ctx->fix_cap = 0;
    return rc;
}
