// api.hip -- the extern "C" surface declared in include/bhr.h.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "bhr_internal.h"

namespace {

thread_local char g_err[512] = "";

template <typename T>
int32_t dev_alloc(T **p, size_t count) {
    *p = nullptr;
    if (count == 0) return BHR_OK;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return bhr_fail(BHR_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    return BHR_OK;
}

int32_t use_device(bhr_ctx *ctx) { return bhr_enter(ctx); }

// points the launchers' view of the frame buffers (ctx->d_bg ...) at slot k
void activate_slot(bhr_ctx *ctx, int k) {
    const bhr_frame_slot &f = ctx->slots[k];
    ctx->d_bg = f.d_bg;
    ctx->d_disk = f.d_disk;
    ctx->d_hblur = f.d_hblur;
    ctx->d_pa = f.d_pa;
    ctx->d_pb = f.d_pb;
    ctx->d_sum = f.d_sum;
    ctx->d_blur = f.d_blur;
    ctx->d_final = f.d_final;
    ctx->d_final_u8 = f.d_final_u8;
    ctx->d_queue = f.d_queue;
    ctx->d_glow_hw = f.d_glow_hw;          // allocated on first use by flare.hip, which stores them back into the slot
    ctx->d_glow_wh = f.d_glow_wh;
    ctx->d_flare_c0 = f.d_flare_c0;
    ctx->d_flare_c12 = f.d_flare_c12;
    ctx->d_flare_sums = f.d_flare_sums;
    ctx->flare_glow_rows = f.flare_glow_rows;
    ctx->active_slot = k;
}

// experiment (BHR_STREAM_PAD="a,b,c"): idle streams created in front of frame slot 0's, slot 1's and the second march
// streams -- HIP hands streams to its hardware queues in creation order, and which queues the frame slots land on decides
// how their launches interleave
// ---- do two streams sit on ONE hardware queue? ------------------------------------------------------------------------------
// HIP hands its streams to a small pool of hardware queues (four per priority by default) by rules of its own; two streams
// on one queue run their kernels strictly one after the other, two on different queues side by side -- which decides how the
// two frame slots' launches interleave.  The probe: a one-lane kernel on `a` that spins until released (or 4 ms
// of the 100 MHz real-time counter), a one-lane kernel on `b` that raises a flag in pinned memory.  The flag rises while the
// spinner is still held -> different queues.
__global__ void probe_spin_kernel(volatile int *release, volatile int *spinning) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    *spinning = 1;
    __threadfence_system();
    const unsigned long long r0 = wall_clock64();
    while (*release == 0 && wall_clock64() - r0 < 400000ull) __builtin_amdgcn_s_sleep(8);
    (void)t0;
}
__global__ void probe_flag_kernel(volatile int *flag) { *flag = 1; __threadfence_system(); }

}  // namespace
int32_t bhr_streams_share_queue(hipStream_t a, hipStream_t b, int32_t *share) {
    if (a == b) { *share = 1; return BHR_OK; }
    volatile int *h = nullptr;
    BHR_HIP(hipHostMalloc((void **)&h, 3 * sizeof(int), hipHostMallocDefault));
    h[0] = h[1] = h[2] = 0;                               // release, spinning, flag
    hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(1), 0, a, h + 0, h + 1);
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    while (h[1] == 0 && ms() < 3.0) {}                    // the spinner is on the chip
    hipLaunchKernelGGL(probe_flag_kernel, dim3(1), dim3(1), 0, b, h + 2);
    const double t1 = ms();
    while (h[2] == 0 && ms() - t1 < 1.0) {}               // a kernel on a free queue lands in ~10 us
    *share = h[2] == 0;
    h[0] = 1;
    const hipError_t e1 = hipStreamSynchronize(a), e2 = hipStreamSynchronize(b);
    (void)hipHostFree((void *)h);
    BHR_HIP(e1);
    BHR_HIP(e2);
    return BHR_OK;
}
namespace {

static void pad_streams(const bhr_ctx *ctx, int which) {
    for (int k = 0; k < ctx->opt.stream_pad[which] && k < 8; ++k) { hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }   // leaked on purpose
}

// the environment, once (bhr_create): nothing on the bhr_render path calls getenv
static void read_options(bhr_options *o) {
    memset(o, 0, sizeof(*o));
    auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e && e[0] ? atoi(e) : dflt; };
    o->frame_slots = num("BHR_FRAME_SLOTS", 2);
    if (o->frame_slots < 1 || o->frame_slots > BHR_MAX_FRAME_SLOTS) o->frame_slots = 2;
    o->bloom_split = num("BHR_BLOOM_SPLIT", -1);
    if (o->bloom_split > 1) o->bloom_split = 1;
    o->bloom_tiles = num("BHR_BLOOM_TILES", 0);
    if (o->bloom_tiles < 0 || o->bloom_tiles > 8) o->bloom_tiles = 0;
    o->hybrid_repair = num("BHR_HYBRID_REPAIR", -1);
    if (o->hybrid_repair > 1) o->hybrid_repair = 1;
    o->hybrid_band_set = 0;
    if (const char *e = getenv("BHR_HYBRID_BAND")) {
        double lo = 0, hi = 0;
        if (sscanf(e, "%lf,%lf", &lo, &hi) == 2 && lo >= 0 && hi >= 0) { o->hybrid_band[0] = lo; o->hybrid_band[1] = hi; o->hybrid_band_set = 1; }
    }
    // share of its own span of b a SMALL tile is padded by in the strict-band test (hybrid.hip: tile_pad; tools/exp_hybrid_pad.py)
    o->hybrid_pad = 0.5;
    if (const char *e = getenv("BHR_HYBRID_PAD")) { const double v = atof(e); if (v >= 0.0 && v <= 4.0) o->hybrid_pad = v; }
    o->hybrid_streams = num("BHR_HYBRID_STREAMS", -1);
    if (o->hybrid_streams != 1 && o->hybrid_streams != 2) o->hybrid_streams = -1;
    o->calibrate_streams = num("BHR_CALIBRATE_STREAMS", 1) != 0;
    o->hybrid_classify = num("BHR_HYBRID_CLASSIFY", 1) != 0;
    o->hybrid_swap = num("BHR_HYBRID_SWAP", 1) != 0;
    o->mip_lds = num("BHR_MIP_LDS", 0) != 0;
    { const char *e = getenv("BHR_TILE_ORDER"); o->tile_order_rows = e && e[0] == 'r'; }
    o->tile_block = num("BHR_TILE_BLOCK", 256);
    if (o->tile_block != 64 && o->tile_block != 128 && o->tile_block != 256) o->tile_block = 256;
    o->group_threads = num("BHR_GROUP_THREADS", -1);
    { const char *e = getenv("BHR_GROUP_SCHEDULE"); o->group_schedule = e && e[0] ? (e[0] == 's' ? 0 : 1) : -1; }
    o->aux_priority = 0;
    o->aux_per_slot = 1;
    if (const char *e = getenv("BHR_AUX_STREAMS")) (void)sscanf(e, "%d,%d", &o->aux_priority, &o->aux_per_slot);
    if (const char *e = getenv("BHR_STREAM_PAD")) (void)sscanf(e, "%d,%d,%d", &o->stream_pad[0], &o->stream_pad[1], &o->stream_pad[2]);
}

int32_t alloc_slot(bhr_ctx *ctx, int k) {
    bhr_frame_slot &f = ctx->slots[k];
    if (f.allocated) return BHR_OK;
    const size_t W = ctx->cfg.width, rows = ctx->rows, R = ctx->bloom_R, px3 = rows * W * 3;
    if (!f.stream) {
        if (k == 0 && ctx->n_slots == 1) f.stream = ctx->scene_stream;
        else { pad_streams(ctx, k == 0 ? 0 : 1); BHR_HIP(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking)); }
    }
    if (!f.done) BHR_HIP(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    int32_t rc = BHR_OK;
    (void)R;
    if ((rc = dev_alloc(&f.d_bg, px3)) || (rc = dev_alloc(&f.d_disk, px3)) || (rc = dev_alloc(&f.d_blur, px3)) ||
        (rc = dev_alloc(&f.d_final, px3)) || (rc = dev_alloc(&f.d_final_u8, px3)) || (rc = dev_alloc(&f.d_queue, 1))) {
        void *bufs[] = {f.d_bg, f.d_disk, f.d_blur, f.d_final, f.d_final_u8, f.d_queue};   // a later retry starts clean
        for (void *b : bufs)
            if (b) (void)hipFree(b);
        f.d_bg = f.d_disk = f.d_blur = f.d_final = nullptr;
        f.d_final_u8 = nullptr;
        f.d_queue = nullptr;
        return rc;
    }
    f.allocated = 1;
    return BHR_OK;
}

// The bloom intermediates of slot k, on first use of either post-pass: the planar f32 H-blur planes of the exact kernels, or
// the packed f16 operands of the split kernels (bloom.hip).  Zero filled: halo rows outside the image stay zero for the
// life of the context, and so do the padding groups of the packed layouts.
int32_t ensure_bloom_buffers(bhr_ctx *ctx, int k, bool split) {
    bhr_frame_slot &f = ctx->slots[k];
    const size_t W = ctx->cfg.width, rows = ctx->rows, R = ctx->bloom_R;
    if (split) {
        if (f.d_pa && f.d_pb) return BHR_OK;
        bhr_split_geom g;
        bhr_split_geometry(ctx, &g);
        BHR_HIP(hipMalloc(&f.d_pa, g.pa_halfs * 2));
        BHR_HIP(hipMalloc(&f.d_pb, g.pb_halfs * 2));
        BHR_TRY(dev_alloc(&f.d_sum, rows * W * 3));
        BHR_HIP(hipMemsetAsync(f.d_pa, 0, g.pa_halfs * 2, ctx->scene_stream));
        BHR_HIP(hipMemsetAsync(f.d_pb, 0, g.pb_halfs * 2, ctx->scene_stream));
    } else {
        if (f.d_hblur_base) return BHR_OK;
        const size_t n = 3 * (rows + 2 * R) * W + 2 * BHR_HBLUR_PAD_ROWS * W;
        BHR_TRY(dev_alloc(&f.d_hblur_base, n));
        f.d_hblur = f.d_hblur_base + BHR_HBLUR_PAD_ROWS * W;
        BHR_HIP(hipMemsetAsync(f.d_hblur_base, 0, n * sizeof(float), ctx->scene_stream));
    }
    BHR_HIP(hipStreamSynchronize(ctx->scene_stream));
    return BHR_OK;
}

void free_slot(bhr_ctx *ctx, int k) {
    bhr_frame_slot &f = ctx->slots[k];
    void *bufs[] = {f.d_bg, f.d_disk, f.d_blur, f.d_final, f.d_final_u8, f.d_hblur_base, f.d_pa, f.d_pb, f.d_sum, f.d_queue,
                    f.d_glow_hw, f.d_glow_wh, f.d_flare_c0, f.d_flare_c12, f.d_flare_sums};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (f.done) (void)hipEventDestroy(f.done);
    if (f.stream && f.stream != ctx->scene_stream) (void)hipStreamDestroy(f.stream);
    memset(&f, 0, sizeof(f));
}

int32_t ensure_pinned(bhr_ctx *ctx, size_t bytes) { return bhr_ensure_pinned(ctx, bytes); }
}  // namespace
int32_t bhr_ensure_pinned(bhr_ctx *ctx, size_t bytes) {
    if (ctx->h_pinned_bytes >= bytes) return BHR_OK;
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    ctx->h_pinned = nullptr;
    ctx->h_pinned_bytes = 0;
    BHR_HIP(hipHostMalloc((void **)&ctx->h_pinned, bytes, hipHostMallocDefault));
    ctx->h_pinned_bytes = bytes;
    return BHR_OK;
}
namespace {

// device -> caller memory through the pinned staging buffer (synchronises the stream)
int32_t download(bhr_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
    BHR_TRY(ensure_pinned(ctx, bytes));
    BHR_HIP(hipMemcpyAsync(ctx->h_pinned, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->h_pinned, bytes);
    return BHR_OK;
}

int32_t upload(bhr_ctx *ctx, void *d_dst, const void *src, size_t bytes) {
    // pageable source: hipMemcpyAsync stages internally and is ordered on the stream
    BHR_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    return BHR_OK;
}

__global__ void quantize_u8_kernel(const float *__restrict__ src, uint8_t *__restrict__ dst, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // save_image: (np.clip(x, 0, 1) * 255).astype(np.uint8) -- truncation (render.py:423)
    for (; i < n; i += stride) dst[i] = (uint8_t)(int)(fminf(fmaxf(src[i], 0.0f), 1.0f) * 255.0f);
}

// sum of the lanes of `n` counter cells -> host
__global__ void steps_fold_kernel(const unsigned long long *__restrict__ cells, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long sh[BHR_STEP_LANES];
    sh[threadIdx.x] = cells[(size_t)blockIdx.x * BHR_STEP_CELL + (size_t)threadIdx.x * BHR_STEP_STRIDE];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < BHR_STEP_LANES; ++k) t += sh[k];
        out[blockIdx.x] = t;
    }
}
int32_t fold_cells(bhr_ctx *ctx, const unsigned long long *cells, int n, unsigned long long *host_out) {
    hipLaunchKernelGGL(steps_fold_kernel, dim3(n), dim3(BHR_STEP_LANES), 0, ctx->stream, cells, ctx->d_steps_fold);
    BHR_HIP(hipGetLastError());
    BHR_HIP(hipMemcpyAsync(host_out, ctx->d_steps_fold, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    return BHR_OK;
}

}  // namespace

// the frame's u8 rows in d_final_u8, on the context's stream: already there when its V pass stored them, else FINAL -> u8
int32_t bhr_launch_quantize(bhr_ctx *ctx) { return bhr_ensure_outputs(ctx, BHR_OUT_U8); }

int32_t bhr_ensure_outputs(bhr_ctx *ctx, uint32_t need) {
    bhr_frame_slot &f = ctx->slots[ctx->active_slot];
    uint32_t missing = need & ~f.have;
    if (!missing) return BHR_OK;
    if ((missing & BHR_OUT_U8) && ((f.have | missing) & BHR_OUT_F32)) {
        // the f32 frame is (or is about to be) the authority -- it may carry a lens flare the V pass knows nothing of
        if (missing & BHR_OUT_F32) BHR_TRY(bhr_ensure_outputs(ctx, BHR_OUT_F32));
        const long long n = (long long)ctx->rows * ctx->cfg.width * 3;
        hipLaunchKernelGGL(quantize_u8_kernel, dim3(2048), dim3(256), 0, ctx->stream, ctx->d_final, ctx->d_final_u8, n);
        BHR_HIP(hipGetLastError());
        f.have |= BHR_OUT_U8;
        missing &= ~(BHR_OUT_U8 | BHR_OUT_F32);
        if (!missing) return BHR_OK;
    }
    // re-run the frame's V pass for what nobody asked for up front (its inputs -- bg, disk, the H-blur planes -- are still the
    // slot's): same kernels, same bits
    const int32_t split = ctx->bloom_split;
    ctx->bloom_split = f.frame_split;
    const int32_t rc = bhr_launch_bloom_v_rows(ctx, f.frame_with_bloom, 0, ctx->rows, missing, nullptr, nullptr);
    ctx->bloom_split = split;
    BHR_TRY(rc);
    f.have |= missing;
    return BHR_OK;
}

// group / tile renders and the stand-alone passes work on slot 0 whatever bhr_render left active
int32_t bhr_activate_slot(bhr_ctx *ctx, int32_t k) {
    if (k < 0 || k >= BHR_MAX_FRAME_SLOTS) return bhr_fail(BHR_ERR_INVALID, "frame slot %d", k);
    BHR_TRY(alloc_slot(ctx, k));
    activate_slot(ctx, k);
    return BHR_OK;
}

int32_t bhr_frame_begin(bhr_ctx *ctx, uint32_t flags) {
    int mode = ctx->cfg.math_mode;
    if (flags & BHR_FORCE_FAST) mode = BHR_MATH_FAST;
    if (flags & BHR_FORCE_STRICT) mode = BHR_MATH_STRICT;
    if (flags & BHR_FORCE_HYBRID) mode = BHR_MATH_HYBRID;
    if (mode == BHR_MATH_HYBRID && (ctx->disk_source != BHR_DISK_TEXTURE || (flags & BHR_PERSISTENT))) mode = BHR_MATH_STRICT;
    // the frame's post-pass follows its march: exact f32 chains under strict, the split-f16 matrix-core kernels (bloom.hip)
    // under fast and hybrid; BHR_BLOOM_SPLIT=0 / 1 forces either for every arithmetic
    int split = ctx->opt.bloom_split >= 0 ? ctx->opt.bloom_split : (mode != BHR_MATH_STRICT ? 1 : 0);
    if (!ctx->split_ok || (flags & BHR_SKIP_BLOOM)) split = 0;
    ctx->bloom_split = split;
    if (!(flags & BHR_SKIP_BLOOM)) {
        BHR_TRY(ensure_bloom_buffers(ctx, ctx->active_slot, split != 0));
        bhr_frame_slot &f = ctx->slots[ctx->active_slot];
        ctx->d_hblur = f.d_hblur;
        ctx->d_pa = f.d_pa;
        ctx->d_pb = f.d_pb;
        ctx->d_sum = f.d_sum;
    }
    bhr_frame_slot &f = ctx->slots[ctx->active_slot];
    f.have = 0;
    f.sum_valid = 0;
    f.frame_split = split;
    f.frame_with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    return BHR_OK;
}

int32_t bhr_frame_post(bhr_ctx *ctx, int32_t with_bloom, uint32_t want) {
    BHR_TRY(bhr_launch_bloom_v_rows(ctx, with_bloom, 0, ctx->rows, want, nullptr, nullptr));
    ctx->slots[ctx->active_slot].have = want;
    return BHR_OK;
}

namespace {

void free_scene(bhr_ctx *ctx) {
    if (ctx->d_mips) (void)hipFree(ctx->d_mips);
    ctx->d_mips = nullptr;
}

void free_bg(bhr_ctx *ctx) {
    if (ctx->d_comp) (void)hipFree(ctx->d_comp);
    if (ctx->d_edge) (void)hipFree(ctx->d_edge);
    if (ctx->d_omega) (void)hipFree(ctx->d_omega);
    if (ctx->d_row_stats) (void)hipFree(ctx->d_row_stats);
    ctx->d_comp = ctx->d_edge = ctx->d_omega = ctx->d_row_stats = nullptr;
    ctx->bg_ready = 0;
}

// Allocates the packed mip stack for an (n_r, n_phi) texture.
int32_t alloc_mips(bhr_ctx *ctx, int32_t n_r, int32_t n_phi) {
    int64_t off = 0;
    int32_t h = n_r, w = n_phi;
    for (int l = 0; l < BHR_NUM_MIP_LEVELS; ++l) {
        ctx->mip_off[l] = (int32_t)off;
        ctx->mip_h[l] = h;
        ctx->mip_w[l] = w;
        off += (int64_t)h * w;
        // generate_disk_mipmaps stops when h < 2 or w < 2 (render.py:1118-1119)
        if (h < 2 || w < 2) { h = 0; w = 0; } else { h /= 2; w /= 2; }
    }
    if (off >= (1ll << 31)) return bhr_fail(BHR_ERR_INVALID, "disk texture too large (%d x %d)", n_r, n_phi);
    ctx->mip_texels = off;
    ctx->n_r = n_r;
    ctx->n_phi = n_phi;
    BHR_TRY(dev_alloc(&ctx->d_mips, (size_t)off));
    BHR_HIP(hipMemsetAsync(ctx->d_mips, 0, (size_t)off * sizeof(float4), ctx->stream));
    return BHR_OK;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0f;
    return ms;
}

}  // namespace

int32_t bhr_fail(int32_t code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int32_t bhr_aux_fork(bhr_ctx *ctx) {
    const int k = ctx->active_slot >= 0 && ctx->active_slot < BHR_MAX_FRAME_SLOTS ? ctx->active_slot : 0;
    if (!ctx->aux_streams[0]) {
        int lo = 0, hi = 0;
        BHR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));       // lo = least priority
        // BHR_AUX_STREAMS="<priority>,<per slot>" (experiments of DESIGN 7): -1 least / 0 normal / 1 highest; 0 one stream
        // for both frame slots / 1 one each
        const int prio_sel = ctx->opt.aux_priority, per_slot = ctx->opt.aux_per_slot;
        ctx->aux_per_slot = per_slot != 0;
        const int prio = prio_sel == 0 ? 0 : (prio_sel > 0 ? hi : lo);
        pad_streams(ctx, 2);
        for (int q = 0; q < BHR_MAX_FRAME_SLOTS; ++q) {
            BHR_HIP(hipStreamCreateWithPriority(&ctx->aux_streams[q], hipStreamNonBlocking, prio));
            BHR_HIP(hipEventCreateWithFlags(&ctx->aux_fork[q], hipEventDisableTiming));
            BHR_HIP(hipEventCreateWithFlags(&ctx->aux_done[q], hipEventDisableTiming));
        }
    }
    ctx->aux_stream = ctx->aux_streams[ctx->aux_per_slot ? k : 0];
    BHR_HIP(hipEventRecord(ctx->aux_fork[k], ctx->stream));
    BHR_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork[k], 0));
    return BHR_OK;
}

int32_t bhr_aux_join(bhr_ctx *ctx) {
    const int k = ctx->active_slot >= 0 && ctx->active_slot < BHR_MAX_FRAME_SLOTS ? ctx->active_slot : 0;
    BHR_HIP(hipEventRecord(ctx->aux_done[k], ctx->aux_stream));
    BHR_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_done[k], 0));
    return BHR_OK;
}

int32_t bhr_enter(bhr_ctx *ctx) {
    BHR_HIP(hipSetDevice(ctx->cfg.device));
    for (int k = 0; k < BHR_MAX_FRAME_SLOTS; ++k) {
        bhr_frame_slot &f = ctx->slots[k];
        if (f.in_flight && f.stream != ctx->scene_stream) BHR_HIP(hipStreamWaitEvent(ctx->scene_stream, f.done, 0));
        f.in_flight = 0;
    }
    ctx->stream = ctx->scene_stream;
    return BHR_OK;
}

int32_t bhr_enter_components(bhr_ctx *ctx) {
    BHR_HIP(hipSetDevice(ctx->cfg.device));
    ctx->stream = ctx->scene_stream;
    return BHR_OK;
}

int32_t bhr_enter_scene_write(bhr_ctx *ctx) {
    BHR_HIP(hipSetDevice(ctx->cfg.device));
    for (int k = 0; k < BHR_MAX_FRAME_SLOTS; ++k) {
        bhr_frame_slot &f = ctx->slots[k];
        if (f.in_flight && f.stream != ctx->scene_stream && f.march_done) BHR_HIP(hipStreamWaitEvent(ctx->scene_stream, f.march_done, 0));
    }
    ctx->stream = ctx->scene_stream;
    return BHR_OK;
}

int32_t bhr_enter_frame(bhr_ctx *ctx) {
    bhr_frame_slot &f = ctx->slots[ctx->active_slot];
    if (!f.allocated || !f.stream || f.stream == ctx->scene_stream) return bhr_enter(ctx);
    BHR_HIP(hipSetDevice(ctx->cfg.device));
    BHR_HIP(hipEventRecord(ctx->scene_ev, ctx->scene_stream));      // e.g. a bhr_write_layer(FINAL) since the render
    BHR_HIP(hipStreamWaitEvent(f.stream, ctx->scene_ev, 0));
    ctx->stream = f.stream;
    return BHR_OK;
}

int32_t bhr_leave_frame(bhr_ctx *ctx) {
    bhr_frame_slot &f = ctx->slots[ctx->active_slot];
    if (ctx->stream != ctx->scene_stream && ctx->stream == f.stream) {
        BHR_HIP(hipEventRecord(f.done, f.stream));                  // joins now wait for this work too
        f.in_flight = 1;
    }
    ctx->stream = ctx->scene_stream;
    return BHR_OK;
}

extern "C" {

const char *bhr_last_error(void) { return g_err; }
int32_t bhr_abi_version(void) { return BHR_ABI_VERSION; }

int32_t bhr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t bhr_create(const bhr_config *cfg, bhr_ctx **out) {
    if (!cfg || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_create: null argument");
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_create: bad image size %dx%d", cfg->width, cfg->height);
    if (cfg->row0 < 0 || cfg->row1 > cfg->height || cfg->row0 >= cfg->row1)
        return bhr_fail(BHR_ERR_INVALID, "bhr_create: bad row block [%d,%d) for height %d", cfg->row0, cfg->row1, cfg->height);
    if (!(cfg->step_size > 0.0f)) return bhr_fail(BHR_ERR_INVALID, "bhr_create: step_size must be positive");
    if (!(cfg->r_disk_inner < cfg->r_disk_outer)) return bhr_fail(BHR_ERR_INVALID, "bhr_create: r_disk_inner must be < r_disk_outer");
    int ndev = bhr_device_count();
    if (ndev <= 0) return bhr_fail(BHR_ERR_NO_DEVICE, "bhr_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return bhr_fail(BHR_ERR_NO_DEVICE, "bhr_create: device %d out of range (have %d)", cfg->device, ndev);

    bhr_ctx *ctx = new bhr_ctx();
    memset(ctx, 0, sizeof(*ctx));
    ctx->cfg = *cfg;
    ctx->rows = cfg->row1 - cfg->row0;
    ctx->bloom_R = (int32_t)(cfg->width * 0.02);  // int(self.width * 0.02), render.py:3914
    ctx->mip_lds_from = -1;

    auto bail = [&](int32_t rc) { bhr_destroy(ctx); return rc; };
    if (hipSetDevice(cfg->device) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "hipSetDevice(%d) failed", cfg->device));
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "hipStreamCreate failed"));
    ctx->scene_stream = ctx->stream;
    read_options(&ctx->opt);
    ctx->n_slots = ctx->opt.frame_slots;                 // 2 (default): frames alternate between two slots / streams
    ctx->split_ok = bhr_split_nt(ctx->bloom_R) <= 12;    // the split-f16 bloom's table: radius <= 176 (widths to 8849)
    ctx->out_want = BHR_OUT_F32;
    if (hipEventCreateWithFlags(&ctx->scene_ev, hipEventDisableTiming) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "hipEventCreate failed"));
    if (hipEventCreateWithFlags(&ctx->sync_ev, hipEventDisableTiming) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "hipEventCreate failed"));
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "hipEventCreate failed"));
    for (auto &e : ctx->ring_ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "hipEventCreate failed"));

    const size_t W = cfg->width, H = cfg->height, rows = ctx->rows, R = ctx->bloom_R;
    const size_t px3 = rows * W * 3;
    int32_t rc;
    (void)px3;
    if ((rc = alloc_slot(ctx, 0))) return bail(rc);
    activate_slot(ctx, 0);
    if ((rc = dev_alloc(&ctx->d_wtab, 3 * (R + 1 + 64)))) return bail(rc);
    if ((rc = dev_alloc(&ctx->d_wext, 3 * (2 * (R + 4) + 8)))) return bail(rc);
    if ((rc = dev_alloc(&ctx->d_wsum_h, 6 * W))) return bail(rc);
    if ((rc = dev_alloc(&ctx->d_wsum_v, 6 * H))) return bail(rc);
    if ((rc = dev_alloc(&ctx->d_ray_steps, BHR_STEP_CELL))) return bail(rc);
    if ((rc = dev_alloc(&ctx->d_steps_ring, (size_t)BHR_TIMING_RING * BHR_STEP_CELL))) return bail(rc);
    if ((rc = dev_alloc(&ctx->d_steps_fold, BHR_TIMING_RING))) return bail(rc);
    if (hipMemsetAsync(ctx->d_ray_steps, 0, sizeof(unsigned long long) * BHR_STEP_CELL, ctx->stream) != hipSuccess ||
        hipMemsetAsync(ctx->d_steps_ring, 0, sizeof(unsigned long long) * BHR_TIMING_RING * BHR_STEP_CELL, ctx->stream) != hipSuccess)
        return bail(bhr_fail(BHR_ERR_HIP, "hipMemsetAsync failed"));
    int32_t v = 0, l = 0;
    if (cfg->math_mode != BHR_MATH_FAST && cfg->math_mode != BHR_MATH_STRICT && cfg->math_mode != BHR_MATH_HYBRID)
        return bail(bhr_fail(BHR_ERR_INVALID, "bhr_create: math_mode %d", cfg->math_mode));
    if ((cfg->math_mode != BHR_MATH_FAST ? (cfg->anti_alias != 0 ? bhr_march_resources_strict_ilp(&v, &l, 1)
                                                                   : bhr_march_resources_strict_ilp(&v, &l, 0))
                                           : bhr_march_resources(&v, &l, cfg->anti_alias != 0)) == BHR_OK) {
        ctx->counters.march_vgprs = v;
        ctx->counters.march_lds_bytes = l;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(bhr_fail(BHR_ERR_HIP, "stream sync failed"));
    *out = ctx;
    return BHR_OK;
}

void bhr_destroy(bhr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    for (auto &f : ctx->slots)
        if (f.stream) (void)hipStreamSynchronize(f.stream);
    if (ctx->scene_stream) (void)hipStreamSynchronize(ctx->scene_stream);
    ctx->stream = ctx->scene_stream;
    free_scene(ctx);
    free_bg(ctx);
    for (int k = 0; k < BHR_MAX_FRAME_SLOTS; ++k) free_slot(ctx, k);
    bhr_png_dev_free(ctx);
    bhr_population_free(ctx);
    bhr_hybrid_free(ctx);
    bhr_pipe_free(ctx);
    for (int q = 0; q < BHR_MAX_FRAME_SLOTS; ++q)
        if (ctx->aux_streams[q]) { (void)hipStreamSynchronize(ctx->aux_streams[q]); (void)hipStreamDestroy(ctx->aux_streams[q]); }
    for (int q = 0; q < ctx->n_calib_idle; ++q) (void)hipStreamDestroy(ctx->calib_idle[q]);
    ctx->n_calib_idle = 0;
    for (int q = 0; q < BHR_MAX_FRAME_SLOTS; ++q) {
        if (ctx->aux_fork[q]) (void)hipEventDestroy(ctx->aux_fork[q]);
        if (ctx->aux_done[q]) (void)hipEventDestroy(ctx->aux_done[q]);
    }
    if (ctx->d_gather_u8) (void)hipFree(ctx->d_gather_u8);
    free(ctx->h_tile_order);
    if (ctx->scene_ev) (void)hipEventDestroy(ctx->scene_ev);
    if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
    void *bufs[] = {ctx->d_skybox,
                    ctx->d_wtab, ctx->d_wsum_h, ctx->d_wsum_v, ctx->d_ray_steps, ctx->d_noise_in,
                    ctx->d_noise_out, ctx->d_steps_ring, ctx->d_steps_fold, ctx->d_pool, ctx->d_pairs, ctx->d_stats_scratch, ctx->d_wext, ctx->d_w16, ctx->d_dv2_params,
                    ctx->d_flare_prog, ctx->d_tile_order, ctx->d_row_steps, ctx->d_gather};   // the flare's per-frame scratch belongs to the slots
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->ring_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Waits by POLLING an event for the first 200 ms, then blocks.  hipStreamSynchronize sleeps on an interrupt, and the wake-up is
// usually 0.1 ms after the GPU is done but every so often 2 ms and more (bench line of a 20-step run: the 20 frames' own event
// span 6.6 ms in both of two runs, the host back after 6.7 and after 8.7 ms -- a quarter of the frame rate of a run that short;
// profiles/r04e_sync_wakeup.txt).  A frame loop that waits once per batch can afford a core for its wait.
static int32_t poll_sync(bhr_ctx *ctx, hipStream_t stream) {
    BHR_HIP(hipEventRecord(ctx->sync_ev, stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ctx->sync_ev);
        if (e == hipSuccess) return BHR_OK;
        if (e != hipErrorNotReady) BHR_HIP(e);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
    }
    BHR_HIP(hipStreamSynchronize(stream));
    return BHR_OK;
}

int32_t bhr_sync(bhr_ctx *ctx) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "null ctx");
    BHR_TRY(use_device(ctx));
    return poll_sync(ctx, ctx->stream);
}

int32_t bhr_set_skybox(bhr_ctx *ctx, const float *rgb, int32_t tex_h, int32_t tex_w) {
    if (!ctx || !rgb || tex_h <= 0 || tex_w <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_set_skybox: bad argument");
    BHR_TRY(use_device(ctx));
    if (ctx->d_skybox && (ctx->sky_h != tex_h || ctx->sky_w != tex_w)) {
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_skybox);
        ctx->d_skybox = nullptr;
    }
    if (!ctx->d_skybox) BHR_TRY(dev_alloc(&ctx->d_skybox, (size_t)tex_h * tex_w * 3));
    ctx->sky_h = tex_h;
    ctx->sky_w = tex_w;
    return upload(ctx, ctx->d_skybox, rgb, (size_t)tex_h * tex_w * 3 * sizeof(float));
}

int32_t bhr_set_disk_texture(bhr_ctx *ctx, const float *rgba, int32_t n_r, int32_t n_phi) {
    if (!ctx || !rgba || n_r <= 0 || n_phi <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_set_disk_texture: bad argument");
    BHR_TRY(use_device(ctx));
    if (ctx->d_mips && (ctx->n_r != n_r || ctx->n_phi != n_phi))
        return bhr_fail(BHR_ERR_INVALID, "Texture size mismatch: expected %dx%d, got %dx%d", ctx->n_r, ctx->n_phi, n_r, n_phi);
    if (!ctx->d_mips) BHR_TRY(alloc_mips(ctx, n_r, n_phi));
    BHR_TRY(upload(ctx, ctx->d_mips, rgba, (size_t)n_r * n_phi * sizeof(float4)));
    return bhr_launch_build_mips(ctx);
}

int32_t bhr_get_disk_texture(bhr_ctx *ctx, float *rgba_out) { return bhr_get_disk_mip(ctx, 0, rgba_out); }

int32_t bhr_get_disk_mip(bhr_ctx *ctx, int32_t level, float *rgba_out) {
    if (!ctx || !rgba_out) return bhr_fail(BHR_ERR_INVALID, "bhr_get_disk_mip: bad argument");
    if (!ctx->d_mips) return bhr_fail(BHR_ERR_STATE, "bhr_get_disk_mip: no disk texture");
    if (level < 0 || level >= BHR_NUM_MIP_LEVELS) return bhr_fail(BHR_ERR_INVALID, "bhr_get_disk_mip: level %d", level);
    BHR_TRY(use_device(ctx));
    size_t n = (size_t)ctx->mip_h[level] * ctx->mip_w[level];
    if (n == 0) return BHR_OK;
    return download(ctx, rgba_out, ctx->d_mips + ctx->mip_off[level], n * sizeof(float4));
}

int32_t bhr_num_mip_levels(bhr_ctx *ctx) {
    if (!ctx || !ctx->d_mips) return 0;
    int n = 0;
    for (int l = 0; l < BHR_NUM_MIP_LEVELS; ++l)
        if (ctx->mip_h[l] > 0 && ctx->mip_w[l] > 0) ++n;
    return n;
}

int32_t bhr_bg_init(bhr_ctx *ctx, int32_t n_r, int32_t n_phi, int32_t az_freq, float az_shear, const float *edge,
                    const float *omega_rows) {
    if (!ctx || !edge || !omega_rows || n_r <= 0 || n_phi <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_bg_init: bad argument");
    BHR_TRY(use_device(ctx));
    if (ctx->d_mips && (ctx->n_r != n_r || ctx->n_phi != n_phi))
        return bhr_fail(BHR_ERR_INVALID, "bhr_bg_init: (%d,%d) does not match the disk texture (%d,%d)", n_r, n_phi, ctx->n_r, ctx->n_phi);
    if (!ctx->d_mips) BHR_TRY(alloc_mips(ctx, n_r, n_phi));
    const size_t plane = (size_t)n_r * n_phi;
    if (!ctx->d_comp || ctx->bg_n_r != n_r || ctx->bg_n_phi != n_phi) {
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        free_bg(ctx);
        BHR_TRY(dev_alloc(&ctx->d_comp, 13 * plane));
        BHR_TRY(dev_alloc(&ctx->d_edge, (size_t)n_r));
        BHR_TRY(dev_alloc(&ctx->d_omega, (size_t)n_r));
        BHR_TRY(dev_alloc(&ctx->d_row_stats, (size_t)n_r * 2));
        BHR_HIP(hipMemsetAsync(ctx->d_comp, 0, 13 * plane * sizeof(float), ctx->stream));
    }
    ctx->bg_n_r = n_r;
    ctx->bg_n_phi = n_phi;
    ctx->az_freq = az_freq;
    ctx->az_shear = az_shear;
    BHR_TRY(upload(ctx, ctx->d_edge, edge, (size_t)n_r * sizeof(float)));
    BHR_TRY(upload(ctx, ctx->d_omega, omega_rows, (size_t)n_r * sizeof(float)));
    // initial stats (render.py:3531-3542): stats (0.5, 0.5); row stats from the undisturbed
    // temp_base profile max(tb, 0.25), max(0.8 tb, 0.10) with tb = clip(1 - r, 0, 1)^1.3 * 0.25
    ctx->stats[0] = 0.5f;
    ctx->stats[1] = 0.5f;
    std::vector<float> rs((size_t)n_r * 2);
    for (int i = 0; i < n_r; ++i) {
        double rn = n_r > 1 ? (double)i / (double)(n_r - 1) : 0.0;  // np.linspace(0, 1, n_r)
        double c = 1.0 - rn;
        if (c < 0) c = 0;
        if (c > 1) c = 1;
        double tb = pow(c, 1.3) * 0.25;
        double a = tb > 0.25 ? tb : 0.25;
        double b = tb * 0.8 > 0.10 ? tb * 0.8 : 0.10;
        rs[(size_t)i * 2 + 0] = (float)a;
        rs[(size_t)i * 2 + 1] = (float)b;
    }
    BHR_TRY(upload(ctx, ctx->d_row_stats, rs.data(), rs.size() * sizeof(float)));
    ctx->bg_ready = 1;
    return BHR_OK;
}

int32_t bhr_generate_background(bhr_ctx *ctx, float t) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "null ctx");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call init_background_layer() first");
    BHR_TRY(bhr_enter_components(ctx));          // writes comp only: runs beside the frame in flight
    BHR_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    BHR_TRY(bhr_launch_background(ctx, t));
    BHR_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    return BHR_OK;
}

int32_t bhr_set_entity_staging(bhr_ctx *ctx, const float *staging) {
    if (!ctx || !staging) return bhr_fail(BHR_ERR_INVALID, "bhr_set_entity_staging: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call init_background_layer() first");
    BHR_TRY(use_device(ctx));
    const size_t plane = (size_t)ctx->bg_n_r * ctx->bg_n_phi;
    return upload(ctx, ctx->d_comp + 5 * plane, staging, 6 * plane * sizeof(float));
}

int32_t bhr_set_comp(bhr_ctx *ctx, const float *comp13) {
    if (!ctx || !comp13) return bhr_fail(BHR_ERR_INVALID, "bhr_set_comp: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "bhr_set_comp: call bhr_bg_init first");
    BHR_TRY(use_device(ctx));
    const size_t plane = (size_t)ctx->bg_n_r * ctx->bg_n_phi;
    return upload(ctx, ctx->d_comp, comp13, 13 * plane * sizeof(float));
}

int32_t bhr_read_comp(bhr_ctx *ctx, float *comp13_out) {
    if (!ctx || !comp13_out) return bhr_fail(BHR_ERR_INVALID, "bhr_read_comp: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "bhr_read_comp: call bhr_bg_init first");
    BHR_TRY(use_device(ctx));
    const size_t plane = (size_t)ctx->bg_n_r * ctx->bg_n_phi;
    return download(ctx, comp13_out, ctx->d_comp, 13 * plane * sizeof(float));
}

int32_t bhr_fill_comp_slice(bhr_ctx *ctx, int32_t idx, float value) {
    if (!ctx || idx < 0 || idx >= 13) return bhr_fail(BHR_ERR_INVALID, "bhr_fill_comp_slice: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "bhr_fill_comp_slice: call bhr_bg_init first");
    BHR_TRY(use_device(ctx));
    const size_t plane = (size_t)ctx->bg_n_r * ctx->bg_n_phi;
    return bhr_launch_fill(ctx, ctx->d_comp + idx * plane, (int64_t)plane, value);
}

int32_t bhr_set_compose_stats(bhr_ctx *ctx, float density_p98, float struct_scale, const float *row_stats) {
    if (!ctx || !row_stats) return bhr_fail(BHR_ERR_INVALID, "bhr_set_compose_stats: bad argument");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "bhr_set_compose_stats: call bhr_bg_init first");
    BHR_TRY(use_device(ctx));
    ctx->stats[0] = density_p98;
    ctx->stats[1] = struct_scale;
    return upload(ctx, ctx->d_row_stats, row_stats, (size_t)ctx->bg_n_r * 2 * sizeof(float));
}

int32_t bhr_compose_texture(bhr_ctx *ctx, float t_offset, int32_t enable_rt, float color_temp) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "null ctx");
    if (!ctx->bg_ready) return bhr_fail(BHR_ERR_STATE, "Must call upload_parametric_state() / init_background_layer() before composing");
    BHR_TRY(bhr_enter_scene_write(ctx));
    BHR_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
    BHR_TRY(bhr_launch_compose(ctx, t_offset, enable_rt, color_temp));
    BHR_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
    return BHR_OK;
}

int32_t bhr_eval_noise(bhr_ctx *ctx, const float *coords, int64_t n, int32_t mode, int32_t octaves,
                       float persistence, float lacunarity, float *out) {
    if (!ctx || !coords || !out || n < 0) return bhr_fail(BHR_ERR_INVALID, "bhr_eval_noise: bad argument");
    if (n == 0) return BHR_OK;
    BHR_TRY(use_device(ctx));
    if (ctx->noise_cap < n) {
        BHR_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_noise_in) (void)hipFree(ctx->d_noise_in);
        if (ctx->d_noise_out) (void)hipFree(ctx->d_noise_out);
        ctx->d_noise_in = ctx->d_noise_out = nullptr;
        ctx->noise_cap = 0;
        BHR_TRY(dev_alloc(&ctx->d_noise_in, (size_t)n * 3));
        BHR_TRY(dev_alloc(&ctx->d_noise_out, (size_t)n));
        ctx->noise_cap = n;
    }
    BHR_TRY(upload(ctx, ctx->d_noise_in, coords, (size_t)n * 3 * sizeof(float)));
    BHR_TRY(bhr_launch_noise(ctx, n, mode, octaves, persistence, lacunarity));
    return download(ctx, out, ctx->d_noise_out, (size_t)n * sizeof(float));
}

// One frame: march -> bloom H -> bloom V + combine (-> lens flare).  Frames alternate between the context's two
// frame slots (bhr_frame_slot): frame n + 1 is launched on the other slot's stream into its own buffers and overlaps
// the tail and the post-passes of frame n.  The scene is only read; everything that writes it or reads a frame goes
// through bhr_enter, which orders the scene stream behind both slots.
namespace {
int32_t render_on_slot(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags, int k, int ring) {
    bhr_frame_slot &f = ctx->slots[k];
    const int with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    BHR_TRY(bhr_frame_begin(ctx, flags));
    BHR_TRY(bhr_launch_march(ctx, cam, flags));  // records the ring slot's march events
    // the march's end is the timing ring's event (recorded by the launcher): an event of its own between the march and the
    // H pass is another ~5 us barrier packet in the frame's stream (kernel-trace gaps: 10 us with two records, 0 with none)
    f.march_done = ctx->ring_ev[ring * 3 + 1];
    if (with_bloom) BHR_TRY(bhr_launch_bloom_h(ctx));
    // the V kernel clears the counter cell BHR_MAX_FRAME_SLOTS frames ahead: no frame that may be in flight on another
    // slot's stream is counting into it (the next frames' marches may already be running)
    ctx->v_zero_cell = ctx->d_steps_ring + (size_t)((ring + BHR_MAX_FRAME_SLOTS) % BHR_TIMING_RING) * BHR_STEP_CELL;
    // what the frame stores: the layers the context's consumers asked for (bhr_set_outputs); the lens flare works on the f32
    // frame, so a flared frame keeps it and quantises afterwards
    uint32_t want = ctx->out_want;
    if (flags & BHR_LENS_FLARE) want = (want | BHR_OUT_F32) & ~BHR_OUT_U8;
    const int32_t rc_v = bhr_frame_post(ctx, with_bloom, want);
    ctx->v_zero_cell = nullptr;
    BHR_TRY(rc_v);
    if (flags & BHR_LENS_FLARE) {
        if (ctx->rows != ctx->cfg.height)
            return bhr_fail(BHR_ERR_INVALID, "bhr_render: the lens flare needs whole-frame sums; use bhr_group_render for row blocks");
        // every slot has its own flare scratch (glow, sums): the frames' flare passes overlap like the rest
        BHR_TRY(bhr_launch_flare_glow(ctx, true));
        BHR_TRY(bhr_launch_flare_sums(ctx));
        BHR_TRY(bhr_launch_flare_apply(ctx, nullptr));
        if (ctx->out_want & BHR_OUT_U8) BHR_TRY(bhr_ensure_outputs(ctx, BHR_OUT_U8));
    }
    BHR_HIP(hipEventRecord(ctx->ring_ev[ring * 3 + 2], f.stream));
    BHR_HIP(hipEventRecord(f.done, f.stream));
    return BHR_OK;
}
}  // namespace

// Which stream should frame slot 1 submit to?  Two slots keep two frames in flight so that one frame's post-pass and the
// ragged end of its march run under the next frame's march -- and how well they do is decided by which HARDWARE queues HIP
// gave the two slots' streams: measured on one binary, 2200-2360 fps (pairs of queues on which the second frame's
// workgroups interleave with the first's from the start) or 2740-2770 (pairs on which they fill in behind), by nothing but
// the number of idle streams the process had created before (tools/sweep_streams.py; round 3 shipped whatever the library's
// own creation order happened to give).  HIP does not tell which queue a stream got and the good pairs are not simply
// "different queues" (bhr_streams_share_queue finds those): so the context MEASURES.  Once eight two-slot frames have been
// asked for, six candidate streams (created back to back: they go round HIP's queues) take turns as slot 1's stream for 24
// frames of the caller's own view, three times; the fastest (by its worst turn) stays, the rest idle.  ~0.2 s, once per context, frames
// identical to the one asked for; BHR_CALIBRATE_STREAMS=0 / option "calibrate_streams" 0 keeps the first stream, and so does a
// context whose frames take more than 4 ms (the first turn tells: 4k / 8k frames would spend seconds here for a per cent).
static int32_t calibrate_slot_streams(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    constexpr int NC = 6, FRAMES = 24;
    ctx->calibrating = 1;
    const auto head0 = ctx->ring_head;
    const int slot0 = ctx->next_slot;
    hipStream_t cand[NC] = {};
    double best_ms[NC];
    int32_t rc = alloc_slot(ctx, 0);
    if (rc == BHR_OK) rc = alloc_slot(ctx, 1);
    if (rc != BHR_OK) { ctx->calibrating = 0; return rc; }
    cand[0] = ctx->slots[1].stream;
    int n_cand = 1;
    for (; n_cand < NC; ++n_cand)
        if (hipStreamCreateWithFlags(&cand[n_cand], hipStreamNonBlocking) != hipSuccess) break;
    for (int c = 0; c < NC; ++c) best_ms[c] = 1e30;
    // Between turns everything is waited for and the timing ring is put back where the caller's frames had brought it: a turn
    // runs at most 64 frames through the cells in front of the caller's, never round the ring into the ones his own frames are
    // on record in (with the ring left to run, three turns of six candidates are 628 frames of a ring of 512).  The next
    // frames' counter cells are cleared as the V pass of their predecessors would have left them.
    auto drain = [&]() -> int32_t {
        for (int k = 0; k < 2; ++k) BHR_TRY(poll_sync(ctx, ctx->slots[k].stream));      // (polled: a turn is 8 ms, an interrupt's wake-up up to 2)
        for (int q = 0; q < BHR_MAX_FRAME_SLOTS; ++q)
            BHR_HIP(hipMemsetAsync(ctx->d_steps_ring + (size_t)((head0 + q) % BHR_TIMING_RING) * BHR_STEP_CELL, 0, sizeof(unsigned long long) * BHR_STEP_CELL, ctx->scene_stream));
        BHR_TRY(poll_sync(ctx, ctx->scene_stream));
        ctx->ring_head = head0;
        ctx->next_slot = slot0;
        return BHR_OK;
    };
    auto frames = [&](int n) -> int32_t {
        for (int i = 0; i < n; ++i) BHR_TRY(bhr_render(ctx, cam, flags));
        return BHR_OK;
    };
    // frames of several milliseconds (4k, 8k) gain little from which pair of queues the two slots got and would make this a
    // matter of seconds: one turn tells, the first stream stays
    bool long_frames = false;
    for (int pass = 0; pass < 3 && rc == BHR_OK && !long_frames; ++pass)
        for (int c = 0; c < n_cand && rc == BHR_OK; ++c) {
            rc = drain();
            if (rc != BHR_OK) break;
            ctx->slots[1].stream = cand[c];
            rc = frames(pass == 0 && c == 0 ? 40 : 6);                 // the first candidate also brings the clocks up
            if (rc == BHR_OK) rc = drain();
            const auto t0 = std::chrono::steady_clock::now();
            if (rc == BHR_OK) rc = frames(FRAMES);
            if (rc == BHR_OK) rc = drain();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rc == BHR_OK && (pass == 0 || ms > best_ms[c])) best_ms[c] = ms;      // the WORST of its three turns: a pair has to be good every time
            if (pass == 0 && c == 0 && ms > 4.0 * FRAMES) { long_frames = true; break; }
        }
    // the level a good pair reaches: the fastest single turn of any candidate
    double good_ms = 1e30;
    for (int c = 0; c < n_cand; ++c) good_ms = best_ms[c] < good_ms ? best_ms[c] : good_ms;
    int best = 0;
    for (int tries = 0; tries < 4 && rc == BHR_OK && !long_frames; ++tries) {
        best = 0;
        // the fastest, with no preference for the earlier ones: a candidate within 1 % of the fastest was, in 8 of 8 bench runs
        // that kept it over the fastest, a pair that later dropped to its slow state (2400 instead of 2900 fps) -- the stream
        // that measures best is also the one that stays there (18 of 18 runs)
        for (int c = 1; c < n_cand; ++c)
            if (best_ms[c] < best_ms[best]) best = c;
        // a longer turn with the one chosen: kept if it holds the good level (a pair can sit 3 % under it for a while; the
        // next candidate then gets its chance)
        rc = drain();
        if (rc != BHR_OK) break;
        ctx->slots[1].stream = cand[best];
        rc = frames(6);
        if (rc == BHR_OK) rc = drain();
        const auto t0 = std::chrono::steady_clock::now();
        if (rc == BHR_OK) rc = frames(2 * FRAMES);
        if (rc == BHR_OK) rc = drain();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 2;
        if (rc != BHR_OK || ms <= good_ms * 1.015) break;
        best_ms[best] = ms > best_ms[best] ? ms : best_ms[best] * 1.02;
    }
    if (rc != BHR_OK) best = 0;
    (void)drain();
    ctx->slots[1].stream = cand[best];
    // the others stay, idle, until the context goes: the choice was measured with them in place (with them destroyed -- HIP
    // then gives up hardware queues nobody references -- the kept pair ran 12 % slower than it had been measured, on one of
    // twelve stream layouts)
    for (int c = 0; c < n_cand; ++c)
        if (c != best && cand[c]) ctx->calib_idle[ctx->n_calib_idle++] = cand[c];
    ctx->calib_choice = best;
    for (int c = 0; c < 8; ++c) ctx->calib_fps[c] = c < n_cand && best_ms[c] < 1e29 ? (int32_t)(FRAMES * 1e3 / best_ms[c]) : 0;
    // (the last drain() has left the timing ring where the caller's frames had brought it)
    ctx->ring_head = head0;
    ctx->next_slot = slot0;
    ctx->calibrating = 0;
    ctx->streams_calibrated = 1;
    return rc;
}

int32_t bhr_render(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    if (!ctx || !cam) return bhr_fail(BHR_ERR_INVALID, "bhr_render: null argument");
    BHR_HIP(hipSetDevice(ctx->cfg.device));
    if (ctx->n_slots > 1 && ctx->opt.calibrate_streams && !ctx->streams_calibrated && !ctx->calibrating &&
        !(flags & (BHR_PERSISTENT | BHR_ROW_COSTS)) && ++ctx->two_slot_frames > 8)
        BHR_TRY(calibrate_slot_streams(ctx, cam, flags));
    ctx->stream = ctx->scene_stream;
    // launches that use per-context scratch (work queue of the persistent schedule, row-cost profile) stay on one slot
    const bool exclusive = (flags & (BHR_PERSISTENT | BHR_ROW_COSTS)) != 0;
    if (exclusive) BHR_TRY(bhr_enter(ctx));
    const int k = (ctx->n_slots > 1 && !exclusive) ? ctx->next_slot : 0;
    BHR_TRY(alloc_slot(ctx, k));
    bhr_frame_slot &f = ctx->slots[k];
    if (!(flags & BHR_SKIP_BLOOM)) BHR_TRY(bhr_bloom_prepare(ctx));   // one-off tables, on the scene stream
    if (f.stream != ctx->scene_stream) {
        BHR_HIP(hipEventRecord(ctx->scene_ev, ctx->scene_stream));    // everything the scene stream has been given so far
        BHR_HIP(hipStreamWaitEvent(f.stream, ctx->scene_ev, 0));
    }
    const int ring = (int)(ctx->ring_head % BHR_TIMING_RING);
    ctx->cur_slot = ring;
    activate_slot(ctx, k);
    ctx->stream = f.stream;
    const int32_t rc = render_on_slot(ctx, cam, flags, k, ring);
    ctx->stream = ctx->scene_stream;
    f.in_flight = 1;
    BHR_TRY(rc);
    if (ctx->n_slots > 1 && !exclusive) ctx->next_slot = (k + 1) % ctx->n_slots;
    ctx->last_slot = ring;
    ctx->ring_head += 1;
    ctx->last_flags = (int32_t)flags;
    ctx->timing_valid = 1;
    return BHR_OK;
}

int32_t bhr_read_layer(bhr_ctx *ctx, int32_t layer, float *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_read_layer: bad argument");
    BHR_TRY(use_device(ctx));
    const float *src = nullptr;
    switch (layer) {
        case BHR_LAYER_FINAL: src = ctx->d_final; BHR_TRY(bhr_ensure_outputs(ctx, BHR_OUT_F32)); break;
        case BHR_LAYER_BG: src = ctx->d_bg; break;
        case BHR_LAYER_DISK: src = ctx->d_disk; break;
        case BHR_LAYER_BLUR: src = ctx->d_blur; BHR_TRY(bhr_ensure_outputs(ctx, BHR_OUT_BLUR)); break;
        default: return bhr_fail(BHR_ERR_INVALID, "bhr_read_layer: unknown layer %d", layer);
    }
    return download(ctx, out, src, (size_t)ctx->rows * ctx->cfg.width * 3 * sizeof(float));
}

int32_t bhr_write_layer(bhr_ctx *ctx, int32_t layer, const float *in) {
    if (!ctx || !in) return bhr_fail(BHR_ERR_INVALID, "bhr_write_layer: bad argument");
    BHR_TRY(use_device(ctx));
    float *dst = nullptr;
    bhr_frame_slot &f = ctx->slots[ctx->active_slot];
    switch (layer) {
        case BHR_LAYER_FINAL: dst = ctx->d_final; f.have = (f.have | BHR_OUT_F32) & ~BHR_OUT_U8; break;   // the u8 rows follow the written frame
        case BHR_LAYER_BG: dst = ctx->d_bg; f.sum_valid = 0; break;      // a later V pass adds the two layers itself
        case BHR_LAYER_DISK: dst = ctx->d_disk; f.sum_valid = 0; break;
        case BHR_LAYER_BLUR: dst = ctx->d_blur; f.have |= BHR_OUT_BLUR; break;
        default: return bhr_fail(BHR_ERR_INVALID, "bhr_write_layer: unknown layer %d", layer);
    }
    return upload(ctx, dst, in, (size_t)ctx->rows * ctx->cfg.width * 3 * sizeof(float));
}

int32_t bhr_bloom(bhr_ctx *ctx) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "bhr_bloom: null ctx");
    if (ctx->rows != ctx->cfg.height)
        return bhr_fail(BHR_ERR_INVALID, "bhr_bloom: needs a whole-frame context (rows %d of %d)", ctx->rows, ctx->cfg.height);
    BHR_TRY(use_device(ctx));
    BHR_TRY(bhr_frame_begin(ctx, 0));                          // the context's arithmetic decides the kernels, as for a rendered frame
    if (ctx->bloom_split) BHR_TRY(bhr_launch_bloom_pack(ctx)); // the disk layer may be the caller's (bhr_write_layer)
    BHR_TRY(bhr_launch_bloom_h(ctx));
    return bhr_frame_post(ctx, 1, BHR_OUT_F32 | BHR_OUT_BLUR);
}

int32_t bhr_debug_read(bhr_ctx *ctx, int32_t which, void *out, int64_t bytes, int32_t *geom) {
    if (!ctx || bytes < 0) return bhr_fail(BHR_ERR_INVALID, "bhr_debug_read: bad argument");
    BHR_TRY(use_device(ctx));
    bhr_split_geom g;
    bhr_split_geometry(ctx, &g);
    if (geom) { geom[0] = g.NT; geom[1] = g.n_tx; geom[2] = g.WP; geom[3] = g.YB; geom[4] = g.GP; geom[5] = g.g0; geom[6] = g.t_first; geom[7] = g.n_ty; geom[8] = g.pbr; geom[9] = g.GR; }
    if (which == 4) {                                   // slot 1's stream calibration: geom[0] = done, [1] = candidate kept, [2..7] = candidates' fps
        if (!geom) return bhr_fail(BHR_ERR_INVALID, "bhr_debug_read: calibration report needs geom");
        geom[0] = ctx->streams_calibrated; geom[1] = ctx->calib_choice;
        for (int c = 0; c < 6; ++c) geom[2 + c] = ctx->calib_fps[c];
        return BHR_OK;
    }
    if (which == 3) {                                   // geom[0..9]: do pairs of the context's streams share a hardware queue (-1: no such stream)
        if (!geom) return bhr_fail(BHR_ERR_INVALID, "bhr_debug_read: stream map needs geom");
        hipStream_t st[5] = {ctx->scene_stream, ctx->slots[0].stream, ctx->n_slots > 1 ? ctx->slots[1].stream : nullptr, ctx->aux_streams[0], ctx->aux_streams[1]};
        int q = 0;                                      // pairs in the order (0,1) (0,2) (0,3) (0,4) (1,2) (1,3) (1,4) (2,3) (2,4) (3,4); 0 scene, 1 / 2 frame slots, 3 / 4 second march streams
        for (int i = 0; i < 5; ++i)
            for (int j = i + 1; j < 5; ++j, ++q) {
                geom[q] = -1;
                if (st[i] && st[j]) BHR_TRY(bhr_streams_share_queue(st[i], st[j], &geom[q]));
            }
        return BHR_OK;
    }
    if (which == 2) {                                   // the partitioned launch order of the last hybrid march (int32 tile indices)
        const int32_t *list = nullptr;
        int32_t n = 0;
        BHR_TRY(bhr_hybrid_active_list(ctx, &list, &n));
        if (geom) geom[0] = n;
        if (!out || bytes == 0) return BHR_OK;
        if (bytes > (int64_t)n * 4) return bhr_fail(BHR_ERR_INVALID, "bhr_debug_read: %lld bytes asked, the list holds %d tiles", (long long)bytes, n);
        return download(ctx, out, list, (size_t)bytes);
    }
    const void *src = which == 0 ? ctx->d_pa : which == 1 ? ctx->d_pb : nullptr;
    const size_t have = which == 0 ? g.pa_halfs * 2 : g.pb_halfs * 2;
    if (!out || bytes == 0) return BHR_OK;
    if (!src) return bhr_fail(BHR_ERR_STATE, "bhr_debug_read: buffer %d does not exist (no split-f16 frame yet)", which);
    if ((size_t)bytes > have) return bhr_fail(BHR_ERR_INVALID, "bhr_debug_read: %lld bytes asked, the buffer holds %zu", (long long)bytes, have);
    return download(ctx, out, src, (size_t)bytes);
}

int32_t bhr_set_option(bhr_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return bhr_fail(BHR_ERR_INVALID, "bhr_set_option: bad argument");
    const std::string n(name);
    const int v = (int)value;
    bhr_options &o = ctx->opt;
    if (n == "bloom_split") o.bloom_split = v < 0 ? -1 : (v ? 1 : 0);
    else if (n == "bloom_tiles") o.bloom_tiles = v < 0 || v > 8 ? 0 : v;
    else if (n == "hybrid_repair") o.hybrid_repair = v < 0 ? -1 : (v ? 1 : 0);
    else if (n == "hybrid_band_lo") { if (!o.hybrid_band_set) o.hybrid_band[1] = 0.36; o.hybrid_band[0] = value; o.hybrid_band_set = 1; }
    else if (n == "hybrid_band_hi") { if (!o.hybrid_band_set) o.hybrid_band[0] = 0.085; o.hybrid_band[1] = value; o.hybrid_band_set = 1; }
    else if (n == "hybrid_band_default") o.hybrid_band_set = 0;
    else if (n == "hybrid_pad") { if (value >= 0.0 && value <= 4.0) o.hybrid_pad = value; }
    else if (n == "hybrid_streams") o.hybrid_streams = v == 1 ? 1 : (v == 2 ? 2 : -1);
    else if (n == "calibrate_streams") o.calibrate_streams = v != 0;
    else if (n == "hybrid_classify") o.hybrid_classify = v != 0;
    else if (n == "hybrid_swap") o.hybrid_swap = v != 0;
    else if (n == "mip_lds") o.mip_lds = v != 0;
    else if (n == "tile_order_rows") o.tile_order_rows = v != 0;
    else if (n == "group_threads") o.group_threads = v < 0 ? -1 : (v ? 1 : 0);
    else if (n == "group_schedule") o.group_schedule = v < 0 ? -1 : (v ? 1 : 0);
    else return bhr_fail(BHR_ERR_INVALID, "bhr_set_option: unknown option '%s'", name);
    return BHR_OK;
}

int32_t bhr_set_outputs(bhr_ctx *ctx, uint32_t mask) {
    if (!ctx || !(mask & (BHR_OUTPUT_F32 | BHR_OUTPUT_BLUR | BHR_OUTPUT_U8)) || (mask & ~(BHR_OUTPUT_F32 | BHR_OUTPUT_BLUR | BHR_OUTPUT_U8)))
        return bhr_fail(BHR_ERR_INVALID, "bhr_set_outputs: mask %u", mask);
    ctx->out_want = mask;
    return BHR_OK;
}

int32_t bhr_lens_flare(bhr_ctx *ctx) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "bhr_lens_flare: null ctx");
    if (ctx->rows != ctx->cfg.height)
        return bhr_fail(BHR_ERR_INVALID, "bhr_lens_flare: needs a whole-frame context (rows %d of %d)", ctx->rows, ctx->cfg.height);
    BHR_TRY(use_device(ctx));
    BHR_TRY(bhr_ensure_outputs(ctx, BHR_OUT_F32));
    ctx->slots[ctx->active_slot].have &= ~BHR_OUT_U8;          // the u8 rows follow the flared frame
    BHR_TRY(bhr_launch_flare_glow(ctx, true));
    BHR_TRY(bhr_launch_flare_sums(ctx));
    return bhr_launch_flare_apply(ctx, nullptr);
}

int32_t bhr_lens_flare_sums(bhr_ctx *ctx, double *out3) {
    if (!ctx || !out3) return bhr_fail(BHR_ERR_INVALID, "bhr_lens_flare_sums: bad argument");
    if (ctx->rows != ctx->cfg.height)
        return bhr_fail(BHR_ERR_INVALID, "bhr_lens_flare_sums: needs a whole-frame context (rows %d of %d)", ctx->rows, ctx->cfg.height);
    BHR_TRY(use_device(ctx));
    BHR_TRY(bhr_launch_flare_glow(ctx, true));
    BHR_TRY(bhr_launch_flare_sums(ctx));
    return download(ctx, out3, ctx->d_flare_sums, 3 * sizeof(double));
}

int32_t bhr_read_gathered(bhr_ctx *ctx, float *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_read_gathered: bad argument");
    if (!ctx->d_gather) return bhr_fail(BHR_ERR_STATE, "bhr_read_gathered: no bhr_group_render(..., BHR_GATHER_PEER) has gathered into this context");
    BHR_TRY(use_device(ctx));
    return download(ctx, out, ctx->d_gather, (size_t)ctx->cfg.height * ctx->cfg.width * 3 * sizeof(float));
}

int32_t bhr_read_final_u8(bhr_ctx *ctx, uint8_t *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_read_final_u8: bad argument");
    BHR_TRY(use_device(ctx));
    BHR_TRY(bhr_ensure_outputs(ctx, BHR_OUT_U8));
    return download(ctx, out, ctx->d_final_u8, (size_t)ctx->rows * ctx->cfg.width * 3);
}

int32_t bhr_get_counters(bhr_ctx *ctx, bhr_counters *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_get_counters: bad argument");
    BHR_TRY(use_device(ctx));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->timing_valid) {
        unsigned long long steps = 0;
        BHR_TRY(fold_cells(ctx, ctx->last_steps_ptr ? ctx->last_steps_ptr : ctx->d_ray_steps, 1, &steps));
        ctx->counters.ray_steps = steps;
        // the last launch's three events: its ring slot's (bhr_render) or the context's scalar ones (group render)
        const hipEvent_t *e = ctx->last_slot >= 0 ? ctx->ring_ev + ctx->last_slot * 3 : ctx->ev;
        ctx->counters.march_ms = ctx->march_end_recorded ? ev_ms(e[0], e[1]) : -1.0f;      // a group render without BHR_GROUP_TIME_MARCH: not timed
        ctx->counters.bloom_ms = ctx->march_end_recorded ? ev_ms(e[1], e[2]) : -1.0f;
        ctx->counters.frame_ms = ev_ms(e[0], e[2]);
    }
    {
        // the cells after the head have been cleared for the frames to come: at most RING - MAX_SLOTS frames are on record
        const int64_t n = ctx->ring_head < BHR_TIMING_RING - BHR_MAX_FRAME_SLOTS ? ctx->ring_head : BHR_TIMING_RING - BHR_MAX_FRAME_SLOTS;
        float ms_m = 0.0f, ms_b = 0.0f;
        unsigned long long steps_sum = 0;
        if (n > 0) {
            std::vector<unsigned long long> hs(BHR_TIMING_RING);
            BHR_TRY(fold_cells(ctx, ctx->d_steps_ring, BHR_TIMING_RING, hs.data()));
            for (int64_t k = 0; k < n; ++k) {
                const int slot = (int)((ctx->ring_head - 1 - k) % BHR_TIMING_RING);
                ms_m += ev_ms(ctx->ring_ev[slot * 3 + 0], ctx->ring_ev[slot * 3 + 1]);
                ms_b += ev_ms(ctx->ring_ev[slot * 3 + 1], ctx->ring_ev[slot * 3 + 2]);
                steps_sum += hs[slot];
            }
        }
        // union of the march intervals, on the clock of the oldest timed frame's start event (events of the two
        // slot streams are comparable: hipEventElapsedTime works across streams of one device)
        float busy = 0.0f, span = 0.0f;
        if (n > 0) {
            const int first = (int)((ctx->ring_head - n) % BHR_TIMING_RING);
            const hipEvent_t origin = ctx->ring_ev[first * 3 + 0];
            std::vector<std::pair<float, float>> iv((size_t)n);
            for (int64_t k = 0; k < n; ++k) {
                const int slot = (int)((ctx->ring_head - n + k) % BHR_TIMING_RING);
                iv[(size_t)k] = {ev_ms(origin, ctx->ring_ev[slot * 3 + 0]), ev_ms(origin, ctx->ring_ev[slot * 3 + 1])};
                span = std::max(span, ev_ms(origin, ctx->ring_ev[slot * 3 + 2]));
            }
            std::sort(iv.begin(), iv.end());
            float lo = iv[0].first, hi = iv[0].second;
            for (size_t k = 1; k < iv.size(); ++k) {
                if (iv[k].first > hi) { busy += hi - lo; lo = iv[k].first; hi = iv[k].second; }
                else hi = std::max(hi, iv[k].second);
            }
            busy += hi - lo;
        }
        ctx->counters.march_busy_ms = busy;
        ctx->counters.span_ms = span;
        ctx->counters.frames_timed = (int32_t)n;
        ctx->counters.march_ms_sum = ms_m;
        ctx->counters.bloom_ms_sum = ms_b;
        ctx->counters.ray_steps_sum = steps_sum;
    }
    if (ctx->bg_ready) {
        if (hipEventQuery(ctx->ev[5]) == hipSuccess) ctx->counters.background_ms = ev_ms(ctx->ev[4], ctx->ev[5]);
        if (hipEventQuery(ctx->ev[7]) == hipSuccess) ctx->counters.compose_ms = ev_ms(ctx->ev[6], ctx->ev[7]);
    }
    *out = ctx->counters;
    return BHR_OK;
}

int32_t bhr_timing_dump(bhr_ctx *ctx, float *out, int32_t n) {
    if (!ctx || !out || n <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_timing_dump: bad argument");
    BHR_TRY(use_device(ctx));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t have = ctx->ring_head < BHR_TIMING_RING - BHR_MAX_FRAME_SLOTS ? ctx->ring_head : BHR_TIMING_RING - BHR_MAX_FRAME_SLOTS;
    if (n > have) return bhr_fail(BHR_ERR_INVALID, "bhr_timing_dump: %d frames asked, %lld on record", n, (long long)have);
    const hipEvent_t origin = ctx->ring_ev[(int)((ctx->ring_head - n) % BHR_TIMING_RING) * 3];
    for (int k = 0; k < n; ++k) {
        const int slot = (int)((ctx->ring_head - n + k) % BHR_TIMING_RING);
        for (int e = 0; e < 3; ++e) out[k * 3 + e] = ev_ms(origin, ctx->ring_ev[slot * 3 + e]);
    }
    return BHR_OK;
}

int32_t bhr_selftest(bhr_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_selftest: bad argument");
    BHR_TRY(use_device(ctx));
    unsigned long long *d = nullptr;
    BHR_TRY(dev_alloc(&d, 4));
    int32_t rc = bhr_selftest_strict(ctx, d);
    if (rc == BHR_OK) rc = download(ctx, out, d, 4 * sizeof(unsigned long long));
    (void)hipFree(d);
    return rc;
}

int32_t bhr_mip_lds_level(bhr_ctx *ctx) { return ctx ? ctx->mip_lds_from : -1; }

int32_t bhr_get_row_costs(bhr_ctx *ctx, uint64_t *out, int32_t n) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_get_row_costs: bad argument");
    const int32_t bands = (ctx->rows + 7) / 8;
    if (n != bands) return bhr_fail(BHR_ERR_INVALID, "bhr_get_row_costs: the context has %d 8-row bands, caller asked for %d", bands, n);
    if (!ctx->d_row_steps || !(ctx->last_flags & BHR_ROW_COSTS))
        return bhr_fail(BHR_ERR_STATE, "bhr_get_row_costs: the last bhr_render did not carry BHR_ROW_COSTS");
    BHR_TRY(use_device(ctx));
    std::vector<uint64_t> both((size_t)2 * bands);
    BHR_TRY(download(ctx, both.data(), ctx->d_row_steps, both.size() * sizeof(uint64_t)));
    for (int32_t k = 0; k < bands; ++k) out[k] = both[(size_t)k] + both[(size_t)bands + k];
    return BHR_OK;
}

int32_t bhr_get_row_costs_split(bhr_ctx *ctx, uint64_t *fast_out, uint64_t *strict_out, int32_t n) {
    if (!ctx || !fast_out || !strict_out) return bhr_fail(BHR_ERR_INVALID, "bhr_get_row_costs_split: bad argument");
    const int32_t bands = (ctx->rows + 7) / 8;
    if (n != bands) return bhr_fail(BHR_ERR_INVALID, "bhr_get_row_costs_split: the context has %d 8-row bands, caller asked for %d", bands, n);
    if (!ctx->d_row_steps || !(ctx->last_flags & BHR_ROW_COSTS))
        return bhr_fail(BHR_ERR_STATE, "bhr_get_row_costs_split: the last bhr_render did not carry BHR_ROW_COSTS");
    BHR_TRY(use_device(ctx));
    BHR_TRY(download(ctx, fast_out, ctx->d_row_steps, (size_t)bands * sizeof(uint64_t)));
    return download(ctx, strict_out, ctx->d_row_steps + bands, (size_t)bands * sizeof(uint64_t));
}

int32_t bhr_timing_reset(bhr_ctx *ctx) {
    if (!ctx) return bhr_fail(BHR_ERR_INVALID, "null ctx");
    BHR_TRY(use_device(ctx));
    BHR_HIP(hipMemsetAsync(ctx->d_steps_ring, 0, sizeof(unsigned long long) * BHR_TIMING_RING * BHR_STEP_CELL, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ring_head = 0;
    return BHR_OK;
}

}  // extern "C"
