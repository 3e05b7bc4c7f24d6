/*
 * bhr.h -- C ABI of libbhr_hip.so, the MI355X (gfx950) Schwarzschild ray tracer.
 *
 * The reference (hwuu/black-hole-renderer) has no FFI: its renderer boundary is
 * the Python class TaichiRenderer (render.py:2189-4028) whose device half is
 * JIT-compiled Taichi.  This header is the boundary a maintainer would bind in
 * its place; each entry point names the reference interface it replaces.  The
 * ctypes binding that mirrors TaichiRenderer lives in
 * black-hole-renderer_amd/renderer.py; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every call returns 0 on success or a negative bhr_status; bhr_last_error()
 *    returns a thread-local message for the last failure on the calling thread;
 *  - a bhr_ctx owns one HIP device, one stream and all its device buffers.  It
 *    is not thread-safe; distinct contexts may be driven from distinct threads
 *    or processes (one process per GPU, or one process driving N devices);
 *  - host pointers are plain row-major float32 arrays owned by the caller and
 *    copied during the call; nothing in this ABI is a torch/Taichi type;
 *  - images handed back to the host are (rows, width, 3) float32, x fastest,
 *    i.e. already in the (H, W, 3) order TaichiRenderer.render() returns after
 *    its final transpose (render.py:3923);
 *  - the library never falls back to a CPU path: without a HIP device
 *    bhr_create() fails with BHR_ERR_NO_DEVICE.
 */
#ifndef BHR_H
#define BHR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHR_ABI_VERSION 1

#if defined(__GNUC__)
#define BHR_API __attribute__((visibility("default")))
#else
#define BHR_API
#endif

typedef struct bhr_ctx bhr_ctx;

typedef enum {
    BHR_OK = 0,
    BHR_ERR_INVALID = -1,      /* bad argument / shape mismatch (reference: AssertionError, ValueError) */
    BHR_ERR_NO_DEVICE = -2,    /* no usable HIP device */
    BHR_ERR_HIP = -3,          /* a HIP runtime call failed; see bhr_last_error() */
    BHR_ERR_STATE = -4,        /* call sequence violated (e.g. background before bhr_bg_init) */
    BHR_ERR_NOMEM = -5
} bhr_status;

/* Constructor arguments of TaichiRenderer.__init__ (render.py:2199-2208), plus
 * the device ordinal and the row block [row0,row1) of the image this context
 * renders (0,height for a whole frame; row blocks are the multi-GPU tiles). */
typedef struct {
    int32_t width, height;
    int32_t row0, row1;
    float step_size;            /* h_base                   (default 0.1)  */
    float r_max;                /* escape radius floor      (default 10)   */
    float r_disk_inner;         /* default 2.0  */
    float r_disk_outer;         /* default 15.0 */
    float disk_tilt_deg;        /* default 0    */
    int32_t anti_alias;         /* 0 = "disabled", 1 = "lod_radius"        */
    float aa_strength;          /* default 1.0  */
    float disk_rotation_speed;  /* t_offset = frame * this (render.py:3897) */
    int32_t device;             /* HIP device ordinal */
    int32_t math_mode;          /* BHR_MATH_FAST (0), BHR_MATH_STRICT (1) or BHR_MATH_HYBRID (2) */
} bhr_config;

/* math_mode: FAST uses v_rsq/v_rcp/v_sqrt and FMA contraction inside the RK4 loop (the analogue
 * of Taichi's fast_math=True default).  STRICT evaluates render.py:2854-3006 operation by operation
 * with IEEE sqrt/divide: ray paths, step counts and hit points are bit-identical to a strict f32
 * evaluation of the reference, at roughly twice the march time. */
#define BHR_MATH_FAST 0
#define BHR_MATH_STRICT 1
/* HYBRID: the strict arithmetic where the geodesic is unstable, the fast arithmetic elsewhere.  The only rays that
 * amplify rounding are those whose impact parameter b = |pos x dir| lies near the critical b_c = (3 sqrt 3 / 2) r_s of
 * the photon sphere (they wind around it; the deflection grows like -ln|b / b_c - 1|).  8x8-pixel tiles whose rays
 * have b within a band around b_c are marched by the STRICT kernel -- bit-identical paths, as math_mode 1 -- and all
 * other tiles by the FAST kernel, as two launches over complementary tile lists.  Pixels stay within the 1e-4 bar of
 * the reference's statements on every fixture; ray-step totals within 2e-4.  Schedules / disk sources
 * without a tile-list form (BHR_PERSISTENT, Disk V2) run STRICT.  With anti_alias = 1 the mip level is a
 * truncated function of the ray differentials, which rounding noise flips on level boundaries: the fast tiles' kernel
 * then carries guards -- a lane within a guard band of a level boundary (or of another switch of the algorithm: a disk
 * crossing in the terminating step, a step that ends on the disk plane, the disk's edges) hands its pixel to a third
 * launch that marches it with the strict arithmetic.  The guards are also on for tilted disks (the plane function of a
 * step that ends on a tilted plane rounds to exactly 0 for ~1e-6 of the crossings, which the reference's sign test never
 * registers); with tilt 0 and no anti-aliasing they are off (BHR_HYBRID_REPAIR=1 / 0 in the environment forces them). */
#define BHR_MATH_HYBRID 2

/* Camera uniforms exactly as TaichiRenderer.render() uploads them
 * (render.py:3880-3892): build_camera() in f64 on the host, cast to f32. */
typedef struct {
    float pos[3], right[3], up[3], forward[3];
    float pixel_width, pixel_height;
    float r_escape;             /* max(r_max, 2*|cam_pos|)  (render.py:3884) */
    float t_offset;             /* frame * disk_rotation_speed */
} bhr_camera;

/* bhr_render flags */
#define BHR_SKIP_DIFFERENTIALS 1u  /* render(skip_differentials=True): plain bilinear disk lookup */
#define BHR_SKIP_BLOOM         2u  /* render(skip_bloom=True) */
#define BHR_PERSISTENT         4u  /* persistent waves + queue refill instead of the tile schedule */
#define BHR_FORCE_FAST         8u  /* this call only: fast arithmetic regardless of bhr_config.math_mode */
#define BHR_FORCE_STRICT      16u  /* this call only: strict arithmetic regardless of bhr_config.math_mode */
#define BHR_ROW_COSTS         64u  /* also accumulate ray-steps per 8-row band (bhr_get_row_costs): the cost profile row blocks are balanced with */
#define BHR_LENS_FLARE        32u  /* add the lens flare to the final layer on the device (render.py:3920-4028) */
#define BHR_FORCE_HYBRID     256u  /* this call only: hybrid arithmetic regardless of bhr_config.math_mode */
#define BHR_GATHER_U8        512u  /* bhr_group_render: gather the QUANTISED rows ((H, W, 3) u8, save_image's truncation, render.py:423)
                                      on ctxs[0]'s device -- a quarter of the f32 bytes over xGMI; the pipelined schedule
                                      ships every row chunk as soon as its V pass has written it */
#define BHR_GROUP_SERIAL    1024u  /* bhr_group_render: the serial schedule (march -> H -> halo -> V -> gather, each behind the other
                                      on the tile's stream); same bytes as the pipelined one */
#define BHR_GROUP_PIPELINED 2048u  /* bhr_group_render: the pipelined schedule (halo pull under the V pass of the middle rows, row
                                      chunks pushed while the next chunk's V kernel runs).  Neither flag: pipelined where the
                                      tiles sit on distinct devices, serial where they share one */
#define BHR_GROUP_ASYNC     8192u  /* bhr_group_render: return once the frame is SUBMITTED (frames whose rows are stored by the kernels
                                      themselves: a GATHER flag, split-f16 post-pass, no lens flare, no host output; any other frame
                                      synchronises as before).  The next frame may be submitted at once: a tile's H pass waits, on
                                      the device, for its neighbours' previous V passes before it stores into their halo rows.
                                      bhr_group_sync / bhr_read_gathered* / any synchronous call waits for the frames in flight;
                                      per-tile counters describe the LAST frame submitted */
#define BHR_GROUP_TIME_MARCH 4096u /* bhr_group_render / bhr_tile_render: also record every tile's march-end event (bhr_counters.march_ms /
                                      bloom_ms of the tiles; frame_ms is always available).  Off by default: the record is a ~5 us bubble
                                      between the march and the H pass of every tile */
#define BHR_GATHER_PEER      128u  /* bhr_group_render: gather the tiles into one (H, W, 3) buffer on ctxs[0]'s device with
                                      hipMemcpyPeerAsync (xGMI), one copy per tile on the tile's own stream */

/* selectors for bhr_read_layer */
typedef enum {
    BHR_LAYER_FINAL = 0,  /* clip(bg + disk + blur, 0, 1)           render.py:3918 */
    BHR_LAYER_BG = 1,     /* image_field  = skybox * (1 - alpha)    render.py:3017 */
    BHR_LAYER_DISK = 2,   /* disk_layer_field = clamp(accum, 0, 1)  render.py:3018 */
    BHR_LAYER_BLUR = 3    /* blur_field after the V pass            render.py:3108 */
} bhr_layer;

typedef struct {
    uint64_t ray_steps;      /* executed while-loop iterations of the last march (render.py:2854) */
    uint64_t rays;           /* pixels marched by the last bhr_render */
    float march_ms;          /* HIP-event time of the march kernel, on the ctx stream */
    float bloom_ms;          /* H pass + V pass + combine */
    float frame_ms;          /* first launch .. last launch of bhr_render */
    float background_ms;     /* last bhr_generate_background */
    float compose_ms;        /* last bhr_compose_texture (compose + mip chain) */
    int32_t march_vgprs;     /* registers per lane of the march kernel that ran (hipFuncGetAttributes) */
    int32_t march_lds_bytes;
    /* sums over the bhr_render calls since bhr_timing_reset (at most the last
     * BHR_TIMING_RING - 2 calls), each launch bracketed by its own HIP events on its frame slot's stream */
    int32_t frames_timed;
    float march_ms_sum;
    float bloom_ms_sum;
    uint64_t ray_steps_sum;  /* ray_steps of one frame x frames_timed is NOT assumed: summed per frame */
    /* With two frames in flight the march launches of successive frames overlap, so march_ms_sum counts shared time
     * twice.  march_busy_ms = length of the UNION of the timed frames' [march start, march end] intervals: the time
     * during which at least one march kernel of this context was running; span_ms = first march start .. last frame end. */
    float march_busy_ms;
    float span_ms;
} bhr_counters;

#define BHR_TIMING_RING 512

BHR_API const char *bhr_last_error(void);
BHR_API int32_t bhr_abi_version(void);
BHR_API int32_t bhr_device_count(void);

/* ---- lifetime: TaichiRenderer.__init__ / garbage collection -------------- */
BHR_API int32_t bhr_create(const bhr_config *cfg, bhr_ctx **out);
BHR_API void bhr_destroy(bhr_ctx *ctx);
BHR_API int32_t bhr_sync(bhr_ctx *ctx);

/* ---- scene data ----------------------------------------------------------- */
/* texture_field.from_numpy(skybox): (tex_h, tex_w, 3) f32   render.py:2232-2233 */
BHR_API int32_t bhr_set_skybox(bhr_ctx *ctx, const float *rgb, int32_t tex_h, int32_t tex_w);
/* The Milky-Way glow of generate_skybox (render.py:296-341) added to the uploaded sky on the device, followed by
 * its clip(0, 1): upload the sky WITHOUT the glow (nebula + stars, the order-sensitive host part), call this once.
 * bhr_get_skybox reads the texture back ((tex_h, tex_w, 3) f32).  Asynchronous / synchronous. */
BHR_API int32_t bhr_skybox_add_glow(bhr_ctx *ctx);
/* generate_skybox's nebula and star splats (render.py:167-295) on the device, into the (tex_h, tex_w, 3) skybox a
 * previous bhr_set_skybox allocated.  The host supplies what comes out of NumPy's random stream: the 1/16-resolution
 * nebula noise as u8 (coarse_h, coarse_w, 3) with Pillow's fixed-point BILINEAR coefficients for both passes (kh:
 * (tex_w, ksize_h) 22-bit weights, bounds_h: (tex_w, 2) first source column and tap count; kv / bounds_v likewise for
 * rows), and per star its centre (cx, cy: f32 texel coordinates), colour (n, 3) and blob values (n, (2 patch_r + 1)^2).
 * The device reproduces Pillow's resize, `sky = 0.003 + resized / 255.0 * 0.04` and np.add.at's accumulation order bit
 * for bit.  Follow with bhr_skybox_add_glow.  Synchronises. */
BHR_API int32_t bhr_skybox_build(bhr_ctx *ctx, int32_t tex_h, int32_t tex_w, const uint8_t *coarse_rgb, int32_t coarse_h,
                                 int32_t coarse_w, const int32_t *kh, const int32_t *bounds_h, int32_t ksize_h,
                                 const int32_t *kv, const int32_t *bounds_v, int32_t ksize_v, int32_t n_stars,
                                 const float *cx, const float *cy, const float *colors, const float *vals, int32_t patch_r);
BHR_API int32_t bhr_get_skybox(bhr_ctx *ctx, float *out);
/* disk_texture_field.from_numpy + generate_disk_mipmaps(levels=4) + padded
 * upload (render.py:2235-2251, update_disk_texture 2292-2312).  (n_r, n_phi, 4)
 * f32.  The first call fixes (n_r, n_phi); later calls must match
 * (reference: AssertionError at render.py:2299 -> BHR_ERR_INVALID).  The mip
 * chain is built on the device and stored packed (1.33x, not 5x). */
BHR_API int32_t bhr_set_disk_texture(bhr_ctx *ctx, const float *rgba, int32_t n_r, int32_t n_phi);
/* disk_texture_field.to_numpy() */
BHR_API int32_t bhr_get_disk_texture(bhr_ctx *ctx, float *rgba_out);
/* disk_mips_field.to_numpy(): level `level` only, (n_r>>level, n_phi>>level, 4) */
BHR_API int32_t bhr_get_disk_mip(bhr_ctx *ctx, int32_t level, float *rgba_out);
BHR_API int32_t bhr_num_mip_levels(bhr_ctx *ctx);

/* ---- procedural disk-texture pipeline ------------------------------------- */
/* init_background_layer (render.py:3491-3547): allocates comp (13,n_r,n_phi),
 * uploads edge[n_r], omega_rows[n_r], initial stats; az_freq/az_shear are the
 * two RNG draws the host makes. */
BHR_API int32_t bhr_bg_init(bhr_ctx *ctx, int32_t n_r, int32_t n_phi, int32_t az_freq, float az_shear,
                    const float *edge, const float *omega_rows);
/* generate_background(t) -> _generate_background_kernel  (render.py:3549-3562, 3332-3451) */
BHR_API int32_t bhr_generate_background(bhr_ctx *ctx, float t);
/* accumulate_entity_layer's upload half: staging (6,n_r,n_phi) -> comp[5..10]
 * (render.py:3651-3653, 3455-3471) */
BHR_API int32_t bhr_set_entity_staging(bhr_ctx *ctx, const float *staging);
/* upload_parametric_state's comp upload: all 13 planes (render.py:2336-2353) */
BHR_API int32_t bhr_set_comp(bhr_ctx *ctx, const float *comp13);
/* _comp_field.to_numpy() (render.py:3666) */
BHR_API int32_t bhr_read_comp(bhr_ctx *ctx, float *comp13_out);
/* _zero_comp_slice / _fill_comp_slice (render.py:3475-3487) */
BHR_API int32_t bhr_fill_comp_slice(bhr_ctx *ctx, int32_t idx, float value);
/* _param_stats_field / _param_row_stats_field uploads (render.py:3708-3712):
 * stats = {density_p98, struct_scale}; row_stats (n_r, 2) = {max, p70} */
BHR_API int32_t bhr_set_compose_stats(bhr_ctx *ctx, float density_p98, float struct_scale, const float *row_stats);
/* _compose_disk_texture_kernel + mip chain (render.py:3755-3767, 3805-3817) */
BHR_API int32_t bhr_compose_texture(bhr_ctx *ctx, float t_offset, int32_t enable_rt, float color_temp);
/* eval_noise (render.py:3769-3790): coords (n,3); mode 0 simplex, 1 fbm */
BHR_API int32_t bhr_eval_noise(bhr_ctx *ctx, const float *coords, int64_t n, int32_t mode, int32_t octaves,
                       float persistence, float lacunarity, float *out);

/* ---- the hot path: TaichiRenderer.render() (render.py:3865-3923) ----------
 * Launches the fused ray-march kernel for rows [row0,row1), then (unless
 * BHR_SKIP_BLOOM) the bloom H pass, V pass and final combine.  Asynchronous; results stay
 * in HBM until read.  Successive calls alternate between the context's two frame slots (own stream and frame
 * buffers, shared read-only scene), so frame n + 1 overlaps the tail and the post-passes of frame n; every other
 * entry point is ordered (on the device) behind the frames in flight: reads return the last frame rendered, scene
 * updates never race a march.  BHR_FRAME_SLOTS=1 in the environment of bhr_create: one frame at a time.  (Every BHR_*
 * environment switch of the library is read once, by bhr_create.) */
BHR_API int32_t bhr_render(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags);
/* image_field/disk_layer_field/blur_field .to_numpy() and the final image,
 * for the context's rows: (row1-row0, width, 3) f32.  Synchronises. */
BHR_API int32_t bhr_read_layer(bhr_ctx *ctx, int32_t layer, float *out);
/* field.from_numpy() for a frame layer: replaces the context's rows of FINAL, BG, DISK or BLUR with
 * caller data, e.g. to post-process a frame composed elsewhere.  Synchronises. */
BHR_API int32_t bhr_write_layer(bhr_ctx *ctx, int32_t layer, const float *in);
/* self._bloom_kernel(disk_layer_field, bright_field, blur_field, 0, 0.4, int(0.02 W), (W / 640)^2) followed by
 * clip(img + disk + blur, 0, 1) (render.py:3914-3918), standalone on the layers currently in the context:
 * BLUR <- bloom(DISK), FINAL <- clip(BG + DISK + BLUR, 0, 1).  Whole-frame context (row blocks need their
 * neighbours' halo rows: bhr_group_render).  Asynchronous. */
BHR_API int32_t bhr_bloom(bhr_ctx *ctx);
/* What bhr_render keeps in memory of a frame besides the bg / disk layers.  TaichiRenderer.render() returns the f32 frame
 * (BHR_OUTPUT_F32, the default); the video loop, the PNG sink and the u8 row-block gather only ever read the quantised
 * rows (BHR_OUTPUT_U8: save_image's truncation fused into the V pass's epilogue, 3 bytes per pixel instead of 12);
 * blur_field (BHR_OUTPUT_BLUR) is internal to the reference's render().  A layer that was not kept is produced on demand
 * by whichever call needs it (bhr_read_layer, bhr_read_final_u8, the sinks): the frame's V pass runs again for it, same
 * kernels, same bits.  mask: any non-empty combination. */
#define BHR_OUTPUT_F32 1u
#define BHR_OUTPUT_BLUR 2u
#define BHR_OUTPUT_U8 4u
BHR_API int32_t bhr_set_outputs(bhr_ctx *ctx, uint32_t mask);
/* The library's switches.  Each has an environment variable that bhr_create reads ONCE (no entry point calls getenv
 * afterwards) and can be changed per context later with this call -- what tests and A/B tools use:
 *   "bloom_split"     BHR_BLOOM_SPLIT     -1 post-pass by arithmetic (exact f32 under strict, split f16 under fast / hybrid), 0 / 1 force
 *   "bloom_tiles"     BHR_BLOOM_TILES     0 output tiles per wave of the split post-pass by launch size, 1..8 force (A/B runs)
 *   "hybrid_repair"   BHR_HYBRID_REPAIR   -1 guards + strict fix list by view (anti-aliased or tilted), 0 / 1 force
 *   "hybrid_band_lo" / "hybrid_band_hi" / "hybrid_band_default"   BHR_HYBRID_BAND="lo,hi"   strict band around b_c, in r_s
 *   "hybrid_pad"      BHR_HYBRID_PAD      share of its own span of b a tile spanning <= 0.1 r_s is padded by in the strict-band test (default 0.5; larger tiles: up to all of it)
 *   "hybrid_streams"  BHR_HYBRID_STREAMS  -1 (default) the two lists of a hybrid march on one stream where two frame slots overlap
 *                                         frames and on two where a frame runs alone; 1 / 2 force
 *   "calibrate_streams" BHR_CALIBRATE_STREAMS 1 (default) a context with two frame slots times six candidate streams for slot 1 on
 *                                         its ninth frame and keeps the fastest (~0.15 s once; csrc/api.hip: calibrate_slot_streams)
 *   "hybrid_swap"     BHR_HYBRID_SWAP     1 (default) the fast list of a two-stream hybrid march on the frame's own stream (the
 *                                         post-pass follows it on one hardware queue), the strict list on the second; 0 swapped
 *   "hybrid_classify" BHR_HYBRID_CLASSIFY 1 (default) a view change classifies the tiles and partitions the launch order on the
 *                                         device (~0.05 ms whatever the size), 0 on the submitting thread (the same lists)
 *   "mip_lds"         BHR_MIP_LDS         1 anti-aliased fast frames stage the coarse mip levels in LDS
 *   "tile_order_rows" BHR_TILE_ORDER=row  1 row-major march launch order
 *   "group_threads"   BHR_GROUP_THREADS   -1 one submitting thread per tile where the tiles sit on distinct devices, 0 / 1 force
 *   "group_schedule"  BHR_GROUP_SCHEDULE  -1 by flags, else pipelined where a halo copy can hide (exact-f32 post-pass on distinct
 *                                         devices) and serial otherwise; 0 serial, 1 pipelined (explicit flags still win)
 * (bhr_create only: BHR_FRAME_SLOTS, BHR_TILE_BLOCK, BHR_AUX_STREAMS, BHR_STREAM_PAD.) */
BHR_API int32_t bhr_set_option(bhr_ctx *ctx, const char *name, double value);
/* Diagnostics (tests): the split-f16 post-pass's packed intermediates of the last frame as raw bytes -- which = 0 the H pass's
 * input (csrc/bloom.hip: pa), 1 its output / the V pass's input (pb) -- and the layout's geometry: geom[10] = {NT, n_tx, WP, YB,
 * GP, g0, t_first, n_ty, pbr, GR}; which = 2 the launch order of the last math-hybrid march as int32 tile indices, strict tiles
 * first (geom[0] = tiles in it; bhr_hybrid_info tells how many are strict).  out == NULL or bytes == 0: geometry only.
 * Synchronises. */
BHR_API int32_t bhr_debug_read(bhr_ctx *ctx, int32_t which, void *out, int64_t bytes, int32_t *geom);
/* TaichiRenderer._apply_lens_flare(final, disk) (render.py:3925-4028) on the device, standalone:
 * FINAL <- clip(FINAL + flare(DISK), 0, 1) for a whole-frame context.  bhr_render / bhr_group_render
 * with BHR_LENS_FLARE run the same kernels after the combine.  Asynchronous. */
BHR_API int32_t bhr_lens_flare(bhr_ctx *ctx);
/* The three frame sums the flare is built from, as the reference computes them (render.py:3931-3939):
 * out3 = { np.sum(glow) (f32 value), np.sum(x * glow), np.sum(y * glow) } with glow = max(DISK, axis=2),
 * accumulated in NumPy's own summation order, hence bit-identical to it.  Synchronises. */
BHR_API int32_t bhr_lens_flare_sums(bhr_ctx *ctx, double *out3);
/* save_image()'s quantisation (clip*255 truncated to u8, render.py:423) done
 * on the device: (row1-row0, width, 3) u8.  Synchronises. */
BHR_API int32_t bhr_read_final_u8(bhr_ctx *ctx, uint8_t *out);
BHR_API int32_t bhr_get_counters(bhr_ctx *ctx, bhr_counters *out);
/* Last BHR_MATH_HYBRID march of this context: out_tiles = {tiles marched strict, tiles of the row block},
 * out_band = {lo, hi}: the strict band [b_c - lo, b_c + hi] of impact parameters, in r_s. */
BHR_API int32_t bhr_hybrid_info(bhr_ctx *ctx, int32_t out_tiles[2], double out_band[2]);
/* Guards of the last BHR_MATH_HYBRID frame: out = {pixels its fast list handed over to the strict fix kernel, capacity of that
 * list}.  The count runs past the capacity -- pixels beyond it keep their fast value -- so out[0] > out[1] says that a view
 * needs a larger list than an eighth of the block's pixels.  {0, 0}: the frame ran without guards.  Synchronises. */
BHR_API int32_t bhr_hybrid_repairs(bhr_ctx *ctx, int32_t out[2]);
/* Device self-test of the strict march's hand-written exact sqrt / divide / divide-by-6 against the
 * compiler's IEEE sequences: out[0..2] = mismatches (sqrt over every f32 in [2^-80, 2^80); 1/x over
 * the same range plus a/b over 2^30 random pairs; x/6 over the same range), out[3] = comparisons made. */
BHR_API int32_t bhr_selftest(bhr_ctx *ctx, uint64_t out[4]);
/* forget the per-frame timing ring (call before a timed region) */
BHR_API int32_t bhr_timing_reset(bhr_ctx *ctx);
/* The last n timed bhr_render calls, oldest first: out[3 k .. 3 k + 2] = march start, march end, frame end of frame k in
 * milliseconds after the oldest frame's march start (HIP events on the frame slots' streams).  Synchronises. */
BHR_API int32_t bhr_timing_dump(bhr_ctx *ctx, float *out, int32_t n);
/* BHR_MIP_LDS=1 (environment of bhr_create): anti-aliased frames of the fast arithmetic stage the coarse levels of the disk
 * texture's mip stack in LDS -- as many of levels 3, 2, 1 as fit 44 KB -- and sample them from there (BASELINE.json's "mipmap
 * levels staged through LDS"; not the default: DESIGN.md section 4).  Returns the first staged level of the last such march,
 * -1 if it staged none (switch off, another arithmetic, or a texture whose level 3 alone exceeds the budget). */
BHR_API int32_t bhr_mip_lds_level(bhr_ctx *ctx);
/* Cost of each band of 8 rows in the last bhr_render(..., BHR_ROW_COSTS), in ray-step units: the ray-steps marched
 * plus 320 per wave-wide shading pass.  n = ceil(rows / 8) values.
 * The step count of a ray depends on the camera, the step size and the escape radius only -- not on the
 * textures -- so a small probe frame gives the cost profile of a large one (multigpu.balanced_row_blocks). */
BHR_API int32_t bhr_get_row_costs(bhr_ctx *ctx, uint64_t *out, int32_t n);
/* The same profile split by the arithmetic that took the steps: a math_mode 2 (hybrid) frame marches its tiles near the
 * photon ring with the strict kernel, ~2.2x the cost per step of the fast one -- row blocks of hybrid frames are balanced
 * on fast + 2.2 strict (multigpu.probe_row_costs).  A strict frame has everything in strict_out, a fast one in fast_out. */
BHR_API int32_t bhr_get_row_costs_split(bhr_ctx *ctx, uint64_t *fast_out, uint64_t *strict_out, int32_t n);

/* ---- multi-GPU row-block tiling (one process driving N devices) -----------
 * ctxs[k] renders rows [row0_k,row1_k) of the same image; blocks must be
 * contiguous, ordered and cover [0,height).  Marches all tiles concurrently,
 * exchanges the R = int(0.02*W) H-blurred halo rows between neighbours with
 * hipMemcpyPeerAsync, runs the V pass per tile and gathers the final tiles:
 * with BHR_GATHER_PEER into a full-frame f32 buffer on ctxs[0]'s device, with BHR_GATHER_U8 into a quantised u8 one
 * (peer copies over xGMI, no collective), and, if out_host != NULL, into out_host (H, W, 3) through per-device pinned
 * buffers.  Two schedules (csrc/group.hip), same bytes: pipelined -- the halo pull runs under the V pass of the rows
 * that need no halo, finished row chunks are pushed while the next chunk's V pass runs -- and serial; see
 * BHR_GROUP_SERIAL / BHR_GROUP_PIPELINED.  Synchronises. */
BHR_API int32_t bhr_group_render(bhr_ctx **ctxs, int32_t n, const bhr_camera *cam, uint32_t flags, float *out_host);
/* The same with only the tiles k with live[k] != 0 rendering; the others keep the buffers (halo rows, gathered rows,
 * glow rows) of the last call in which they were live.  live == NULL: all.  Times one tile of N end to end on one
 * device (bench.py tile_scaling): its counters' frame_ms then spans first march launch .. its rows landed on tile 0. */
BHR_API int32_t bhr_group_render_subset(bhr_ctx **ctxs, int32_t n, const bhr_camera *cam, uint32_t flags, float *out_host,
                                        const int32_t *live);
/* Waits for every frame the tiles have in flight (BHR_GROUP_ASYNC). */
BHR_API int32_t bhr_group_sync(bhr_ctx **ctxs, int32_t n);
/* The quantised frame the last bhr_group_render(..., BHR_GATHER_U8) gathered on this context's device: (H, W, 3) u8. */
BHR_API int32_t bhr_read_gathered_u8(bhr_ctx *ctx, uint8_t *out);

/* ---- multi-GPU row-block tiling, one PROCESS per tile ------------------------------------------------------------
 * The same frame and the same pipelined schedule with every tile in its own process (one rank per GPU under
 * torch.distributed.run, each seeing only its device).  Device buffers cross the process boundary as HIP IPC memory
 * handles, ordering as two counters per rank in host shared memory; no collective, no inter-process event.
 *   1. every rank: bhr_tile_export(ctx, BHR_GATHER_U8 and / or BHR_GATHER_PEER, &mine) -- rank 0 (the tile with row0 = 0)
 *      allocates the frame buffers and exports them with its H-blur planes;
 *   2. the ranks exchange the bhr_tile_handles records (any host channel: gloo all_gather, a file ...) and agree on a
 *      zero-initialised shared-memory area of world * BHR_TILE_SHM_WORDS uint64 words;
 *   3. every rank: bhr_tile_connect(ctx, rank, world, all, shm) -- opens the neighbours' planes and rank 0's buffers;
 *   4. per frame, every rank: bhr_tile_render(ctx, cam, flags) -- returns when EVERY rank's rows have landed;
 *      rank 0 then reads the frame with bhr_read_gathered / bhr_read_gathered_u8.
 * Tiles must be at least R = int(0.02 W) rows high; the lens flare (frame sums) needs bhr_group_render. */
#define BHR_TILE_SHM_WORDS 8
typedef struct {
    uint8_t hblur[64];          /* hipIpcMemHandle_t of the tile's H-blur planes (3, rows + 2R, W) */
    uint8_t gather_f32[64];     /* rank 0: the (H, W, 3) f32 frame buffer */
    uint8_t gather_u8[64];      /* rank 0: the (H, W, 3) u8 frame buffer */
    int32_t row0, rows, device;
    int32_t has_gather_f32, has_gather_u8;
    int32_t reserved[3];
} bhr_tile_handles;
BHR_API int32_t bhr_tile_export(bhr_ctx *ctx, uint32_t gather_flags, bhr_tile_handles *out);
BHR_API int32_t bhr_tile_connect(bhr_ctx *ctx, int32_t rank, int32_t world, const bhr_tile_handles *all, uint64_t *shm);
BHR_API int32_t bhr_tile_render(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags);
/* The frame the last bhr_group_render(..., BHR_GATHER_PEER) gathered on this context's device: (H, W, 3) f32. */
BHR_API int32_t bhr_read_gathered(bhr_ctx *ctx, float *out);

#ifdef __cplusplus
}
#endif
#endif /* BHR_H */
