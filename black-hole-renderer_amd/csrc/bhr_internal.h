// bhr_internal.h -- shared declarations of libbhr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bhr.h"
#include "../../include/bhr_disk_v2.h"
#include "../../include/bhr_lifecycle.h"

#define BHR_NUM_MIP_LEVELS 5  // generate_disk_mipmaps(levels=4) => 5 stored levels (render.py:2239-2240)
#define BHR_WAVE 64

// ---- constants of the reference (render.py:37-59) ---------------------------
#define BHR_RS 1.0f
#define BHR_G_FACTOR_CAP 1.5f
#define BHR_G_LUMINOSITY_POWER 1.5f
#define BHR_G_BRIGHTNESS_GAIN 0.38f
#define BHR_DISK_COLOR_TEMPERATURE 6000.0f
#define BHR_DISK_ALPHA_GAIN 6.0f
#define BHR_DISK_RADIAL_BRIGHTNESS_POWER 1.2f
#define BHR_DISK_RADIAL_BRIGHTNESS_MIN 0.2f
#define BHR_DISK_RADIAL_BRIGHTNESS_MAX 8.0f

// Ray-step counters: every wave adds its count with one atomic.  32 400 atomics to ONE address serialise in
// the memory system and put a floor of 0.45 ms under an fhd launch (measured; the fast march spends 0.36 ms);
// a counter is therefore a cell of 128 lanes, 256 bytes apart, indexed by block, summed when read.
// rows of zeros in front of and behind the (3, rows + 2R, W) H-blur planes of the exact-f32 post-pass (the IPC handles of
// csrc/group.hip name the allocation: the planes start this many rows in)
#define BHR_HBLUR_PAD_ROWS 16
#define BHR_STEP_LANES 128
#define BHR_STEP_STRIDE 32          // in u64 words
#define BHR_STEP_CELL (BHR_STEP_LANES * BHR_STEP_STRIDE)

// hybrid march: guard bands around the algorithm's switches; a lane inside one is re-marched strict (march.hip: march_tile_hybrid)
#ifndef BHR_LOD_GUARD
#define BHR_LOD_GUARD 1e-2f         // |lod - level boundary|: the fast differentials are good to ~1e-5 in lod at the BASELINE views, but a camera
                                    // 15-50 r_s away behind a long lens carries them through hundreds of steps -- fuzzed 512x320 views flipped
                                    // levels with 2e-3 (3 of 6 gone at 5e-3, all at 1e-2; no measurable cost at 4k: tools/dbg_flipviews.sh)
#endif
#ifndef BHR_R2_GUARD
#define BHR_R2_GUARD 4e-5f          // |r^2 - r_term^2| / r_term^2 of a plane-crossing step (fast positions are good to ~1e-6 relative)
#endif
#ifndef BHR_EDGE_GUARD
#define BHR_EDGE_GUARD 2e-5f        // |hit_r - r_edge| / r_edge at the disk's edges
#endif
#ifndef BHR_F_GUARD
#define BHR_F_GUARD 2e-6f           // |plane function at new_pos| / |new_pos|: a step that ends on the disk plane
#endif
#define BHR_FLUSH_COST 5u            // cost of one wave-wide shading pass in wave-steps (row-cost profile)
#define BHR_VOLUME_OPAQUE 0.9999f   // finite-thickness disk: accumulated opacity at which a ray stops sampling

#define BHR_PI_F 3.14159274101257324f      // (float)pi
#define BHR_TWO_PI_F 6.28318548202514648f  // (float)(2*pi)

// Scene textures as the march kernel sees them.
struct BhrScene {
    const float *skybox;   // (sky_h, sky_w, 3)
    int32_t sky_h, sky_w;
    const float4 *mips;    // packed levels 0..4, level l at mip_off[l], dims (mip_h[l], mip_w[l])
    int32_t mip_off[BHR_NUM_MIP_LEVELS];
    int32_t mip_h[BHR_NUM_MIP_LEVELS];
    int32_t mip_w[BHR_NUM_MIP_LEVELS];
    int32_t n_r, n_phi;
};

// Kernel argument block of the march (passed by value -> SGPRs).
struct BhrMarchArgs {
    float cp[3], cr[3], cu[3], cf[3];
    float pw, ph, r_esc, r_esc2;
    float e1[3], r0, A;      // fast build: e1 = cam/|cam|, r0 = |cam|, A = n.e1 with n = (0, -tan_t, 1)
    float h_base, r_inner, r_outer, t_offset;
    float tilt_rad, tan_t, sin_t, cos_t;
    float aa_strength;
    float max_affine, max_affine_u;   // max_affine_u = max_affine / h_base (fast build)
    int32_t max_iter;
    int32_t width, height;   // full image
    int32_t row0, rows;      // this context's row block
    BhrScene sc;
    float *bg;               // (rows, width, 3)
    float *disk;             // (rows, width, 3)
    _Float16 *diskp;         // non-null: the disk layer once more, cut into f16 halves in the bloom H pass's operand order (bloom.hip: pa)
    int32_t dp_yb, dp_gp, dp_g0;
    float *sum;              // with diskp: bg + disk per value (the V pass's combine reads one plane instead of two)
    unsigned long long *ray_steps;
    unsigned int *queue;     // persistent-wave work counter (zeroed before launch)
    const bhr_disk_v2_params *dv2;   // non-null: analytic Disk V2 source instead of the texture
    double dv2_norm_shear, dv2_norm_hotspot, dv2_t_peak;
    double vol_absorption, vol_grazing_gain, vol_h_max, vol_r_max;   // finite-thickness Disk V2
    int32_t vol_substeps;
    const int32_t *tile_order;   // launch slot -> tile (nullptr: row-major)
    unsigned long long *row_steps;   // BHR_ROW_COSTS: ray-steps per 8-row band (nullptr: not collected)
    unsigned long long *wave_stamps; // diagnostic (env BHR_WAVE_STAMPS): per wave s_memrealtime at start / end, steps, XCC|CU id
    int32_t n_tiles;         // 8x8 pixel tiles in the row block
    int32_t tiles_x;
    int32_t n_list;          // launch slots of this launch (= n_tiles, or the length of a hybrid / row-band sub-list)
    int32_t mip_lds_from;    // BHR_MIP_LDS: mip levels mip_lds_from .. BHR_NUM_MIP_LEVELS - 1 are staged in LDS (march_tile_mipstaged_kernel); -1: none
    unsigned int *fix_count; // hybrid march: pixels the guard kernel handed over to the strict fix kernel
    int32_t *fix_list;
    int32_t fix_cap;
};

// A partial march launch: the tiles of `d_list` only.  Set by the callers that split one march into several launches --
// bhr_launch_march_hybrid (strict list + fast list) and the pipelined row-block path (halo bands first) -- and consumed
// by the three compilations of the launcher in march.hip.
struct bhr_march_part {
    const int32_t *d_list;   // device list of this launch
    const int32_t *h_list;   // the same list on the host (the hybrid launcher partitions it further)
    int32_t n;
    int32_t id;              // which base list: 0 whole block, 1 halo bands, 2 the rows between them
    int32_t active, first, last;
    int32_t math_resolved;   // the arithmetic has been chosen by the caller (the two launches of a hybrid march)
    int32_t repair;          // 1: the fast object's guard kernel (marks lanes on a discontinuity, appends them to the fix list); 2: the strict fix kernel over that list
};

// what a frame's V pass stores (bhr_launch_bloom_v_rows); bhr_ensure_outputs re-runs it for layers nobody asked for up front
#define BHR_OUT_F32 1u    // clip(bg + disk + blur) as f32: what TaichiRenderer.render() returns
#define BHR_OUT_BLUR 2u   // blur_field
#define BHR_OUT_U8 4u     // the frame quantised as save_image does (render.py:423)

// The library's environment switches, read ONCE by bhr_create (nothing on the bhr_render path calls getenv).
struct bhr_options {
    int32_t frame_slots;        // BHR_FRAME_SLOTS: frames in flight per context (1 or 2, default 2)
    int32_t bloom_split;        // BHR_BLOOM_SPLIT: -1 by arithmetic (default), 0 exact f32 kernels always, 1 split-f16 always
    int32_t bloom_tiles;        // BHR_BLOOM_TILES: output tiles per wave of the split-f16 post-pass (1..8; 0 = by launch size), A/B runs
    int32_t hybrid_repair;      // BHR_HYBRID_REPAIR: -1 by view (default), 0 / 1 guards + strict fix list off / on
    double hybrid_band[2];      // BHR_HYBRID_BAND="lo,hi": strict band around b_c in r_s (default 0.085, 0.36)
    int32_t hybrid_band_set;
    double hybrid_pad;          // share of its own span of b a small tile is padded by in the strict-band test (BHR_HYBRID_PAD, default 0.5; hybrid.hip: tile_pad)
    int32_t hybrid_streams;     // BHR_HYBRID_STREAMS: 1 both lists of a hybrid march on one stream, 2 on two, -1 (default) 1 where two frame slots overlap frames, else 2
    int32_t calibrate_streams;  // BHR_CALIBRATE_STREAMS: 1 (default) a two-slot context picks slot 1's stream by timing candidates (api.hip)
    int32_t hybrid_swap;        // BHR_HYBRID_SWAP: 1 (default) the fast list on the frame's stream and the strict one on the second, 0 the other way round
    int32_t hybrid_classify;    // BHR_HYBRID_CLASSIFY: 1 (default) the tiles are classified and the launch order partitioned on the device, 0 on the host
    int32_t mip_lds;            // BHR_MIP_LDS=1: anti-aliased fast frames stage the coarse mip levels in LDS
    int32_t tile_order_rows;    // BHR_TILE_ORDER=row: row-major march launch order (A/B runs)
    int32_t tile_block;         // BHR_TILE_BLOCK: threads per march workgroup (64 / 128 / 256)
    int32_t group_threads;      // BHR_GROUP_THREADS: -1 by device layout (default), 0 / 1 one submitting thread / one per tile
    int32_t group_schedule;     // BHR_GROUP_SCHEDULE: -1 by flags (default), 0 serial, 1 pipelined
    int32_t aux_priority, aux_per_slot;   // BHR_AUX_STREAMS="<priority>,<per slot>": the second march stream(s) of hybrid frames
    int32_t stream_pad[3];      // BHR_STREAM_PAD="a,b,c": idle streams created in front of slot 0's, slot 1's, the second march streams (experiment)
};

// geometry of a context's split-f16 bloom buffers (bloom.hip)
struct bhr_split_geom {
    int32_t NT;            // 16-tap chunks either side of a tile: ceil(R / 16) + 1
    int32_t n_tx, WP;      // 32-pixel tiles along x, W rounded up to them
    int32_t YB, GP, g0;    // pa: 32-row blocks, 8-pixel groups per block (padded), zero groups in front
    int32_t t_first, n_ty; // global 32-row tile of the block's first row, tiles it touches
    int32_t pbr, GR;       // pb: global row of plane row 0 (multiple of 16, may be negative), 8-row groups
    size_t pa_halfs, pb_halfs;
    int32_t table_bytes;
};

// Frame slot: the buffers one frame in flight owns.  bhr_render alternates between two slots, each with its own
// stream, so that the tail and the bloom of frame n run under the march of frame n + 1; the scene (skybox, mip
// stack, comp planes ...) is shared and read-only while frames are in flight (bhr_enter orders every other entry
// point behind them).  Slot 1 is allocated at the second bhr_render; BHR_FRAME_SLOTS=1 keeps one slot on the
// context's own stream (round 1 behaviour, isolated per-kernel timing).
// two is the measured optimum (fhd strict: 1291 fps with one frame in flight, 1452 with two, 1389 / 1400 with three / four)
#define BHR_MAX_FRAME_SLOTS 2
struct bhr_frame_slot {
    hipStream_t stream;
    float *d_bg, *d_disk, *d_hblur, *d_blur, *d_final;
    float *d_hblur_base;       // the allocation d_hblur points BHR_HBLUR_PAD_ROWS rows into (zero rows in front of plane 0 and behind plane 2); exact-f32 bloom, on first use
    void *d_pa, *d_pb;         // split-f16 bloom (bloom.hip): the march's packed copy of the disk layer, the packed H-blur planes; on first use
    float *d_sum;              // ... and bg + disk of the frame (rows, W, 3); sum_valid: written by this frame's march / pack kernel
    int32_t sum_valid;
    uint8_t *d_final_u8;
    uint32_t have;             // BHR_OUT_* layers of the slot's last frame that are in memory (the V pass stores what was asked for; the rest on demand)
    int32_t frame_split, frame_with_bloom;   // how that frame's post-pass ran (bhr_ensure_outputs re-runs its V pass)
    unsigned int *d_queue;
    // lens flare scratch of the frame (flare.hip): glow rows, their transpose, chunk sums, the three frame sums
    float *d_glow_hw, *d_glow_wh, *d_flare_c0;
    double *d_flare_c12, *d_flare_sums;
    int64_t flare_glow_rows;
    hipEvent_t done;        // end of the slot's last bhr_render (and of frame work queued behind it: bhr_leave_frame)
    hipEvent_t march_done;  // end of its last march: the last reader of the scene (bhr_enter_scene_write); borrowed from the timing ring
    int32_t allocated;
    int32_t in_flight;      // rendered since the last join
};

struct bhr_ctx {
    bhr_config cfg;
    int32_t rows;
    hipStream_t stream;                 // the stream launchers use: the scene stream, or a slot's during bhr_render
    hipStream_t scene_stream;           // scene updates, read-backs, group renders
    bhr_frame_slot slots[BHR_MAX_FRAME_SLOTS];
    int32_t n_slots, next_slot, active_slot;
    int32_t two_slot_frames, streams_calibrated, calibrating;   // calibrate_slot_streams (api.hip)
    int32_t calib_choice, calib_fps[8];
    hipStream_t calib_idle[8];
    int32_t n_calib_idle;
    hipEvent_t scene_ev;                // scene stream -> slot stream ordering, recorded at every bhr_render
    hipEvent_t ev[8];
    hipEvent_t sync_ev;                 // bhr_sync polls it (no timing)
    // per-frame timing ring: 3 events per bhr_render (march start, march end, frame end)
    hipEvent_t ring_ev[BHR_TIMING_RING * 3];
    unsigned long long *d_steps_ring;   // one ray-step counter cell (BHR_STEP_CELL words) per ring slot
    unsigned long long *d_steps_fold;   // folded cells (BHR_TIMING_RING words)
    unsigned long long *v_zero_cell;    // counter cell the next bloom V launch clears (next ring slot), or null
    int64_t ring_head;                  // frames recorded since reset
    unsigned long long *last_steps_ptr; // counter the last march accumulated into
    int32_t cur_slot;                   // ring slot of the bhr_render in flight (-1: untimed launch)
    int32_t last_slot;                  // ring slot of the last completed bhr_render, -1 after a group render

    // scene
    float *d_skybox;
    int32_t sky_h, sky_w;
    float4 *d_mips;
    int32_t n_r, n_phi;
    int32_t mip_off[BHR_NUM_MIP_LEVELS], mip_h[BHR_NUM_MIP_LEVELS], mip_w[BHR_NUM_MIP_LEVELS];
    int64_t mip_texels;

    // texture pipeline
    int32_t bg_ready, bg_n_r, bg_n_phi, az_freq;
    float az_shear;
    float *d_comp;       // (13, n_r, n_phi)
    float *d_edge, *d_omega, *d_row_stats;
    float stats[2];
    float *d_noise_in, *d_noise_out;
    int64_t noise_cap;
    // device lifecycle (lifecycle.hip)
    float *d_pool;             // entity profile pool (bump allocated)
    int64_t pool_used, pool_cap;
    void *d_pairs;             // per-call pair tables
    size_t pairs_cap;
    float *d_stats_scratch;    // density | temp_struct | histogram | row results
    size_t stats_scratch_elems;
    int32_t stats_prepared;
    // analytic disk source
    int32_t disk_source;
    bhr_disk_v2_params *d_dv2_params;
    double dv2_norm[3];        // max|raw shear|, max|raw hotspot| on the reference grid, peak T_mid
    double vol_opts[4];        // absorption Ca, grazing gain kg, max half thickness, max spherical radius of the volume
    int32_t vol_substeps;

    // frame buffers for rows [row0,row1)
    float *d_bg, *d_disk;      // (rows, W, 3)
    float *d_hblur;            // planar (3, rows + 2R, W): rows [row0-R, row1+R)
    float *d_blur;             // (rows, W, 3)
    float *d_final;            // (rows, W, 3)
    uint8_t *d_final_u8;       // (rows, W, 3)
    float *d_wtab;             // bloom weights (3, R + pad)
    float *d_wext;             // unfolded weights (3, 2 R4 + 8)
    unsigned short *d_w16;     // split-f16 weight table: 3 channels x 2 halves x 8 shifted copies (bloom.hip: bloom_tables_kernel)
    void *d_pa, *d_pb;         // the active slot's packed bloom operands (null until a split frame needs them)
    float *d_sum;
    int32_t mip_lds_from;      // first mip level the last anti-aliased fast march staged in LDS (BHR_MIP_LDS), -1: none
    int32_t bloom_split;       // post-pass of the current frame: 0 exact f32 kernels (strict), 1 split-f16 matrix-core kernels (fast / hybrid)
    int32_t split_ok;          // the context's radius fits the split kernels' table (R <= 176)
    uint32_t out_want;         // BHR_OUT_* the frames of this context store (bhr_set_outputs; default: the f32 frame)
    // rows of this block that neighbouring row blocks need for their V pass: the H pass writes them straight into those
    // blocks' planes (set by group.hip for the duration of a group / tile render)
    struct { void *pb; int32_t pbr, gr; } mirrors[6];
    int32_t n_mirrors;
    bhr_options opt;           // the BHR_* environment switches, read once by bhr_create
    float *d_wsum_h;           // (3, W) in-bounds weight sums, then (3, W) the split H pass's multiplier 2^-10 / sum
    float *d_wsum_v;           // (3, H), then (3, H) the split V pass's multiplier 2^-24 / sum
    int32_t bloom_R, bloom_ready;
    unsigned long long *d_ray_steps;
    unsigned int *d_queue;
    unsigned long long *d_row_steps;   // ray-steps per 8-row band of the last BHR_ROW_COSTS launch
    int32_t *d_tile_order;     // march launch order of the 8x8 tiles
    int32_t *h_tile_order;     // host copy (malloc)
    int32_t tile_order_n;
    bhr_march_part part;       // partial launch in progress (inactive: whole block)
    // second march stream (lowest priority): the other half of a split march -- the fast tiles of a hybrid march, the middle
    // rows of a pipelined row block -- runs beside the first half instead of behind its ragged end (bhr_aux_fork / _join)
    hipStream_t aux_stream;    // the active slot's second march stream (set by bhr_aux_fork)
    hipStream_t aux_streams[BHR_MAX_FRAME_SLOTS];
    int32_t aux_per_slot;
    hipEvent_t aux_fork[BHR_MAX_FRAME_SLOTS], aux_done[BHR_MAX_FRAME_SLOTS];
    void *hybrid;              // hybrid.hip: tile classification cache
    unsigned int *fix_count;   // fix list of the hybrid march being launched (owned by hybrid.hip, per frame slot)
    int32_t *fix_list;
    int32_t fix_cap;
    void *pipe;                // group.hip: streams, events and band lists of the pipelined row-block path
    uint8_t *d_gather_u8;      // (H, W, 3) u8: quantised frame gathered from the tiles (BHR_GATHER_U8), on tile 0
    // lens flare (flare.hip)
    float *d_glow_hw;          // glow rows (rows, W); (H, W) on the context that sums the frame
    int64_t flare_glow_rows;
    float *d_glow_wh;          // (W, H): the reference's memory order, summed the way NumPy sums it
    float *d_flare_c0;         // per 8192-element chunk: sum glow (f32)
    double *d_flare_c12;       // per chunk: sum x glow | sum y glow
    int32_t *d_flare_prog;     // pairwise tree of the ragged last chunk
    double *d_flare_sums;      // S0, S1, S2
    float *d_gather;           // (H, W, 3): full frame gathered from the tiles of a group render (BHR_GATHER_PEER), on tile 0
    void *png_dev;             // device PNG encoder state (png_device.hip), created on first use
    void *pop_host;            // pinned staging of bhr_accumulate_population (lifecycle.hip)
    float *h_pinned;           // staging for readbacks
    size_t h_pinned_bytes;

    bhr_counters counters;
    int32_t last_flags;
    int32_t timing_valid;
    int32_t group_time_march;  // group / tile renders: also record the march-end event (BHR_GROUP_TIME_MARCH); off, the tile's stream carries no event between march and H pass
    int32_t march_end_recorded;
};

// error plumbing (api.hip)
int32_t bhr_fail(int32_t code, const char *fmt, ...);
#define BHR_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return bhr_fail(BHR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                            \
    } while (0)

#define BHR_TRY(expr)                    \
    do {                                 \
        int32_t rc__ = (expr);           \
        if (rc__ != BHR_OK) return rc__; \
    } while (0)

// Every API entry point except bhr_render: selects the device and orders the scene stream behind the frames in
// flight, so that scene writes, read-backs and stand-alone passes see (and never race with) a finished frame.
int32_t bhr_enter(bhr_ctx *ctx);
// Entry points that touch only the component planes / entity pool (comp, d_pool: read by no frame kernel) -- the
// per-frame background and entity-layer passes -- select the scene stream WITHOUT joining the frames in flight, so that
// the next frame's texture work runs beside the current march; bhr_compose_texture (full bhr_enter) is the join.
int32_t bhr_enter_components(bhr_ctx *ctx);
// Work that only reads the last rendered frame (quantise, PNG encode, copy out) rides the stream that rendered it:
// bhr_enter_frame points ctx->stream at that slot's stream (ordered behind the scene stream), bhr_leave_frame
// re-records the slot's completion event and restores the scene stream.  The scene stream stays free meanwhile.
// bhr_compose_texture rewrites the disk texture and its mips, which only the MARCH of a frame in flight reads: the scene
// stream waits for the marches, not for the bloom / flare / PNG work behind them.
int32_t bhr_enter_scene_write(bhr_ctx *ctx);
int32_t bhr_enter_frame(bhr_ctx *ctx);
int32_t bhr_leave_frame(bhr_ctx *ctx);

// launchers (each lives next to its kernels)
int32_t bhr_launch_march(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags);         // dispatches on math_mode
int32_t bhr_launch_march_strict(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags);  // march_strict.o
int32_t bhr_march_resources_strict(int32_t *vgprs, int32_t *lds, int32_t diff);
int32_t bhr_launch_march_strict_ilp(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags);   // march_strict_ilp.o
int32_t bhr_march_resources_strict_ilp(int32_t *vgprs, int32_t *lds, int32_t diff);
int32_t bhr_selftest_strict(bhr_ctx *ctx, unsigned long long *d_out4);
int32_t bhr_ensure_tile_order(bhr_ctx *ctx);                                               // march.o: builds d_/h_tile_order
int32_t bhr_launch_march_hybrid(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags);     // hybrid.hip
void bhr_hybrid_free(bhr_ctx *ctx);
int32_t bhr_hybrid_active_list(bhr_ctx *ctx, const int32_t **list, int32_t *n);   // hybrid.o: the active slot's partitioned launch order (tests)
int32_t bhr_bloom_prepare(bhr_ctx *ctx);
int32_t bhr_split_nt(int32_t R);
void bhr_split_geometry(const bhr_ctx *ctx, bhr_split_geom *g);
int32_t bhr_launch_bloom_pack(bhr_ctx *ctx);                               // d_disk -> d_pa
int32_t bhr_launch_bloom_h(bhr_ctx *ctx);
// V pass + combine over local rows [r0, r1) storing the BHR_OUT_* layers in `want`; gather_u8 / gather_f32 non-null: the
// u8 / f32 rows go into that (H, W, 3) frame buffer (a row-block gather, possibly on a peer device) instead of the context's own
int32_t bhr_launch_bloom_v_rows(bhr_ctx *ctx, int32_t with_bloom, int32_t r0, int32_t r1, uint32_t want, uint8_t *gather_u8, float *gather_f32);
int32_t bhr_bloom_v_tile_rows(bhr_ctx *ctx);                               // output rows per V-pass block
int32_t bhr_activate_slot(bhr_ctx *ctx, int32_t k);                        // api.hip: points the launchers at frame slot k (allocating it)
// api.hip: decides the frame's arithmetic / post-pass kernels and makes sure their buffers exist (before the march is launched)
int32_t bhr_frame_begin(bhr_ctx *ctx, uint32_t flags);
// api.hip: the whole-block V pass of a frame into the context's own buffers, recording what it stored
int32_t bhr_frame_post(bhr_ctx *ctx, int32_t with_bloom, uint32_t want);
// api.hip: makes the BHR_OUT_* layers in `need` of the active slot's last frame exist (re-runs its V pass for what is missing)
int32_t bhr_ensure_outputs(bhr_ctx *ctx, uint32_t need);
void bhr_pipe_free(bhr_ctx *ctx);                                          // group.hip
int32_t bhr_ensure_pinned(bhr_ctx *ctx, size_t bytes);                     // api.hip
// fork: the aux stream waits for everything ctx->stream has been given so far; join: ctx->stream waits for the aux stream
int32_t bhr_streams_share_queue(hipStream_t a, hipStream_t b, int32_t *share);   // api.o: probe (two one-lane kernels, <= 4 ms)
int32_t bhr_aux_fork(bhr_ctx *ctx);
int32_t bhr_aux_join(bhr_ctx *ctx);
int32_t bhr_launch_flare_glow(bhr_ctx *ctx, bool whole_frame);       // flare.hip
int32_t bhr_launch_flare_sums(bhr_ctx *ctx);
int32_t bhr_launch_flare_apply(bhr_ctx *ctx, const double *sums);    // sums == nullptr: device-resident totals
int32_t bhr_launch_quantize(bhr_ctx *ctx);                           // api.hip: the frame's u8 rows, on the stream (from the V pass, or FINAL -> u8)
// png_device.hip: (rows, W, 3) u8 at d_rgb -> PNG file bytes at d_out on ctx->stream; d_meta (4 words) = {length, error, ..}
int32_t bhr_launch_png_encode(bhr_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_out, int64_t cap, uint32_t *d_meta);
void bhr_png_dev_free(bhr_ctx *ctx);
void bhr_population_free(bhr_ctx *ctx);                              // lifecycle.hip
int32_t bhr_launch_build_mips(bhr_ctx *ctx);
int32_t bhr_launch_background(bhr_ctx *ctx, float t);
int32_t bhr_launch_compose(bhr_ctx *ctx, float t_offset, int32_t enable_rt, float color_temp);
int32_t bhr_launch_fill(bhr_ctx *ctx, float *dst, int64_t n, float v);
int32_t bhr_launch_noise(bhr_ctx *ctx, int64_t n, int32_t mode, int32_t octaves, float pers, float lac);
int32_t bhr_march_resources(int32_t *vgprs, int32_t *lds, int32_t diff);
int32_t bhr_launch_disk_v2(bhr_ctx *ctx, const bhr_disk_v2_params *p, const double *d_r, const double *d_z,
                           const double *d_phi, int64_t n, int32_t field, double *d_out, double *d_aux,
                           double *d_maxabs, double norm0, double norm1);
