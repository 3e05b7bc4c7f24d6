// group.hip -- one frame in N row blocks (BASELINE.json configs[3]): bhr_group_render / bhr_group_render_subset.
//
// The reference has no multi-GPU code; what has to be kept is the frame: march -> bloom H -> bloom V ->
// clip(bg + disk + blur) (-> lens flare) of render.py:3865-3923, 3022-3114.  A row block needs R = int(0.02 W) rows of
// its neighbours' H-blurred disk layer for the V pass; everything else is local.  Two schedules produce the same bytes:
//
//  * serial (BHR_GROUP_SERIAL, round 1-2): per tile march -> H pass -> halo pull -> V pass -> gather, each step behind
//    the previous one on the tile's stream.  At 8 tiles of an 8k frame the steps after the march (H 0.12, halo 0.1,
//    V 0.34, f32 gather 0.33 ms) are a 0.9 ms tail behind a 3.0 ms march.
//  * pipelined (default): per tile three streams.
//      march stream  the march in TWO launches over complementary tile lists: the halo bands (the R rows next to each
//                    neighbour) first, the rows between them second;
//      post stream   H pass of the bands as soon as the first launch is done -> `halo_ready`; H pass of the rest after
//                    the second; V pass + combine in row chunks, each chunk's quantised bytes written by the V kernel's
//                    epilogue (BHR_GATHER_U8);
//      copy stream   pulls the neighbours' halo rows behind THEIR `halo_ready` -- under the march of the middle rows --
//                    and pushes every finished chunk into the frame buffer on tile 0's device while the next chunk's V
//                    kernel runs.  u8 rows: 12.4 MB per 8k tile instead of 49.8 MB of f32.
//    The lens flare needs the frame's three sums (a host read-back on tile 0), so with BHR_LENS_FLARE the chunks stay on
//    the device until the flare has been applied; march / H / halo / V are pipelined all the same.
//
// All exchanges are hipMemcpyPeerAsync between neighbours or onto tile 0 (xGMI point to point), no collective.  One
// process drives the devices; `live` (bhr_group_render_subset) restricts a call to some tiles while the others keep the
// buffers of the previous full render -- how one tile of eight is timed end to end on a single GPU (bench.py
// tile_scaling.tile_tail_ms).
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "bhr_internal.h"

namespace {

constexpr int PIPE_MAX_CHUNKS = 16;

struct TilePipe {
    hipStream_t post, copy;
    hipEvent_t march_a, march_b, halo_ready, h_all, halo_in, v_done[PIPE_MAX_CHUNKS], landed, post_done, glow_ready;
    int32_t *d_band, *h_band;     // halo-band tiles first, then the rest; launch order kept inside both
    int32_t n_band, n_rest;
    int32_t band_top;             // local rows [0, band_top) form the upper halo band (0: none)
    int32_t band_bot;             // local rows [band_bot, rows) the lower one (rows: none)
    int32_t key;                  // 4 | has_up | has_down << 1 once the lists are built
    // one-process-per-tile variant (bhr_tile_connect): the neighbours' H-blur planes and tile 0's frame buffers, opened
    // from their IPC handles; counters in host shared memory for the hand-shakes
    int32_t linked, rank, world;
    float *nb_hblur[2];           // [0] the tile above, [1] the tile below (nullptr: none)
    int32_t nb_rows[2];
    float *gather_f32;            // frame buffers on tile 0's device (its own pointers on rank 0)
    uint8_t *gather_u8;
    void *opened[4];              // what hipIpcCloseMemHandle has to release
    volatile uint64_t *shm;       // BHR_TILE_SHM_WORDS words per rank: [0] frames whose halo bands are H-blurred, [1] frames done
    uint64_t frame;               // frames rendered through bhr_tile_render
};

template <typename T>
int32_t dev_alloc(T **p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return bhr_fail(BHR_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    return BHR_OK;
}

int32_t ensure_pipe(bhr_ctx *ctx, bool has_up, bool has_down) {
    TilePipe *p = (TilePipe *)ctx->pipe;
    if (!p) {
        p = new TilePipe();
        memset(p, 0, sizeof(*p));
        ctx->pipe = p;
        int lo = 0, hi = 0;
        BHR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));       // hi = greatest priority (numerically lowest)
        // the post and copy streams carry short, latency-critical work next to a long march: highest priority
        const char *e = getenv("BHR_PIPE_PRIORITY");
        const int prio = (e && atoi(e) == 0) ? lo : hi;
        BHR_HIP(hipStreamCreateWithPriority(&p->post, hipStreamNonBlocking, prio));
        BHR_HIP(hipStreamCreateWithPriority(&p->copy, hipStreamNonBlocking, prio));
        hipEvent_t *evs[] = {&p->march_a, &p->march_b, &p->halo_ready, &p->h_all, &p->halo_in, &p->landed, &p->post_done, &p->glow_ready};
        for (hipEvent_t *ev : evs) BHR_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
        for (auto &ev : p->v_done) BHR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int key = 4 | (has_up ? 1 : 0) | (has_down ? 2 : 0);
    if (p->key == key) return BHR_OK;
    BHR_TRY(bhr_ensure_tile_order(ctx));
    const int rows = ctx->rows, R = ctx->bloom_R, n_tiles = ctx->tile_order_n, tiles_x = (ctx->cfg.width + 7) / 8;
    const int band = ((R + 7) / 8) * 8;                           // whole 8-row tile rows that cover R rows
    p->band_top = has_up ? (band < rows ? band : rows) : 0;
    int bot = has_down ? ((rows - R) > 0 ? ((rows - R) / 8) * 8 : 0) : rows;
    if (bot < p->band_top) bot = p->band_top;
    p->band_bot = bot;
    if (!p->h_band) {
        p->h_band = (int32_t *)malloc((size_t)n_tiles * sizeof(int32_t));
        if (!p->h_band) return bhr_fail(BHR_ERR_NOMEM, "tile pipe: out of host memory");
        BHR_TRY(dev_alloc(&p->d_band, (size_t)n_tiles));
    }
    int n = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int k = 0; k < n_tiles; ++k) {
            const int t = ctx->h_tile_order[k], y = (t / tiles_x) * 8;
            const bool in_band = y < p->band_top || y >= p->band_bot;
            if (in_band == (pass == 0)) p->h_band[n++] = t;
        }
        if (pass == 0) p->n_band = n;
    }
    p->n_rest = n_tiles - p->n_band;
    BHR_HIP(hipMemcpyAsync(p->d_band, p->h_band, (size_t)n_tiles * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    p->key = key;
    return BHR_OK;
}

// Direct xGMI copies between the tiles' devices: without peer access hipMemcpyPeerAsync stages through
// host memory.  Tried once per ordered device pair; a refusal is not an error (the staged copy still works).
void enable_peer_access(bhr_ctx **ctxs, int32_t n) {
    static bool tried[64][64];
    for (int k = 0; k < n; ++k)
        for (int q = 0; q < n; ++q) {
            const int a = ctxs[k]->cfg.device, b = ctxs[q]->cfg.device;
            if (a == b || a < 0 || b < 0 || a >= 64 || b >= 64 || tried[a][b]) continue;
            tried[a][b] = true;
            int can = 0;
            if (hipSetDevice(a) != hipSuccess || hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                (void)hipGetLastError();
                continue;
            }
            if (hipDeviceEnablePeerAccess(b, 0) != hipSuccess) (void)hipGetLastError();   // e.g. already enabled
        }
}

// runs f(k) for every live tile: one host thread per tile when the tiles sit on distinct devices (a single thread
// needs ~30 us per device, a quarter of a millisecond of skew at 8 devices), in order otherwise
template <typename F>
int32_t for_tiles(int n, const int32_t *live, bool threaded, F f) {
    if (!threaded) {
        for (int k = 0; k < n; ++k)
            if (!live || live[k]) BHR_TRY(f(k));
        return BHR_OK;
    }
    std::vector<int32_t> rcs((size_t)n, BHR_OK);
    std::vector<std::string> errs((size_t)n);
    std::vector<std::thread> th;
    int first = -1;
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        if (first < 0) { first = k; continue; }
        th.emplace_back([&, k] { rcs[(size_t)k] = f(k); if (rcs[(size_t)k] != BHR_OK) errs[(size_t)k] = bhr_last_error(); });
    }
    if (first >= 0) {
        rcs[(size_t)first] = f(first);
        if (rcs[(size_t)first] != BHR_OK) errs[(size_t)first] = bhr_last_error();
    }
    for (auto &t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (rcs[(size_t)k] != BHR_OK) return bhr_fail(rcs[(size_t)k], "tile %d: %s", k, errs[(size_t)k].c_str());
    return BHR_OK;
}

// Halo pull of tile k: up to R rows of the H-blurred planes (planar (3, rows + 2R, W)) from the tiles above and below
// (a tile thinner than R passes the request on to the next one), queued on `stream` behind the producers' events.
// pipelined: wait for the neighbour's `halo_ready` when its halo band covers the rows, for `h_all` otherwise.
int32_t queue_halo_pull(bhr_ctx **ctxs, int n, int k, hipStream_t stream, bool pipelined) {
    bhr_ctx *me = ctxs[k];
    const size_t R = me->bloom_R, W = me->cfg.width, my_rows = me->rows;
    for (int side = 0; side < 2; ++side) {
        size_t need = R, got = 0;                  // side 0: rows above me come from tiles k-1, k-2, ...; side 1: below
        int q = side == 0 ? k - 1 : k + 1;
        while (need > 0 && q >= 0 && q < n) {
            bhr_ctx *nb = ctxs[q];
            const size_t take = (size_t)nb->rows < need ? (size_t)nb->rows : need;
            if (pipelined) {
                const TilePipe *np = (const TilePipe *)nb->pipe;
                if (np) {                          // a tile that has never rendered pipelined has nothing in flight
                    const bool in_band = side == 0 ? (int)take <= nb->rows - np->band_bot : (int)take <= np->band_top;
                    BHR_HIP(hipStreamWaitEvent(stream, in_band ? np->halo_ready : np->h_all, 0));
                }
            } else {
                BHR_HIP(hipStreamWaitEvent(stream, nb->ev[3], 0));
            }
            const size_t nb_plane = ((size_t)nb->rows + 2 * R) * W, my_plane = (my_rows + 2 * R) * W;
            const size_t src_row = side == 0 ? R + nb->rows - take : R;          // neighbour's own rows live at [R, R + rows)
            const size_t dst_row = side == 0 ? R - got - take : R + my_rows + got;
            for (int c = 0; c < 3; ++c)
                BHR_HIP(hipMemcpyPeerAsync(me->d_hblur + c * my_plane + dst_row * W, me->cfg.device,
                                           nb->d_hblur + c * nb_plane + src_row * W, nb->cfg.device, take * W * sizeof(float), stream));
            need -= take;
            got += take;
            q += side == 0 ? -1 : 1;
        }
    }
    return BHR_OK;
}

// Lens flare of a row-block frame (render.py:3920-4028): tile 0 collects every tile's glow rows, sums the frame in
// NumPy's order, the three totals go to every tile's apply launch.  Runs on the tiles' main streams.
int32_t flare_pass(bhr_ctx **ctxs, int n, const int32_t *live) {
    bhr_ctx *head = ctxs[0];
    const int W = head->cfg.width;
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_TRY(bhr_launch_flare_glow(ctxs[k], k == 0));
        BHR_HIP(hipEventRecord(ctxs[k]->ev[3], ctxs[k]->stream));
    }
    BHR_HIP(hipSetDevice(head->cfg.device));
    for (int k = 1; k < n; ++k) {
        if (live && !live[k]) continue;            // a tile that is not live left its glow rows on tile 0 last time
        BHR_HIP(hipStreamWaitEvent(head->stream, ctxs[k]->ev[3], 0));
        BHR_HIP(hipMemcpyPeerAsync(head->d_glow_hw + (size_t)ctxs[k]->cfg.row0 * W, head->cfg.device, ctxs[k]->d_glow_hw,
                                   ctxs[k]->cfg.device, (size_t)ctxs[k]->rows * W * sizeof(float), head->stream));
    }
    BHR_TRY(bhr_launch_flare_sums(head));
    double tot[3];
    BHR_HIP(hipMemcpyAsync(tot, head->d_flare_sums, sizeof(tot), hipMemcpyDeviceToHost, head->stream));
    BHR_HIP(hipStreamSynchronize(head->stream));
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_TRY(bhr_launch_flare_apply(ctxs[k], tot));
    }
    return BHR_OK;
}

int32_t ensure_gather(bhr_ctx *head, uint32_t flags) {
    const size_t px3 = (size_t)head->cfg.height * head->cfg.width * 3;
    BHR_HIP(hipSetDevice(head->cfg.device));
    if ((flags & BHR_GATHER_PEER) && !head->d_gather) BHR_TRY(dev_alloc(&head->d_gather, px3));
    if ((flags & BHR_GATHER_U8) && !head->d_gather_u8) BHR_TRY(dev_alloc(&head->d_gather_u8, px3));
    return BHR_OK;
}

// push local rows [r0, r1) of tile `t` into the frame buffers on `head` (f32 and / or u8), on `stream`
int32_t queue_push(bhr_ctx *head, bhr_ctx *t, uint32_t flags, int r0, int r1, hipStream_t stream) {
    const size_t W3 = (size_t)t->cfg.width * 3, off = (size_t)r0 * W3, cnt = (size_t)(r1 - r0) * W3;
    const size_t dst = (size_t)(t->cfg.row0 + r0) * W3;
    if (flags & BHR_GATHER_U8)
        BHR_HIP(hipMemcpyPeerAsync(head->d_gather_u8 + dst, head->cfg.device, t->d_final_u8 + off, t->cfg.device, cnt, stream));
    if (flags & BHR_GATHER_PEER)
        BHR_HIP(hipMemcpyPeerAsync(head->d_gather + dst, head->cfg.device, t->d_final + off, t->cfg.device, cnt * sizeof(float), stream));
    return BHR_OK;
}

int32_t finish(bhr_ctx **ctxs, int n, const int32_t *live, float *out_host) {
    const size_t W = ctxs[0]->cfg.width;
    // gather to the host -- every device copies into its own pinned buffer concurrently, the host then assembles the
    // frame (a pageable destination would serialise the DMA streams)
    if (out_host)
        for (int k = 0; k < n; ++k) {
            if (live && !live[k]) continue;
            BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
            const size_t bytes = (size_t)ctxs[k]->rows * W * 3 * sizeof(float);
            BHR_TRY(bhr_ensure_pinned(ctxs[k], bytes));
            BHR_HIP(hipMemcpyAsync(ctxs[k]->h_pinned, ctxs[k]->d_final, bytes, hipMemcpyDeviceToHost, ctxs[k]->stream));
        }
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_HIP(hipStreamSynchronize(ctxs[k]->stream));
        if (out_host)
            memcpy(out_host + (size_t)ctxs[k]->cfg.row0 * W * 3, ctxs[k]->h_pinned, (size_t)ctxs[k]->rows * W * 3 * sizeof(float));
    }
    return BHR_OK;
}

// ---- serial schedule (rounds 1-2) ---------------------------------------------------------------------------------
int32_t render_serial(bhr_ctx **ctxs, int n, const bhr_camera *cam, uint32_t flags, float *out_host, const int32_t *live,
                      bool threaded) {
    const int with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    BHR_TRY(for_tiles(n, live, threaded, [&](int k) -> int32_t {
        BHR_TRY(bhr_enter(ctxs[k]));
        ctxs[k]->cur_slot = -1;
        ctxs[k]->last_slot = -1;
        BHR_TRY(bhr_launch_march(ctxs[k], cam, flags));
        if (with_bloom) BHR_TRY(bhr_launch_bloom_h(ctxs[k]));
        BHR_HIP(hipEventRecord(ctxs[k]->ev[3], ctxs[k]->stream));
        return BHR_OK;
    }));
    if (with_bloom && n > 1)
        for (int k = 0; k < n; ++k) {
            if (live && !live[k]) continue;
            BHR_TRY(bhr_enter(ctxs[k]));
            BHR_TRY(queue_halo_pull(ctxs, n, k, ctxs[k]->stream, false));
        }
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_TRY(bhr_enter(ctxs[k]));
        BHR_TRY(bhr_launch_bloom_v(ctxs[k], with_bloom));
        ctxs[k]->last_flags = (int32_t)flags;
        ctxs[k]->timing_valid = 1;
    }
    if (flags & BHR_LENS_FLARE) BHR_TRY(flare_pass(ctxs, n, live));
    if (flags & (BHR_GATHER_PEER | BHR_GATHER_U8)) {
        BHR_TRY(ensure_gather(ctxs[0], flags));
        for (int k = 0; k < n; ++k) {
            if (live && !live[k]) continue;
            BHR_TRY(bhr_enter(ctxs[k]));
            if (flags & BHR_GATHER_U8) BHR_TRY(bhr_launch_quantize(ctxs[k]));
            BHR_TRY(queue_push(ctxs[0], ctxs[k], flags, 0, ctxs[k]->rows, ctxs[k]->stream));
        }
    }
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_HIP(hipEventRecord(ctxs[k]->ev[2], ctxs[k]->stream));
    }
    return finish(ctxs, n, live, out_host);
}

// ---- pipelined schedule ---------------------------------------------------------------------------------------------
int32_t render_pipelined(bhr_ctx **ctxs, int n, const bhr_camera *cam, uint32_t flags, float *out_host, const int32_t *live,
                         bool threaded) {
    const int with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    const bool flare = (flags & BHR_LENS_FLARE) != 0;
    const bool gather = (flags & (BHR_GATHER_PEER | BHR_GATHER_U8)) != 0;
    bhr_ctx *head = ctxs[0];
    if (gather) BHR_TRY(ensure_gather(head, flags));
    int n_chunks_want = 3;
    if (const char *e = getenv("BHR_TILE_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= PIPE_MAX_CHUNKS) n_chunks_want = v; }

    // phase 1: march (halo bands first) and H passes
    BHR_TRY(for_tiles(n, live, threaded, [&](int k) -> int32_t {
        bhr_ctx *c = ctxs[k];
        BHR_TRY(bhr_enter(c));
        c->cur_slot = -1;
        c->last_slot = -1;
        BHR_TRY(ensure_pipe(c, with_bloom && k > 0, with_bloom && k < n - 1));
        TilePipe *p = (TilePipe *)c->pipe;
        const bool split = with_bloom && p->n_band > 0 && p->n_rest > 0 && !(flags & (BHR_PERSISTENT | BHR_ROW_COSTS));
        auto march = [&]() -> int32_t {
            if (!split) {
                BHR_TRY(bhr_launch_march(c, cam, flags));
                BHR_HIP(hipEventRecord(p->march_a, c->stream));
                BHR_HIP(hipEventRecord(p->march_b, c->stream));
                return BHR_OK;
            }
            // the halo bands on the tile's stream, the rows between them on its low-priority second stream: they start
            // together, the bands' workgroups are dispatched first, the rest fills the slots the bands leave -- no drain
            // between the two launches.  Bracket (start event, counter clear / end event) = empty first / last part.
            bhr_march_part part;
            memset(&part, 0, sizeof(part));
            part.active = 1;
            part.id = 1; part.first = 1;
            c->part = part;
            BHR_TRY(bhr_launch_march(c, cam, flags));                              // prologue
            const bool two = !(getenv("BHR_PIPE_MARCH_STREAMS") && atoi(getenv("BHR_PIPE_MARCH_STREAMS")) == 1);
            if (two) BHR_TRY(bhr_aux_fork(c));
            part.d_list = p->d_band; part.h_list = p->h_band; part.n = p->n_band; part.first = 0; part.last = 0;
            c->part = part;
            BHR_TRY(bhr_launch_march(c, cam, flags));
            BHR_HIP(hipEventRecord(p->march_a, c->stream));
            hipStream_t main_stream = c->stream;
            if (two) c->stream = c->aux_stream;
            part.d_list = p->d_band + p->n_band; part.h_list = p->h_band + p->n_band; part.n = p->n_rest; part.id = 2;
            c->part = part;
            const int32_t rc_rest = bhr_launch_march(c, cam, flags);
            c->stream = main_stream;
            BHR_TRY(rc_rest);
            if (two) BHR_TRY(bhr_aux_join(c));
            BHR_HIP(hipEventRecord(p->march_b, c->stream));
            part.d_list = nullptr; part.h_list = nullptr; part.n = 0; part.last = 1;
            c->part = part;
            BHR_TRY(bhr_launch_march(c, cam, flags));                              // epilogue: the end event
            return BHR_OK;
        };
        const int32_t rc_m = march();
        c->part.active = 0;
        BHR_TRY(rc_m);
        // H passes on the post stream: the halo bands as soon as their march is done, the rest behind the second launch
        auto h_passes = [&]() -> int32_t {
            BHR_HIP(hipStreamWaitEvent(p->post, p->march_a, 0));
            if (!split) BHR_HIP(hipStreamWaitEvent(p->post, p->march_b, 0));
            if (with_bloom && split) {
                BHR_TRY(bhr_launch_bloom_h_rows(c, 0, p->band_top));
                BHR_TRY(bhr_launch_bloom_h_rows(c, p->band_bot, c->rows));
            } else if (with_bloom) {
                BHR_TRY(bhr_launch_bloom_h(c));
            }
            BHR_HIP(hipEventRecord(p->halo_ready, p->post));
            BHR_HIP(hipStreamWaitEvent(p->post, p->march_b, 0));
            if (with_bloom && split) BHR_TRY(bhr_launch_bloom_h_rows(c, p->band_top, p->band_bot));
            BHR_HIP(hipEventRecord(p->h_all, p->post));
            return BHR_OK;
        };
        hipStream_t main_stream = c->stream;
        c->stream = p->post;
        const int32_t rc_h = h_passes();
        c->stream = main_stream;
        return rc_h;
    }));

    // phase 2: halo pulls on the copy streams, behind the neighbours' halo_ready
    if (with_bloom && n > 1)
        for (int k = 0; k < n; ++k) {
            if (live && !live[k]) continue;
            TilePipe *p = (TilePipe *)ctxs[k]->pipe;
            BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
            BHR_TRY(queue_halo_pull(ctxs, n, k, p->copy, true));
            BHR_HIP(hipEventRecord(p->halo_in, p->copy));
        }

    // phase 3: V pass + combine in row chunks on the post stream; every finished chunk is pushed by the copy stream
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        bhr_ctx *c = ctxs[k];
        TilePipe *p = (TilePipe *)c->pipe;
        BHR_HIP(hipSetDevice(c->cfg.device));
        if (with_bloom && n > 1) BHR_HIP(hipStreamWaitEvent(p->post, p->halo_in, 0));
        const int vb = bhr_bloom_v_tile_rows(c);
        int chunk = (c->rows + n_chunks_want - 1) / n_chunks_want;
        chunk = ((chunk + vb - 1) / vb) * vb;
        const bool push_chunks = gather && !flare;
        auto v_passes = [&]() -> int32_t {
            int ci = 0;
            for (int r0 = 0; r0 < c->rows; r0 += chunk, ++ci) {
                const int r1 = r0 + chunk < c->rows ? r0 + chunk : c->rows;
                BHR_TRY(bhr_launch_bloom_v_rows(c, with_bloom, r0, r1, (push_chunks && (flags & BHR_GATHER_U8)) ? c->d_final_u8 : nullptr));
                if (!push_chunks) continue;
                BHR_HIP(hipEventRecord(p->v_done[ci], p->post));
                BHR_HIP(hipStreamWaitEvent(p->copy, p->v_done[ci], 0));
                BHR_TRY(queue_push(head, c, flags, r0, r1, p->copy));
            }
            return BHR_OK;
        };
        hipStream_t main_stream = c->stream;
        c->stream = p->post;
        const int32_t rc_v = v_passes();
        c->stream = main_stream;
        BHR_TRY(rc_v);
        BHR_HIP(hipEventRecord(p->post_done, p->post));
        BHR_HIP(hipEventRecord(p->landed, p->copy));
        BHR_HIP(hipStreamWaitEvent(c->stream, p->post_done, 0));
        BHR_HIP(hipStreamWaitEvent(c->stream, p->landed, 0));
        c->last_flags = (int32_t)flags;
        c->timing_valid = 1;
    }
    if (flare) {
        BHR_TRY(flare_pass(ctxs, n, live));
        if (gather)
            for (int k = 0; k < n; ++k) {
                if (live && !live[k]) continue;
                BHR_TRY(bhr_enter(ctxs[k]));
                if (flags & BHR_GATHER_U8) BHR_TRY(bhr_launch_quantize(ctxs[k]));
                BHR_TRY(queue_push(head, ctxs[k], flags, 0, ctxs[k]->rows, ctxs[k]->stream));
            }
    }
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_HIP(hipEventRecord(ctxs[k]->ev[2], ctxs[k]->stream));      // frame_ms = first march launch .. rows landed
    }
    return finish(ctxs, n, live, out_host);
}

}  // namespace

void bhr_pipe_free(bhr_ctx *ctx) {
    TilePipe *p = (TilePipe *)ctx->pipe;
    if (!p) return;
    for (void *o : p->opened)
        if (o) (void)hipIpcCloseMemHandle(o);
    if (p->post) { (void)hipStreamSynchronize(p->post); (void)hipStreamDestroy(p->post); }
    if (p->copy) { (void)hipStreamSynchronize(p->copy); (void)hipStreamDestroy(p->copy); }
    hipEvent_t evs[] = {p->march_a, p->march_b, p->halo_ready, p->h_all, p->halo_in, p->landed, p->post_done, p->glow_ready};
    for (hipEvent_t ev : evs)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->v_done)
        if (ev) (void)hipEventDestroy(ev);
    if (p->d_band) (void)hipFree(p->d_band);
    free(p->h_band);
    delete p;
    ctx->pipe = nullptr;
}

extern "C" {

int32_t bhr_group_render_subset(bhr_ctx **ctxs, int32_t n, const bhr_camera *cam, uint32_t flags, float *out_host, const int32_t *live) {
    if (!ctxs || n <= 0 || !cam) return bhr_fail(BHR_ERR_INVALID, "bhr_group_render: bad argument");
    const int W = ctxs[0]->cfg.width, H = ctxs[0]->cfg.height;
    int expect = 0;
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k]) return bhr_fail(BHR_ERR_INVALID, "bhr_group_render: null ctx %d", k);
        if (ctxs[k]->cfg.width != W || ctxs[k]->cfg.height != H || ctxs[k]->cfg.row0 != expect)
            return bhr_fail(BHR_ERR_INVALID, "bhr_group_render: tile %d does not continue the image (row0 %d, expected %d)", k, ctxs[k]->cfg.row0, expect);
        expect = ctxs[k]->cfg.row1;
    }
    if (expect != H) return bhr_fail(BHR_ERR_INVALID, "bhr_group_render: tiles cover %d of %d rows", expect, H);
    if (out_host && live)
        for (int k = 0; k < n; ++k)
            if (!live[k]) return bhr_fail(BHR_ERR_INVALID, "bhr_group_render_subset: a host gather needs every tile live");
    enable_peer_access(ctxs, n);
    int n_live = 0;
    bool distinct_devices = true;
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        ++n_live;
        for (int q = 0; q < k; ++q)
            if ((!live || live[q]) && ctxs[k]->cfg.device == ctxs[q]->cfg.device) distinct_devices = false;
    }
    bool threaded = n_live > 1 && distinct_devices;
    if (const char *e = getenv("BHR_GROUP_THREADS")) threaded = n_live > 1 && atoi(e) != 0;   // test knob: force / forbid
    bool serial = (flags & BHR_GROUP_SERIAL) != 0;
    if (const char *e = getenv("BHR_GROUP_SCHEDULE")) serial = e[0] == 's';                   // "serial" | "pipelined": A/B runs
    return serial ? render_serial(ctxs, n, cam, flags, out_host, live, threaded)
                  : render_pipelined(ctxs, n, cam, flags, out_host, live, threaded);
}

int32_t bhr_group_render(bhr_ctx **ctxs, int32_t n, const bhr_camera *cam, uint32_t flags, float *out_host) {
    return bhr_group_render_subset(ctxs, n, cam, flags, out_host, nullptr);
}

// ---- one process per tile (bench.py --strong under torchrun when a rank sees only its own GPU) -------------------------
// Same frame, same kernels, same three streams as the pipelined schedule above; what changes is who talks to whom.  Every
// rank owns ONE tile.  Device memory crosses the process boundary through hipIpcMemHandle (the neighbours' H-blur planes
// for the halo pull, tile 0's frame buffers for the push); ordering crosses it through two counters per rank in host
// shared memory: a rank waits on the HOST for its own halo_ready event, publishes the frame number, and its neighbours
// queue their halo pulls once they have seen it -- the data is complete by then, no inter-process event is needed.  The
// frame ends with every rank synchronising its streams and publishing `done`; rank 0 returns when all have.
int32_t bhr_tile_export(bhr_ctx *ctx, uint32_t gather_flags, bhr_tile_handles *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_export: bad argument");
    BHR_TRY(bhr_enter(ctx));
    memset(out, 0, sizeof(*out));
    static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(out->hblur), "handle size");
    hipIpcMemHandle_t h;
    BHR_HIP(hipIpcGetMemHandle(&h, ctx->d_hblur));
    memcpy(out->hblur, &h, sizeof(h));
    out->row0 = ctx->cfg.row0;
    out->rows = ctx->rows;
    out->device = ctx->cfg.device;
    if (ctx->cfg.row0 == 0) {
        BHR_TRY(ensure_gather(ctx, gather_flags));
        if (ctx->d_gather) {
            BHR_HIP(hipIpcGetMemHandle(&h, ctx->d_gather));
            memcpy(out->gather_f32, &h, sizeof(h));
            out->has_gather_f32 = 1;
        }
        if (ctx->d_gather_u8) {
            BHR_HIP(hipIpcGetMemHandle(&h, ctx->d_gather_u8));
            memcpy(out->gather_u8, &h, sizeof(h));
            out->has_gather_u8 = 1;
        }
    }
    return BHR_OK;
}

int32_t bhr_tile_connect(bhr_ctx *ctx, int32_t rank, int32_t world, const bhr_tile_handles *all, uint64_t *shm) {
    if (!ctx || !all || !shm || world < 1 || rank < 0 || rank >= world) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: bad argument");
    BHR_TRY(bhr_enter(ctx));
    int expect = 0;
    for (int k = 0; k < world; ++k) {
        if (all[k].row0 != expect) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: tile %d starts at row %d, expected %d", k, all[k].row0, expect);
        expect += all[k].rows;
        if (world > 1 && all[k].rows < ctx->bloom_R)
            return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: tile %d has %d rows, fewer than the bloom radius %d (use bhr_group_render)", k, all[k].rows, ctx->bloom_R);
    }
    if (expect != ctx->cfg.height || all[rank].row0 != ctx->cfg.row0 || all[rank].rows != ctx->rows)
        return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: the handles do not describe this frame / this rank's tile");
    const int with_up = rank > 0, with_down = rank < world - 1;
    BHR_TRY(ensure_pipe(ctx, with_up, with_down));
    TilePipe *p = (TilePipe *)ctx->pipe;
    for (void *&o : p->opened) {
        if (o) (void)hipIpcCloseMemHandle(o);
        o = nullptr;
    }
    auto open = [&](const uint8_t *raw, void **out, int slot) -> int32_t {
        hipIpcMemHandle_t h;
        memcpy(&h, raw, sizeof(h));
        BHR_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
        p->opened[slot] = *out;
        return BHR_OK;
    };
    p->nb_hblur[0] = p->nb_hblur[1] = nullptr;
    if (with_up) { BHR_TRY(open(all[rank - 1].hblur, (void **)&p->nb_hblur[0], 0)); p->nb_rows[0] = all[rank - 1].rows; }
    if (with_down) { BHR_TRY(open(all[rank + 1].hblur, (void **)&p->nb_hblur[1], 1)); p->nb_rows[1] = all[rank + 1].rows; }
    p->gather_f32 = ctx->d_gather;
    p->gather_u8 = ctx->d_gather_u8;
    if (rank != 0) {
        p->gather_f32 = nullptr;
        p->gather_u8 = nullptr;
        if (all[0].has_gather_f32) BHR_TRY(open(all[0].gather_f32, (void **)&p->gather_f32, 2));
        if (all[0].has_gather_u8) BHR_TRY(open(all[0].gather_u8, (void **)&p->gather_u8, 3));
    }
    p->rank = rank;
    p->world = world;
    p->shm = shm;
    p->frame = 0;
    p->linked = 1;
    return BHR_OK;
}

namespace {
// spins until the counter has reached `want`; gives up after ~20 s (a rank that died must not hang the others for ever)
int32_t wait_counter(volatile uint64_t *c, uint64_t want, const char *what, int peer) {
    for (uint64_t spins = 0; __atomic_load_n(c, __ATOMIC_ACQUIRE) < want; ++spins) {
        if ((spins & 0xffff) == 0xffff) {
            std::this_thread::yield();
            if (spins > (1ull << 33)) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: timed out waiting for rank %d (%s)", peer, what);
        }
    }
    return BHR_OK;
}
}  // namespace

int32_t bhr_tile_render(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    if (!ctx || !cam) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_render: bad argument");
    TilePipe *p = (TilePipe *)ctx->pipe;
    if (!p || !p->linked) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: call bhr_tile_connect first");
    if (flags & BHR_LENS_FLARE) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_render: the lens flare needs the one-process path (bhr_group_render)");
    if ((flags & BHR_GATHER_PEER) && !p->gather_f32) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: no f32 frame buffer was exported by rank 0");
    if ((flags & BHR_GATHER_U8) && !p->gather_u8) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: no u8 frame buffer was exported by rank 0");
    const int with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    const int rank = p->rank, world = p->world;
    const size_t R = ctx->bloom_R, W = ctx->cfg.width, rows = ctx->rows;
    const uint64_t frame = ++p->frame;
    volatile uint64_t *mine = p->shm + (size_t)rank * BHR_TILE_SHM_WORDS;
    BHR_TRY(bhr_enter(ctx));
    ctx->cur_slot = -1;
    ctx->last_slot = -1;
    const bool split = with_bloom && p->n_band > 0 && p->n_rest > 0 && !(flags & (BHR_PERSISTENT | BHR_ROW_COSTS));
    // march: halo bands on the tile's stream, the rest on the low-priority second stream (as render_pipelined)
    auto march = [&]() -> int32_t {
        if (!split) {
            BHR_TRY(bhr_launch_march(ctx, cam, flags));
            BHR_HIP(hipEventRecord(p->march_a, ctx->stream));
            BHR_HIP(hipEventRecord(p->march_b, ctx->stream));
            return BHR_OK;
        }
        bhr_march_part part;
        memset(&part, 0, sizeof(part));
        part.active = 1; part.id = 1; part.first = 1;
        ctx->part = part;
        BHR_TRY(bhr_launch_march(ctx, cam, flags));
        BHR_TRY(bhr_aux_fork(ctx));
        part.d_list = p->d_band; part.h_list = p->h_band; part.n = p->n_band; part.first = 0;
        ctx->part = part;
        BHR_TRY(bhr_launch_march(ctx, cam, flags));
        BHR_HIP(hipEventRecord(p->march_a, ctx->stream));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->aux_stream;
        part.d_list = p->d_band + p->n_band; part.h_list = p->h_band + p->n_band; part.n = p->n_rest; part.id = 2;
        ctx->part = part;
        const int32_t rc = bhr_launch_march(ctx, cam, flags);
        ctx->stream = main_stream;
        BHR_TRY(rc);
        BHR_TRY(bhr_aux_join(ctx));
        BHR_HIP(hipEventRecord(p->march_b, ctx->stream));
        part.d_list = nullptr; part.h_list = nullptr; part.n = 0; part.last = 1;
        ctx->part = part;
        BHR_TRY(bhr_launch_march(ctx, cam, flags));
        return BHR_OK;
    };
    const int32_t rc_m = march();
    ctx->part.active = 0;
    BHR_TRY(rc_m);
    hipStream_t main_stream = ctx->stream;
    auto on_post = [&](auto f) -> int32_t {
        ctx->stream = p->post;
        const int32_t rc = f();
        ctx->stream = main_stream;
        return rc;
    };
    BHR_TRY(on_post([&]() -> int32_t {
        BHR_HIP(hipStreamWaitEvent(p->post, p->march_a, 0));
        if (!split) BHR_HIP(hipStreamWaitEvent(p->post, p->march_b, 0));
        if (with_bloom && split) {
            BHR_TRY(bhr_launch_bloom_h_rows(ctx, 0, p->band_top));
            BHR_TRY(bhr_launch_bloom_h_rows(ctx, p->band_bot, ctx->rows));
        } else if (with_bloom) {
            BHR_TRY(bhr_launch_bloom_h(ctx));
        }
        BHR_HIP(hipEventRecord(p->halo_ready, p->post));
        BHR_HIP(hipStreamWaitEvent(p->post, p->march_b, 0));
        if (with_bloom && split) BHR_TRY(bhr_launch_bloom_h_rows(ctx, p->band_top, p->band_bot));
        BHR_HIP(hipEventRecord(p->h_all, p->post));
        return BHR_OK;
    }));
    if (with_bloom && world > 1) {
        // my halo bands are blurred: tell the neighbours; then pull theirs once they have said the same
        BHR_HIP(hipEventSynchronize(p->halo_ready));
        __atomic_store_n(mine + 0, frame, __ATOMIC_RELEASE);
        for (int side = 0; side < 2; ++side) {
            if (!p->nb_hblur[side]) continue;
            const int nb = side == 0 ? rank - 1 : rank + 1;
            BHR_TRY(wait_counter(p->shm + (size_t)nb * BHR_TILE_SHM_WORDS, frame, "halo bands", nb));
            const size_t nb_rows = p->nb_rows[side], nb_plane = (nb_rows + 2 * R) * W, my_plane = (rows + 2 * R) * W;
            const size_t src_row = side == 0 ? R + nb_rows - R : R;        // its last / first R rows (own rows live at [R, R + rows))
            const size_t dst_row = side == 0 ? 0 : R + rows;
            for (int c = 0; c < 3; ++c)
                BHR_HIP(hipMemcpyAsync(ctx->d_hblur + c * my_plane + dst_row * W, p->nb_hblur[side] + c * nb_plane + src_row * W,
                                       R * W * sizeof(float), hipMemcpyDeviceToDevice, p->copy));
        }
        BHR_HIP(hipEventRecord(p->halo_in, p->copy));
        BHR_HIP(hipStreamWaitEvent(p->post, p->halo_in, 0));
    }
    // V pass + combine in row chunks, every finished chunk pushed into tile 0's frame buffers by the copy stream
    int n_chunks_want = 3;
    if (const char *e = getenv("BHR_TILE_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= PIPE_MAX_CHUNKS) n_chunks_want = v; }
    const int vb = bhr_bloom_v_tile_rows(ctx);
    int chunk = (ctx->rows + n_chunks_want - 1) / n_chunks_want;
    chunk = ((chunk + vb - 1) / vb) * vb;
    const bool gather = (flags & (BHR_GATHER_PEER | BHR_GATHER_U8)) != 0;
    BHR_TRY(on_post([&]() -> int32_t {
        int ci = 0;
        for (int r0 = 0; r0 < ctx->rows; r0 += chunk, ++ci) {
            const int r1 = r0 + chunk < ctx->rows ? r0 + chunk : ctx->rows;
            BHR_TRY(bhr_launch_bloom_v_rows(ctx, with_bloom, r0, r1, (flags & BHR_GATHER_U8) ? ctx->d_final_u8 : nullptr));
            if (!gather) continue;
            BHR_HIP(hipEventRecord(p->v_done[ci], p->post));
            BHR_HIP(hipStreamWaitEvent(p->copy, p->v_done[ci], 0));
            const size_t W3 = W * 3, off = (size_t)r0 * W3, cnt = (size_t)(r1 - r0) * W3, dst = (size_t)(ctx->cfg.row0 + r0) * W3;
            if (flags & BHR_GATHER_U8)
                BHR_HIP(hipMemcpyAsync(p->gather_u8 + dst, ctx->d_final_u8 + off, cnt, hipMemcpyDeviceToDevice, p->copy));
            if (flags & BHR_GATHER_PEER)
                BHR_HIP(hipMemcpyAsync(p->gather_f32 + dst, ctx->d_final + off, cnt * sizeof(float), hipMemcpyDeviceToDevice, p->copy));
        }
        return BHR_OK;
    }));
    BHR_HIP(hipEventRecord(p->post_done, p->post));
    BHR_HIP(hipEventRecord(p->landed, p->copy));
    BHR_HIP(hipStreamWaitEvent(ctx->stream, p->post_done, 0));
    BHR_HIP(hipStreamWaitEvent(ctx->stream, p->landed, 0));
    BHR_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    ctx->last_flags = (int32_t)flags;
    ctx->timing_valid = 1;
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    // end of frame: my rows have landed and nobody reads my halo rows any more once every rank has said `done`
    __atomic_store_n(mine + 1, frame, __ATOMIC_RELEASE);
    for (int k = 0; k < world; ++k)
        BHR_TRY(wait_counter(p->shm + (size_t)k * BHR_TILE_SHM_WORDS + 1, frame, "end of frame", k));
    return BHR_OK;
}

int32_t bhr_read_gathered_u8(bhr_ctx *ctx, uint8_t *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_read_gathered_u8: bad argument");
    if (!ctx->d_gather_u8) return bhr_fail(BHR_ERR_STATE, "bhr_read_gathered_u8: no bhr_group_render(..., BHR_GATHER_U8) has gathered into this context");
    BHR_TRY(bhr_enter(ctx));
    const size_t bytes = (size_t)ctx->cfg.height * ctx->cfg.width * 3;
    BHR_TRY(bhr_ensure_pinned(ctx, bytes));
    BHR_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_gather_u8, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->h_pinned, bytes);
    return BHR_OK;
}

}  // extern "C"
