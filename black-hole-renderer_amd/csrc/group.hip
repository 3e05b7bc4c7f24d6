// group.hip -- one frame in N row blocks (BASELINE.json configs[3]): bhr_group_render / bhr_group_render_subset.
//
// The reference has no multi-GPU code; what has to be kept is the frame: march -> bloom H -> bloom V ->
// clip(bg + disk + blur) (-> lens flare) of render.py:3865-3923, 3022-3114.  A row block needs R = int(0.02 W) rows of
// its neighbours' H-blurred disk layer for the V pass; everything else is local.  Two schedules produce the same bytes:
//
//  * serial (BHR_GROUP_SERIAL, round 1-2): per tile march -> H pass -> halo pull -> V pass -> gather, each step behind
//    the previous one on the tile's stream.  At 8 tiles of an 8k frame the steps after the march (H 0.12, halo 0.1,
//    V 0.34, f32 gather 0.33 ms) are a 0.9 ms tail behind a 3.0 ms march.
//  * pipelined: per tile three streams.
//      tile stream   march -> H pass -> `halo_ready` -> V pass + combine in row chunks: the rows at least R away from a
//                    neighbouring tile FIRST (they need no halo), the rows next to the edges once the halo has arrived;
//                    each chunk's quantised bytes are written by the V kernel's epilogue (BHR_GATHER_U8);
//      copy stream   pulls the neighbours' halo rows behind THEIR `halo_ready`, under the V pass of the middle rows;
//      push stream   pushes every finished chunk into the frame buffer on tile 0's device while the next chunk's V
//                    kernel runs.  u8 rows: 12.4 MB per 8k tile instead of 49.8 MB of f32.
//    Default where the tiles sit on DISTINCT devices (the copies cross xGMI: ~0.09 ms halo + ~0.08 ms u8 rows of an 8k
//    tile); tiles sharing one device default to the serial schedule -- their copies are HBM to HBM and every cross-stream
//    hand-over costs more than it hides (tools/exp_tile_tail.py).  BHR_GROUP_SCHEDULE / BHR_GROUP_SERIAL override.
//    Tried and removed: marching the halo bands in a launch of their own so that the pull starts under the march of the
//    middle rows, H / V on a high-priority stream of their own -- the second launch's ragged start and the extra
//    hand-overs cost an 8k tile 0.2-0.5 ms against the 0.09 ms they hide.
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
    hipStream_t copy, push;       // halo pulls; pushes of finished row chunks
    hipEvent_t halo_ready, halo_in, v_done[PIPE_MAX_CHUNKS], landed;
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

int32_t ensure_pipe(bhr_ctx *ctx) {
    if (ctx->pipe) return BHR_OK;
    TilePipe *p = new TilePipe();
    memset(p, 0, sizeof(*p));
    ctx->pipe = p;
    BHR_HIP(hipStreamCreateWithFlags(&p->copy, hipStreamNonBlocking));
    BHR_HIP(hipStreamCreateWithFlags(&p->push, hipStreamNonBlocking));
    hipEvent_t *evs[] = {&p->halo_ready, &p->halo_in, &p->landed};
    for (hipEvent_t *ev : evs) BHR_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    for (auto &ev : p->v_done) BHR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return BHR_OK;
}

// Row chunks of a tile's V pass, in launch order.  Rows at least R away from a neighbouring tile need no halo rows: they
// go first, under the halo pull; the rows next to the tile's edges follow once the halo has arrived.  Every chunk is
// pushed to tile 0 while the next one's V kernel runs.  Chunk boundaries fall on multiples of the V kernel's block height.
struct Chunk { int r0, r1, needs_halo; };
int plan_chunks(const bhr_ctx *c, bool has_up, bool has_down, int vb, Chunk (&out)[PIPE_MAX_CHUNKS]) {
    const int rows = c->rows, R = c->bloom_R;
    int want = 3;
    if (const char *e = getenv("BHR_TILE_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= PIPE_MAX_CHUNKS - 2) want = v; }
    int n = 0;
    auto add = [&](int r0, int r1, int halo, int pieces) {
        if (r1 <= r0) return;
        int step = ((r1 - r0 + pieces - 1) / pieces + vb - 1) / vb * vb;
        for (int r = r0; r < r1 && n < PIPE_MAX_CHUNKS; r += step) out[n++] = Chunk{r, r + step < r1 ? r + step : r1, halo};
    };
    int m0 = has_up ? (R + vb - 1) / vb * vb : 0;
    if (m0 > rows) m0 = rows;
    int m1 = has_down ? m0 + ((rows - R - m0) > 0 ? (rows - R - m0) / vb * vb : 0) : rows;
    if (m1 > rows) m1 = rows;
    if (m1 < m0) m1 = m0;
    if (!has_up && !has_down) { add(0, rows, 0, want); return n; }
    add(m0, m1, 0, want > 2 ? want - 2 : 1);       // the middle: no halo needed
    add(0, m0, 1, 1);
    add(m1, rows, 1, 1);
    return n;
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
// pipelined: behind the neighbour's `halo_ready` (its H pass), serial: behind its ev[3].
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
                if (np) BHR_HIP(hipStreamWaitEvent(stream, np->halo_ready, 0));   // a tile that never rendered pipelined has nothing in flight
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

    // phase 1: march and H pass on the tile's stream
    BHR_TRY(for_tiles(n, live, threaded, [&](int k) -> int32_t {
        bhr_ctx *c = ctxs[k];
        BHR_TRY(bhr_enter(c));
        c->cur_slot = -1;
        c->last_slot = -1;
        BHR_TRY(ensure_pipe(c));
        TilePipe *p = (TilePipe *)c->pipe;
        BHR_TRY(bhr_launch_march(c, cam, flags));
        if (with_bloom) BHR_TRY(bhr_launch_bloom_h(c));
        BHR_HIP(hipEventRecord(p->halo_ready, c->stream));
        return BHR_OK;
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

    // phase 3: V pass + combine in row chunks -- the rows that need no halo first, under the halo pull -- every finished
    // chunk pushed by the push stream while the next chunk's V kernel runs
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        bhr_ctx *c = ctxs[k];
        TilePipe *p = (TilePipe *)c->pipe;
        BHR_HIP(hipSetDevice(c->cfg.device));
        const bool halo = with_bloom && n > 1;
        Chunk chunks[PIPE_MAX_CHUNKS];
        const int n_chunks = plan_chunks(c, halo && k > 0, halo && k < n - 1, bhr_bloom_v_tile_rows(c), chunks);
        const bool push_chunks = gather && !flare;
        bool waited = false;
        for (int ci = 0; ci < n_chunks; ++ci) {
            if (chunks[ci].needs_halo && !waited) { BHR_HIP(hipStreamWaitEvent(c->stream, p->halo_in, 0)); waited = true; }
            BHR_TRY(bhr_launch_bloom_v_rows(c, with_bloom, chunks[ci].r0, chunks[ci].r1, (push_chunks && (flags & BHR_GATHER_U8)) ? c->d_final_u8 : nullptr));
            if (!push_chunks) continue;
            BHR_HIP(hipEventRecord(p->v_done[ci], c->stream));
            BHR_HIP(hipStreamWaitEvent(p->push, p->v_done[ci], 0));
            BHR_TRY(queue_push(head, c, flags, chunks[ci].r0, chunks[ci].r1, p->push));
        }
        if (halo && !waited) BHR_HIP(hipStreamWaitEvent(c->stream, p->halo_in, 0));   // nothing of this frame stays in flight
        BHR_HIP(hipEventRecord(p->landed, p->push));
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
    if (p->copy) { (void)hipStreamSynchronize(p->copy); (void)hipStreamDestroy(p->copy); }
    if (p->push) { (void)hipStreamSynchronize(p->push); (void)hipStreamDestroy(p->push); }
    hipEvent_t evs[] = {p->halo_ready, p->halo_in, p->landed};
    for (hipEvent_t ev : evs)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->v_done)
        if (ev) (void)hipEventDestroy(ev);
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
    // tiles that share a device copy HBM to HBM: nothing to hide, the serial schedule has fewer hand-overs
    bool serial = (flags & BHR_GROUP_SERIAL) != 0 || (!(flags & BHR_GROUP_PIPELINED) && !distinct_devices);
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
    BHR_HIP(hipIpcGetMemHandle(&h, ctx->slots[ctx->active_slot].d_hblur_base));   // handles name allocations: the planes start BHR_HBLUR_PAD_ROWS rows in
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
    BHR_TRY(ensure_pipe(ctx));
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
    const size_t pad = (size_t)BHR_HBLUR_PAD_ROWS * ctx->cfg.width;     // the handle opens the allocation; the planes start `pad` floats in
    if (with_up) { BHR_TRY(open(all[rank - 1].hblur, (void **)&p->nb_hblur[0], 0)); p->nb_hblur[0] += pad; p->nb_rows[0] = all[rank - 1].rows; }
    if (with_down) { BHR_TRY(open(all[rank + 1].hblur, (void **)&p->nb_hblur[1], 1)); p->nb_hblur[1] += pad; p->nb_rows[1] = all[rank + 1].rows; }
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
    // march -> H pass on the tile's stream
    BHR_TRY(bhr_launch_march(ctx, cam, flags));
    if (with_bloom) BHR_TRY(bhr_launch_bloom_h(ctx));
    BHR_HIP(hipEventRecord(p->halo_ready, ctx->stream));
    const bool halo = with_bloom && world > 1;
    const bool gather = (flags & (BHR_GATHER_PEER | BHR_GATHER_U8)) != 0;
    Chunk chunks[PIPE_MAX_CHUNKS];
    const int n_chunks = plan_chunks(ctx, halo && rank > 0, halo && rank < world - 1, bhr_bloom_v_tile_rows(ctx), chunks);
    // V pass + combine of chunk ci on the tile's stream, its rows pushed into tile 0's frame buffers by the push stream
    auto v_chunk = [&](int ci) -> int32_t {
        const int r0 = chunks[ci].r0, r1 = chunks[ci].r1;
        BHR_TRY(bhr_launch_bloom_v_rows(ctx, with_bloom, r0, r1, (flags & BHR_GATHER_U8) ? ctx->d_final_u8 : nullptr));
        if (!gather) return BHR_OK;
        BHR_HIP(hipEventRecord(p->v_done[ci], ctx->stream));
        BHR_HIP(hipStreamWaitEvent(p->push, p->v_done[ci], 0));
        const size_t W3 = W * 3, off = (size_t)r0 * W3, cnt = (size_t)(r1 - r0) * W3, dst = (size_t)(ctx->cfg.row0 + r0) * W3;
        if (flags & BHR_GATHER_U8)
            BHR_HIP(hipMemcpyAsync(p->gather_u8 + dst, ctx->d_final_u8 + off, cnt, hipMemcpyDeviceToDevice, p->push));
        if (flags & BHR_GATHER_PEER)
            BHR_HIP(hipMemcpyAsync(p->gather_f32 + dst, ctx->d_final + off, cnt * sizeof(float), hipMemcpyDeviceToDevice, p->push));
        return BHR_OK;
    };
    // the rows that need no halo are queued now: they run while this rank waits for its neighbours below
    for (int ci = 0; ci < n_chunks; ++ci)
        if (!chunks[ci].needs_halo) BHR_TRY(v_chunk(ci));
    if (halo) {
        // my rows are blurred: tell the neighbours; then pull theirs once they have said the same
        BHR_HIP(hipEventSynchronize(p->halo_ready));
        __atomic_store_n(mine + 0, frame, __ATOMIC_RELEASE);
        for (int side = 0; side < 2; ++side) {
            if (!p->nb_hblur[side]) continue;
            const int nb = side == 0 ? rank - 1 : rank + 1;
            BHR_TRY(wait_counter(p->shm + (size_t)nb * BHR_TILE_SHM_WORDS, frame, "halo rows", nb));
            const size_t nb_rows = p->nb_rows[side], nb_plane = (nb_rows + 2 * R) * W, my_plane = (rows + 2 * R) * W;
            const size_t src_row = side == 0 ? nb_rows : R;                  // its last / first R rows (own rows live at [R, R + rows))
            const size_t dst_row = side == 0 ? 0 : R + rows;
            for (int c = 0; c < 3; ++c)
                BHR_HIP(hipMemcpyAsync(ctx->d_hblur + c * my_plane + dst_row * W, p->nb_hblur[side] + c * nb_plane + src_row * W,
                                       R * W * sizeof(float), hipMemcpyDeviceToDevice, p->copy));
        }
        BHR_HIP(hipEventRecord(p->halo_in, p->copy));
        BHR_HIP(hipStreamWaitEvent(ctx->stream, p->halo_in, 0));
    }
    for (int ci = 0; ci < n_chunks; ++ci)
        if (chunks[ci].needs_halo) BHR_TRY(v_chunk(ci));
    BHR_HIP(hipEventRecord(p->landed, p->push));
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
