// group.hip -- one frame in N row blocks (BASELINE.json configs[3]): bhr_group_render / bhr_group_render_subset (one process
// drives the devices) and bhr_tile_export / _connect / _render (one process per tile).
//
// The reference has no multi-GPU code; what has to be kept is the frame: march -> bloom H -> bloom V ->
// clip(bg + disk + blur) (-> lens flare) of render.py:3865-3923, 3022-3114.  A row block needs R = int(0.02 W) rows of
// its neighbours' H-blurred disk layer for the V pass; everything else is local.
//
// Round 4: the kernels do the exchange.  Rounds 2-3 moved the halo rows and the finished rows with hipMemcpyPeerAsync
// stages of their own (halo pull, u8 / f32 push), each behind a kernel boundary and a cross-stream hand-over: 0.2-0.28 ms
// of tail behind a 1.2 ms march of an 8k tile.  Now
//   * the H pass of a split-f16 frame (fast / hybrid arithmetic, bloom.hip) stores the rows a neighbouring block needs
//     straight into that block's planes -- peer-mapped pointers in one process, hipIpcMemHandle mappings between
//     processes -- next to its own copy (`mirrors`);
//   * the V pass's epilogue stores the quantised u8 rows (BHR_GATHER_U8) or the f32 rows (BHR_GATHER_PEER) straight into
//     the frame buffer on tile 0's device;
//   so a tile's stream carries march -> H -> V and nothing else: the only cross-tile dependency left is "my V pass waits
//   for my neighbours' H passes" -- one event wait (one shared-memory counter between processes).  The exact-f32
//   post-pass of strict frames keeps its planar planes and pulls its halo rows with copies as before; its V pass stores
//   into the frame buffer directly all the same.
// Two schedules, same bytes: serial (the V pass in one launch behind the wait) and pipelined (the rows at least R away
// from a neighbouring block go through the V pass FIRST, under the neighbours' H passes / the halo pull).  Default:
// pipelined where there is a copy stage to hide (exact-f32 post-pass on distinct devices), serial otherwise.  The lens flare needs the frame's three
// sums (a host read-back on tile 0), so with BHR_LENS_FLARE the tiles keep their f32 rows until the flare has been
// applied and ship them with copies afterwards.
//
// `live` (bhr_group_render_subset) restricts a call to some tiles while the others keep the buffers of the previous full
// render -- how one tile of eight is timed end to end on a single GPU (bench.py tile_scaling.tile_tail_ms).
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "bhr_internal.h"

namespace {

constexpr int PIPE_MAX_CHUNKS = 16;

struct TilePipe {
    hipStream_t copy;             // halo pulls of the exact-f32 post-pass
    hipEvent_t halo_ready, halo_in, frame_done;   // frame_done: the tile's last kernel of its last group frame
    int32_t in_flight;            // that frame was submitted with BHR_GROUP_ASYNC and nobody has waited for it yet
    // one-process-per-tile variant (bhr_tile_connect): the neighbours' planes and tile 0's frame buffers, opened from their
    // IPC handles; counters in host shared memory for the hand-shakes
    int32_t linked, rank, world, split;
    float *nb_hblur[2];           // exact-f32 post-pass: the planes of the tile above [0] / below [1] (nullptr: none)
    int32_t nb_rows[2];
    struct { void *pb; int32_t pbr, gr; } nb_split[2];   // split-f16 post-pass: their packed planes
    float *gather_f32;            // frame buffers on tile 0's device (its own pointers on rank 0)
    uint8_t *gather_u8;
    void *opened[4];              // what hipIpcCloseMemHandle has to release
    volatile uint64_t *shm;       // BHR_TILE_SHM_WORDS words per rank: [0] frames whose halo rows are H-blurred, [1] frames done
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
    hipEvent_t *evs[] = {&p->halo_ready, &p->halo_in, &p->frame_done};
    for (hipEvent_t *ev : evs) BHR_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return BHR_OK;
}

// Row chunks of a tile's V pass, in launch order.  Rows at least R away from a neighbouring tile need no halo rows: they
// go first, under the neighbours' H passes; the rows next to the tile's edges follow once the halo has arrived.  Chunk
// boundaries fall on multiples of the V kernel's block height (in image rows where the kernel's tiles are global).
struct Chunk { int r0, r1, needs_halo; };
int plan_chunks(const bhr_ctx *c, bool has_up, bool has_down, int vb, bool pipelined, Chunk (&out)[PIPE_MAX_CHUNKS]) {
    const int rows = c->rows;
    if (!pipelined || (!has_up && !has_down)) { out[0] = Chunk{0, rows, (has_up || has_down) ? 1 : 0}; return 1; }
    // reach of a neighbour's rows into this tile: the bloom radius, or the padded radius of the split kernels' 16-tap chunks
    const int R = c->bloom_split ? 16 * (bhr_split_nt(c->bloom_R) - 1) : c->bloom_R;
    int n = 0;
    // first / last local row whose V-pass tile is clear of the neighbours: tiles are `vb` image rows, aligned globally
    int m0 = 0, m1 = rows;
    if (has_up) m0 = ((c->cfg.row0 + R + vb - 1) / vb) * vb - c->cfg.row0;
    if (has_down) m1 = ((c->cfg.row0 + rows - R) / vb) * vb - c->cfg.row0;
    if (m0 > rows) m0 = rows;
    if (m1 < m0) m1 = m0;
    if (m1 > m0) out[n++] = Chunk{m0, m1, 0};       // the middle: no halo needed
    if (m0 > 0) out[n++] = Chunk{0, m0, 1};
    if (m1 < rows) out[n++] = Chunk{m1, rows, 1};
    return n;
}

// Direct xGMI stores / copies between the tiles' devices.  Tried once per ordered device pair; a refusal is not an error
// for the copies (hipMemcpyPeerAsync then stages through the host) but rules the direct stores out (peer_ok).
bool g_peer_ok[64][64];
void enable_peer_access(bhr_ctx **ctxs, int32_t n) {
    static bool tried[64][64];
    for (int k = 0; k < n; ++k)
        for (int q = 0; q < n; ++q) {
            const int a = ctxs[k]->cfg.device, b = ctxs[q]->cfg.device;
            if (a < 0 || b < 0 || a >= 64 || b >= 64) continue;
            if (a == b) { g_peer_ok[a][b] = true; continue; }
            if (tried[a][b]) continue;
            tried[a][b] = true;
            int can = 0;
            if (hipSetDevice(a) != hipSuccess || hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                (void)hipGetLastError();
                continue;
            }
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess) (void)hipGetLastError();
            g_peer_ok[a][b] = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        }
}
bool peer_ok(const bhr_ctx *from, const bhr_ctx *to) {
    const int a = from->cfg.device, b = to->cfg.device;
    return a == b || (a >= 0 && b >= 0 && a < 64 && b < 64 && g_peer_ok[a][b]);
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

// does tile q's V pass read rows of tile k (k != q)?  Its planes reach `reach` rows past its own.
bool needs_rows_of(const bhr_ctx *q, const bhr_ctx *k, int reach) {
    return k->cfg.row1 > q->cfg.row0 - reach && k->cfg.row0 < q->cfg.row1 + reach;
}

// Exact-f32 post-pass: halo pull of tile k -- up to R rows of the planar (3, rows + 2R, W) H-blur planes from the tiles above
// and below (a tile thinner than R passes the request on to the next one) -- queued on `stream` behind the producers' H passes.
int32_t queue_halo_pull(bhr_ctx **ctxs, int n, int k, hipStream_t stream, const int32_t *live) {
    bhr_ctx *me = ctxs[k];
    const size_t R = me->bloom_R, W = me->cfg.width, my_rows = me->rows;
    for (int side = 0; side < 2; ++side) {
        size_t need = R, got = 0;                  // side 0: rows above me come from tiles k-1, k-2, ...; side 1: below
        int q = side == 0 ? k - 1 : k + 1;
        while (need > 0 && q >= 0 && q < n) {
            bhr_ctx *nb = ctxs[q];
            const size_t take = (size_t)nb->rows < need ? (size_t)nb->rows : need;
            const TilePipe *np = (const TilePipe *)nb->pipe;
            if (np && (!live || live[q])) BHR_HIP(hipStreamWaitEvent(stream, np->halo_ready, 0));   // a resting tile has nothing in flight
            if (!nb->d_hblur) return bhr_fail(BHR_ERR_STATE, "group render: tile %d has no H-blur planes to pull from (render every tile once first)", q);
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
        ctxs[k]->slots[ctxs[k]->active_slot].have &= ~BHR_OUT_U8;
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

// copy the rows of tile `t` into the frame buffers on `head` (f32 and / or u8), on `stream`: the flare path and tiles whose
// device cannot store into tile 0's memory
int32_t queue_push(bhr_ctx *head, bhr_ctx *t, uint32_t flags, hipStream_t stream) {
    const size_t W3 = (size_t)t->cfg.width * 3, cnt = (size_t)t->rows * W3, dst = (size_t)t->cfg.row0 * W3;
    if (flags & BHR_GATHER_U8)
        BHR_HIP(hipMemcpyPeerAsync(head->d_gather_u8 + dst, head->cfg.device, t->d_final_u8, t->cfg.device, cnt, stream));
    if (flags & BHR_GATHER_PEER)
        BHR_HIP(hipMemcpyPeerAsync(head->d_gather + dst, head->cfg.device, t->d_final, t->cfg.device, cnt * sizeof(float), stream));
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

// the planes of the other tiles that hold rows of tile k: its H pass writes them there itself
int32_t set_mirrors(bhr_ctx **ctxs, int n, int k) {
    bhr_ctx *c = ctxs[k];
    c->n_mirrors = 0;
    if (!c->bloom_split) return BHR_OK;
    const int reach = 16 * (bhr_split_nt(c->bloom_R) - 1);
    for (int q = 0; q < n; ++q) {
        if (q == k || !needs_rows_of(ctxs[q], c, reach + 32)) continue;   // + 32: planes start on 32-row tile boundaries
        bhr_ctx *nb = ctxs[q];
        void *pb = nb->slots[0].d_pb;
        if (!pb) continue;                                                  // a tile that has never rendered a split frame
        if (!peer_ok(c, nb)) return bhr_fail(BHR_ERR_STATE, "group render: device %d cannot store into device %d's memory (no peer access)", c->cfg.device, nb->cfg.device);
        bhr_split_geom g;
        bhr_split_geometry(nb, &g);
        if (c->n_mirrors >= (int)(sizeof(c->mirrors) / sizeof(c->mirrors[0])))
            return bhr_fail(BHR_ERR_INVALID, "group render: tile %d feeds more than %d neighbouring tiles (row blocks thinner than the bloom radius / 3?)", k, c->n_mirrors);
        c->mirrors[c->n_mirrors].pb = pb;
        c->mirrors[c->n_mirrors].pbr = g.pbr;
        c->mirrors[c->n_mirrors].gr = g.GR;
        c->n_mirrors += 1;
    }
    return BHR_OK;
}

int32_t render_tiles(bhr_ctx **ctxs, int n, const bhr_camera *cam, uint32_t flags, float *out_host, const int32_t *live,
                     bool threaded, int schedule) {
    const int with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    const bool flare = (flags & BHR_LENS_FLARE) != 0;
    const bool gather = (flags & (BHR_GATHER_PEER | BHR_GATHER_U8)) != 0;
    bhr_ctx *head = ctxs[0];
    if (gather) BHR_TRY(ensure_gather(head, flags));

    // The split-f16 post-pass stores halo rows straight into the neighbours' memory: every pair of tiles must be able to (peer
    // access between their devices).  Where a pair cannot, the whole group falls back to the exact-f32 post-pass for this frame,
    // whose halo rows travel by copies (staged through the host where there is no peer access): slower, still correct.
    bool all_peer = true;
    for (int k = 0; k < n && all_peer; ++k)
        for (int q = 0; q < n && all_peer; ++q)
            if (k != q && !peer_ok(ctxs[k], ctxs[q])) all_peer = false;
    struct RestoreSplit {                     // the option is the context's own: put back whatever the frame does
        bhr_ctx **c; int n; int32_t saved[64]; bool on;
        ~RestoreSplit() { if (on) for (int k = 0; k < n && k < 64; ++k) c[k]->opt.bloom_split = saved[k]; }
    } restore{ctxs, n, {}, !all_peer && n <= 64};
    if (restore.on)
        for (int k = 0; k < n; ++k) { restore.saved[k] = ctxs[k]->opt.bloom_split; ctxs[k]->opt.bloom_split = 0; }

    // every tile chooses its post-pass kernels and makes sure their buffers exist BEFORE any H pass may store into them
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        bhr_ctx *c = ctxs[k];
        BHR_TRY(bhr_enter(c));
        BHR_TRY(bhr_activate_slot(c, 0));
        c->cur_slot = -1;
        c->last_slot = -1;
        BHR_TRY(ensure_pipe(c));
        BHR_TRY(bhr_frame_begin(c, flags));
        c->group_time_march = (flags & BHR_GROUP_TIME_MARCH) ? 1 : 0;
    }
    const bool halo = with_bloom && n > 1;
    if (halo)
        for (int k = 0; k < n; ++k)
            if (!live || live[k]) BHR_TRY(set_mirrors(ctxs, n, k));
    // schedule < 0: by what there is to hide.  The exact-f32 post-pass PULLS its halo rows (a copy stage: worth running under
    // the V pass of the middle rows on distinct devices); the split-f16 post-pass has none -- its neighbours' H kernels store
    // the rows themselves -- and its V pass in ONE launch behind the wait is never later than in chunks: the frame ends with
    // the slowest tile's own V pass either way, and one launch saves the chunks' fixed costs (~20 us per tile at 8k).
    bool any_split = false;
    for (int k = 0; k < n; ++k)
        if (!live || live[k]) any_split = any_split || ctxs[k]->bloom_split != 0;
    const bool pipelined = schedule >= 0 ? schedule != 0 : (schedule == -2 && !any_split);      // -2: tiles on distinct devices

    // phase 1: march and H pass (+ the neighbours' halo rows of a split frame) on the tile's stream
    BHR_TRY(for_tiles(n, live, threaded, [&](int k) -> int32_t {
        bhr_ctx *c = ctxs[k];
        BHR_HIP(hipSetDevice(c->cfg.device));
        TilePipe *p = (TilePipe *)c->pipe;
        BHR_TRY(bhr_launch_march(c, cam, flags));
        // frames in flight (BHR_GROUP_ASYNC): this H pass stores into its neighbours' halo rows, which their previous frame's V
        // passes may still be reading -- wait for those on the device
        if (with_bloom && c->bloom_split)
            for (int q = 0; q < n; ++q) {
                TilePipe *pq = q == k ? nullptr : (TilePipe *)ctxs[q]->pipe;
                if (pq && pq->in_flight && needs_rows_of(ctxs[q], c, 16 * (bhr_split_nt(c->bloom_R) - 1) + 32)) BHR_HIP(hipStreamWaitEvent(c->stream, pq->frame_done, 0));
            }
        if (with_bloom) BHR_TRY(bhr_launch_bloom_h(c));
        // `halo_ready` (this tile's H pass is done: its neighbours may run the V pass of their edge rows).  An event record is a
        // ~5 us bubble in the stream: the pipelined schedule records it BEHIND the V pass of the middle rows (phase 3) -- the
        // neighbours are busy with their own middle rows until then -- the serial one here
        if (!pipelined || !with_bloom || !c->bloom_split) BHR_HIP(hipEventRecord(p->halo_ready, c->stream));
        return BHR_OK;
    }));
    for (int k = 0; k < n; ++k)
        if (!live || live[k]) ctxs[k]->n_mirrors = 0;

    // phase 2 (exact-f32 post-pass only): halo pulls behind the neighbours' H passes -- under the V pass of the middle rows
    // (copy stream) in the pipelined schedule
    if (halo)
        for (int k = 0; k < n; ++k) {
            if ((live && !live[k]) || ctxs[k]->bloom_split) continue;
            TilePipe *p = (TilePipe *)ctxs[k]->pipe;
            BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
            hipStream_t s = pipelined ? p->copy : ctxs[k]->stream;
            if (pipelined) BHR_HIP(hipStreamWaitEvent(s, p->halo_ready, 0));        // the pulls overwrite halo rows: behind this tile's own H pass too
            BHR_TRY(queue_halo_pull(ctxs, n, k, s, live));
            BHR_HIP(hipEventRecord(p->halo_in, s));
        }

    // phase 3: V pass + combine, storing straight into the frame buffers on tile 0's device where it may.  Two host passes
    // over the tiles: (a) the chunks that need no halo rows, then `halo_ready` where it is still to be recorded; (b) the wait
    // for the neighbours and the chunks next to the edges -- every event a tile waits for has been recorded for THIS frame
    // by then.
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < n; ++k) {
            if (live && !live[k]) continue;
            bhr_ctx *c = ctxs[k];
            TilePipe *p = (TilePipe *)c->pipe;
            BHR_HIP(hipSetDevice(c->cfg.device));
            const bool direct = gather && !flare && !out_host && peer_ok(c, head);
            uint32_t want = 0;
            if (direct) want = ((flags & BHR_GATHER_U8) ? BHR_OUT_U8 : 0u) | ((flags & BHR_GATHER_PEER) ? BHR_OUT_F32 : 0u);
            else want = BHR_OUT_F32 | ((gather && !flare && (flags & BHR_GATHER_U8)) ? BHR_OUT_U8 : 0u);
            Chunk chunks[PIPE_MAX_CHUNKS];
            const int n_chunks = plan_chunks(c, halo && k > 0, halo && k < n - 1, bhr_bloom_v_tile_rows(c), pipelined, chunks);
            if (pass == 0) {
                for (int ci = 0; ci < n_chunks; ++ci)
                    if (!chunks[ci].needs_halo)
                        BHR_TRY(bhr_launch_bloom_v_rows(c, with_bloom, chunks[ci].r0, chunks[ci].r1, want, direct ? head->d_gather_u8 : nullptr,
                                                        direct ? head->d_gather : nullptr));
                if (pipelined && with_bloom && c->bloom_split) BHR_HIP(hipEventRecord(p->halo_ready, c->stream));
                continue;
            }
            if (halo) {
                if (!c->bloom_split) {
                    BHR_HIP(hipStreamWaitEvent(c->stream, p->halo_in, 0));
                } else {
                    const int reach = 16 * (bhr_split_nt(c->bloom_R) - 1) + 32;
                    for (int q = 0; q < n; ++q) {
                        if (q == k || (live && !live[q]) || !needs_rows_of(c, ctxs[q], reach)) continue;
                        BHR_HIP(hipStreamWaitEvent(c->stream, ((TilePipe *)ctxs[q]->pipe)->halo_ready, 0));
                    }
                }
            }
            for (int ci = 0; ci < n_chunks; ++ci)
                if (chunks[ci].needs_halo)
                    BHR_TRY(bhr_launch_bloom_v_rows(c, with_bloom, chunks[ci].r0, chunks[ci].r1, want, direct ? head->d_gather_u8 : nullptr,
                                                    direct ? head->d_gather : nullptr));
            c->slots[c->active_slot].have = direct ? 0u : want;           // what sits in the tile's OWN buffers
            c->last_flags = (int32_t)flags;
            c->timing_valid = 1;
        }
    if (flare) BHR_TRY(flare_pass(ctxs, n, live));
    if (gather)
        for (int k = 0; k < n; ++k) {
            if (live && !live[k]) continue;
            bhr_ctx *c = ctxs[k];
            if (!flare && !out_host && peer_ok(c, head)) continue;    // stored directly
            BHR_HIP(hipSetDevice(c->cfg.device));
            if (flags & BHR_GATHER_U8) BHR_TRY(bhr_ensure_outputs(c, BHR_OUT_U8));
            BHR_TRY(queue_push(head, c, flags, c->stream));
        }
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_HIP(hipEventRecord(ctxs[k]->ev[2], ctxs[k]->stream));      // frame_ms = first march launch .. rows landed
    }
    // BHR_GROUP_ASYNC: where the kernels themselves have stored everything the frame produces, the call ends here; the frame
    // buffer's owner waits for every tile on the DEVICE, so whatever is queued on its stream next (bhr_read_gathered*) sees
    // the whole frame
    bool async = (flags & BHR_GROUP_ASYNC) && gather && !flare && !out_host && with_bloom;
    for (int k = 0; k < n && async; ++k)
        if ((!live || live[k]) && (!ctxs[k]->bloom_split || !peer_ok(ctxs[k], head))) async = false;
    for (int k = 0; k < n; ++k) {
        if (live && !live[k]) continue;
        TilePipe *p = (TilePipe *)ctxs[k]->pipe;
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_HIP(hipEventRecord(p->frame_done, ctxs[k]->stream));
        p->in_flight = async ? 1 : 0;
    }
    if (async) {
        BHR_HIP(hipSetDevice(head->cfg.device));
        for (int k = 0; k < n; ++k)
            if ((!live || live[k]) && ctxs[k] != head) BHR_HIP(hipStreamWaitEvent(head->stream, ((TilePipe *)ctxs[k]->pipe)->frame_done, 0));
        return BHR_OK;
    }
    return finish(ctxs, n, live, out_host);
}

// nothing may be queued against the mappings when they are closed
void drain(bhr_ctx *ctx, TilePipe *p) {
    (void)hipSetDevice(ctx->cfg.device);
    if (ctx->scene_stream) (void)hipStreamSynchronize(ctx->scene_stream);
    if (ctx->stream && ctx->stream != ctx->scene_stream) (void)hipStreamSynchronize(ctx->stream);
    if (p && p->copy) (void)hipStreamSynchronize(p->copy);
}

void close_mappings(bhr_ctx *ctx, TilePipe *p) {
    drain(ctx, p);
    for (void *&o : p->opened) {
        if (o) (void)hipIpcCloseMemHandle(o);
        o = nullptr;
    }
    p->nb_hblur[0] = p->nb_hblur[1] = nullptr;
    p->nb_split[0].pb = p->nb_split[1].pb = nullptr;
    p->linked = 0;
}

}  // namespace

void bhr_pipe_free(bhr_ctx *ctx) {
    TilePipe *p = (TilePipe *)ctx->pipe;
    if (!p) return;
    close_mappings(ctx, p);
    if (p->copy) (void)hipStreamDestroy(p->copy);
    hipEvent_t evs[] = {p->halo_ready, p->halo_in, p->frame_done};
    for (hipEvent_t ev : evs)
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
    const bhr_options &opt = ctxs[0]->opt;
    bool threaded = n_live > 1 && distinct_devices;
    if (opt.group_threads >= 0) threaded = n_live > 1 && opt.group_threads != 0;              // BHR_GROUP_THREADS: test knob, force / forbid
    // explicit flags win; without them BHR_GROUP_SCHEDULE decides, else render_tiles does (by post-pass and device layout)
    int schedule = distinct_devices ? -2 : -1;
    if (opt.group_schedule >= 0) schedule = opt.group_schedule != 0;
    if (flags & BHR_GROUP_SERIAL) schedule = 0;
    else if (flags & BHR_GROUP_PIPELINED) schedule = 1;
    return render_tiles(ctxs, n, cam, flags, out_host, live, threaded, schedule);
}

int32_t bhr_group_render(bhr_ctx **ctxs, int32_t n, const bhr_camera *cam, uint32_t flags, float *out_host) {
    return bhr_group_render_subset(ctxs, n, cam, flags, out_host, nullptr);
}

int32_t bhr_group_sync(bhr_ctx **ctxs, int32_t n) {
    if (!ctxs || n <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_group_sync: bad argument");
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k]) return bhr_fail(BHR_ERR_INVALID, "bhr_group_sync: null context %d", k);
        BHR_HIP(hipSetDevice(ctxs[k]->cfg.device));
        BHR_HIP(hipStreamSynchronize(ctxs[k]->stream));
        if (ctxs[k]->pipe) ((TilePipe *)ctxs[k]->pipe)->in_flight = 0;
    }
    return BHR_OK;
}

// ---- one process per tile (bench.py --strong under torchrun when a rank sees only its own GPU) -------------------------
// Same frame, same kernels as above; what changes is who talks to whom.  Every rank owns ONE tile.  Device memory crosses
// the process boundary through hipIpcMemHandle -- the neighbours' planes (the H pass of a split frame stores its halo rows
// into them; an exact-f32 frame pulls from them) and tile 0's frame buffers (the V pass stores into them) -- ordering
// crosses it through two counters per rank in host shared memory: a rank waits on the HOST for its own H pass, publishes
// the frame number, and launches the V pass of its edge rows once its neighbours have published theirs -- the halo rows
// are complete by then, no inter-process event is needed.  The frame ends with every rank synchronising its stream and
// publishing `done`; a rank returns when all have (nobody stores into a neighbour's planes while they are still read).
int32_t bhr_tile_export(bhr_ctx *ctx, uint32_t gather_flags, bhr_tile_handles *out) {
    if (!ctx || !out) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_export: bad argument");
    BHR_TRY(bhr_enter(ctx));
    BHR_TRY(bhr_activate_slot(ctx, 0));                                      // tile renders use slot 0, whatever bhr_render left active
    BHR_TRY(bhr_frame_begin(ctx, 0));                                        // the context's arithmetic picks the post-pass, allocates its planes
    memset(out, 0, sizeof(*out));
    static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(out->hblur), "handle size");
    hipIpcMemHandle_t h;
    if (ctx->bloom_split) {
        bhr_split_geom g;
        bhr_split_geometry(ctx, &g);
        BHR_HIP(hipIpcGetMemHandle(&h, ctx->slots[0].d_pb));
        out->reserved[0] = 1;
        out->reserved[1] = g.pbr;
        out->reserved[2] = g.GR;
    } else {
        BHR_HIP(hipIpcGetMemHandle(&h, ctx->slots[0].d_hblur_base));        // handles name allocations: the planes start BHR_HBLUR_PAD_ROWS rows in
    }
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
    BHR_TRY(bhr_activate_slot(ctx, 0));
    BHR_TRY(bhr_frame_begin(ctx, 0));
    const int split = ctx->bloom_split;
    const int min_rows = split ? 16 * bhr_split_nt(ctx->bloom_R) + 16 : ctx->bloom_R;   // a tile's planes must not reach past its neighbours
    int expect = 0;
    for (int k = 0; k < world; ++k) {
        if (all[k].row0 != expect) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: tile %d starts at row %d, expected %d", k, all[k].row0, expect);
        expect += all[k].rows;
        if (world > 1 && all[k].rows < min_rows)
            return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: tile %d has %d rows, fewer than the %d the bloom's halo spans (use bhr_group_render)", k, all[k].rows, min_rows);
        if (all[k].reserved[0] != split) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: tile %d runs the other post-pass arithmetic", k);
    }
    if (expect != ctx->cfg.height || all[rank].row0 != ctx->cfg.row0 || all[rank].rows != ctx->rows)
        return bhr_fail(BHR_ERR_INVALID, "bhr_tile_connect: the handles do not describe this frame / this rank's tile");
    const int with_up = rank > 0, with_down = rank < world - 1;
    BHR_TRY(ensure_pipe(ctx));
    TilePipe *p = (TilePipe *)ctx->pipe;
    close_mappings(ctx, p);                                                  // a re-connect: drained first
    auto open = [&](const uint8_t *raw, void **out, int slot) -> int32_t {
        hipIpcMemHandle_t h;
        memcpy(&h, raw, sizeof(h));
        BHR_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
        p->opened[slot] = *out;
        return BHR_OK;
    };
    const size_t pad = (size_t)BHR_HBLUR_PAD_ROWS * ctx->cfg.width;     // the handle opens the allocation; the f32 planes start `pad` floats in
    for (int side = 0; side < 2; ++side) {
        if (!(side == 0 ? with_up : with_down)) continue;
        const bhr_tile_handles &nb = all[side == 0 ? rank - 1 : rank + 1];
        void *base = nullptr;
        BHR_TRY(open(nb.hblur, &base, side));
        if (split) {
            p->nb_split[side].pb = base;
            p->nb_split[side].pbr = nb.reserved[1];
            p->nb_split[side].gr = nb.reserved[2];
        } else {
            p->nb_hblur[side] = (float *)base + pad;
            p->nb_rows[side] = nb.rows;
        }
    }
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
    p->split = split;
    p->shm = shm;
    p->frame = 0;
    p->linked = 1;
    return BHR_OK;
}

namespace {
// waits until the counter has reached `want`; gives up after 20 s of wall-clock time (a rank that died must not hang the
// others for ever)
int32_t wait_counter(volatile uint64_t *c, uint64_t want, const char *what, int peer) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(20);
    for (uint64_t spins = 0; __atomic_load_n(c, __ATOMIC_ACQUIRE) < want; ++spins) {
        if ((spins & 0xfff) == 0xfff) {
            std::this_thread::yield();
            if (std::chrono::steady_clock::now() > deadline)
                return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: timed out waiting for rank %d (%s)", peer, what);
        }
    }
    return BHR_OK;
}

int32_t tile_render_linked(bhr_ctx *ctx, TilePipe *p, const bhr_camera *cam, uint32_t flags) {
    const int with_bloom = (flags & BHR_SKIP_BLOOM) ? 0 : 1;
    const int rank = p->rank, world = p->world;
    const size_t R = ctx->bloom_R, W = ctx->cfg.width, rows = ctx->rows;
    const uint64_t frame = ++p->frame;
    volatile uint64_t *mine = p->shm + (size_t)rank * BHR_TILE_SHM_WORDS;
    BHR_TRY(bhr_enter(ctx));
    BHR_TRY(bhr_activate_slot(ctx, 0));
    ctx->cur_slot = -1;
    ctx->last_slot = -1;
    BHR_TRY(bhr_frame_begin(ctx, flags));
    ctx->group_time_march = (flags & BHR_GROUP_TIME_MARCH) ? 1 : 0;
    if (with_bloom && ctx->bloom_split != p->split)
        return bhr_fail(BHR_ERR_INVALID, "bhr_tile_render: the flags select the other post-pass arithmetic than the one the tiles were connected for");
    const bool halo = with_bloom && world > 1;
    ctx->n_mirrors = 0;
    if (halo && p->split)
        for (int side = 0; side < 2; ++side)
            if (p->nb_split[side].pb) {
                ctx->mirrors[ctx->n_mirrors].pb = p->nb_split[side].pb;
                ctx->mirrors[ctx->n_mirrors].pbr = p->nb_split[side].pbr;
                ctx->mirrors[ctx->n_mirrors].gr = p->nb_split[side].gr;
                ctx->n_mirrors += 1;
            }
    // march -> H pass (the neighbours' halo rows with it) on the tile's stream
    int32_t rc = bhr_launch_march(ctx, cam, flags);
    if (rc == BHR_OK && with_bloom) rc = bhr_launch_bloom_h(ctx);
    ctx->n_mirrors = 0;
    BHR_TRY(rc);
    BHR_HIP(hipEventRecord(p->halo_ready, ctx->stream));
    uint32_t want = ((flags & BHR_GATHER_U8) ? BHR_OUT_U8 : 0u) | ((flags & BHR_GATHER_PEER) ? BHR_OUT_F32 : 0u);
    const bool direct = want != 0;
    if (!direct) want = ctx->out_want;
    Chunk chunks[PIPE_MAX_CHUNKS];
    const int n_chunks = plan_chunks(ctx, halo && rank > 0, halo && rank < world - 1, bhr_bloom_v_tile_rows(ctx), true, chunks);
    auto v_chunk = [&](int ci) -> int32_t {
        return bhr_launch_bloom_v_rows(ctx, with_bloom, chunks[ci].r0, chunks[ci].r1, want, direct ? p->gather_u8 : nullptr, direct ? p->gather_f32 : nullptr);
    };
    // the rows that need no halo are queued now: they run while this rank waits for its neighbours below
    for (int ci = 0; ci < n_chunks; ++ci)
        if (!chunks[ci].needs_halo) BHR_TRY(v_chunk(ci));
    if (halo) {
        // my rows are blurred (and, split frames, stored in the neighbours' planes): tell them; go on once they have said the same
        BHR_HIP(hipEventSynchronize(p->halo_ready));
        __atomic_store_n(mine + 0, frame, __ATOMIC_RELEASE);
        for (int side = 0; side < 2; ++side) {
            const bool have_nb = p->split ? p->nb_split[side].pb != nullptr : p->nb_hblur[side] != nullptr;
            if (!have_nb) continue;
            const int nb = side == 0 ? rank - 1 : rank + 1;
            BHR_TRY(wait_counter(p->shm + (size_t)nb * BHR_TILE_SHM_WORDS, frame, "halo rows", nb));
            if (p->split) continue;                                          // its H pass has put the rows here already
            const size_t nb_rows = p->nb_rows[side], nb_plane = (nb_rows + 2 * R) * W, my_plane = (rows + 2 * R) * W;
            const size_t src_row = side == 0 ? nb_rows : R;                  // its last / first R rows (own rows live at [R, R + rows))
            const size_t dst_row = side == 0 ? 0 : R + rows;
            for (int c = 0; c < 3; ++c)
                BHR_HIP(hipMemcpyAsync(ctx->d_hblur + c * my_plane + dst_row * W, p->nb_hblur[side] + c * nb_plane + src_row * W,
                                       R * W * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    for (int ci = 0; ci < n_chunks; ++ci)
        if (chunks[ci].needs_halo) BHR_TRY(v_chunk(ci));
    BHR_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    ctx->slots[ctx->active_slot].have = direct ? 0u : want;
    ctx->last_flags = (int32_t)flags;
    ctx->timing_valid = 1;
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    // end of frame: my rows have landed and nobody reads my halo rows any more once every rank has said `done`
    __atomic_store_n(mine + 1, frame, __ATOMIC_RELEASE);
    for (int k = 0; k < world; ++k)
        BHR_TRY(wait_counter(p->shm + (size_t)k * BHR_TILE_SHM_WORDS + 1, frame, "end of frame", k));
    return BHR_OK;
}
}  // namespace

int32_t bhr_tile_render(bhr_ctx *ctx, const bhr_camera *cam, uint32_t flags) {
    if (!ctx || !cam) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_render: bad argument");
    TilePipe *p = (TilePipe *)ctx->pipe;
    if (!p || !p->linked) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: call bhr_tile_connect first (a failed frame breaks the link: connect again)");
    if (flags & BHR_LENS_FLARE) return bhr_fail(BHR_ERR_INVALID, "bhr_tile_render: the lens flare needs the one-process path (bhr_group_render)");
    if ((flags & BHR_GATHER_PEER) && !p->gather_f32) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: no f32 frame buffer was exported by rank 0");
    if ((flags & BHR_GATHER_U8) && !p->gather_u8) return bhr_fail(BHR_ERR_STATE, "bhr_tile_render: no u8 frame buffer was exported by rank 0");
    const int32_t rc = tile_render_linked(ctx, p, cam, flags);
    if (rc != BHR_OK) {
        // a failed frame leaves the counters of the ranks out of step and may leave stores / copies into the neighbours' mappings
        // queued: drain, and refuse further frames until the ranks have connected again (later calls fail fast instead of
        // spinning out a time-out each)
        const std::string msg = bhr_last_error();
        drain(ctx, p);
        p->linked = 0;
        return bhr_fail(rc, "%s", msg.c_str());
    }
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
