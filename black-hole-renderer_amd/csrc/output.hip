// output.hip -- PNG encoder and the pipelined frame sink (include/bhr_output.h).
//
// Host code: filtering + zlib deflate on worker threads, fed by asynchronous device-to-host copies of
// the device-quantised frame.  Reference counterpart: save_image (render.py:420-425) and the PIL pool
// of render_video (render.py:4412-4413, 4458-4467).
#include "bhr_internal.h"
#include "../../include/bhr_output.h"

#include <pthread.h>
#include <signal.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

inline void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

// One PNG chunk: length, type, data, CRC over type + data.  Returns the bytes written.
size_t put_chunk(uint8_t *out, const char type[4], const uint8_t *data, uint32_t len) {
    put32(out, len);
    memcpy(out + 4, type, 4);
    if (len) memcpy(out + 8, data, len);
    uint32_t crc = (uint32_t)crc32(0L, out + 4, len + 4);
    put32(out + 8 + len, crc);
    return (size_t)len + 12;
}

// Filter one row (bpp = 3) with the type that minimises the sum of absolute signed residuals -- the
// heuristic of the PNG specification (12.8).  dst gets the filter byte + the filtered row.  `up` is the
// previous row or a row of zeros; the loops are branch-free so that the compiler vectorises them.
void filter_row(const uint8_t *__restrict__ cur, const uint8_t *__restrict__ up, int nbytes, uint8_t *__restrict__ dst,
                uint8_t *__restrict__ scratch) {
    uint8_t *cand[5];
    for (int f = 0; f < 5; ++f) cand[f] = scratch + (size_t)f * nbytes;
    const int head = nbytes < 3 ? nbytes : 3;
    memcpy(cand[0], cur, nbytes);
    {
        uint8_t *o = cand[1];
        for (int i = 0; i < head; ++i) o[i] = cur[i];
        for (int i = 3; i < nbytes; ++i) o[i] = (uint8_t)(cur[i] - cur[i - 3]);
    }
    {
        uint8_t *o = cand[2];
        for (int i = 0; i < nbytes; ++i) o[i] = (uint8_t)(cur[i] - up[i]);
    }
    {
        uint8_t *o = cand[3];
        for (int i = 0; i < head; ++i) o[i] = (uint8_t)(cur[i] - (up[i] >> 1));
        for (int i = 3; i < nbytes; ++i) o[i] = (uint8_t)(cur[i] - (((int)cur[i - 3] + (int)up[i]) >> 1));
    }
    {
        uint8_t *o = cand[4];
        for (int i = 0; i < head; ++i) o[i] = (uint8_t)(cur[i] - up[i]);          // paeth(0, b, 0) = b
        for (int i = 3; i < nbytes; ++i) {
            const int a = cur[i - 3], b = up[i], c = up[i - 3];
            const int p = a + b - c;
            const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
            const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            o[i] = (uint8_t)(cur[i] - pred);
        }
    }
    long best_cost = -1;
    int best = 0;
    for (int f = 0; f < 5; ++f) {
        const uint8_t *o = cand[f];
        unsigned cost = 0;
        for (int i = 0; i < nbytes; ++i) {
            const int v = (int8_t)o[i];
            cost += (unsigned)(v < 0 ? -v : v);
        }
        if (best_cost < 0 || (long)cost < best_cost) { best_cost = (long)cost; best = f; }
    }
    dst[0] = (uint8_t)best;
    memcpy(dst + 1, cand[best], nbytes);
}

struct Band {
    std::vector<uint8_t> comp;
    uLong adler = 1;
    size_t raw_len = 0;
    int rc = Z_OK;
};

// Filter + raw-deflate rows [r0, r1).  Non-final bands end on a sync flush (byte aligned, no final bit),
// so that the bands concatenate into one valid deflate stream.
void deflate_band(const uint8_t *rgb, int w, int r0, int r1, int level, bool last, Band *band) {
    const int nbytes = 3 * w;
    const size_t line = (size_t)nbytes + 1;
    std::vector<uint8_t> filt((size_t)(r1 - r0) * line), scratch((size_t)5 * nbytes), zero_row((size_t)nbytes, 0);
    for (int r = r0; r < r1; ++r)
        filter_row(rgb + (size_t)r * nbytes, r > 0 ? rgb + (size_t)(r - 1) * nbytes : zero_row.data(), nbytes,
                   filt.data() + (size_t)(r - r0) * line, scratch.data());
    band->raw_len = filt.size();
    {   // adler32 takes uInt lengths
        uLong a = adler32(0L, Z_NULL, 0);
        size_t done = 0;
        while (done < filt.size()) {
            const size_t n = filt.size() - done < (1u << 30) ? filt.size() - done : (1u << 30);
            a = adler32(a, filt.data() + done, (uInt)n);
            done += n;
        }
        band->adler = a;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    band->rc = deflateInit2(&zs, level, Z_DEFLATED, -15, 9, level == 0 ? Z_DEFAULT_STRATEGY : Z_FILTERED);
    if (band->rc != Z_OK) return;
    band->comp.resize(filt.size() + filt.size() / 256 + 4096);
    size_t in_done = 0, out_done = 0;
    for (;;) {
        const size_t in_n = filt.size() - in_done < (1u << 30) ? filt.size() - in_done : (1u << 30);
        const bool final_piece = in_done + in_n == filt.size();
        zs.next_in = filt.data() + in_done;
        zs.avail_in = (uInt)in_n;
        const int flush = final_piece ? (last ? Z_FINISH : Z_SYNC_FLUSH) : Z_NO_FLUSH;
        int rc;
        do {   // until this piece is consumed and, on a flush, everything pending has been emitted
            if (band->comp.size() - out_done < (1u << 16)) band->comp.resize(band->comp.size() * 2);
            const size_t room = band->comp.size() - out_done < (1u << 30) ? band->comp.size() - out_done : (1u << 30);
            zs.next_out = band->comp.data() + out_done;
            zs.avail_out = (uInt)room;
            rc = deflate(&zs, flush);
            out_done += room - zs.avail_out;
            if (rc == Z_STREAM_ERROR) { band->rc = rc; deflateEnd(&zs); return; }
        } while (zs.avail_out == 0 || (flush == Z_FINISH && rc != Z_STREAM_END));
        in_done += in_n;
        if (final_piece) break;
    }
    deflateEnd(&zs);
    band->comp.resize(out_done);
}

int32_t encode_png(const uint8_t *rgb, int w, int h, int level, int threads, uint8_t *out, int64_t cap, int64_t *out_len) {
    if (!rgb || !out || !out_len || w <= 0 || h <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_png_encode: bad argument");
    if (level < 0 || level > 9) return bhr_fail(BHR_ERR_INVALID, "bhr_png_encode: zlib level %d outside 0..9", level);
    if (cap < bhr_png_bound(w, h)) return bhr_fail(BHR_ERR_INVALID, "bhr_png_encode: buffer smaller than bhr_png_bound");
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int bands = threads;
    if (bands > h / 16) bands = h / 16 > 0 ? h / 16 : 1;     // a band shorter than 16 rows is not worth a stream
    std::vector<Band> band(bands);
    std::vector<std::thread> pool;
    for (int b = 0; b < bands; ++b) {
        const int r0 = (int)((int64_t)h * b / bands), r1 = (int)((int64_t)h * (b + 1) / bands);
        if (b + 1 < bands) pool.emplace_back(deflate_band, rgb, w, r0, r1, level, false, &band[b]);
        else deflate_band(rgb, w, r0, r1, level, true, &band[b]);
    }
    for (auto &t : pool) t.join();
    for (int b = 0; b < bands; ++b)
        if (band[b].rc != Z_OK) return bhr_fail(BHR_ERR_HIP, "bhr_png_encode: zlib error %d in band %d", band[b].rc, b);

    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t *p = out;
    memcpy(p, sig, 8);
    p += 8;
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)w);
    put32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   // 8-bit, colour type 2 (RGB), no interlace
    p += put_chunk(p, "IHDR", ihdr, 13);

    // zlib stream = 2-byte header + the spliced raw deflate bands + adler32 of all filtered bytes
    size_t zlen = 2 + 4;
    for (auto &b : band) zlen += b.comp.size();
    if ((int64_t)(p - out) + (int64_t)zlen + 12 * (int64_t)(zlen / (1u << 30) + 2) + 12 > cap)
        return bhr_fail(BHR_ERR_INVALID, "bhr_png_encode: output exceeds the bound (incompressible frame?)");
    std::vector<uint8_t> z(zlen);
    z[0] = 0x78;
    z[1] = level >= 7 ? 0xDA : (level >= 6 ? 0x9C : (level >= 2 ? 0x5E : 0x01));
    size_t at = 2;
    uLong adler = 1;
    for (auto &b : band) {
        memcpy(z.data() + at, b.comp.data(), b.comp.size());
        at += b.comp.size();
        adler = adler32_combine(adler, b.adler, (z_off_t)b.raw_len);
    }
    put32(z.data() + at, (uint32_t)adler);
    for (size_t done = 0; done < zlen;) {
        const size_t n = zlen - done < (1u << 30) ? zlen - done : (1u << 30);
        p += put_chunk(p, "IDAT", z.data() + done, (uint32_t)n);
        done += n;
    }
    p += put_chunk(p, "IEND", nullptr, 0);
    *out_len = (int64_t)(p - out);
    return BHR_OK;
}

int32_t write_file_atomic(const char *path, const uint8_t *data, size_t len) {
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return bhr_fail(BHR_ERR_INVALID, "cannot open %s for writing", tmp.c_str());
    const size_t put = fwrite(data, 1, len, f);
    const int rc = fclose(f);
    if (put != len || rc != 0) { remove(tmp.c_str()); return bhr_fail(BHR_ERR_INVALID, "short write to %s", tmp.c_str()); }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return bhr_fail(BHR_ERR_INVALID, "cannot rename %s to %s", tmp.c_str(), path); }
    return BHR_OK;
}

}  // namespace

struct bhr_sink {
    bhr_ctx *ctx = nullptr;
    int w = 0, h = 0, level = 1;
    size_t frame_bytes = 0;
    bool on_device = false;        // level BHR_PNG_DEVICE: the slot receives finished PNG bytes (png_device.hip)
    size_t host_bytes = 0;         // size of a slot's pinned buffer: the raw frame, or the bound of the device encoder
    struct Slot {
        uint8_t *host = nullptr;
        hipEvent_t ev = nullptr;
        uint8_t *dev = nullptr;    // device encoder: PNG bytes in HBM
        uint32_t *d_meta = nullptr, *h_meta = nullptr;   // {length, error, ..}: device, and its pinned copy
    };
    std::vector<Slot> slots;
    std::deque<int> free_slots;
    struct Job { int slot; std::string path; };
    std::deque<Job> jobs;
    int in_flight = 0;
    bool stop = false;
    std::mutex mu;
    std::condition_variable cv_job, cv_free, cv_idle;
    std::vector<std::thread> workers;
    int32_t err = BHR_OK;
    std::string err_text;
    int64_t frames = 0, bytes = 0;

    void work() {
        (void)hipSetDevice(ctx->cfg.device);
        std::vector<uint8_t> png(on_device ? 0 : (size_t)bhr_png_bound(w, h));
        hipStream_t copy_stream = nullptr;          // device encoder: each worker fetches exactly the bytes of its file
        if (on_device) (void)hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) break;
                job = jobs.front();
                jobs.pop_front();
            }
            int32_t rc = BHR_OK;
            int64_t len = 0;
            Slot &sl = slots[job.slot];
            hipError_t e = hipEventSynchronize(sl.ev);
            if (e != hipSuccess) rc = bhr_fail(BHR_ERR_HIP, "frame sink: hipEventSynchronize: %s", hipGetErrorString(e));
            if (on_device) {
                if (rc == BHR_OK && (sl.h_meta[1] != 0 || sl.h_meta[0] > host_bytes))
                    rc = bhr_fail(BHR_ERR_STATE, "frame sink: the device encoder overran its bound (%u bytes)", sl.h_meta[0]);
                if (rc == BHR_OK) {
                    len = (int64_t)sl.h_meta[0];
                    e = copy_stream ? hipMemcpyAsync(sl.host, sl.dev, (size_t)len, hipMemcpyDeviceToHost, copy_stream) : hipErrorUnknown;
                    if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
                    if (e != hipSuccess) rc = bhr_fail(BHR_ERR_HIP, "frame sink: fetching the encoded frame: %s", hipGetErrorString(e));
                }
                if (rc == BHR_OK) rc = write_file_atomic(job.path.c_str(), sl.host, (size_t)len);
            } else {
                if (rc == BHR_OK) rc = encode_png(sl.host, w, h, level, 1, png.data(), (int64_t)png.size(), &len);
                if (rc == BHR_OK) rc = write_file_atomic(job.path.c_str(), png.data(), (size_t)len);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc != BHR_OK && err == BHR_OK) { err = rc; err_text = bhr_last_error(); }
                if (rc == BHR_OK) { frames += 1; bytes += len; }
                free_slots.push_back(job.slot);
                in_flight -= 1;
            }
            cv_free.notify_one();
            cv_idle.notify_all();
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

// ---- YUV4MPEG2 stream ---------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int q8(float x) { return (int)(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f); }   // render.py:4463

// One thread per 2x2 block: four luma samples, one Cb, one Cr.  FINAL is (h, w, 3) f32; out = Y (h*w) | Cb | Cr.
__global__ void rgb_to_yuv420_kernel(const float *__restrict__ rgb, uint8_t *__restrict__ out, int w, int h) {
    const int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y * blockDim.y + threadIdx.y;
    const int cw = w >> 1, ch = h >> 1;
    if (bx >= cw || by >= ch) return;
    uint8_t *Y = out, *U = out + (size_t)w * h, *V = U + (size_t)cw * ch;
    int sr = 0, sg = 0, sb = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = 2 * bx + dx, y = 2 * by + dy;
            const float *p = rgb + ((size_t)y * w + x) * 3;
            const int r = q8(p[0]), g = q8(p[1]), b = q8(p[2]);
            Y[(size_t)y * w + x] = (uint8_t)(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16);
            sr += r; sg += g; sb += b;
        }
    const int r = (sr + 2) >> 2, g = (sg + 2) >> 2, b = (sb + 2) >> 2;
    U[(size_t)by * cw + bx] = (uint8_t)(((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128);
    V[(size_t)by * cw + bx] = (uint8_t)(((112 * r - 94 * g - 18 * b + 128) >> 8) + 128);
}

}  // namespace

struct bhr_y4m {
    bhr_ctx *ctx = nullptr;
    int w = 0, h = 0;
    size_t frame_bytes = 0;
    FILE *f = nullptr;
    // every ring slot owns its device planes: the conversion rides the stream of the frame slot that rendered the frame
    // (bhr_enter_frame), so two frames in flight convert side by side and the scene stream stays free
    struct Slot { uint8_t *host = nullptr; uint8_t *dev = nullptr; hipEvent_t ev = nullptr; };
    std::vector<Slot> slots;
    std::deque<int> free_slots, jobs;      // jobs in submission order: ONE writer keeps the frame order
    int in_flight = 0;
    bool stop = false;
    std::mutex mu;
    std::condition_variable cv_job, cv_free, cv_idle;
    std::thread writer;
    int32_t err = BHR_OK;
    std::string err_text;
    int64_t frames = 0, bytes = 0;

    void work() {
        (void)hipSetDevice(ctx->cfg.device);
        // a reader that goes away (an encoder exiting on a FIFO / pipe) must surface as a short write, not kill a C
        // embedder with SIGPIPE: the signal is blocked in this thread, write() then fails with EPIPE
        sigset_t block;
        sigemptyset(&block);
        sigaddset(&block, SIGPIPE);
        (void)pthread_sigmask(SIG_BLOCK, &block, nullptr);
        for (;;) {
            int slot;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                slot = jobs.front();
                jobs.pop_front();
            }
            int32_t rc = BHR_OK;
            const hipError_t e = hipEventSynchronize(slots[slot].ev);
            if (e != hipSuccess) rc = bhr_fail(BHR_ERR_HIP, "y4m stream: hipEventSynchronize: %s", hipGetErrorString(e));
            if (rc == BHR_OK && (fwrite("FRAME\n", 1, 6, f) != 6 || fwrite(slots[slot].host, 1, frame_bytes, f) != frame_bytes))
                rc = bhr_fail(BHR_ERR_INVALID, "y4m stream: short write (reader gone?)");
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc != BHR_OK && err == BHR_OK) { err = rc; err_text = bhr_last_error(); }
                if (rc == BHR_OK) { frames += 1; bytes += 6 + (int64_t)frame_bytes; }
                free_slots.push_back(slot);
                in_flight -= 1;
            }
            cv_free.notify_one();
            cv_idle.notify_all();
        }
    }
};

extern "C" {

int32_t bhr_y4m_open(bhr_ctx *ctx, const char *path, int32_t fps_num, int32_t fps_den, int32_t n_slots, bhr_y4m **out) {
    if (!ctx || !path || !out || fps_num <= 0 || fps_den <= 0 || n_slots < 1 || n_slots > 256)
        return bhr_fail(BHR_ERR_INVALID, "bhr_y4m_open: bad argument");
    *out = nullptr;
    if ((ctx->cfg.width & 1) || (ctx->rows & 1))
        return bhr_fail(BHR_ERR_INVALID, "bhr_y4m_open: 4:2:0 needs even width and height, got %dx%d", ctx->cfg.width, ctx->rows);
    BHR_TRY(bhr_enter(ctx));
    bhr_y4m *s = new bhr_y4m();
    s->ctx = ctx;
    s->w = ctx->cfg.width;
    s->h = ctx->rows;
    s->frame_bytes = (size_t)s->w * s->h * 3 / 2;
    s->f = fopen(path, "wb");
    if (!s->f) { delete s; return bhr_fail(BHR_ERR_INVALID, "bhr_y4m_open: cannot open %s for writing", path); }
    setvbuf(s->f, nullptr, _IOFBF, 1 << 22);
    fprintf(s->f, "YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C420jpeg XCOLORRANGE=LIMITED\n", s->w, s->h, fps_num, fps_den);
    hipError_t e = hipSuccess;
    s->slots.resize(n_slots);
    for (int k = 0; k < n_slots && e == hipSuccess; ++k) {
        e = hipHostMalloc((void **)&s->slots[k].host, s->frame_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->slots[k].dev, s->frame_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->slots[k].ev, hipEventDisableTiming);
        s->free_slots.push_back(k);
    }
    s->writer = std::thread(&bhr_y4m::work, s);
    if (e != hipSuccess) {
        bhr_y4m_close(s);
        return bhr_fail(BHR_ERR_HIP, "bhr_y4m_open: %s", hipGetErrorString(e));
    }
    *out = s;
    return BHR_OK;
}

int32_t bhr_y4m_submit(bhr_y4m *s) {
    if (!s) return bhr_fail(BHR_ERR_INVALID, "bhr_y4m_submit: null stream");
    int slot;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        if (s->err != BHR_OK) return bhr_fail(s->err, "y4m stream: %s", s->err_text.c_str());
        s->cv_free.wait(lk, [&] { return !s->free_slots.empty(); });
        slot = s->free_slots.front();
        s->free_slots.pop_front();
        s->in_flight += 1;
    }
    bhr_ctx *ctx = s->ctx;
    int32_t rc = bhr_enter_frame(ctx);       // the stream that rendered the last frame (behind its post-passes)
    hipError_t e = hipSuccess;
    if (rc == BHR_OK) rc = bhr_ensure_outputs(ctx, BHR_OUT_F32);     // the conversion reads the f32 frame (a context that keeps only u8 rows gets it on demand)
    if (rc == BHR_OK) {
        dim3 block(32, 8), grid(((s->w >> 1) + 31) / 32, ((s->h >> 1) + 7) / 8);
        hipLaunchKernelGGL(rgb_to_yuv420_kernel, grid, block, 0, ctx->stream, ctx->d_final, s->slots[slot].dev, s->w, s->h);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(s->slots[slot].host, s->slots[slot].dev, s->frame_bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(s->slots[slot].ev, ctx->stream);
        const int32_t rc_leave = bhr_leave_frame(ctx);
        if (rc == BHR_OK) rc = rc_leave;
    }
    if (e != hipSuccess || rc != BHR_OK) {
        std::lock_guard<std::mutex> lk(s->mu);
        s->free_slots.push_back(slot);
        s->in_flight -= 1;
        return e != hipSuccess ? bhr_fail(BHR_ERR_HIP, "bhr_y4m_submit: %s", hipGetErrorString(e)) : rc;
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->jobs.push_back(slot);
    }
    s->cv_job.notify_one();
    return BHR_OK;
}

int32_t bhr_y4m_drain(bhr_y4m *s, int64_t *frames_written, int64_t *bytes_written) {
    if (!s) return bhr_fail(BHR_ERR_INVALID, "bhr_y4m_drain: null stream");
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv_idle.wait(lk, [&] { return s->in_flight == 0; });
    if (s->f) fflush(s->f);
    if (frames_written) *frames_written = s->frames;
    if (bytes_written) *bytes_written = s->bytes;
    if (s->err != BHR_OK) {
        const int32_t code = s->err;
        const std::string text = s->err_text;
        lk.unlock();
        return bhr_fail(code, "y4m stream: %s", text.c_str());
    }
    return BHR_OK;
}

void bhr_y4m_close(bhr_y4m *s) {
    if (!s) return;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_idle.wait(lk, [&] { return s->in_flight == 0; });
        s->stop = true;
    }
    s->cv_job.notify_all();
    if (s->writer.joinable()) s->writer.join();
    if (s->f) fclose(s->f);
    (void)hipSetDevice(s->ctx->cfg.device);
    (void)bhr_enter(s->ctx);                             // joins the frame slots' streams: a slot's planes may still be a copy's source
    (void)hipStreamSynchronize(s->ctx->scene_stream);
    for (auto &sl : s->slots) {
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
    }
    delete s;
}

int64_t bhr_png_bound(int32_t w, int32_t h) {
    if (w <= 0 || h <= 0) return 0;
    const int64_t raw = (int64_t)h * (3 * (int64_t)w + 1);
    return raw + raw / 128 + 65536;
}

int32_t bhr_png_encode(const uint8_t *rgb, int32_t w, int32_t h, int32_t level, int32_t threads, uint8_t *out, int64_t cap,
                       int64_t *out_len) {
    return encode_png(rgb, w, h, level, threads, out, cap, out_len);
}

int32_t bhr_png_write(const char *path, const uint8_t *rgb, int32_t w, int32_t h, int32_t level, int32_t threads) {
    if (!path) return bhr_fail(BHR_ERR_INVALID, "bhr_png_write: null path");
    if (w <= 0 || h <= 0) return bhr_fail(BHR_ERR_INVALID, "bhr_png_write: bad size %dx%d", w, h);
    std::vector<uint8_t> png((size_t)bhr_png_bound(w, h));
    int64_t len = 0;
    BHR_TRY(encode_png(rgb, w, h, level, threads, png.data(), (int64_t)png.size(), &len));
    return write_file_atomic(path, png.data(), (size_t)len);
}

int32_t bhr_sink_create(bhr_ctx *ctx, int32_t slots, int32_t workers, int32_t level, bhr_sink **out) {
    if (!ctx || !out || slots < 1 || slots > 256 || workers < 1 || workers > 256 || level < BHR_PNG_DEVICE || level > 9)
        return bhr_fail(BHR_ERR_INVALID, "bhr_sink_create: bad argument (slots %d, workers %d, level %d)", slots, workers, level);
    if (level == BHR_PNG_DEVICE && ctx->cfg.width > bhr_png_device_max_width())
        return bhr_fail(BHR_ERR_INVALID, "bhr_sink_create: the device PNG encoder takes frames up to %d pixels wide, this one has %d; "
                        "use a zlib level (host encoder)", bhr_png_device_max_width(), ctx->cfg.width);
    BHR_HIP(hipSetDevice(ctx->cfg.device));
    bhr_sink *s = new bhr_sink();
    s->ctx = ctx;
    s->w = ctx->cfg.width;
    s->h = ctx->rows;
    s->level = level;
    s->on_device = level == BHR_PNG_DEVICE;
    s->frame_bytes = (size_t)s->w * s->h * 3;
    s->host_bytes = s->on_device ? (size_t)bhr_png_device_bound(s->w, s->h) : s->frame_bytes;
    s->slots.resize(slots);
    for (int k = 0; k < slots; ++k) {
        hipError_t e = hipHostMalloc((void **)&s->slots[k].host, s->host_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->slots[k].ev, hipEventDisableTiming);
        if (e == hipSuccess && s->on_device) {
            e = hipMalloc((void **)&s->slots[k].dev, s->host_bytes);
            if (e == hipSuccess) e = hipMalloc((void **)&s->slots[k].d_meta, 4 * sizeof(uint32_t));
            if (e == hipSuccess) e = hipHostMalloc((void **)&s->slots[k].h_meta, 4 * sizeof(uint32_t), hipHostMallocDefault);
        }
        if (e != hipSuccess) {
            bhr_sink_destroy(s);
            return bhr_fail(BHR_ERR_HIP, "bhr_sink_create: %s", hipGetErrorString(e));
        }
        s->free_slots.push_back(k);
    }
    for (int k = 0; k < workers; ++k) s->workers.emplace_back(&bhr_sink::work, s);
    *out = s;
    return BHR_OK;
}

int32_t bhr_sink_submit(bhr_sink *s, const char *path) {
    if (!s || !path) return bhr_fail(BHR_ERR_INVALID, "bhr_sink_submit: bad argument");
    int slot;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_free.wait(lk, [&] { return !s->free_slots.empty(); });
        slot = s->free_slots.front();
        s->free_slots.pop_front();
        s->in_flight += 1;
    }
    bhr_ctx *ctx = s->ctx;
    hipError_t e = hipSuccess;
    int32_t rc = bhr_enter_frame(ctx);      // quantise, encode and copy ride the stream that rendered the frame
    if (rc == BHR_OK) rc = bhr_launch_quantize(ctx);
    if (rc == BHR_OK && s->on_device) {
        bhr_sink::Slot &sl = s->slots[slot];
        rc = bhr_launch_png_encode(ctx, ctx->d_final_u8, sl.dev, (int64_t)s->host_bytes, sl.d_meta);
        if (rc == BHR_OK) {
            e = hipMemcpyAsync(sl.h_meta, sl.d_meta, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(sl.ev, ctx->stream);
        }
    } else if (rc == BHR_OK) {
        e = hipMemcpyAsync(s->slots[slot].host, ctx->d_final_u8, s->frame_bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(s->slots[slot].ev, ctx->stream);
    }
    {
        const int32_t rc_leave = bhr_leave_frame(ctx);
        if (rc == BHR_OK) rc = rc_leave;
    }
    if (e != hipSuccess || rc != BHR_OK) {
        // device work queued before the failing step may still write this slot's buffers: the slot goes back to the free
        // list only once the frame slots' streams have drained (advisor finding, round 2)
        std::string msg = e != hipSuccess ? std::string("bhr_sink_submit: ") + hipGetErrorString(e) : std::string(bhr_last_error());
        (void)bhr_enter(ctx);
        (void)hipStreamSynchronize(ctx->scene_stream);
        std::lock_guard<std::mutex> lk(s->mu);
        s->free_slots.push_back(slot);
        s->in_flight -= 1;
        return bhr_fail(e != hipSuccess ? BHR_ERR_HIP : rc, "%s", msg.c_str());
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->jobs.push_back(bhr_sink::Job{slot, path});
    }
    s->cv_job.notify_one();
    return BHR_OK;
}

int32_t bhr_sink_drain(bhr_sink *s, int64_t *frames_written, int64_t *bytes_written) {
    if (!s) return bhr_fail(BHR_ERR_INVALID, "bhr_sink_drain: null sink");
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv_idle.wait(lk, [&] { return s->in_flight == 0; });
    if (frames_written) *frames_written = s->frames;
    if (bytes_written) *bytes_written = s->bytes;
    if (s->err != BHR_OK) {
        const int32_t code = s->err;
        const std::string text = s->err_text;
        s->err = BHR_OK;
        lk.unlock();
        return bhr_fail(code, "frame sink: %s", text.c_str());
    }
    return BHR_OK;
}

void bhr_sink_destroy(bhr_sink *s) {
    if (!s) return;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_idle.wait(lk, [&] { return s->in_flight == 0; });
        s->stop = true;
    }
    s->cv_job.notify_all();
    for (auto &t : s->workers) t.join();
    (void)hipSetDevice(s->ctx->cfg.device);
    for (auto &sl : s->slots) {
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.d_meta) (void)hipFree(sl.d_meta);
        if (sl.h_meta) (void)hipHostFree(sl.h_meta);
    }
    delete s;
}

}  // extern "C"
