/* bhr_output.h -- frame output of libbhr_hip.so: PNG encoding and the pipelined frame sink.
 *
 * Replaces the reference's host output path: save_image (render.py:420-425, PIL) for still images and
 * the two-thread PIL pool of render_video (render.py:4412-4413, 4458-4467), which caps the video driver
 * at a few frames per second once a frame renders in a millisecond.  Files are ordinary 8-bit RGB PNGs;
 * the decoded pixels equal (np.clip(frame, 0, 1) * 255).astype(np.uint8) of the reference exactly,
 * the compressed bytes differ (they also differ between PIL versions).
 */
#ifndef BHR_OUTPUT_H
#define BHR_OUTPUT_H

#include "bhr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Encode an (h, w, 3) u8 image.  level: zlib level 0..9 (PIL's default is 6).  threads > 1 deflates
 * row bands in parallel and splices them into one stream.  out must hold bhr_png_bound(w, h) bytes. */
BHR_API int64_t bhr_png_bound(int32_t w, int32_t h);
BHR_API int32_t bhr_png_encode(const uint8_t *rgb, int32_t w, int32_t h, int32_t level, int32_t threads,
                               uint8_t *out, int64_t cap, int64_t *out_len);
/* Image.fromarray(rgb).save(path): encode and write atomically (path.tmp, then rename). */
BHR_API int32_t bhr_png_write(const char *path, const uint8_t *rgb, int32_t w, int32_t h, int32_t level, int32_t threads);

/* PNG encoding on the device (csrc/png_device.hip).  The frame is filtered (the five PNG filters, the minimum sum of
 * absolute residuals per scanline) and entropy coded in HBM: one dynamic-Huffman deflate block and one IDAT chunk per
 * scanline, the prefix code of each scanline picked from a menu of 16 static codes (no LZ77 matches), chunk CRC-32 and
 * stream Adler-32 computed on the device.  Any PNG reader decodes the result to exactly the pixels of
 * bhr_read_final_u8; the files are ~15-25 % larger than zlib level 1 makes them and cost a fraction of a millisecond
 * of device time instead of ~50 ms of a host core per fhd frame.  Replaces the same reference code as the host
 * encoder (render.py:420-425, 4412-4467).
 *   bhr_png_device_bound: capacity that always suffices for a w x h frame.
 *   bhr_png_encode_device: quantise the context's FINAL layer and encode it; `out` (host) receives the file bytes.
 *   BHR_PNG_DEVICE as the `level` of bhr_sink_create: the sink encodes on the device, its workers only fetch the
 *     finished bytes (an exact-length copy on their own stream) and write the file. */
#define BHR_PNG_DEVICE (-1)
BHR_API int64_t bhr_png_device_bound(int32_t w, int32_t h);
/* Widest frame the device encoder takes (one scanline is coded by one block out of LDS): 17 000-odd pixels.
 * bhr_png_encode_device and bhr_sink_create(.., BHR_PNG_DEVICE, ..) fail with BHR_ERR_INVALID beyond it. */
BHR_API int32_t bhr_png_device_max_width(void);
BHR_API int32_t bhr_png_encode_device(bhr_ctx *ctx, uint8_t *out, int64_t cap, int64_t *out_len);
/* Entry k of the code menu, as the kernels use it (host only, no GPU needed; for inspection and tests):
 * codes[257] = (bit-reversed code << 4) | length for literals 0..255 and end-of-block, hdr_words[64] / *hdr_bits =
 * the deflate block header announcing that code, LSB first.  *n_tables receives the menu size. */
BHR_API int32_t bhr_png_device_menu(int32_t k, uint32_t *codes, uint32_t *hdr_words, uint32_t *hdr_bits, int32_t *n_tables);

/* Frame sink.  bhr_sink_submit quantises the context's FINAL layer on the device (save_image's
 * truncation), starts an asynchronous copy into one of `slots` pinned host buffers and returns; `workers`
 * host threads wait for the copy, encode and write `path`.  The caller goes on to render the next frame
 * on the same stream meanwhile.  submit blocks only while every slot is busy. */
typedef struct bhr_sink bhr_sink;
BHR_API int32_t bhr_sink_create(bhr_ctx *ctx, int32_t slots, int32_t workers, int32_t level, bhr_sink **out);
BHR_API int32_t bhr_sink_submit(bhr_sink *sink, const char *path);
/* Wait until every submitted frame is on disk; returns the first error a worker met, if any.
 * frames_written / bytes_written may be NULL. */
BHR_API int32_t bhr_sink_drain(bhr_sink *sink, int64_t *frames_written, int64_t *bytes_written);
BHR_API void bhr_sink_destroy(bhr_sink *sink);

/* Video stream without the PNG detour (replaces render.py:4497-4503, where the reference re-reads every PNG and
 * feeds libx264 through imageio/pyav with pixelformat yuv420p).  bhr_y4m_submit converts the context's FINAL layer
 * on the device -- the reference's u8 quantisation (render.py:4463), then BT.601 limited-range Y'CbCr with the
 * chroma of each 2x2 block taken from its rounded mean RGB (4:2:0, centre sited) -- copies the 1.5 bytes/pixel
 * asynchronously into a pinned ring and returns; ONE writer thread appends the frames in submission order to
 * `path` as YUV4MPEG2 ("YUV4MPEG2 W.. H.. F<num>:<den> Ip A1:1 C420jpeg XCOLORRANGE=LIMITED", then "FRAME\n" + Y, Cb, Cr
 * planes per frame).  `path` may be a FIFO or "-"-less file path; `ffmpeg -f yuv4mpegpipe -i path -c:v libx264 -pix_fmt
 * yuv420p out.mp4` turns it into the reference's MP4 (drivers.render_video does that when an ffmpeg binary exists).
 * Width and height must be even.  Integer arithmetic (exactly reproducible on the host):
 *   Y  = ((66 R + 129 G + 25 B + 128) >> 8) + 16,  Cb = ((-38 R - 74 G + 112 B + 128) >> 8) + 128,
 *   Cr = ((112 R - 94 G - 18 B + 128) >> 8) + 128,  chroma R,G,B = (sum of the 2x2 block + 2) >> 2. */
typedef struct bhr_y4m bhr_y4m;
BHR_API int32_t bhr_y4m_open(bhr_ctx *ctx, const char *path, int32_t fps_num, int32_t fps_den, int32_t slots, bhr_y4m **out);
BHR_API int32_t bhr_y4m_submit(bhr_y4m *stream);
/* Wait until every submitted frame has been written; first writer error, if any.  frames_written may be NULL. */
BHR_API int32_t bhr_y4m_drain(bhr_y4m *stream, int64_t *frames_written, int64_t *bytes_written);
/* Drains, closes the file and frees the stream. */
BHR_API void bhr_y4m_close(bhr_y4m *stream);

#ifdef __cplusplus
}
#endif
#endif /* BHR_OUTPUT_H */
