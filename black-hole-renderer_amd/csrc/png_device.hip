// png_device.hip -- PNG encoding of the quantised frame ON THE DEVICE (include/bhr_output.h, BHR_PNG_DEVICE).
//
// The host encoder of output.hip (zlib on worker threads) caps the video driver at ~16 frames/s per host core
// (47 ms per fhd frame at level 1); with a frame rendering in 0.7 ms the PNG files are what the driver waits for.
// Here the frame never leaves the device uncompressed: three launches turn the (rows, W, 3) u8 frame into the bytes
// of a complete PNG file in HBM, and only those (a quarter to a third of the raw size) cross PCIe.
//
// Format.  One scanline = one deflate block = one IDAT chunk:
//   * filter: the five PNG filter types are tried on every scanline, the one with the smallest sum of absolute
//     signed residuals wins (PNG specification 12.8; the same rule as the host encoder);
//   * entropy code: Huffman only, no LZ77 (residuals of a rendered frame are noise around zero; zlib level 1 gains
//     ~15 % from matches on these frames, tools/png_menu_study.py).  The code is not built per block: a MENU of
//     kTables static prefix codes is prepared once on the host -- a spike at zero of weight 1 - q plus a two-sided
//     geometric tail of scale b, for a grid of (q, b), plus a flat 8/9-bit code that bounds the worst case -- and
//     every scanline takes the code that makes it shortest (its histogram x the code lengths).  Each block is a
//     dynamic-Huffman block (BTYPE = 2) whose header, the bit string describing the chosen code, is a constant
//     per menu entry;
//   * every scanline's block is followed by an empty stored block (3 header bits, padding to the byte boundary,
//     00 00 FF FF): the next scanline starts byte aligned, so that scanlines are coded independently and in
//     parallel; the last one carries BFINAL;
//   * one IDAT chunk per scanline: its CRC-32 is computed by the block that coded it (256 slices folded with the
//     x^n mod P operator), no pass over the whole stream is needed; the Adler-32 of the zlib stream is assembled
//     from per-scanline sums by the scan kernel.
// A decoder sees an ordinary PNG (RFC 1950/1951, PNG 1.2): tests decode the files with zlib and PIL.
#include "bhr_internal.h"
#include "../../include/bhr_output.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

namespace {

constexpr int kTables = 16;        // menu entries: 5 x 3 (q, b) codes + the flat code (16: one nibble each in len_nibbles)
constexpr int kSyms = 257;         // literals 0..255 + end of block
constexpr int kHdrWords = 64;      // room for a block header (<= 2048 bits)
constexpr int kThreads = 256;
constexpr int kScanThreads = 1024;
constexpr uint32_t kPoly = 0xEDB88320u;
static_assert(kThreads == 256, "one thread per literal in the plan kernel, one per CRC table entry in the encoder");
static_assert(kTables == 16, "len_nibbles packs one 4-bit length per menu entry into 64 bits");

struct PngTables {
    uint32_t code[kTables][260];   // (bit-reversed code << 4) | length, indexed by symbol
    uint32_t hdr[kTables][kHdrWords];
    uint32_t hdr_bits[kTables];
    unsigned long long len_nibbles[260];   // code lengths of symbol s in all 16 menu entries, entry k at bits 4k..4k+3
    uint32_t crc_table[256];
    uint32_t x2n[32];              // x^(2^k) mod P, reflected (zlib's x2n_table)
    uint8_t head[40];              // signature + IHDR chunk (33 bytes)
};

struct RowPlan {
    uint32_t filter, table, chunk_bytes, data_bits;
    unsigned long long a, b;       // Adler partial sums of the filtered scanline: sum d_i, sum (N - i) d_i
};

struct PngDev {
    PngTables *d_tab = nullptr;
    // scanline plans and chunk offsets, one set per frame slot: encodes of successive frames run on different streams
    RowPlan *d_plan[BHR_MAX_FRAME_SLOTS] = {nullptr, nullptr};
    uint32_t *d_offs[BHR_MAX_FRAME_SLOTS] = {nullptr, nullptr};
    uint32_t *d_meta = nullptr;    // [0] file length, [1] error (1: output buffer too small), [2] adler
    int32_t plan_rows = 0;
    int32_t head_w = 0, head_h = 0;
    bool lds_attr = false;         // large dynamic LDS enabled for the kernels on this context's device
    uint8_t *d_out = nullptr;      // scratch for bhr_png_encode_device
    int64_t out_cap = 0;
};

// ---------------------------------------------------------------------------------------------- host: the code menu
// Optimal prefix code lengths for `freq` (all > 0), limited to max_len by halving the frequencies until the tree fits.
std::vector<int> huffman_lengths(std::vector<uint64_t> freq, int max_len) {
    const int n = (int)freq.size();
    for (;;) {
        typedef std::pair<uint64_t, int> Item;    // weight, node (ties: lower node first -> deterministic)
        std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
        std::vector<int> parent(2 * n, -1);
        for (int i = 0; i < n; ++i) heap.push(Item(freq[i], i));
        int next = n;
        while (heap.size() > 1) {
            const Item a = heap.top(); heap.pop();
            const Item b = heap.top(); heap.pop();
            parent[a.second] = parent[b.second] = next;
            heap.push(Item(a.first + b.first, next));
            ++next;
        }
        std::vector<int> len(n);
        int longest = 0;
        for (int i = 0; i < n; ++i) {
            int d = 0;
            for (int k = i; parent[k] >= 0; k = parent[k]) ++d;
            len[i] = d > 0 ? d : 1;
            longest = std::max(longest, len[i]);
        }
        if (longest <= max_len) return len;
        for (auto &f : freq) f = std::max<uint64_t>((f + 1) >> 1, 1);
    }
}

// Canonical code of RFC 1951 3.2.2 for the given lengths (0 = unused), bit-reversed for LSB-first emission.
std::vector<uint32_t> canonical_reversed(const std::vector<int> &len) {
    int count[16] = {0}, next[16] = {0};
    for (int l : len) count[l] += 1;
    count[0] = 0;
    int code = 0;
    for (int bits = 1; bits < 16; ++bits) {
        code = (code + count[bits - 1]) << 1;
        next[bits] = code;
    }
    std::vector<uint32_t> out(len.size(), 0);
    for (size_t s = 0; s < len.size(); ++s) {
        if (!len[s]) continue;
        uint32_t c = (uint32_t)next[len[s]]++, r = 0;
        for (int k = 0; k < len[s]; ++k) r |= ((c >> k) & 1u) << (len[s] - 1 - k);
        out[s] = r;
    }
    return out;
}

struct BitString {
    std::vector<uint32_t> words;
    uint32_t bits = 0;
    void put(uint32_t v, int n) {
        for (int k = 0; k < n; ++k, ++bits) {
            if ((bits >> 5) >= words.size()) words.push_back(0);
            words[bits >> 5] |= ((v >> k) & 1u) << (bits & 31);
        }
    }
};

// Header of a dynamic-Huffman block (RFC 1951 3.2.7) announcing `lit_len` for symbols 0..256 and ONE distance code
// of length one (never used: there are no matches).  Runs of equal lengths use symbol 16 (repeat 3..6 times).
BitString block_header(const std::vector<int> &lit_len) {
    std::vector<int> seq(lit_len.begin(), lit_len.end());
    seq.push_back(1);                                     // the distance code
    struct Tok { int sym, extra; };
    std::vector<Tok> toks;
    for (size_t i = 0; i < seq.size();) {
        size_t j = i;
        while (j < seq.size() && seq[j] == seq[i]) ++j;
        size_t run = j - i;
        toks.push_back(Tok{seq[i], 0});
        run -= 1;
        while (run >= 3) {
            const size_t r = std::min<size_t>(run, 6);
            toks.push_back(Tok{16, (int)r - 3});
            run -= r;
        }
        for (; run > 0; --run) toks.push_back(Tok{seq[i], 0});
        i = j;
    }
    std::vector<uint64_t> cl_freq(19, 0);
    for (const Tok &t : toks) cl_freq[t.sym] += 1;
    // the code-length code must be complete: build it over the symbols that occur (at least two)
    std::vector<int> used;
    for (int s = 0; s < 19; ++s)
        if (cl_freq[s]) used.push_back(s);
    if (used.size() < 2) used.push_back(used[0] == 0 ? 1 : 0);
    std::vector<uint64_t> f;
    for (int s : used) f.push_back(std::max<uint64_t>(cl_freq[s], 1));
    const std::vector<int> l = huffman_lengths(f, 7);
    std::vector<int> cl_len(19, 0);
    for (size_t k = 0; k < used.size(); ++k) cl_len[used[k]] = l[k];
    const std::vector<uint32_t> cl_code = canonical_reversed(cl_len);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl_len[order[hclen - 1]] == 0) --hclen;
    BitString b;
    b.put(0, 1);                  // BFINAL = 0: the stream ends with the last scanline's stored block
    b.put(2, 2);                  // BTYPE = 10
    b.put(kSyms - 257, 5);        // HLIT
    b.put(0, 5);                  // HDIST: one distance code
    b.put((uint32_t)hclen - 4, 4);
    for (int k = 0; k < hclen; ++k) b.put((uint32_t)cl_len[order[k]], 3);
    for (const Tok &t : toks) {
        b.put(cl_code[t.sym], cl_len[t.sym]);
        if (t.sym == 16) b.put((uint32_t)t.extra, 2);
    }
    return b;
}

uint32_t multmodp_host(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int k = 0; k < 32; ++k) {
        if (a & (0x80000000u >> k)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}

bool fill_tables(PngTables *t) {
    memset(t, 0, sizeof(*t));
    static const double qs[5] = {0.15, 0.3, 0.45, 0.6, 0.8}, bs[3] = {0.7, 1.5, 4.0};
    for (int k = 0; k < kTables; ++k) {
        std::vector<uint64_t> freq(kSyms, 1);
        if (k < 15) {
            const double q = qs[k / 3], b = bs[k % 3];
            double norm = 0.0;
            for (int s = 1; s < 256; ++s) norm += std::exp(-(double)(s < 128 ? s : 256 - s) / b);
            for (int s = 1; s < 256; ++s)
                freq[s] = std::max<uint64_t>((uint64_t)(std::exp(-(double)(s < 128 ? s : 256 - s) / b) / norm * q * 16777216.0), 1);
            freq[0] = (uint64_t)((1.0 - q) * 16777216.0);
            freq[256] = 16777216 / 5761;          // one end-of-block per scanline
        }
        const std::vector<int> len = huffman_lengths(freq, 15);
        const std::vector<uint32_t> code = canonical_reversed(len);
        for (int s = 0; s < kSyms; ++s) {
            t->code[k][s] = (code[s] << 4) | (uint32_t)len[s];
            t->len_nibbles[s] |= (unsigned long long)len[s] << (4 * k);
        }
        const BitString h = block_header(len);
        if (h.words.size() > (size_t)kHdrWords) return false;
        t->hdr_bits[k] = h.bits;
        for (size_t w = 0; w < h.words.size() && w < (size_t)kHdrWords; ++w) t->hdr[k][w] = h.words[w];
    }
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kPoly : c >> 1;
        t->crc_table[n] = c;
    }
    uint32_t p = 1u << 30;                         // x^1
    t->x2n[0] = p;
    for (int n = 1; n < 32; ++n) t->x2n[n] = p = multmodp_host(p, p);
    return true;
}

void fill_head(uint8_t *head, int w, int h) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    memcpy(head, sig, 8);
    uint8_t *c = head + 8;
    const uint32_t fields[2] = {(uint32_t)w, (uint32_t)h};
    c[0] = 0; c[1] = 0; c[2] = 0; c[3] = 13;
    memcpy(c + 4, "IHDR", 4);
    for (int k = 0; k < 2; ++k)
        for (int b = 0; b < 4; ++b) c[8 + 4 * k + b] = (uint8_t)(fields[k] >> (24 - 8 * b));
    c[16] = 8; c[17] = 2; c[18] = 0; c[19] = 0; c[20] = 0;     // 8-bit, colour type 2 (RGB), deflate, adaptive filters, no interlace
    const uint32_t crc = (uint32_t)crc32(0L, c + 4, 17);
    for (int b = 0; b < 4; ++b) c[21 + b] = (uint8_t)(crc >> (24 - 8 * b));
}

// ---------------------------------------------------------------------------------------------- device
__device__ __forceinline__ int paeth_pred(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Scanline buffers in LDS: 16 zero bytes, the n bytes of the row, zeros up to a multiple of 16.  The zero prefix
// stands for the pixel left of the first one (PNG: "bytes to the left of the first pixel are zero"), so no position needs
// a branch; rows are processed four bytes (one aligned LDS word per operand) at a time.
constexpr int kRowPrefix = 16;
__device__ __forceinline__ int row_stride(int n) { return kRowPrefix + ((n + 15) & ~15); }

// The scanline and the one above it -> LDS (cur, up point behind the prefix).  16-byte loads when the rows are 16-byte
// aligned (every width that is a multiple of 16: all BASELINE sizes); otherwise bytes, eight loads in flight per thread.
__device__ __forceinline__ void load_rows(const uint8_t *__restrict__ rgb, int row, int n, uint8_t *cur, uint8_t *up) {
    const uint8_t *g = rgb + (size_t)row * n;
    const int npad = (n + 15) & ~15;
    if (threadIdx.x < 4) {
        ((uint32_t *)(cur - kRowPrefix))[threadIdx.x] = 0u;
        ((uint32_t *)(up - kRowPrefix))[threadIdx.x] = 0u;
    }
    if ((n & 15) == 0 && (((uintptr_t)rgb) & 15) == 0) {
        const uint4 *g4 = (const uint4 *)g, *u4 = (const uint4 *)(g - n);
        uint4 *c4 = (uint4 *)cur, *p4 = (uint4 *)up;
        for (int v = threadIdx.x; v < (n >> 4); v += kThreads) {
            c4[v] = g4[v];
            p4[v] = row > 0 ? u4[v] : make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    for (int i0 = 0; i0 < npad; i0 += 8 * kThreads) {
        uint8_t a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kThreads + (int)threadIdx.x;
            a[k] = i < n ? g[i] : (uint8_t)0;
            b[k] = (i < n && row > 0) ? g[i - n] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kThreads + (int)threadIdx.x;
            if (i < npad) { cur[i] = a[k]; up[i] = b[k]; }          // the tail up to npad is zero
        }
    }
}

// Operands of bytes i .. i + 3 (i a multiple of 4): x = the bytes, a = three to the left, b = above, c = above-left.
struct Quad { uint32_t x, a, b, c; };
__device__ __forceinline__ Quad load_quad(const uint8_t *cur, const uint8_t *up, int i) {
    const uint32_t x = *(const uint32_t *)(cur + i), xp = *(const uint32_t *)(cur + i - 4);
    const uint32_t b = *(const uint32_t *)(up + i), bp = *(const uint32_t *)(up + i - 4);
    Quad q;
    q.x = x;
    q.b = b;
    q.a = __byte_perm(xp, x, 0x4321);      // bytes i-3, i-2, i-1, i
    q.c = __byte_perm(bp, b, 0x4321);
    return q;
}

// residual of byte k of the quad under filter f (0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth), 0..255
__device__ __forceinline__ uint32_t residual(int f, const Quad &q, int k) {
    const int x = (int)((q.x >> (8 * k)) & 255u), a = (int)((q.a >> (8 * k)) & 255u);
    const int b = (int)((q.b >> (8 * k)) & 255u), c = (int)((q.c >> (8 * k)) & 255u);
    int pred;
    switch (f) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        default: pred = paeth_pred(a, b, c); break;
    }
    return (uint32_t)(x - pred) & 255u;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// K1: per scanline -- filter choice, histogram, code choice, size, Adler partial sums.
__global__ __launch_bounds__(kThreads) void png_plan_kernel(const uint8_t *__restrict__ rgb, int n, int h,
                                                            const PngTables *__restrict__ tab, RowPlan *__restrict__ plan) {
    extern __shared__ uint8_t smem[];
    __shared__ uint32_t hist[kSyms + 3];
    __shared__ uint32_t cost[5];
    __shared__ uint32_t bits[kTables];
    __shared__ unsigned long long ab[2];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint8_t *cur = smem + kRowPrefix, *up = cur + row_stride(n);
    for (int i = tid; i < kSyms + 3; i += kThreads) hist[i] = 0;
    if (tid < 5) cost[tid] = 0;
    if (tid < kTables) bits[tid] = 0;
    if (tid < 2) ab[tid] = 0;
    load_rows(rgb, row, n, cur, up);
    __syncthreads();
    {   // sum of |signed residual| for the five filters; neighbouring lanes read neighbouring words
        uint32_t c[5] = {0, 0, 0, 0, 0};
#pragma unroll 2
        for (int i = 4 * tid; i < n; i += 4 * kThreads) {
            const Quad q = load_quad(cur, up, i);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n)
#pragma unroll
                    for (int f = 0; f < 5; ++f) {
                        const uint32_t d = residual(f, q, k);
                        c[f] += min(d, 256u - d);
                    }
        }
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const uint32_t t = wave_sum(c[f]);
            if (lane == 0) atomicAdd(&cost[f], t);
        }
    }
    __syncthreads();
    int best = 0;
#pragma unroll
    for (int f = 1; f < 5; ++f)
        if (cost[f] < cost[best]) best = f;
    {
        const unsigned long long N = (unsigned long long)n + 1;        // filter byte + n residuals
        uint32_t a = 0, zeros = 0;                                     // most residuals are zero: counted in a register
        unsigned long long b = 0;
#pragma unroll 2
        for (int i = 4 * tid; i < n; i += 4 * kThreads) {
            const Quad q = load_quad(cur, up, i);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) {
                    const uint32_t d = residual(best, q, k);
                    if (d) atomicAdd(&hist[d], 1u); else ++zeros;
                    a += d;
                    b += (unsigned long long)((uint32_t)n - (uint32_t)(i + k)) * d;   // stream position of residual i is i + 1
                }
        }
        const uint32_t z = wave_sum(zeros), ta = wave_sum(a);
        const unsigned long long tb = wave_sum64(b);
        if (lane == 0) {
            if (z) atomicAdd(&hist[0], z);
            atomicAdd(&ab[0], (unsigned long long)ta);
            atomicAdd(&ab[1], tb);
        }
        if (tid == 0) {
            atomicAdd(&hist[best], 1u);
            atomicAdd(&ab[0], (unsigned long long)best);
            atomicAdd(&ab[1], N * (unsigned long long)best);
        }
    }
    __syncthreads();
    {   // bits of the scanline under each menu entry: histogram x code lengths (16 nibbles per symbol)
        const uint32_t cnt = hist[tid];                                // kThreads == 256 literals
        const unsigned long long nib = tab->len_nibbles[tid];
#pragma unroll
        for (int k = 0; k < kTables; ++k) {
            const uint32_t t = wave_sum(cnt * (uint32_t)((nib >> (4 * k)) & 15ull));
            if (lane == 0) atomicAdd(&bits[k], t);
        }
    }
    __syncthreads();
    if (tid < kTables) {
        // + end of block + block header; then the smallest total, the lowest entry on ties
        const uint32_t total = bits[tid] + (uint32_t)((tab->len_nibbles[256] >> (4 * tid)) & 15ull) + tab->hdr_bits[tid];
        uint32_t best_bits = total;
        int kb = tid;
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) {
            const uint32_t ob = __shfl_xor(best_bits, d, 64);
            const int ok = __shfl_xor(kb, d, 64);
            if (ob < best_bits || (ob == best_bits && ok < kb)) { best_bits = ob; kb = ok; }
        }
        if (tid == 0) {
            // header + symbols + end of block, then the stored block: 3 bits, padding, LEN, NLEN
            uint32_t payload = (best_bits + 3 + 7) / 8 + 4;
            if (row == 0) payload += 2;                                // zlib header
            if (row == h - 1) payload += 4;                            // Adler-32
            RowPlan p;
            p.filter = (uint32_t)best;
            p.table = (uint32_t)kb;
            p.chunk_bytes = payload + 12;
            p.data_bits = best_bits - tab->hdr_bits[kb];
            p.a = ab[0];
            p.b = ab[1];
            plan[row] = p;
        }
    }
}

// K2: chunk offsets (exclusive scan), Adler-32 of the whole filtered stream, file head and tail.
__global__ __launch_bounds__(kScanThreads) void png_scan_kernel(const RowPlan *__restrict__ plan, int n, int h,
                                                            const PngTables *__restrict__ tab, uint32_t *__restrict__ offs,
                                                            uint32_t *__restrict__ meta, uint8_t *__restrict__ out, long long cap) {
    __shared__ unsigned long long sc[kScanThreads], sa[kScanThreads];
    __shared__ unsigned long long carry_off, carry_a, s2_acc;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_off = 33; carry_a = 1; s2_acc = 0; }       // after signature + IHDR; Adler s1 starts at 1
    __syncthreads();
    const unsigned long long N = (unsigned long long)n + 1, M = 65521ull;
    unsigned long long s2_mine = 0;
    for (int base = 0; base < h; base += kScanThreads) {
        const int r = base + tid;
        const unsigned long long len = r < h ? plan[r].chunk_bytes : 0ull, a = r < h ? plan[r].a : 0ull;
        sc[tid] = len;
        sa[tid] = a;
        __syncthreads();
        for (int d = 1; d < kScanThreads; d <<= 1) {                      // inclusive scans
            const unsigned long long vc = tid >= d ? sc[tid - d] : 0ull, va = tid >= d ? sa[tid - d] : 0ull;
            __syncthreads();
            sc[tid] += vc;
            sa[tid] += va;
            __syncthreads();
        }
        if (r < h) {
            offs[r] = (uint32_t)(carry_off + sc[tid] - len);
            const unsigned long long s1_before = (carry_a + sa[tid] - a) % M;
            s2_mine += (N % M * s1_before + plan[r].b % M) % M;
        }
        __syncthreads();
        if (tid == 0) { carry_off += sc[kScanThreads - 1]; carry_a += sa[kScanThreads - 1]; }
        __syncthreads();
    }
    {
        const unsigned long long t = wave_sum64(s2_mine);
        if ((tid & 63) == 0) atomicAdd(&s2_acc, t);
    }
    __syncthreads();
    const unsigned long long total = carry_off + 12;                  // + IEND
    const bool too_big = total > (unsigned long long)cap || total > 0xFFFFFFF0ull;
    if (tid == 0) {
        meta[0] = (uint32_t)total;
        meta[1] = too_big ? 1u : 0u;
        meta[2] = (uint32_t)(((s2_acc % M) << 16) | (carry_a % M));
    }
    if (!too_big) {
        const uint32_t iend_lo = 0x00000000u, iend_mid = 0x444E4549u, iend_hi = 0x826042AEu;   // 0 | "IEND" | CRC, as stored
        if (tid < 33) out[tid] = tab->head[tid];
        else if (tid < 45) {
            const int k = tid - 33;
            const uint32_t w = k < 4 ? iend_lo : (k < 8 ? iend_mid : iend_hi);
            out[carry_off + k] = (uint8_t)(w >> (8 * (k & 3)));
        }
    }
}

__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (int k = 0; k < 32; ++k) {
        if (a & (0x80000000u >> k)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}

// K3: code one scanline into its IDAT chunk.
__global__ __launch_bounds__(kThreads) void png_encode_kernel(const uint8_t *__restrict__ rgb, int n, int h,
                                                              const PngTables *__restrict__ tab, const RowPlan *__restrict__ plan,
                                                              const uint32_t *__restrict__ offs, const uint32_t *__restrict__ meta,
                                                              uint8_t *__restrict__ out, int chunk_words) {
    extern __shared__ uint8_t smem[];
    __shared__ uint32_t codes[kSyms + 3];
    __shared__ uint32_t crc_tab[256];
    __shared__ uint32_t scan[kThreads];
    __shared__ uint32_t crc_acc;
    if (meta[1]) return;                                               // the plan does not fit the output buffer
    const int row = blockIdx.x, tid = threadIdx.x;
    // LDS: [the two rows | later: the chunk being assembled] [fb: filter byte + residuals (n + 1)].  The chunk reuses the
    // rows' space once they have been filtered (8k: 69 KB per block instead of 95, two blocks per CU).
    const int rows_bytes = max(2 * row_stride(n), 4 * chunk_words);
    uint8_t *cur = smem + kRowPrefix, *up = cur + row_stride(n), *fb = smem + ((rows_bytes + 15) & ~15);
    uint32_t *cw = (uint32_t *)smem;                                   // the chunk, word addressed
    uint8_t *cb = (uint8_t *)cw;
    const RowPlan p = plan[row];
    for (int i = tid; i < kSyms; i += kThreads) codes[i] = tab->code[p.table][i];
    crc_tab[tid] = tab->crc_table[tid];
    if (tid == 0) crc_acc = 0;
    load_rows(rgb, row, n, cur, up);
    __syncthreads();
    for (int i = 4 * tid; i < n; i += 4 * kThreads) {
        const Quad q = load_quad(cur, up, i);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n) fb[i + 1 + k] = (uint8_t)residual((int)p.filter, q, k);
    }
    if (tid == 0) fb[0] = (uint8_t)p.filter;
    __syncthreads();

    for (int i = tid; i < chunk_words; i += kThreads) cw[i] = 0;       // the rows are dead: their space becomes the chunk
    const int N = n + 1;
    // contiguous slices of whole words of fb: four symbols per LDS read, four independent code lookups
    const int per = (((N + kThreads - 1) / kThreads) + 3) & ~3, i0 = min(tid * per, N), i1 = min(i0 + per, N);
    uint32_t my_bits = 0;
    for (int i = i0; i < i1; i += 4) {
        const uint32_t w = *(const uint32_t *)(fb + i);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < i1) my_bits += codes[(w >> (8 * k)) & 255u] & 15u;
    }
    scan[tid] = my_bits;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {
        const uint32_t v = tid >= d ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const uint32_t payload = p.chunk_bytes - 12;
    const uint32_t p0 = 8u + (row == 0 ? 2u : 0u);                    // byte where the deflate bits begin
    const uint32_t hdr_bits = tab->hdr_bits[p.table];
    const uint32_t data0 = 8u * p0 + hdr_bits;
    // block header: a constant bit string of the menu entry, shifted to its byte position
    for (uint32_t w = tid; w * 32u < hdr_bits; w += kThreads) {
        const uint32_t v = tab->hdr[p.table][w], at = 8u * p0 + 32u * w, sh = at & 31u;
        atomicOr(&cw[at >> 5], v << sh);
        if (sh) atomicOr(&cw[(at >> 5) + 1], v >> (32u - sh));
    }
    {   // the symbols of this thread's slice
        uint32_t pos = data0 + scan[tid] - my_bits, wi = pos >> 5, nb = pos & 31u;
        unsigned long long acc = 0;
        for (int i = i0; i < i1; i += 4) {
            const uint32_t w = *(const uint32_t *)(fb + i);
            uint32_t c4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c4[k] = codes[(w >> (8 * k)) & 255u];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < i1) {
                    acc |= (unsigned long long)(c4[k] >> 4) << nb;
                    nb += c4[k] & 15u;
                    if (nb >= 32u) {
                        atomicOr(&cw[wi], (uint32_t)acc);
                        acc >>= 32;
                        nb -= 32u;
                        ++wi;
                    }
                }
        }
        if (nb && acc) atomicOr(&cw[wi], (uint32_t)acc);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t pos = data0 + scan[kThreads - 1];
        const uint32_t eob = codes[256];
        {   // end of block + stored-block header (BFINAL on the last scanline, BTYPE 00)
            unsigned long long v = (unsigned long long)(eob >> 4) | ((unsigned long long)(row == h - 1 ? 1u : 0u) << (eob & 15u));
            const uint32_t nbits = (eob & 15u) + 3u, sh = pos & 31u;
            v <<= sh;
            cw[pos >> 5] |= (uint32_t)v;
            if (sh + nbits > 32u) cw[(pos >> 5) + 1] |= (uint32_t)(v >> 32);
            pos += nbits;
        }
        uint32_t at = (pos + 7u) >> 3;                                 // byte aligned
        cb[at] = 0; cb[at + 1] = 0; cb[at + 2] = 0xFF; cb[at + 3] = 0xFF;
        at += 4;
        if (row == h - 1) {
            const uint32_t ad = meta[2];
            cb[at] = (uint8_t)(ad >> 24); cb[at + 1] = (uint8_t)(ad >> 16); cb[at + 2] = (uint8_t)(ad >> 8); cb[at + 3] = (uint8_t)ad;
            at += 4;
        }
        // at == 8 + payload by construction of the plan
        cb[0] = (uint8_t)(payload >> 24); cb[1] = (uint8_t)(payload >> 16); cb[2] = (uint8_t)(payload >> 8); cb[3] = (uint8_t)payload;
        cb[4] = 'I'; cb[5] = 'D'; cb[6] = 'A'; cb[7] = 'T';
        if (row == 0) { cb[8] = 0x78; cb[9] = 0x01; }
    }
    __syncthreads();
    {   // CRC-32 of type + data: every thread takes a slice, slices fold with x^(8 * bytes after the slice) mod P
        const uint32_t L = payload + 4u, per_c = (L + kThreads - 1) / kThreads;
        const uint32_t s0 = min((uint32_t)tid * per_c, L), s1 = min(s0 + per_c, L);
        if (s1 > s0) {
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t i = s0; i < s1; ++i) c = crc_tab[(c ^ cb[4 + i]) & 255u] ^ (c >> 8);
            c = ~c;
            uint32_t rem = L - s1, xp = 0x80000000u, k = 3;           // x2nmodp(rem, 3)
            while (rem) {
                if (rem & 1u) xp = multmodp(tab->x2n[k & 31u], xp);
                rem >>= 1;
                ++k;
            }
            atomicXor(&crc_acc, multmodp(xp, c));
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t c = crc_acc, at = 8u + payload;
        cb[at] = (uint8_t)(c >> 24); cb[at + 1] = (uint8_t)(c >> 16); cb[at + 2] = (uint8_t)(c >> 8); cb[at + 3] = (uint8_t)c;
    }
    __syncthreads();
    uint8_t *g = out + offs[row];
    for (uint32_t i = tid; i < p.chunk_bytes; i += kThreads) g[i] = cb[i];
}

PngDev *dev_of(bhr_ctx *ctx) { return (PngDev *)ctx->png_dev; }

int chunk_words_for(int n) { return (int)(((size_t)n + 1) * 9 / 8 / 4 + 96); }   // flat code: <= 9 bits/byte, + header, framing

}  // namespace

// Widest frame the encode kernel can hold: two padded rows (or the chunk) + the filtered row in 150 KB of LDS.
extern "C" int32_t bhr_png_device_max_width(void) {
    int lo = 1, hi = 1 << 16;
    auto fits = [](int w) {
        const size_t n = 3 * (size_t)w, npad = (n + 15) & ~(size_t)15;
        const size_t rows = 2 * (16 + npad), chunk = 4 * (size_t)chunk_words_for((int)n);
        return ((std::max(rows, chunk) + 15) & ~(size_t)15) + ((n + 1 + 15) & ~(size_t)15) <= 150 * 1024;
    };
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (fits(mid)) lo = mid; else hi = mid - 1;
    }
    return lo;
}

extern "C" int64_t bhr_png_device_bound(int32_t w, int32_t h) {
    if (w <= 0 || h <= 0) return 0;
    return (int64_t)h * (4 * (int64_t)chunk_words_for(3 * w)) + 64;
}

void bhr_png_dev_free(bhr_ctx *ctx) {
    PngDev *d = dev_of(ctx);
    if (!d) return;
    void *bufs[] = {d->d_tab, d->d_meta, d->d_out};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (int k = 0; k < BHR_MAX_FRAME_SLOTS; ++k) {
        if (d->d_plan[k]) (void)hipFree(d->d_plan[k]);
        if (d->d_offs[k]) (void)hipFree(d->d_offs[k]);
    }
    delete d;
    ctx->png_dev = nullptr;
}

// Tables and per-scanline scratch for this context's frame size (idempotent).
int32_t bhr_png_dev_prepare(bhr_ctx *ctx) {
    const int w = ctx->cfg.width, h = ctx->rows;
    PngDev *d = dev_of(ctx);
    if (!d) {
        d = new PngDev();
        ctx->png_dev = d;
    }
    if (!d->d_tab || d->head_w != w || d->head_h != h) {
        std::vector<PngTables> t(1);
        if (!fill_tables(&t[0])) return bhr_fail(BHR_ERR_STATE, "device PNG encoder: a block header exceeds %d words", kHdrWords);
        fill_head(t[0].head, w, h);
        if (!d->d_tab) BHR_HIP(hipMalloc((void **)&d->d_tab, sizeof(PngTables)));
        BHR_HIP(hipMemcpyAsync(d->d_tab, &t[0], sizeof(PngTables), hipMemcpyHostToDevice, ctx->stream));
        BHR_HIP(hipStreamSynchronize(ctx->stream));        // the host copy goes out of scope
        d->head_w = w;
        d->head_h = h;
    }
    if (d->plan_rows < h) {
        for (int k = 0; k < BHR_MAX_FRAME_SLOTS; ++k) {
            if (d->d_plan[k]) (void)hipFree(d->d_plan[k]);
            if (d->d_offs[k]) (void)hipFree(d->d_offs[k]);
            d->d_plan[k] = nullptr;
            d->d_offs[k] = nullptr;
        }
        d->plan_rows = 0;
        for (int k = 0; k < BHR_MAX_FRAME_SLOTS; ++k) {
            BHR_HIP(hipMalloc((void **)&d->d_plan[k], sizeof(RowPlan) * (size_t)h));
            BHR_HIP(hipMalloc((void **)&d->d_offs[k], sizeof(uint32_t) * (size_t)h));
        }
        d->plan_rows = h;
    }
    if (!d->d_meta) BHR_HIP(hipMalloc((void **)&d->d_meta, 4 * sizeof(uint32_t)));
    return BHR_OK;
}

// Encodes the (rows, W, 3) u8 image at d_rgb into d_out (cap bytes) on ctx->stream; d_meta_out (4 words, device)
// receives {file length, error, adler, 0}.  The plan / offsets scratch is the active frame slot's.
int32_t bhr_launch_png_encode(bhr_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_out, int64_t cap, uint32_t *d_meta_out) {
    BHR_TRY(bhr_png_dev_prepare(ctx));
    PngDev *d = dev_of(ctx);
    const int n = 3 * ctx->cfg.width, h = ctx->rows;
    const int npad = (n + 15) & ~15;
    const int cwords = chunk_words_for(n);
    const size_t row_buf = 16 + (size_t)npad;                       // kRowPrefix + padded row (row_stride)
    const size_t lds_plan = 2 * row_buf;
    const size_t lds_enc = ((std::max(2 * row_buf, (size_t)4 * cwords) + 15) & ~(size_t)15) + ((n + 1 + 15) & ~15);
    if (bhr_png_device_max_width() < ctx->cfg.width)
        return bhr_fail(BHR_ERR_INVALID, "device PNG encoder: a scanline of %d pixels does not fit LDS (at most %d); use the host encoder",
                        ctx->cfg.width, bhr_png_device_max_width());
    if (!d->lds_attr) {
        BHR_HIP(hipFuncSetAttribute((const void *)png_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        BHR_HIP(hipFuncSetAttribute((const void *)png_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        d->lds_attr = true;
    }
    RowPlan *plan = d->d_plan[ctx->active_slot];
    uint32_t *offs = d->d_offs[ctx->active_slot];
    hipLaunchKernelGGL(png_plan_kernel, dim3(h), dim3(kThreads), lds_plan, ctx->stream, d_rgb, n, h, d->d_tab, plan);
    hipLaunchKernelGGL(png_scan_kernel, dim3(1), dim3(kScanThreads), 0, ctx->stream, plan, n, h, d->d_tab, offs, d_meta_out, d_out,
                       (long long)cap);
    hipLaunchKernelGGL(png_encode_kernel, dim3(h), dim3(kThreads), lds_enc, ctx->stream, d_rgb, n, h, d->d_tab, plan, offs,
                       d_meta_out, d_out, cwords);
    BHR_HIP(hipGetLastError());
    return BHR_OK;
}

extern "C" {

int32_t bhr_png_device_menu(int32_t k, uint32_t *codes, uint32_t *hdr_words, uint32_t *hdr_bits, int32_t *n_tables) {
    if (n_tables) *n_tables = kTables;
    if (k < 0 || k >= kTables || !codes || !hdr_words || !hdr_bits)
        return bhr_fail(BHR_ERR_INVALID, "bhr_png_device_menu: entry %d of %d", k, kTables);
    std::vector<PngTables> t(1);
    if (!fill_tables(&t[0])) return bhr_fail(BHR_ERR_STATE, "device PNG encoder: a block header exceeds %d words", kHdrWords);
    memcpy(codes, t[0].code[k], sizeof(uint32_t) * kSyms);
    memcpy(hdr_words, t[0].hdr[k], sizeof(uint32_t) * kHdrWords);
    *hdr_bits = t[0].hdr_bits[k];
    return BHR_OK;
}

int32_t bhr_png_encode_device(bhr_ctx *ctx, uint8_t *out, int64_t cap, int64_t *out_len) {
    if (!ctx || !out || !out_len) return bhr_fail(BHR_ERR_INVALID, "bhr_png_encode_device: null argument");
    BHR_TRY(bhr_enter(ctx));
    BHR_TRY(bhr_png_dev_prepare(ctx));
    PngDev *d = dev_of(ctx);
    const int64_t bound = bhr_png_device_bound(ctx->cfg.width, ctx->rows);
    if (d->out_cap < bound) {
        if (d->d_out) (void)hipFree(d->d_out);
        d->d_out = nullptr;
        d->out_cap = 0;
        BHR_HIP(hipMalloc((void **)&d->d_out, (size_t)bound));
        d->out_cap = bound;
    }
    BHR_TRY(bhr_launch_quantize(ctx));
    BHR_TRY(bhr_launch_png_encode(ctx, ctx->d_final_u8, d->d_out, bound, d->d_meta));
    uint32_t meta[4] = {0, 0, 0, 0};
    BHR_HIP(hipMemcpyAsync(meta, d->d_meta, sizeof(meta), hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    if (meta[1]) return bhr_fail(BHR_ERR_STATE, "bhr_png_encode_device: the encoded frame exceeds bhr_png_device_bound");
    if ((int64_t)meta[0] > cap)
        return bhr_fail(BHR_ERR_INVALID, "bhr_png_encode_device: %u bytes do not fit the caller's %lld", meta[0], (long long)cap);
    BHR_HIP(hipMemcpyAsync(out, d->d_out, meta[0], hipMemcpyDeviceToHost, ctx->stream));
    BHR_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = (int64_t)meta[0];
    return BHR_OK;
}

}  // extern "C"
