"""Study for the device PNG encoder (csrc/png_device.hip): size of a frame under per-row PNG filters + a menu of
static Huffman codes (no LZ77), against zlib on the same filtered stream.  Usage: png_menu_study.py frame.npy"""
import sys, zlib, heapq
import numpy as np


def filter_rows(img):
    h, w, _ = img.shape
    cur = img.reshape(h, w * 3).astype(np.int16)
    up = np.vstack([np.zeros((1, w * 3), np.int16), cur[:-1]])
    a = np.hstack([np.zeros((h, 3), np.int16), cur[:, :-3]])
    c = np.hstack([np.zeros((h, 3), np.int16), up[:, :-3]])
    p = a + up - c
    pa, pb, pc = np.abs(p - a), np.abs(p - up), np.abs(p - c)
    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
    cand = np.stack([cur, cur - a, cur - up, cur - ((a + up) >> 1), cur - pred]).astype(np.uint8)
    cost = np.abs(cand.view(np.int8).astype(np.int32)).sum(axis=2)
    best = cost.argmin(axis=0)
    out = cand[best, np.arange(h)]
    return best.astype(np.uint8), out


def huff_lengths(freq, max_len):
    freq = [int(f) for f in freq]
    while True:
        heap = [(f, i, None, None) for i, f in enumerate(freq) if f > 0]
        heapq.heapify(heap)
        n = len(freq)
        nodes = {}
        k = n
        while len(heap) > 1:
            a = heapq.heappop(heap); b = heapq.heappop(heap)
            nodes[k] = (a[1], b[1])
            heapq.heappush(heap, (a[0] + b[0], k, None, None))
            k += 1
        lens = [0] * n
        def walk(i, d):
            if i < n: lens[i] = max(d, 1)
            else:
                walk(nodes[i][0], d + 1); walk(nodes[i][1], d + 1)
        walk(heap[0][1], 0)
        if max(lens) <= max_len:
            return lens
        freq = [max((f + 1) >> 1, 1) if f > 0 else 0 for f in freq]


def laplace_freq(b, n_row=5761):
    v = np.arange(256); v = np.where(v < 128, v, 256 - v).astype(np.float64)
    f = np.exp(-v / b)
    f = f / f.sum()
    fr = np.maximum((f * (1 << 24)).astype(np.int64), 1)
    return list(fr) + [max((1 << 24) // n_row, 1)]


def main():
    img = np.load(sys.argv[1])
    h, w, _ = img.shape
    ftype, rows = filter_rows(img)
    stream = np.hstack([ftype[:, None], rows])
    raw = stream.tobytes()
    for lv in (1, 6):
        print(f"zlib level {lv}: {len(zlib.compress(raw, lv)) / 1e6:.3f} MB")
    print(f"filters used: {np.bincount(ftype, minlength=5)}")
    hist = np.zeros((h, 257), np.int64)
    for r in range(h):
        hist[r, :256] = np.bincount(stream[r], minlength=256)
    hist[:, 256] = 1
    # entropy bound with a per-row ideal code, and with a per-frame ideal code
    def ent(hh):
        p = hh[hh > 0] / hh.sum(); return -(p * np.log2(p)).sum() * hh.sum()
    print(f"per-row entropy bound {sum(ent(hist[r]) for r in range(h)) / 8e6:.3f} MB, per-frame {ent(hist.sum(0)) / 8e6:.3f} MB")
    menus = {"7": [0.25, 0.5, 1.0, 2.0, 4.0, 10.0], "5": [0.3, 0.8, 2.0, 6.0], "9": [0.2, 0.35, 0.6, 1.0, 1.7, 3.0, 6.0, 14.0]}
    for name, bs in menus.items():
        tabs = [np.array(huff_lengths(laplace_freq(b), 15)) for b in bs] + [np.array(huff_lengths([1] * 257, 15))]
        bits = np.stack([hist @ t for t in tabs])         # (tables, rows)
        best = bits.argmin(axis=0)
        tot = bits.min(axis=0).sum() / 8 + h * (40 + 17)
        print(f"menu {name} {bs}: {tot / 1e6:.3f} MB, tables used {np.bincount(best, minlength=len(tabs))}, lens of 0: {[int(t[0]) for t in tabs]}")
    # optimal per-row dynamic code (what zlib's Z_HUFFMAN_ONLY would do), header ~ 80 B
    tot = 0
    for r in range(0, h, max(h // 200, 1)):
        t = np.array(huff_lengths(hist[r] + 0, 15)); tot += (hist[r] @ t) / 8 + 80
    print(f"per-row optimal Huffman (sampled rows, extrapolated): {tot * max(h // 200, 1) / 1e6:.3f} MB")
    print(f"zlib Z_HUFFMAN_ONLY: {len(zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_HUFFMAN_ONLY).compress(raw)) / 1e6:.3f}+ MB (unflushed)")


main()
