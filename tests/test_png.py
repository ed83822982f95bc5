"""CPU: the PNG decoder of the host input pipeline (csrc/omni_png.hip, omnifusion_amd/png.py) — SURVEY 8f rank 2, the `cv2.imread` calls of
dataset_loader_stanford.py:85,96.  PNG is lossless, so the check is conformance: files written here by a plain-Python encoder (zlib from
the standard library, every scan-line filter type, several IDAT chunks) must come back sample for sample, in cv2.imread's layouts (B G R
order; alpha dropped; 16-bit -> 8-bit by the high byte for the colour read; the unchanged read keeps uint16).  cv2 itself is absent from
this image: its CONVERSIONS for the colour types the dataset does not contain (palette, gray + alpha) follow OpenCV's documentation, unpinned."""
import struct
import zlib

import numpy as np
import pytest

from omnifusion_amd import png


def _chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def encode_png(arr, ctype, depth, filters=(0, 1, 2, 3, 4), idat=3, palette=None, extra=()):
    """arr: [H,W,ch] uint8/uint16 samples in FILE order (R,G,B[,A] / gray[,A] / index); filter type of row y = filters[y % len]."""
    H, W, ch = arr.shape
    bs = depth // 8
    raw = arr.astype(">u2").tobytes() if depth == 16 else arr.astype(np.uint8).tobytes()
    stride, bpp = W * ch * bs, ch * bs
    rows = [bytearray(raw[y * stride:(y + 1) * stride]) for y in range(H)]
    out = bytearray()
    for y in range(H):
        ft = filters[y % len(filters)]
        cur, prev = rows[y], (rows[y - 1] if y else bytearray(stride))
        f = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, c)][ft]
            f[i] = (cur[i] - pred) & 0xff
        out += bytes([ft]) + f
    z = zlib.compress(bytes(out), 6)
    cuts = [len(z) * k // idat for k in range(idat + 1)]
    png_bytes = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0))
    for t, body in extra:
        png_bytes += _chunk(t, body)
    if palette is not None:
        png_bytes += _chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    for k in range(idat):
        png_bytes += _chunk(b"IDAT", z[cuts[k]:cuts[k + 1]])
    return png_bytes + _chunk(b"IEND", b"")


RNG = np.random.default_rng(5)


def test_rgb8_is_cv2_imread_bgr():
    a = RNG.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    got = png.imread(encode_png(a, 2, 8))
    assert got.dtype == np.uint8 and got.shape == (37, 53, 3) and np.array_equal(got, a[:, :, ::-1])
    assert png.png_info(encode_png(a, 2, 8)) == (37, 53, 8, 2)


@pytest.mark.parametrize("ft", [0, 1, 2, 3, 4])
def test_every_filter_type_alone(ft):
    a = RNG.integers(0, 256, (16, 21, 3), dtype=np.uint8)
    assert np.array_equal(png.imread(encode_png(a, 2, 8, filters=(ft,), idat=1)), a[:, :, ::-1])


def test_depth_gray16_unchanged_is_uint16():
    d = RNG.integers(0, 65536, (24, 40, 1), dtype=np.uint16)
    got = png.imread(encode_png(d, 0, 16), unchanged=True)
    assert got.dtype == np.uint16 and np.array_equal(got, d[:, :, 0])
    # the colour read of the same file: gray replicated, 16 -> 8 bits by the high byte
    col = png.imread(encode_png(d, 0, 16))
    assert np.array_equal(col, np.repeat((d >> 8).astype(np.uint8), 3, axis=2))


def test_other_colour_types():
    a = RNG.integers(0, 256, (9, 14, 4), dtype=np.uint8)
    assert np.array_equal(png.imread(encode_png(a, 6, 8)), a[:, :, 2::-1])                       # RGBA: alpha dropped
    g = RNG.integers(0, 256, (9, 14, 1), dtype=np.uint8)
    assert np.array_equal(png.imread(encode_png(g, 0, 8)), np.repeat(g, 3, axis=2))
    assert np.array_equal(png.imread(encode_png(g, 0, 8), unchanged=True), g[:, :, 0])
    ga = RNG.integers(0, 256, (9, 14, 2), dtype=np.uint8)
    assert np.array_equal(png.imread(encode_png(ga, 4, 8)), np.repeat(ga[:, :, :1], 3, axis=2))
    pal = RNG.integers(0, 256, (200, 3), dtype=np.uint8)
    idx = RNG.integers(0, 200, (9, 14, 1), dtype=np.uint8)
    assert np.array_equal(png.imread(encode_png(idx, 3, 8, palette=pal)), pal[idx[:, :, 0]][:, :, ::-1])
    w16 = RNG.integers(0, 65536, (6, 7, 3), dtype=np.uint16)
    assert np.array_equal(png.imread(encode_png(w16, 2, 16)), (w16 >> 8).astype(np.uint8)[:, :, ::-1])
    # ancillary chunks are skipped
    assert np.array_equal(png.imread(encode_png(g, 0, 8, extra=((b"tEXt", b"k\0v"), (b"gAMA", struct.pack(">I", 45455))))), np.repeat(g, 3, axis=2))


def test_batch_on_threads_matches_single_and_pure_python_decoder():
    frames = [RNG.integers(0, 256, (64, 128, 3), dtype=np.uint8) for _ in range(9)]
    files = [encode_png(f, 2, 8, idat=1 + k % 4) for k, f in enumerate(frames)]
    out = png.decode_batch(files, threads=4)
    assert tuple(out.shape) == (9, 64, 128, 3)
    for k, f in enumerate(frames):
        assert np.array_equal(out[k].numpy(), f[:, :, ::-1]) and np.array_equal(png.imread(files[k]), f[:, :, ::-1])
    depth = [RNG.integers(0, 65536, (32, 64, 1), dtype=np.uint16) for _ in range(5)]
    dout = png.decode_batch([encode_png(d, 0, 16) for d in depth], unchanged=True, threads=0)
    assert np.array_equal(dout.numpy().view(np.uint16), np.stack([d[:, :, 0] for d in depth]))


def test_damaged_and_unsupported_files_are_refused():
    a = RNG.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    good = encode_png(a, 2, 8, idat=1)
    with pytest.raises(ValueError, match="not a PNG"):
        png.imread(b"JFIF" + good[4:])
    bad = bytearray(good); bad[60] ^= 0x40                                        # a flipped bit inside IDAT: the chunk CRC catches it
    with pytest.raises(ValueError, match="CRC"):
        png.imread(bytes(bad))
    with pytest.raises(ValueError, match="truncated|IEND|ends early"):
        png.imread(good[:-20])
    inter = bytearray(good); inter[8 + 8 + 12] = 1                                # interlace flag (+ fixed CRC)
    inter[8 + 8 + 13:8 + 8 + 17] = struct.pack(">I", zlib.crc32(bytes(inter[12:8 + 8 + 13])) & 0xffffffff)
    with pytest.raises(NotImplementedError, match="interlaced"):
        png.imread(bytes(inter))
    with pytest.raises(NotImplementedError, match="single-channel"):
        png.imread(good, unchanged=True)
    files = [good, good[:40]]
    with pytest.raises(ValueError, match="image 1"):
        png.decode_batch(files)


def _with_ihdr(good, w, h, depth=8, ctype=2):
    """`good` with its IHDR rewritten (CRC fixed): the header announces a size the data does not have"""
    body = struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)
    return good[:8] + _chunk(b"IHDR", body) + good[8 + 25:]


def test_hostile_headers_cost_nothing_and_legal_oddities_decode():
    """the IHDR is untrusted — its size is checked against the destination (and 2 GiB) BEFORE a byte is allocated for it, a second
    IHDR is refused, no exception crosses the C ABI; a zero-length IDAT chunk is legal PNG (cv2.imread reads such files)."""
    a = RNG.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    good = encode_png(a, 2, 8, idat=1)
    huge = _with_ihdr(good, 0x7fffffff, 0x7fffffff, 16, 6)                        # 2^65 bytes of samples announced by a 100-byte file
    assert png.png_info(huge)[:2] == (0x7fffffff, 0x7fffffff)                     # (info only parses)
    out = np.empty((8, 8, 3), np.uint8)
    from omnifusion_amd import _lib
    import ctypes
    for blob, H, W in ((huge, 8, 8), (_with_ihdr(good, 16, 8), 8, 8), (_with_ihdr(good, 46341, 46341), 46341, 46341)):
        rc = _lib.load().omni_png_decode(blob, ctypes.c_size_t(len(blob)), out.ctypes.data_as(ctypes.c_void_p), H, W, 0)
        assert rc != 0                                                            # refused with a status, the process lives
    with pytest.raises(ValueError, match="destination 8 x 8"):
        png.decode_batch([good, _with_ihdr(good, 16, 8)])
    twice = good[:8 + 25] + good[8:8 + 25] + good[8 + 25:]
    with pytest.raises(ValueError, match="second IHDR"):
        png.imread(twice)
    ihdr_end = 8 + 25
    empty_idat = good[:ihdr_end] + _chunk(b"IDAT", b"") + good[ihdr_end:-12] + _chunk(b"IDAT", b"") + good[-12:]
    assert np.array_equal(png.imread(empty_idat), a[:, :, ::-1])


def test_png_batches_abandoned_iterator_and_buffer_recycling():
    import threading
    import time
    frames = [RNG.integers(0, 256, (16, 32, 3), dtype=np.uint8) for _ in range(12)]
    files = [encode_png(f, 2, 8, idat=2) for f in frames]
    before = threading.active_count()
    it = iter(png.PngBatches(files, 2, threads=2, pinned=False, ring=2))
    first = next(it)
    assert np.array_equal(first[1].numpy(), frames[1][:, :, ::-1])
    it.close()                                                                    # the consumer leaves early: the producer must not stay blocked in put()
    t0 = time.time()
    while threading.active_count() > before and time.time() - t0 < 5:
        time.sleep(0.05)
    assert threading.active_count() == before
    pb = png.PngBatches(files, 1, threads=2, pinned=False, ring=3, workers=2)       # 12 batches through <= 5 buffers
    seen, ptrs = [], set()
    for b in pb:
        seen.append(b.numpy().copy())
        ptrs.add(b.data_ptr())
        pb.recycle(b)                                                             # done with it at once: later batches reuse the buffers
    assert np.array_equal(np.concatenate(seen), np.stack([f[:, :, ::-1] for f in frames]))
    assert len(ptrs) < len(seen)
