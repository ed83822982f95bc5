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


# ------------------------------------------------------------------ round 5: the decoder's own inflate / checksums (csrc/omni_inflate.h) against Python's zlib
def _inflate(z, cap):
    """omni_zlib_inflate -> (status, produced bytes, consumed)"""
    import ctypes
    from omnifusion_amd import _lib as L
    lib = L.load()
    out = np.empty(max(cap, 1), np.uint8)
    prod, used = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = lib.omni_zlib_inflate(ctypes.c_char_p(z), ctypes.c_size_t(len(z)), ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(cap), ctypes.byref(prod), ctypes.byref(used))
    return rc, out[:prod.value].tobytes(), used.value


def _zlib_says(z, cap):
    try:
        do = zlib.decompressobj()
        o = do.decompress(z, cap + 1)
        if len(o) > cap:
            return "full", None
        return ("ok", o) if do.eof else ("short", o)
    except zlib.error:
        return "bad", None


def _payloads():
    r = np.random.default_rng(11)
    yield b""
    yield b"a"
    yield b"abc" * 1000                                                            # overlapping matches (distance 3), then long ones
    yield bytes(100000)                                                            # distance 1, length 258 runs
    yield r.integers(0, 256, 70000, dtype=np.uint8).tobytes()                      # incompressible: stored blocks at level 0, literals otherwise
    yield r.integers(0, 4, 200000, dtype=np.uint8).tobytes()                       # tiny alphabet: 2-bit codes, two literals per table look-up
    yield r.integers(0, 256, 1000, dtype=np.uint8).tobytes() * 200                 # distance 1000 matches
    yield np.cumsum(r.integers(-3, 4, 300000)).astype(np.uint8).tobytes()          # image-like residuals: short matches, every code length
    yield bytes(r.choice([0, 1, 2, 255], 150000, p=[.9, .05, .03, .02]).astype(np.uint8))   # skewed: 1-bit codes and 12+-bit ones (second-level tables)
    yield bytes(range(256)) * 3 + r.integers(0, 256, 40000, dtype=np.uint8).tobytes()[:30000] + bytes(range(255, -1, -1))
    for n in (1, 2, 3, 15, 16, 17, 255, 256, 257, 258, 259, 300, 32767, 32768, 32769, 65536):
        yield r.integers(0, 8, n, dtype=np.uint8).tobytes()


def test_inflate_equals_zlib_on_valid_streams():
    """every block type (stored / fixed / dynamic via the strategies), window sizes, output exactly full, one byte short, trailing bytes after the stream"""
    n = 0
    for d in _payloads():
        for lvl in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for wbits in (15, 9):
                    c = zlib.compressobj(lvl, zlib.DEFLATED, wbits, 8, strat)
                    z = c.compress(d) + c.flush()
                    rc, o, used = _inflate(z, len(d))
                    assert rc == 0 and o == d and used == len(z), (len(d), lvl, strat, wbits, rc)
                    rc, o, used = _inflate(z + b"trailing", len(d) + 7)
                    assert rc == 0 and o == d and used == len(z)
                    if d:
                        rc, o, _ = _inflate(z, len(d) - 1)
                        assert rc != 0 and o == d[:len(o)]
                    n += 1
    assert n > 800
    # a stream of many blocks with empty stored blocks between them (Z_SYNC_FLUSH)
    r = np.random.default_rng(12)
    c, z, parts = zlib.compressobj(6), b"", [r.integers(0, 16, 5000, dtype=np.uint8).tobytes() for _ in range(20)]
    for p in parts:
        z += c.compress(p) + c.flush(zlib.Z_SYNC_FLUSH)
    z += c.flush()
    rc, o, _ = _inflate(z, 100000)
    assert rc == 0 and o == b"".join(parts)


def test_inflate_refuses_what_zlib_refuses():
    """truncated at every byte of the first 200 and a sample beyond, the last 12 bytes one by one; 1-3 flipped bits; random bytes behind a valid
    header: accepted exactly where Python's zlib accepts, with the same bytes, and whatever was produced before a failure is a prefix of the truth"""
    r = np.random.default_rng(13)
    d = np.cumsum(r.integers(-2, 3, 60000)).astype(np.uint8).tobytes()
    for lvl in (1, 6):
        z = zlib.compress(d, lvl)
        for cut in list(range(0, 200)) + list(range(200, len(z), 97)) + [len(z) - k for k in range(1, 12)]:
            rc, o, _ = _inflate(z[:cut], len(d))
            assert rc != 0 and o == d[:len(o)], cut
            assert _zlib_says(z[:cut], len(d))[0] in ("short", "bad")
    both_ok = 0
    for trial in range(1500):
        z = bytearray(zlib.compress(d[:int(r.integers(100, 20000))], int(r.choice([1, 6, 9]))))
        cap = len(zlib.decompress(bytes(z)))
        for _ in range(int(r.integers(1, 4))):
            z[int(r.integers(0, len(z)))] ^= 1 << int(r.integers(0, 8))
        rc, o, _ = _inflate(bytes(z), cap)
        kind, ref = _zlib_says(bytes(z), cap)
        assert (rc == 0) == (kind == "ok"), (trial, rc, kind)
        if rc == 0:
            assert o == ref
            both_ok += 1
    for trial in range(1500):
        z = bytes([0x78, 0x9c]) + r.integers(0, 256, int(r.integers(1, 400)), dtype=np.uint8).tobytes()
        rc, o, _ = _inflate(z, 5000)
        assert (rc == 0) == (_zlib_says(z, 5000)[0] == "ok"), trial
    # header checks: method, window, check bits, preset dictionary
    good = zlib.compress(b"hello hello hello")
    for bad in (bytes([0x79]) + good[1:], bytes([0x88]) + good[1:], bytes([good[0], good[1] ^ 1]) + good[2:], bytes([0x78, 0xbb]) + good[2:]):
        assert _inflate(bad, 100)[0] != 0


def test_checksums_equal_zlib():
    import ctypes
    from omnifusion_amd import _lib as L
    lib = L.load()
    r = np.random.default_rng(14)

    def both(b, crc0=0, adler0=1):
        c, a = ctypes.c_uint(crc0), ctypes.c_uint(adler0)
        assert lib.omni_png_checksums(ctypes.c_char_p(b), ctypes.c_size_t(len(b)), ctypes.byref(c), ctypes.byref(a)) == 0
        return c.value, a.value
    for n in list(range(0, 200)) + [255, 256, 1000, 4099, 5551, 5552, 5553, 65536, 1572864 + 512]:
        b = r.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert both(b) == (zlib.crc32(b), zlib.adler32(b)), n
        assert both(b, 0x12345678, 0x00c0ffee) == (zlib.crc32(b, 0x12345678), zlib.adler32(b, 0x00c0ffee)), n
    b = bytes([255]) * (1 << 21)                                                   # the largest sums: no 32-bit overflow between the modulo steps
    assert both(b) == (zlib.crc32(b), zlib.adler32(b))


@pytest.mark.parametrize("W", [1, 2, 3, 4, 7, 64, 1024])
def test_rgb8_row_kernels_every_filter_on_every_row_position(W):
    """the 8-bit RGB row kernels (one pixel per vector step) at widths below and above their minimum, every filter type on the first row (no row
    above) and below it, smooth content (Paeth ties, all three outcomes) and noise"""
    H = 10
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = np.stack([(3 * xx + 5 * yy) % 256, (250 - 2 * xx + yy) % 256, (xx * yy) % 256], axis=2).astype(np.uint8)
    noise = RNG.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for a in (smooth, noise, np.zeros_like(noise), np.full_like(noise, 255)):
        for rot in range(5):
            filters = tuple((rot + k) % 5 for k in range(5))
            assert np.array_equal(png.imread(encode_png(a, 2, 8, filters=filters, idat=2)), a[:, :, ::-1]), (W, rot)


def test_many_small_idat_chunks_and_empty_ones():
    """libpng writes 8-KiB IDAT chunks; here the stream is cut into 1-byte chunks with empty ones between them"""
    a = RNG.integers(0, 256, (12, 17, 3), dtype=np.uint8)
    one = encode_png(a, 2, 8, idat=1)
    pos, z = 8, b""
    while pos < len(one):
        ln = struct.unpack(">I", one[pos:pos + 4])[0]
        if one[pos + 4:pos + 8] == b"IDAT":
            z += one[pos + 8:pos + 8 + ln]
        pos += 12 + ln
    f = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 17, 12, 8, 2, 0, 0, 0))
    for k in range(len(z)):
        f += _chunk(b"IDAT", z[k:k + 1]) + (_chunk(b"IDAT", b"") if k % 50 == 0 else b"")
    f += _chunk(b"IEND", b"")
    assert np.array_equal(png.imread(f), a[:, :, ::-1])
    # a stream cut inside its Adler-32 trailer still carries every sample (libz's streaming inflate, and libpng with it, reads such files)
    cut = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 17, 12, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", z[:-2]) + _chunk(b"IEND", b"")
    assert np.array_equal(png.imread(cut), a[:, :, ::-1])
    # ... a wrong trailer is a damaged file
    wrong = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 17, 12, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", z[:-1] + bytes([z[-1] ^ 1])) + _chunk(b"IEND", b"")
    with pytest.raises(ValueError, match="corrupt"):
        png.imread(wrong)
    # ... and so is a stream that holds more rows than the header announces
    more = zlib.compress(zlib.decompress(z) + bytes(1 + 17 * 3))
    with pytest.raises(ValueError, match="more image data"):
        png.imread(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 17, 12, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", more) + _chunk(b"IEND", b""))


@pytest.mark.parametrize("ctype,depth", [(0, 8), (0, 16), (2, 8), (2, 16), (4, 8), (4, 16), (6, 8), (6, 16), (3, 8)])
def test_row_kernels_every_pixel_size(ctype, depth):
    """1, 2, 3, 4, 6 and 8 bytes per pixel through the vector row kernels (rows of 16 bytes and more) and the byte loops (shorter rows), every
    filter type on every row position, smooth content (Paeth ties) and noise; the colour read's conversions as in test_other_colour_types"""
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    top = 200 if ctype == 3 else (1 << depth)
    pal = RNG.integers(0, 256, (200, 3), dtype=np.uint8) if ctype == 3 else None
    for W in (1, 5, 16, 33, 200):
        H = 7
        yy, xx = np.mgrid[0:H, 0:W]
        smooth = np.stack([((3 + c) * xx + (5 - c) * yy * (257 if depth == 16 else 1)) % top for c in range(ch)], axis=2)
        noise = RNG.integers(0, top, (H, W, ch))
        for a in (smooth, noise):
            a = a.astype(np.uint16 if depth == 16 else np.uint8)
            for rot in range(5):
                f = encode_png(a, ctype, depth, filters=tuple((rot + k) % 5 for k in range(5)), idat=1, palette=pal)
                a8 = (a >> 8).astype(np.uint8) if depth == 16 else a
                want = pal[a[:, :, 0]][:, :, ::-1] if ctype == 3 else (np.repeat(a8[:, :, :1], 3, axis=2) if ch <= 2 else a8[:, :, 2::-1])
                assert np.array_equal(png.imread(f), want), (ctype, depth, W, rot)
                if ctype == 0:
                    assert np.array_equal(png.imread(f, unchanged=True), a[:, :, 0]), (depth, W, rot)


def test_decode_batch_works_in_a_forked_child_and_beside_a_busy_pool():
    """the decoder pool is process-wide; torch DataLoader workers (the reference's test.py:90-97) are FORKED children that inherit the
    pool object but none of its threads — a batch call in the child queued helpers nobody ran and waited for ever.  The pool is now abandoned across
    fork() (pthread_atfork holds its locks over the fork, the child starts its own pool on first use), and a caller no longer waits for helpers that
    never started.  The child must decode the same bytes as the parent, several times, within seconds."""
    import os
    import threading
    frames = [RNG.integers(0, 256, (48, 96, 3), dtype=np.uint8) for _ in range(12)]
    files = [encode_png(f, 2, 8, idat=2) for f in frames]
    want = png.decode_batch(files, threads=0).numpy().copy()                       # the parent's pool exists and has run
    busy = threading.Event()

    def hammer():                                                                  # the pool is busy (and its locks are taken and released) while the child forks
        while not busy.is_set():
            png.decode_batch(files, threads=0)
    th = [threading.Thread(target=hammer) for _ in range(3)]
    for t in th:
        t.start()
    try:
        for rep in range(3):
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:                                                           # the child: no pytest machinery, just the library
                code = 1
                try:
                    os.close(r)
                    ok = all(np.array_equal(png.decode_batch(files, threads=0).numpy(), want) for _ in range(5))
                    ok = ok and np.array_equal(png.decode_batch(files, threads=4).numpy(), want)
                    os.write(w, b"ok" if ok else b"no")
                    code = 0
                finally:
                    os._exit(code)
            os.close(w)
            done = threading.Event()
            got = []

            def reader():
                got.append(os.read(r, 2)); done.set()
            rt = threading.Thread(target=reader, daemon=True)
            rt.start()
            finished = done.wait(60)
            if not finished:
                os.kill(pid, 9)
            os.waitpid(pid, 0)
            os.close(r)
            assert finished, "the forked child hung in omni_png_decode_batch"
            assert got == [b"ok"], got
    finally:
        busy.set()
        for t in th:
            t.join()
    assert np.array_equal(png.decode_batch(files, threads=0).numpy(), want)       # ... and the parent's pool is intact


def _bits_writer():
    out, acc, n = bytearray(), 0, 0

    def put(v, k):                                                                 # k bits, LSB first (header fields, extra bits)
        nonlocal acc, n
        acc |= v << n; n += k
        while n >= 8:
            out.append(acc & 255); acc >>= 8; n -= 8

    def code(c, k):                                                                # a Huffman code: MSB first
        put(int(format(c, "0%db" % k)[::-1], 2), k)

    def flush():
        nonlocal acc, n
        if n:
            out.append(acc & 255); acc = 0; n = 0
        return bytes(out)
    return put, code, flush


def _canonical(lengths):
    """RFC 1951 3.2.2: symbol -> (code, length)"""
    bl = {}
    for l in lengths:
        if l:
            bl[l] = bl.get(l, 0) + 1
    code, nxt = 0, {}
    for bits in range(1, 16):
        code = (code + bl.get(bits - 1, 0)) << 1
        nxt[bits] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[s] = (nxt[l], l); nxt[l] += 1
    return out


def test_hand_built_streams_15_bit_codes_and_distance_32768():
    """the differential fuzz's payloads are small; the corners zlib's own encoder hardly ever emits are built by hand here — a dynamic
    block whose literal / length codes are 15 bits long (behind the second-level tables) and whose matches reach back the full 32768 bytes with length
    258 — and a multi-megabyte photo-like stream; all checked against Python's zlib."""
    r = np.random.default_rng(21)
    # literal/length alphabet: 0..255 + 256 + 257..285 (286 symbols); a Kraft-complete length set with many 15-bit codes:
    # 2 symbols of length 2 would dominate; use: one symbol of length 1 (literal 0x41), then a chain 2,3,...,14 for 13 symbols and the rest of the budget at 15
    ll = [0] * 286
    ll[0x41] = 1
    chain = [256, 285, 0x42, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x4b, 0x4c]     # lengths 2..14
    for k, sym in enumerate(chain):
        ll[sym] = 2 + k
    rest = [s for s in (list(range(0x50, 0x50 + 3)) + [257]) if ll[s] == 0]                    # remaining Kraft budget 2^-14 = two codes of length 15 ... use 15-bit codes
    # budget left after 1/2 + sum_{l=2..14} 2^-l = 1 - 2^-14 -> exactly two 15-bit codes
    ll[rest[0]] = 15; ll[rest[1]] = 15
    dl = [0] * 30
    dl[29] = 1; dl[0] = 1                                                          # distance codes: 29 (24577..32768, 13 extra bits) and 0 (distance 1)
    lit, dist = _canonical(ll), _canonical(dl)
    # code-length alphabet: we need to transmit lengths {0,1,2..15}: give every code-length symbol 0..15 a 4-bit code (16 symbols x 2^-4 = 1), 16/17/18 unused
    cl = [4] * 16 + [0, 0, 0]
    clc = _canonical(cl)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    put, code, flush = _bits_writer()
    put(1, 1); put(2, 2)                                                           # BFINAL, dynamic
    put(286 - 257, 5); put(30 - 1, 5); put(19 - 4, 4)
    for s in order:
        put(cl[s], 3)
    for l in ll + dl:
        code(*clc[l])
    plain = bytearray()

    def literal(b):
        code(*lit[b]); plain.append(b)
    window = bytes(r.choice([0x41, 0x42, 0x43, 0x50, 0x51], 32768, p=[.6, .2, .1, .05, .05]).astype(np.uint8))
    for b in window:
        literal(b)
    for _ in range(40):                                                            # length 258 (symbol 285, no extra bits) at distance 32768 (code 29 + 13 extra bits all ones)
        code(*lit[285]); code(*dist[29]); put(8191, 13)
        start = len(plain) - 32768
        for k in range(258):
            plain.append(plain[start + k])
        literal(0x50); literal(0x51)                                               # the two 15-bit literals between the matches
        code(*lit[285]); code(*dist[0])                                            # distance 1, length 258: a run
        for k in range(258):
            plain.append(plain[-1])
    code(*lit[256])
    deflate = flush()
    z = b"\x78\x01" + deflate + struct.pack(">I", zlib.adler32(bytes(plain)))
    assert zlib.decompress(z) == bytes(plain)                                      # the hand-built stream is valid by an independent decoder
    rc, o, used = _inflate(z, len(plain))
    assert rc == 0 and o == bytes(plain) and used == len(z)
    rc, o, _ = _inflate(z, len(plain) - 1)
    assert rc != 0
    # photo-like, several megabytes (a real panorama inflates to 1.5-25 MB): smooth gradients + noise through the Paeth filter of the encoder above, and as a raw stream
    h, w = 768, 1536
    yy, xx = np.mgrid[0:h, 0:w]
    img = (np.stack([xx * 0.11 + yy * 0.07, xx * 0.05 - yy * 0.09, (xx + yy) * 0.03], 2) + r.normal(0, 6, (h, w, 3))).astype(np.int64) % 256
    img = img.astype(np.uint8)
    f = encode_png(img, 2, 8, filters=(4,), idat=5)
    assert np.array_equal(png.imread(f), img[:, :, ::-1])
    raw = img.tobytes() * 2                                                        # 7 MB, the second half a 3.5-MB-distant repeat (no match reaches it: fresh codes again)
    for lvl in (1, 6):
        zz = zlib.compress(raw, lvl)
        rc, o, used = _inflate(zz, len(raw))
        assert rc == 0 and o == raw and used == len(zz)
