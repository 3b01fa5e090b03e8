"""The reference's I/O formats (m4ri/io.c:49-357) as libm4ri_amd.so provides them -- pure
host code, no GPU needed.  mzd_from_str, mzd_from_jcf and mzd_fprint_row are compared with the real reference
build; the PNG pair cannot be (the reference build in oracle/_ref has no libpng: __M4RI_HAVE_LIBPNG = 0, so the
PNG codec's parity against libm4ri is UNPINNED): it is checked against the PNG specification instead -- an
independent decoder/encoder in this file (zlib + struct), the pixel convention of io.c:148-178 / :254-287
(leftmost pixel = lowest bit, set bit = black = sample 0), and round trips."""
import ctypes
import os
import struct
import zlib

import numpy as np
import pytest

import m4ri_amd
from m4ri_amd.mzd import Mzd, MzdPtr, from_struct_ptr


def _ref_bind(reference):
    L = reference.L
    L.mzd_from_str.restype, L.mzd_from_str.argtypes = MzdPtr, [ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    L.mzd_from_jcf.restype, L.mzd_from_jcf.argtypes = MzdPtr, [ctypes.c_char_p, ctypes.c_int]
    L.mzd_fprint_row.restype, L.mzd_fprint_row.argtypes = None, [ctypes.c_void_p, MzdPtr, ctypes.c_int]
    return L


def test_from_str_matches_reference(reference):
    RL = _ref_bind(reference)
    rng = np.random.default_rng(1)
    for (m, n) in [(1, 1), (3, 5), (7, 64), (5, 65), (10, 130)]:
        s = "".join(rng.choice(["0", "1"], size=m * n))
        got = m4ri_amd.mzd_from_str(m, n, s)
        want = from_struct_ptr(RL.mzd_from_str(m, n, s.encode()), RL.mzd_free)
        assert got.equal(want)
        assert np.array_equal(got.to_bits().reshape(-1), np.array([c == "1" for c in s], dtype=np.uint8))


def test_from_jcf_matches_reference(tmp_path, reference):
    RL = _ref_bind(reference)
    rng = np.random.default_rng(2)
    for k, (m, n, dens) in enumerate([(5, 7, 0.5), (40, 100, 0.1), (64, 64, 0.3), (30, 200, 0.02)]):
        bits = (rng.random((m, n)) < dens).astype(np.uint8)
        bits[m // 2] = 0  # an empty row cannot be expressed (a row starts with its first negative index): keep one out of the file
        lines = [f"{m} {n} 2", str(int(bits.sum())), ""]
        rows = []
        for i in range(m):
            cols = np.nonzero(bits[i])[0]
            if len(cols) == 0:
                continue
            rows.append(i)
            lines.append(str(-(cols[0] + 1)))
            lines += [str(c + 1) for c in cols[1:]]
        path = tmp_path / f"m{k}.jcf"
        path.write_text("\n".join(lines) + "\n")
        got = m4ri_amd.mzd_from_jcf(str(path))
        want = from_struct_ptr(RL.mzd_from_jcf(str(path).encode(), 0), RL.mzd_free)
        assert got.equal(want)
        expect = np.zeros((m, n), dtype=np.uint8)   # rows are numbered by order of appearance in the file
        for dst, src in enumerate(rows):
            expect[dst] = bits[src]
        assert np.array_equal(got.to_bits(), expect)
    assert m4ri_amd.mzd_from_jcf(str(tmp_path / "missing.jcf")) is None
    bad = tmp_path / "bad.jcf"
    bad.write_text("3 3 5\n1\n\n-1\n")  # p != 2
    assert m4ri_amd.mzd_from_jcf(str(bad)) is None and not RL.mzd_from_jcf(str(bad).encode(), 0)


def test_fprint_row_matches_reference(tmp_path, reference):
    RL = _ref_bind(reference)
    libc = ctypes.CDLL(None)
    libc.fopen.restype, libc.fopen.argtypes = ctypes.c_void_p, [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    for (m, n) in [(3, 1), (2, 64), (4, 65), (3, 200), (2, 128)]:
        A = Mzd.random(m, n, 9)
        outs = []
        for tag, fn in (("ours", m4ri_amd.lib().mzd_fprint_row), ("ref", RL.mzd_fprint_row)):
            p = str(tmp_path / f"{tag}_{m}_{n}.txt").encode()
            fh = libc.fopen(p, b"w")
            for i in range(m):
                fn(fh, A.ptr, i)
            libc.fclose(fh)
            outs.append(open(p, "rb").read())
        assert outs[0] == outs[1] and outs[0].count(b"\n") == m


# ---- an independent 1-bit PNG codec (PNG spec: chunks, CRC, zlib, scanline filters) ------------------------------
def _png_chunks(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, out = 8, []
    while at < len(data):
        ln, typ = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + ln]
        crc, = struct.unpack(">I", data[at + 8 + ln:at + 12 + ln])
        assert crc == zlib.crc32(typ + body) & 0xFFFFFFFF, "chunk CRC"
        out.append((typ, body))
        at += 12 + ln
    return out


def _png_decode_1bit(data):
    ch = _png_chunks(data)
    w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", ch[0][1])
    assert ch[0][0] == b"IHDR" and (depth, ctype, comp, filt, inter) == (1, 0, 0, 0, 0) and ch[-1][0] == b"IEND"
    raw = zlib.decompress(b"".join(b for t, b in ch if t == b"IDAT"))
    rb = (w + 7) // 8
    px = np.zeros((h, w), dtype=np.uint8)
    for i in range(h):
        line = raw[i * (rb + 1):(i + 1) * (rb + 1)]
        assert line[0] == 0
        bits = np.unpackbits(np.frombuffer(line[1:], dtype=np.uint8))  # big-endian bit order: pixel x = bit 7 - x % 8
        px[i] = bits[:w]
    return px, dict((t, b) for t, b in ch if t == b"tEXt" or True)


def _png_encode_1bit(px, filters):
    h, w = px.shape
    rb = (w + 7) // 8
    raw, prev = b"", bytes(rb)
    for i in range(h):
        line = np.packbits(np.concatenate([px[i], np.zeros(rb * 8 - w, dtype=np.uint8)])).tobytes()
        f = filters[i % len(filters)]
        out = bytearray(rb)
        for b in range(rb):
            a, up, ul = (line[b - 1] if b else 0), prev[b], (prev[b - 1] if b else 0)
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = up
            elif f == 3:
                pred = (a + up) // 2
            else:
                p = a + up - ul
                pa, pb, pc = abs(p - a), abs(p - up), abs(p - ul)
                pred = a if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
            out[b] = (line[b] - pred) & 0xFF
        raw += bytes([f]) + bytes(out)
        prev = line

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    comp = zlib.compress(raw)
    half = len(comp) // 2
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 1, 0, 0, 0, 0)) + chunk(b"IDAT", comp[:half])
            + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("m,n", [(1, 1), (3, 8), (5, 9), (10, 64), (7, 65), (33, 130), (64, 200)])
def test_png_write_is_a_valid_png_with_black_for_one(tmp_path, m, n):
    A = Mzd.random(m, n, 100 + n)
    p = str(tmp_path / "a.png")
    assert m4ri_amd.mzd_to_png(A, p, 6, "a comment") == 0
    px, _ = _png_decode_1bit(open(p, "rb").read())
    assert np.array_equal(px, 1 - A.to_bits()), "entry 1 <-> black pixel (sample 0), leftmost pixel = column 0"
    texts = [b for t, b in _png_chunks(open(p, "rb").read()) if t == b"tEXt"]
    assert any(b.startswith(b"Software\x00M4RI") for b in texts) and any(b == b"Comment\x00a comment" for b in texts)
    back = m4ri_amd.mzd_from_png(p)
    assert back.equal(A)


@pytest.mark.parametrize("filters", [[0], [1], [2], [3], [4], [0, 1, 2, 3, 4]])
def test_png_read_handles_every_scanline_filter_and_split_idat(tmp_path, filters):
    rng = np.random.default_rng(5)
    px = (rng.random((37, 131)) < 0.5).astype(np.uint8)
    p = tmp_path / "f.png"
    p.write_bytes(_png_encode_1bit(px, filters))
    A = m4ri_amd.mzd_from_png(str(p))
    assert np.array_equal(A.to_bits(), 1 - px)
    assert not (A.valid_words()[:, -1] & ~np.uint64(A.high_bitmask)).any()


def test_png_read_rejects_what_the_reference_rejects(tmp_path):
    (tmp_path / "x.png").write_bytes(b"not a png at all")
    assert m4ri_amd.mzd_from_png(str(tmp_path / "x.png")) is None
    assert m4ri_amd.mzd_from_png(str(tmp_path / "nope.png")) is None
    data = bytearray(_png_encode_1bit(np.zeros((2, 2), dtype=np.uint8), [0]))
    body = bytearray(struct.pack(">IIBBBBB", 2, 2, 1, 2, 0, 0, 0))  # colour type 2 (RGB)
    data[8:8 + 25] = struct.pack(">I", 13) + b"IHDR" + body + struct.pack(">I", zlib.crc32(b"IHDR" + bytes(body)) & 0xFFFFFFFF)
    (tmp_path / "rgb.png").write_bytes(bytes(data))
    assert m4ri_amd.mzd_from_png(str(tmp_path / "rgb.png")) is None
