"""The reference's I/O formats (m4ri/io.c:49-357) as libm4ri_amd.so provides them -- pure
host code, no GPU needed.  mzd_from_str, mzd_from_jcf and mzd_fprint_row are compared with the real reference
build.  The PNG pair cannot be compared with libm4ri itself (png.h is not in this image, so the reference's io.c builds
with __M4RI_HAVE_LIBPNG = 0); it is pinned against the two third-party PNG implementations that ARE here, for the
reference's pixel convention (io.c:148-178, :254-287: leftmost pixel = lowest bit, set bit = black = sample 0):
  * libpng itself -- the library the reference reads and writes through -- via ctypes on libpng16.so.16 (simplified
    png_image_* API): it decodes what mzd_to_png wrote, and it encodes (1-bit palette, colour type 3) what mzd_from_png reads;
  * Pillow's PNG plugin (its own chunk parser / writer on zlib): 1-bit grayscale both ways;
and beside them an independent decoder/encoder in this file (zlib + struct) for the scanline filters and split IDATs."""
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


# ---- pinned against third-party PNG implementations: libpng (ctypes) and Pillow ----------------------------------------
class _PngImage(ctypes.Structure):  # png.h: png_image (simplified API), PNG_IMAGE_VERSION 1
    _fields_ = [("opaque", ctypes.c_void_p), ("version", ctypes.c_uint32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("format", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("colormap_entries", ctypes.c_uint32),
                ("warning_or_error", ctypes.c_uint32), ("message", ctypes.c_char * 64)]


_PNG_FORMAT_GRAY, _PNG_FORMAT_RGB_COLORMAP = 0, 2 | 8  # png.h: PNG_FORMAT_FLAG_COLOR = 2, PNG_FORMAT_FLAG_COLORMAP = 8


def _libpng():
    import ctypes.util
    name = ctypes.util.find_library("png16") or "libpng16.so.16"
    try:
        L = ctypes.CDLL(name)
    except OSError:
        pytest.fail("libpng16.so.16 is part of this image: the PNG codec is pinned against it")
    L.png_image_begin_read_from_file.argtypes = [ctypes.POINTER(_PngImage), ctypes.c_char_p]
    L.png_image_finish_read.argtypes = [ctypes.POINTER(_PngImage), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.png_image_write_to_file.argtypes = [ctypes.POINTER(_PngImage), ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.png_image_free.argtypes = [ctypes.POINTER(_PngImage)]
    return L


def _libpng_read_gray8(path):
    L, img = _libpng(), _PngImage()
    img.version = 1
    assert L.png_image_begin_read_from_file(ctypes.byref(img), os.fsencode(path)), img.message
    img.format = _PNG_FORMAT_GRAY
    px = np.zeros((img.height, img.width), dtype=np.uint8)
    assert L.png_image_finish_read(ctypes.byref(img), None, px.ctypes.data_as(ctypes.c_void_p), 0, None), img.message
    L.png_image_free(ctypes.byref(img))
    return px


def _libpng_write_palette1(path, index):
    """index: h x w array of 0/1 palette indices -> a colour-type-3 PNG; libpng picks bit depth 1 for a 2-entry colormap."""
    L, img = _libpng(), _PngImage()
    img.version, img.height, img.width = 1, index.shape[0], index.shape[1]
    img.format, img.colormap_entries = _PNG_FORMAT_RGB_COLORMAP, 2
    cmap = np.array([[0, 0, 0], [255, 255, 255]], dtype=np.uint8)
    buf = np.ascontiguousarray(index, dtype=np.uint8)
    assert L.png_image_write_to_file(ctypes.byref(img), os.fsencode(path), 0, buf.ctypes.data_as(ctypes.c_void_p), 0,
                                     cmap.ctypes.data_as(ctypes.c_void_p)), img.message


PNG_SHAPES = [(1, 1), (3, 8), (5, 9), (10, 64), (7, 65), (33, 130), (64, 200), (200, 1023)]


@pytest.mark.parametrize("m,n", PNG_SHAPES)
def test_png_written_here_decodes_in_libpng_and_pillow(tmp_path, m, n):
    """mzd_to_png's files through the reference's own PNG library and through Pillow: entry 1 = black (sample 0),
    entry 0 = white, column 0 = leftmost pixel, row 0 on top (io.c:254-287)."""
    from PIL import Image
    A = Mzd.random(m, n, 300 + n)
    p = str(tmp_path / "a.png")
    for level in (-1, 0, 9):
        assert m4ri_amd.mzd_to_png(A, p, level, "pinned") == 0
        want = (1 - A.to_bits()).astype(np.uint8) * 255
        assert np.array_equal(_libpng_read_gray8(p), want)
        with Image.open(p) as im:
            assert im.mode == "1" and im.size == (n, m) and im.info.get("Software") == "M4RI" and im.info.get("Comment") == "pinned"
            assert np.array_equal(np.array(im.convert("L")), want)


@pytest.mark.parametrize("m,n", PNG_SHAPES)
def test_png_written_by_libpng_and_pillow_reads_here(tmp_path, m, n):
    """mzd_from_png on files made by third parties.  The reference takes the packed bits as they are -- leftmost pixel =
    lowest bit (png_set_packswap) -- and complements them (io.c:148-178): for a grayscale file a black pixel is a 1, for a
    palette file (colour type 3, which it accepts as well) palette INDEX 0 is a 1, whatever the palette says."""
    from PIL import Image
    rng = np.random.default_rng(7 * m + n)
    px = (rng.random((m, n)) < 0.5).astype(np.uint8)
    p = str(tmp_path / "pil.png")
    Image.fromarray(px * 255).convert("1").save(p, optimize=bool(n % 2))     # 1-bit grayscale, Pillow's writer
    A = m4ri_amd.mzd_from_png(p)
    assert (A.nrows, A.ncols) == (m, n) and np.array_equal(A.to_bits(), 1 - px)
    assert not (A.valid_words()[:, -1] & ~np.uint64(A.high_bitmask)).any()
    q = str(tmp_path / "libpng.png")
    _libpng_write_palette1(q, px)                                             # 1-bit palette, libpng's writer
    raw = open(q, "rb").read()
    assert raw[24] == 1 and raw[25] == 3, "libpng wrote bit depth 1, colour type 3"
    B = m4ri_amd.mzd_from_png(q)
    assert (B.nrows, B.ncols) == (m, n) and np.array_equal(B.to_bits(), 1 - px)


def test_png_reader_refuses_a_header_its_data_cannot_back(tmp_path):
    """An IHDR promising 2^31-1 x 2^31-1 pixels over a few bytes of IDAT is refused (NULL), not allocated."""
    data = bytearray(_png_encode_1bit(np.zeros((2, 2), dtype=np.uint8), [0]))
    body = struct.pack(">IIBBBBB", 0x7fffffff, 0x7fffffff, 1, 0, 0, 0, 0)
    data[8:8 + 25] = struct.pack(">I", 13) + b"IHDR" + body + struct.pack(">I", zlib.crc32(b"IHDR" + body) & 0xFFFFFFFF)
    (tmp_path / "huge.png").write_bytes(bytes(data))
    assert m4ri_amd.mzd_from_png(str(tmp_path / "huge.png")) is None


def test_jcf_reader_edge_cases(tmp_path, reference):
    """Free-form whitespace is the format's own (the reference parses with fscanf); a truncated header is refused; an index
    that addresses a cell outside the matrix is fatal -- including the two the reference does not catch (a positive FIRST
    index = row -1, an index 0 = column -1), which die here instead of writing out of bounds."""
    import subprocess
    import sys
    RL = _ref_bind(reference)
    p = tmp_path / "ws.jcf"
    p.write_text("3   4 2 5\n\n\n  -1 3\t4\n-2\n\n-4 1\n")       # 3 rows opened: (0;0,2,3), (1;1), (2;3,0)
    got = m4ri_amd.mzd_from_jcf(str(p))
    want = from_struct_ptr(RL.mzd_from_jcf(str(p).encode(), 0), RL.mzd_free)
    assert got.equal(want) and got.to_bits().tolist() == [[1, 0, 1, 1], [0, 1, 0, 0], [1, 0, 0, 1]]
    (tmp_path / "short.jcf").write_text("3 4 2\n")
    assert m4ri_amd.mzd_from_jcf(str(tmp_path / "short.jcf")) is None and not RL.mzd_from_jcf(str(tmp_path / "short.jcf").encode(), 0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for text, where in (("2 2 2\n1\n\n-3\n", "(0,2)"), ("2 2 2\n3\n\n-1\n-1\n-1\n", "(2,0)"), ("2 2 2\n1\n\n1\n", "(-1,0)"), ("2 2 2\n1\n\n-1\n0\n", "(0,-1)")):
        (tmp_path / "bad.jcf").write_text(text)
        code = f"import m4ri_amd\nm4ri_amd.mzd_from_jcf({str(tmp_path / 'bad.jcf')!r})\nprint('SURVIVED')\n"
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
        assert r.returncode < 0 and f"trying to write to {where} in 2 x 2 matrix" in r.stderr and "SURVIVED" not in r.stdout, (text, r.stderr[-300:])
