"""Host-side mirror of M4RI's mzd_t (reference m4ri/mzd.h:68-139) for Python callers and tests.

`Mzd` owns (or views) a numpy uint64 buffer laid out exactly like an mzd_t's `data` and carries the
64-byte descriptor as a ctypes structure, so the same object can be handed to libm4ri_amd.so, to
the CPU oracle and to the reference build -- all three read the same bytes.

This module is pure numpy/ctypes: it never computes a product.  Products live in libm4ri_amd.so.
"""
from __future__ import annotations

import ctypes

import numpy as np

RADIX = 64
FLAG_EXCESS = 0x2  # mzd.h:144
FLAG_WINDOW = 0x4  # mzd.h:150
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class MzdStruct(ctypes.Structure):
    """Byte-for-byte the reference's mzd_t (sizeof == 64, asserted at mzd.c:143)."""

    _fields_ = [
        ("nrows", ctypes.c_int32),
        ("ncols", ctypes.c_int32),
        ("width", ctypes.c_int64),
        ("rowstride", ctypes.c_int64),
        ("flags", ctypes.c_uint8),
        ("padding", ctypes.c_uint8 * 23),
        ("high_bitmask", ctypes.c_uint64),
        ("data", ctypes.POINTER(ctypes.c_uint64)),
    ]


assert ctypes.sizeof(MzdStruct) == 64
MzdPtr = ctypes.POINTER(MzdStruct)


def left_bitmask(n: int) -> int:
    """__M4RI_LEFT_BITMASK (misc.h:272): the n lowest bits, n == 0 -> all 64."""
    return (0xFFFFFFFFFFFFFFFF >> ((RADIX - n) % RADIX)) & 0xFFFFFFFFFFFFFFFF


def splitmix_words(seed: int, start: int, count: int) -> np.ndarray:
    """Outputs number start .. start+count-1 of the splitmix64 stream seeded `seed` (counter form)."""
    idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


class Mzd:
    """A dense GF(2) matrix in M4RI's bit-packed row-major layout (bit (r,c) = word[r][c//64] >> c%64)."""

    def __init__(self, nrows: int, ncols: int, buf: np.ndarray | None = None, rowstride: int | None = None,
                 offset: int = 0, windowed: bool = False):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.width = (ncols + RADIX - 1) // RADIX if ncols > 0 else 0
        if rowstride is None:
            rowstride = self.width + (self.width & 1)  # even stride, mzd.c:147-148
        self.rowstride = int(rowstride)
        if buf is None:
            buf = np.zeros(max(1, self.nrows * self.rowstride), dtype=np.uint64)
        self.buf, self.offset, self.windowed = buf, int(offset), bool(windowed)
        self.high_bitmask = left_bitmask(ncols % RADIX)
        s = MzdStruct()
        s.nrows, s.ncols, s.width, s.rowstride = self.nrows, self.ncols, self.width, self.rowstride
        s.high_bitmask = self.high_bitmask
        s.flags = (FLAG_WINDOW if windowed else 0) | (FLAG_EXCESS if ncols % RADIX else 0)
        base = buf.ctypes.data + 8 * self.offset
        s.data = ctypes.cast(ctypes.c_void_p(base), ctypes.POINTER(ctypes.c_uint64))
        if self.nrows == 0 or self.ncols == 0:
            s.data = ctypes.cast(ctypes.c_void_p(base if windowed else 0), ctypes.POINTER(ctypes.c_uint64))
        self.struct = s

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def init(cls, nrows: int, ncols: int) -> "Mzd":
        """mzd_init (mzd.c:142-157): zero matrix."""
        return cls(nrows, ncols)

    @classmethod
    def random(cls, nrows: int, ncols: int, seed: int) -> "Mzd":
        m = cls(nrows, ncols)
        m.fill_splitmix(seed)
        return m

    def window(self, lowr: int, lowc: int, highr: int, highc: int) -> "Mzd":
        """mzd_init_window (mzd.c:159-177): O(1) view, lowc must be a multiple of 64."""
        assert lowc % RADIX == 0
        nrows = min(highr - lowr, self.nrows - lowr)
        return Mzd(nrows, highc - lowc, self.buf, self.rowstride,
                   self.offset + lowr * self.rowstride + lowc // RADIX, windowed=True)

    def copy(self) -> "Mzd":
        """mzd_copy(NULL, self): fresh non-window matrix with the same valid bits."""
        out = Mzd(self.nrows, self.ncols)
        if self.nrows and self.width:
            w = self.valid_words().copy()
            w[:, -1] &= np.uint64(self.high_bitmask)
            out.rows()[:, : self.width] = w
        return out

    # -- views ----------------------------------------------------------------------------------
    def rows(self) -> np.ndarray:
        """(nrows x rowstride) strided view of the backing words (includes padding/neighbour words)."""
        return np.lib.stride_tricks.as_strided(self.buf[self.offset:], shape=(self.nrows, self.rowstride),
                                               strides=(8 * self.rowstride, 8), writeable=True)

    def valid_words(self) -> np.ndarray:
        return self.rows()[:, : self.width]

    def fill_splitmix(self, seed: int) -> None:
        """mzd_randomize_custom's fill order (mzd.c:1282-1292): `width` PRNG words per row, last masked in."""
        if self.nrows == 0 or self.width == 0:
            return
        z = splitmix_words(seed, 0, self.nrows * self.width).reshape(self.nrows, self.width)
        v = self.valid_words()
        mask = np.uint64(self.high_bitmask)
        v[:, : self.width - 1] = z[:, : self.width - 1]
        v[:, self.width - 1] ^= (v[:, self.width - 1] ^ z[:, self.width - 1]) & mask

    def masked(self) -> np.ndarray:
        """Valid words with the excess bits of the last word cleared (what mzd_equal compares)."""
        w = self.valid_words().copy()
        if self.width:
            w[:, -1] &= np.uint64(self.high_bitmask)
        return w

    def equal(self, other: "Mzd") -> bool:
        """mzd_equal (mzd.c:1314-1331)."""
        return (self.nrows, self.ncols) == (other.nrows, other.ncols) and bool(np.array_equal(self.masked(), other.masked()))

    def to_bits(self) -> np.ndarray:
        """nrows x ncols array of 0/1 (small matrices only)."""
        if self.nrows == 0 or self.ncols == 0:
            return np.zeros((self.nrows, self.ncols), dtype=np.uint8)
        b = np.unpackbits(self.masked().view(np.uint8).reshape(self.nrows, -1), axis=1, bitorder="little")
        return b[:, : self.ncols]

    @classmethod
    def from_bits(cls, bits: np.ndarray) -> "Mzd":
        nrows, ncols = bits.shape
        m = cls(nrows, ncols)
        if nrows and ncols:
            pad = np.zeros((nrows, m.width * RADIX), dtype=np.uint8)
            pad[:, :ncols] = bits & 1
            m.valid_words()[:, :] = np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(nrows, m.width)
        return m

    def fingerprint(self) -> int:
        """FNV-1a over the valid bytes, row-major, excess masked (== oracle gf2o_fingerprint)."""
        h = 0xCBF29CE484222325
        for byte in self.masked().view(np.uint8).reshape(-1).tolist():
            h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h

    @property
    def ptr(self):
        return ctypes.byref(self.struct)

    def __repr__(self) -> str:
        return f"Mzd({self.nrows}x{self.ncols}, stride={self.rowstride}, window={self.windowed})"


def from_struct_ptr(p, free_fn=None) -> Mzd:
    """Wrap an mzd_t* allocated by a C library (C == NULL results).  Copies into numpy and frees."""
    s = p.contents
    out = Mzd(s.nrows, s.ncols)
    if s.nrows and s.width:
        src = np.ctypeslib.as_array(s.data, shape=(s.nrows * s.rowstride,))
        out.rows()[:, : s.width] = np.lib.stride_tricks.as_strided(src, shape=(s.nrows, s.width), strides=(8 * s.rowstride, 8))
    if free_fn is not None:
        free_fn(p)
    return out
