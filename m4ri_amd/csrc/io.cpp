// io.cpp -- the reference's matrix I/O formats, pure host code:
//   mzd_fprint_row   /root/reference m4ri/io.h:44,  io.c:49-70    "[" 64-bit groups, ':' every 4 bits, '|' between words "]"
//   mzd_from_str     m4ri/io.h:193, io.c:350-357   row-major string of '0' / '1'
//   mzd_from_jcf     m4ri/io.h:180, io.c:297-348   Jean-Guillaume Dumas' sparse text format: "m n p\nnnz\n\n", then signed
//                                                  1-based column indices, a negative one starts the next row
//   mzd_from_png / mzd_to_png   m4ri/io.h:103,129, io.c:72-293   1-bit grayscale PNG, one pixel per entry, black = 1
// The reference reads and writes PNG through libpng (png_set_packswap: leftmost pixel = lowest bit of a byte, i.e.
// the byte stream of a row IS the little-endian byte stream of its words; png_set_invert_mono on write and `~` on read:
// a set bit is a black pixel, sample value 0).  libpng is not in this image; the same files are produced and parsed here
// directly on zlib: IHDR (bit depth 1, colour type 0, no interlace), tEXt chunks Software / Date / Comment like
// io.c:228-242, one IDAT stream of filter-0 scanlines, IEND.  The reader accepts what the reference accepts (bit depth 1,
// colour type 0 or 3, non-interlaced) with any of the five scanline filters.
#include <zlib.h>
#include <dlfcn.h>
#include <cctype>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include "../../include/m4ri_amd.h"

namespace {

[[noreturn]] void die(const char *fmt, ...) {  // misc.c:36-42
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  abort();
}

mzd_t *new_matrix(rci_t r, rci_t c) {  // from the host program's libm4ri when there is one: the caller will mzd_free() it
  typedef mzd_t *(*init_fn)(rci_t, rci_t);
  static init_fn host_init = reinterpret_cast<init_fn>(dlsym(RTLD_DEFAULT, "mzd_init"));
  return host_init ? host_init(r, c) : m4ri_amd_mzd_init(r, c);
}

void drop_matrix(mzd_t *A) { m4ri_amd_result_free(A); }

inline void write_bit(mzd_t *A, rci_t r, rci_t c, int v) {  // mzd.h: mzd_write_bit
  word *w = A->data + (int64_t)r * A->rowstride + c / 64;
  *w      = (*w & ~((word)1 << (c % 64))) | ((word)(v & 1) << (c % 64));
}

void put_be32(std::vector<unsigned char> &v, uint32_t x) {
  v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}

void put_chunk(std::vector<unsigned char> &out, const char type[4], const unsigned char *data, size_t len) {
  put_be32(out, (uint32_t)len);
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  if (len) out.insert(out.end(), data, data + len);
  put_be32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(len + 4)));
}

void put_text(std::vector<unsigned char> &out, const char *key, const char *text) {  // tEXt: keyword, NUL, text
  std::string s(key);
  s.push_back('\0');
  s += text ? text : "";
  put_chunk(out, "tEXt", reinterpret_cast<const unsigned char *>(s.data()), s.size());
}

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

extern "C" {

void mzd_fprint_row(FILE *stream, mzd_t const *M, const rci_t i) {  // io.c:49-70
  fputc('[', stream);
  const word *row = M->data + (int64_t)i * M->rowstride;
  for (wi_t j = 0; j + 1 < M->width; ++j) {
    for (int b = 0; b < 64; ++b) {
      if (b != 0 && (b % 4) == 0) fputc(':', stream);  // misc.c:46-56 m4ri_word_to_str(.., colon = 1)
      fputc(((row[j] >> b) & 1) ? '1' : ' ', stream);
    }
    fputc('|', stream);
  }
  if (M->width > 0) {
    const word last = row[M->width - 1];
    const int wide  = (M->ncols % 64) ? M->ncols % 64 : 64;
    for (int b = 0; b < wide; ++b) {
      if (b != 0 && (b % 4) == 0) fputc(':', stream);
      fputc(((last >> b) & 1) ? '1' : ' ', stream);
    }
  }
  fputs("]\n", stream);
}

void mzd_fprint(FILE *stream, mzd_t const *M) {  // io.h:66-68
  for (rci_t i = 0; i < M->nrows; ++i) mzd_fprint_row(stream, M, i);
}

void mzd_print(mzd_t const *M) { mzd_fprint(stdout, M); }  // io.h:78

mzd_t *mzd_from_str(rci_t m, rci_t n, const char *str) {  // io.c:350-357
  mzd_t *A = new_matrix(m, n);
  size_t idx = 0;
  for (rci_t i = 0; i < A->nrows; ++i)
    for (rci_t j = 0; j < A->ncols; ++j) write_bit(A, i, j, str[idx++] == '1');
  return A;
}

// JCF reader.  The format is a stream of whitespace-separated decimal integers: "rows cols modulus" and an entry count, then
// one signed 1-based column index per entry, a NEGATIVE index opening the next row (io.h:160-180).  Own structure: the file
// is slurped and walked by a small integer tokenizer (the reference drives fscanf directly, io.c:297-348); the diagnostics
// are kept word for word so that callers parsing stdout see the same text.  Where the reference writes outside the matrix
// (a first index that is positive addresses row -1; an index 0 addresses column -1) this reader dies with the message
// the reference uses for an index beyond the last row / column.
mzd_t *mzd_from_jcf(const char *fn, int verbose) {
  struct IntStream {
    std::string text;
    size_t at = 0;
    bool load(const char *path) {
      FILE *fh = fopen(path, "r");
      if (!fh) return false;
      char chunk[1 << 16];
      for (size_t got; (got = fread(chunk, 1, sizeof chunk, fh)) > 0;) text.append(chunk, got);
      fclose(fh);
      return true;
    }
    bool next(long long &value) {  // false at the end of the file or at the first token that is not an integer
      while (at < text.size() && isspace((unsigned char)text[at])) ++at;
      if (at >= text.size()) return false;
      char *stop = nullptr;
      const long long v = strtoll(text.c_str() + at, &stop, 10);
      if (stop == text.c_str() + at) return false;
      at    = (size_t)(stop - text.c_str());
      value = v;
      return true;
    }
  } in;
  if (!in.load(fn)) {
    if (verbose) printf("Could not open file '%s' for reading\n", fn);
    return NULL;
  }
  long long header[4] = {0, 0, 0, 0};  // rows, columns, modulus, number of entries
  for (long long &h : header)
    if (!in.next(h)) {
      if (verbose) printf("File '%s' does not seem to be in JCF format.", fn);
      return NULL;
    }
  const long long nrows = header[0], ncols = header[1], modulus = header[2], entries = header[3];
  if (modulus != 2) {
    if (verbose) printf("Expected p==2 but found p==%d\n", (int)modulus);
    return NULL;
  }
  if (nrows < 0 || ncols < 0 || nrows > 0x7fffffffLL || ncols > 0x7fffffffLL || entries < 0) {
    if (verbose) printf("File '%s' does not seem to be in JCF format.", fn);
    return NULL;
  }
  if (verbose)
    printf("reading %d x %d matrix with at most %" PRId64 " non-zero entries (density at most: %6.5f)\n", (int)nrows, (int)ncols,
           (int64_t)entries, ((double)entries) / ((double)nrows * ncols));
  mzd_t *A      = new_matrix((rci_t)nrows, (rci_t)ncols);
  long long row = -1;  // no row is open until the first negative index
  for (long long token; in.next(token);) {
    if (token < 0) ++row;
    // an index the matrix cannot hold (strtoll saturates on overflow: LLONG_MIN must not be negated) is out of range as it stands
    const bool representable = token >= -0x7fffffffLL && token <= 0x7fffffffLL;
    const long long col      = representable ? (token < 0 ? -token : token) - 1 : 0x7fffffffLL;
    if (row < 0 || row >= nrows || col < 0 || col >= ncols)
      die("trying to write to (%d,%d) in %d x %d matrix\n", (int)row, (int)col, (int)nrows, (int)ncols);
    write_bit(A, (rci_t)row, (rci_t)col, 1);
  }
  return A;
}

int mzd_to_png(const mzd_t *A, const char *fn, int compression_level, const char *comment, int verbose) {  // io.c:193-293
  FILE *fh = fopen(fn, "wb");
  if (!fh) {
    if (verbose) printf("Could not open file '%s' for writing\n", fn);
    return 1;
  }
  std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<unsigned char> ihdr;
  put_be32(ihdr, (uint32_t)A->ncols);
  put_be32(ihdr, (uint32_t)A->nrows);
  const unsigned char tail[5] = {1, 0, 0, 0, 0};  // bit depth 1, grayscale, deflate, adaptive filtering, no interlace
  ihdr.insert(ihdr.end(), tail, tail + 5);
  put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
  char pdate[32];
  time_t ptime     = time(NULL);
  struct tm *ltime = localtime(&ptime);
  snprintf(pdate, sizeof pdate, "%04d/%02d/%02d %02d:%02d:%02d", ltime->tm_year + 1900, ltime->tm_mon + 1, ltime->tm_mday, ltime->tm_hour,
           ltime->tm_min, ltime->tm_sec);
  put_text(out, "Software", "M4RI");
  put_text(out, "Date", pdate);
  put_text(out, "Comment", comment);
  // scanlines: filter byte 0, then the row's bytes: pixel x of a byte is bit 7 - x in the file; the reference hands
  // libpng little-endian word bytes with packswap (bit order reversed inside every byte) and invert_mono (all bits flipped)
  const size_t rowbytes = ((size_t)A->ncols + 7) / 8;
  std::vector<unsigned char> raw((size_t)A->nrows * (rowbytes + 1));
  static unsigned char rev[256];
  static bool rev_ok = false;
  if (!rev_ok) {
    for (int x = 0; x < 256; ++x) { unsigned char r = 0; for (int b = 0; b < 8; ++b) r |= (unsigned char)(((x >> b) & 1) << (7 - b)); rev[x] = r; }
    rev_ok = true;
  }
  for (rci_t i = 0; i < A->nrows; ++i) {
    unsigned char *dst = raw.data() + (size_t)i * (rowbytes + 1);
    *dst++ = 0;
    const word *row = A->data + (int64_t)i * A->rowstride;
    for (size_t b = 0; b < rowbytes; ++b) {
      unsigned char byte = (unsigned char)(row[b / 8] >> (8 * (b % 8)));
      if (b == rowbytes - 1 && (A->ncols % 8)) byte &= (unsigned char)((1u << (A->ncols % 8)) - 1);  // padding pixels of the last byte: entry 0
      dst[b] = (unsigned char)~rev[byte];
    }
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<unsigned char> comp(clen);
  if (compression_level < -1 || compression_level > 9) compression_level = -1;
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), compression_level) != Z_OK) {
    if (verbose) printf("error writing PNG file\n");
    fclose(fh);
    return 1;
  }
  if ((uint64_t)clen > 0x7fffffffull) {  // a PNG chunk length is 31 bits; one IDAT per file is all this writer makes
    if (verbose) printf("error writing PNG file\n");
    fclose(fh);
    return 1;
  }
  put_chunk(out, "IDAT", comp.data(), clen);
  put_chunk(out, "IEND", nullptr, 0);
  const bool ok = fwrite(out.data(), 1, out.size(), fh) == out.size();
  fclose(fh);
  if (!ok && verbose) printf("error writing PNG file\n");
  return ok ? 0 : 1;
}

mzd_t *mzd_from_png(const char *fn, int verbose) {  // io.c:72-191
  FILE *fh = fopen(fn, "rb");
  if (!fh) {
    if (verbose) printf("Could not open file '%s' for reading\n", fn);
    return NULL;
  }
  std::vector<unsigned char> f;
  unsigned char buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, fh)) > 0) f.insert(f.end(), buf, buf + got);
  fclose(fh);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (f.size() < 8) {
    if (verbose) printf("Could not read file '%s'\n", fn);
    return NULL;
  }
  if (memcmp(f.data(), sig, 8) != 0) {
    if (verbose) printf("'%s' is not a PNG file.\n", fn);
    return NULL;
  }
  uint32_t m = 0, n = 0;
  int bit_depth = 0, color_type = 0, interlace = 0;
  bool have_ihdr = false;
  std::vector<unsigned char> idat;
  for (size_t at = 8; at + 12 <= f.size();) {
    const uint32_t len = be32(&f[at]);
    if (at + 12 + (size_t)len > f.size()) break;
    const unsigned char *type = &f[at + 4], *data = &f[at + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      n = be32(data); m = be32(data + 4);
      bit_depth = data[8]; color_type = data[9]; interlace = data[12];
      have_ihdr = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    at += 12 + (size_t)len;
  }
  if (!have_ihdr) {
    if (verbose) printf("failed to initialise PNG read struct.\n");
    return NULL;
  }
  if (interlace != 0) {
    if (verbose) printf("interlaced images not supported\n");
    return NULL;
  }
  if (verbose)
    printf("reading %u x %u matrix (bit depth: %u, channels: %u, color type: %u, compression type: %u)\n", m, n, (unsigned)bit_depth, 1u,
           (unsigned)color_type, 0u);
  if (color_type != 0 && color_type != 3) {
    if (verbose) printf("only graycscale and palette colors are supported.\n");
    return NULL;
  }
  if (bit_depth != 1 || m > 0x7fffffffu || n > 0x7fffffffu) {
    if (verbose) printf("only one bit per pixel is supported.\n");
    return NULL;
  }
  const size_t rowbytes = ((size_t)n + 7) / 8;
  // the header of an untrusted file promises m * (rowbytes + 1) bytes of scanlines: believe it only if the IDAT stream could
  // inflate to that (deflate expands at most ~1032 : 1), instead of allocating up to 2^59 bytes and dying in bad_alloc
  const unsigned __int128 promised = (unsigned __int128)m * (rowbytes + 1);
  if (promised > (unsigned __int128)idat.size() * 1040u + 65536u) {
    if (verbose) printf("Could not read file '%s'\n", fn);
    return NULL;
  }
  std::vector<unsigned char> raw((size_t)promised);
  uLongf rlen = (uLongf)raw.size();
  if (!raw.empty() && (uncompress(raw.data(), &rlen, idat.data(), (uLong)idat.size()) != Z_OK || rlen != raw.size())) {
    if (verbose) printf("Could not read file '%s'\n", fn);
    return NULL;
  }
  mzd_t *A = new_matrix((rci_t)m, (rci_t)n);
  static unsigned char rev[256];
  static bool rev_ok = false;
  if (!rev_ok) {
    for (int x = 0; x < 256; ++x) { unsigned char r = 0; for (int b = 0; b < 8; ++b) r |= (unsigned char)(((x >> b) & 1) << (7 - b)); rev[x] = r; }
    rev_ok = true;
  }
  std::vector<unsigned char> prev(rowbytes, 0), cur(rowbytes, 0);
  for (uint32_t i = 0; i < m; ++i) {
    const unsigned char *src = raw.data() + (size_t)i * (rowbytes + 1);
    const int filter = *src++;
    for (size_t b = 0; b < rowbytes; ++b) {  // bytes per pixel for filtering = 1 at this depth
      const int a = b ? cur[b - 1] : 0, up = prev[b], ul = b ? prev[b - 1] : 0;
      int x = src[b];
      switch (filter) {
        case 1: x += a; break;
        case 2: x += up; break;
        case 3: x += (a + up) / 2; break;
        case 4: x += paeth(a, up, ul); break;
        default: break;
      }
      cur[b] = (unsigned char)x;
    }
    word *row = A->data + (int64_t)i * A->rowstride;
    for (size_t b = 0; b < rowbytes; ++b) {  // io.c:148-178: packswap, then the words complemented under the column mask
      const unsigned char byte = (unsigned char)~rev[cur[b]];
      row[b / 8] |= (word)byte << (8 * (b % 8));
    }
    if (A->width) row[A->width - 1] &= A->high_bitmask;
    prev.swap(cur);
  }
  return A;
}

}  // extern "C"
