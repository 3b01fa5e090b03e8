// gf2_common.h -- shared device/host types for the MI355X GF(2) multiply engine.
//
// Data layout in HBM (identical to M4RI's host layout, /root/reference m4ri/mzd.h:68-139):
//   a matrix is `nrows` rows of `rowstride` 64-bit words; bit (r, c) is
//   (data[r*rowstride + c/64] >> (c%64)) & 1, i.e. LSB = lowest column.
//   Device temporaries keep ALL bits at column index >= ncols equal to zero ("zero excess"),
//   which is what lets every kernel work on whole words without per-bit masks.
#pragma once
#include <stdint.h>
#include <stddef.h>

typedef uint64_t word;

// A device-resident (sub)matrix view: base pointer + geometry, all in 64-bit words.
struct DMat {
  word *p;            // device pointer to word (0,0) of the view
  int64_t nrows;      // rows
  int64_t ncols;      // columns (bits)
  int64_t stride;     // words between consecutive rows
};

static inline int64_t words_of(int64_t ncols) { return (ncols + 63) >> 6; }

static inline DMat dview(const DMat &M, int64_t r0, int64_t c0_bits, int64_t nr, int64_t nc_bits) {
  DMat V;
  V.p      = M.p + r0 * M.stride + (c0_bits >> 6);
  V.nrows  = nr;
  V.ncols  = nc_bits;
  V.stride = M.stride;
  return V;
}

// ---- M4RM leaf geometry (m4rm_leaf.hip) --------------------------------------------------------
// One workgroup owns a C tile of LEAF_ROWS x LEAF_COLS bits held in VGPRs for the whole inner loop.
#define LEAF_K        8                    // bits per Gray/lookup table index (2^8 entries)
#define LEAF_NT       2                    // tables resident in LDS per stage (2 x 64 KiB)
#define LEAF_STAGE    (LEAF_K * LEAF_NT)   // inner-dimension bits consumed per stage
#define LEAF_TW       32                   // tile width in words (2048 columns, 256 B per entry)
#define LEAF_THREADS  512                  // 8 waves, 2 per SIMD

// Batched leaf launch descriptor.  Product b (0 <= b < batch) is
//   C_b (^)= A_b * B_b,  X_b = X + b * x_bs  (word offsets), all the same shape.
struct LeafArgs {
  const word *A; const word *B; word *C;
  const uint32_t *Apk;                    // packed, chunk-major copy of A (generations 2-4), set by their launchers
  int64_t apk_stride, apk_bs;             // dwords between chunks (= padded rows) / batch members of Apk
  int64_t a_stride, b_stride, c_stride;   // words between rows
  int64_t a_bs, b_bs, c_bs;               // words between consecutive batch members
  int32_t m, l, n;                        // C is m x n, inner dimension l (bits)
  int32_t wn;                             // words_of(n)
  int32_t tiles_m, tiles_n;               // tile grid per product
  int32_t ksplit;                         // inner-dimension splits (>=1); >1 => atomic XOR output
  int32_t stages_per_split;               // two-phase kernel: LEAF_STAGE-bit stages per split
  int32_t chunks_per_split;               // double-buffered kernel: 32-bit A chunks per split
  int32_t batch;
  int32_t mode;                           // 0: C = A*B (plain store), 1: C ^= A*B (no-return atomic xor),
                                          // 2 (generation 4): every (tile, split) stores its whole tile into its
                                          //    own slab of Cpart; gf2_launch_reduce_partials folds them into C
  // generation 4 only: the launch covers the tiles [tile_base, tile_base + tile_count) of the
  // batch's linear tile order (tile_m fastest, then tile_n, then batch member); tile_count == 0
  // means all of them.  Lets the engine run the full rounds and a finer-split tail as two launches.
  int64_t tile_base, tile_count;
  word *Cpart;                            // mode 2: slab (tile - tile_base) * ksplit + split, LEAF_PART_WORDS words each
};
#define LEAF_PART_WORDS 32768             // one generation-4 tile: 4096 rows x 8 words, dense
