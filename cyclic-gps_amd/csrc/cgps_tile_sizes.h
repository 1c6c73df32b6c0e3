// Host-side sizing of the fused solve + log-det pipeline (cgps_tile.h): rows per stage-1
// workgroup, workspace bytes.  Kept apart from the kernels so that cgps_workspace_bytes() does
// not have to parse them.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace cgps {

// per-block partial results: {sum x^2, sum log pivots, 1 + first failing row or 0, unused}
constexpr int PARTIAL_STRIDE = 4;

// TileCfg<T, d>::ROWS1 for a run-time d and scalar size s (workspace sizing)
inline int64_t tile_rows1(int d, size_t s) {
  if (d == 8) return s == 4 ? 128 * 256 / 4 : 64 * 256 / 4;
  if (s == 8 && d == 6) return 32 * 256 / 2;
  if (s == 8 && d == 7) return 16 * 128;
  return 16 * 256;
}
// Below ~2^19 rows the op is pure latency and stage 1's sequential chain of C - 1 eliminations
// per lane is most of it.  Small systems therefore take fewer rows per lane: the smallest C of
// {1, 4, 8, C_full} that keeps the grid within one workgroup per CU (more lanes, shorter chains,
// the same number of records for the final stage or fewer).
constexpr int64_t STAGE1_SMALL_TILES = 256;
inline int stage1_rows_per_lane(int64_t N, int c_full, int lanes) {
  // a few hundred rows: ONE workgroup (a second one costs an inter-workgroup hand-off, ~10 us, to save
  // three or seven eliminations of ~1 us per lane)
  if (c_full > 4 && N > lanes && N <= (int64_t)lanes * 4) return 4;
  if (c_full > 8 && N > lanes && N <= (int64_t)lanes * 8) return 8;
  if (c_full > 4 && N <= STAGE1_SMALL_TILES * lanes * 1) return 1;
  if (c_full > 4 && N <= STAGE1_SMALL_TILES * lanes * 4) return 4;
  if (c_full > 8 && N <= STAGE1_SMALL_TILES * lanes * 8) return 8;
  return c_full;
}
inline int64_t tile_cap(int64_t N, int d, size_t s) { return N / tile_rows1(d, s) + 2 + STAGE1_SMALL_TILES; }

inline size_t tile_ws_bytes(int64_t N, int d, size_t s) {
  const int64_t tiles = tile_cap(N, d, s);
  const size_t stride = (size_t)(((3 * d * d + 2 * d + 3) / 4) * 4) * s;
  const size_t pbytes = ((size_t)(2 * tiles + 8) * PARTIAL_STRIDE * sizeof(double) + 255) & ~(size_t)255;
  return pbytes + (((size_t)2 * (tiles + 2) * stride + 255) & ~(size_t)255);
}

}  // namespace cgps
