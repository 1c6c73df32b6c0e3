// Tile-fused multi-level kernels (placeholder until the fused path lands).
#pragma once
#include "cgps_math.h"
namespace cgps {
template <typename T, int D> constexpr bool tile_supported() { return false; }
inline size_t tile_ws_bytes(int64_t, int, size_t) { return 0; }
template <typename T, int D>
int run_tile_mahal_logdet(const T*, const T*, const T*, int64_t, char*, size_t, double*, int*, hipStream_t) { return -1; }
}  // namespace cgps
