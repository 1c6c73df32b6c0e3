// cgps_inverse.hip -- inverse_blocks and the adjoint pass of mahal_and_det
// One translation unit of libcgps (include/cgps.h); host code only decides sizes/offsets and
// enqueues kernels on the caller's stream: nothing here allocates, copies to the host or synchronises.
#include "cgps_host.h"
#include "cgps_tile.h"
#include "cgps_inverse_tile.h"
#include "cgps_inverse_quad.h"

using namespace cgps_host;

namespace {
#ifndef CGPS_INV_FUSED_MAX_BLOCK
#define CGPS_INV_FUSED_MAX_BLOCK 200
#endif
inline bool inverse_deep_enabled() {        // CGPS_NO_DEEP_INVERSE=1: one launch per coarse level (A/B timing)
  static const bool on = [] { const char* e = getenv("CGPS_NO_DEEP_INVERSE"); return !(e && e[0] == '1'); }();
  return on;
}
inline bool inverse_quad_enabled() {        // CGPS_NO_QUAD_INVERSE=1: 8 x 8 blocks one lane per row (A/B timing)
  static const bool on = [] { const char* e = getenv("CGPS_NO_QUAD_INVERSE"); return !(e && e[0] == '1'); }();
  return on;
}
inline bool inverse_lds_enabled() {         // CGPS_NO_LDS_INVERSE=1: large blocks one launch per level (A/B timing)
  static const bool on = [] { const char* e = getenv("CGPS_NO_LDS_INVERSE"); return !(e && e[0] == '1'); }();
  return on;
}
constexpr int64_t INV_FUSED_MIN_ROWS = 1024;   // a fused inverse pass must produce at least this many rows
template <typename T, int D>
int run_inverse(const T* Dp, const T* Fp, const T* Gp, int64_t N, T* Sd, T* So, char* ws, size_t ws_bytes,
                hipStream_t st) {
  const int64_t cap = N / 2 + 1;
  const size_t one = align_up((size_t)2 * D * D * sizeof(T) * cap);
  if (ws_bytes < 2 * one) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, 2 * one);
  Layout L;
  make_layout(N, L);
  T* bufs[2] = {reinterpret_cast<T*>(ws), reinterpret_cast<T*>(ws + one)};
  const T *Sdc = nullptr, *Soc = nullptr;
  // The coarse levels one launch each (latency-bound, little data); once a level that is a
  // multiple of INV_LP above level 0 is reached and the rows get many, INV_LP levels per launch
  // (cgps_inverse_tile.h): those passes read 1/8 of what they write instead of ping-ponging every
  // level's Sigma through HBM.
  // blocks up to CGPS_INV_FUSED_MAX_BLOCK bytes keep the tile's Sigma in registers, larger ones in LDS
  // 8 x 8 blocks: four lanes per row (cgps_inverse_quad.h); fp64 d = 8 one lane per row spills in either form
  const bool IN_QUAD = D == 8 && inverse_quad_enabled();
  const bool FUSED = (size_t)D * D * sizeof(T) <= 400 || IN_QUAD;
  static const size_t reg_max = [] {                  // CGPS_INV_REG_MAX_BLOCK=<bytes>: A/B timing of the two forms
    const char* e = getenv("CGPS_INV_REG_MAX_BLOCK");
    return e ? (size_t)atoi(e) : (size_t)CGPS_INV_FUSED_MAX_BLOCK;
  }();
  const bool IN_LDS = (size_t)D * D * sizeof(T) > reg_max;
  size_t lds = IN_LDS ? cgps::inverse_tile_lds_bytes<T, D>() : (size_t)64 * D * D * sizeof(T);
  auto* tile_kernel = IN_LDS ? &cgps::inverse_tile_lds_kernel<T, D> : &cgps::inverse_tile_kernel<T, D>;
  int tile_threads = cgps::INV_NT;
  if constexpr (D == 8) {
    if (IN_QUAD) {
      lds = cgps::inverse_quad_lds_bytes<T>();
      if constexpr (sizeof(T) == 4) tile_kernel = &cgps::inverse_tile_quad_kernel<T, 64>;
      else tile_kernel = &cgps::inverse_tile_quad_wg_kernel<T>;
      tile_threads = cgps::inverse_quad_threads<T>();
    }
  }
  static PerDevice<int> grid_caps[3];                 // persistent workgroups: what this device holds at once
  const int grid_cap = grid_caps[IN_QUAD ? 2 : (IN_LDS ? 1 : 0)].get([&](int dev) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nb = 2;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tile_kernel, tile_threads, lds);
    return device_cus(dev) * (nb > 0 ? nb : 1);
  });
  int p = 0;
  int l_start = L.nlevels - 1;
  if constexpr (cgps::inverse_deep_supported<T, D>()) {
    // the coarse end in ONE launch (inverse_deep_kernel): from the single row of the coarsest level down
    // to the finest level of at most INVD_TS rows at which the three-levels-per-launch passes can take
    // over (a multiple of INV_LP), or to level 0 of a small system
    if (inverse_deep_enabled()) {
      constexpr int INVD_TS_ = 1 << cgps::invd_tsl<T, D>();
      int lf = -1;
      for (int l = 0; l < L.nlevels; ++l)
        if (L.ms[l] <= INVD_TS_ && (l % cgps::INV_LP == 0 || !FUSED)) { lf = l; break; }
      if (lf < 0)
        for (int l = 0; l < L.nlevels; ++l)
          if (L.ms[l] <= INVD_TS_) { lf = l; break; }
      if (lf >= 0 && L.nlevels - lf <= cgps::INVD_MAXLEV) {
        static PerDevice<int> attr;
        attr.get([](int) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::inverse_deep_kernel<T, D>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)cgps::inverse_deep_lds_bytes<T, D>());
          return 1;
        });
        cgps::InverseDeepLevels dl;
        dl.nlev = L.nlevels - lf;
        for (int j = 0; j < cgps::INVD_MAXLEV; ++j) {
          const int l = lf + j < L.nlevels ? lf + j : L.nlevels - 1;
          dl.offD[j] = L.offD[l]; dl.offF[j] = L.offF[l]; dl.offG[j] = L.offG[l];
        }
        T* od = (lf == 0) ? Sd : bufs[p];
        T* oo = (lf == 0) ? So : bufs[p] + cap * D * D;
        const size_t lds_deep = cgps::inverse_deep_lds_bytes<T, D>();
        hipLaunchKernelGGL((cgps::inverse_deep_kernel<T, D>), dim3(1), dim3(INVD_TS_ / 2), lds_deep, st, Dp, Fp, Gp, dl,
                           (int)L.ms[lf], od, oo);
        Sdc = od; Soc = oo; p ^= 1;
        l_start = lf - 1;
      }
    }
  }
  for (int l = l_start; l >= 0;) {
    const int have = l + 1;                               // Sdc / Soc hold Sigma of this level
    if (FUSED && (IN_QUAD || !IN_LDS || inverse_lds_enabled()) && Sdc != nullptr && have % cgps::INV_LP == 0 && L.ms[have] >= 1 &&
        L.ms[have - cgps::INV_LP] >= INV_FUSED_MIN_ROWS) {
      const int lf = have - cgps::INV_LP;
      const int64_t n = L.ms[lf], tiles = (n + cgps::INV_TS - 1) / cgps::INV_TS;
      cgps::InverseLevels lv;
      for (int t = 0; t < cgps::INV_LP; ++t) {
        lv.offD[t] = L.offD[lf + t]; lv.offF[t] = L.offF[lf + t]; lv.offG[t] = L.offG[lf + t];
      }
      T* od = (lf == 0) ? Sd : bufs[p];
      T* oo = (lf == 0) ? So : bufs[p] + cap * D * D;
      const int64_t grid = tiles < grid_cap ? tiles : grid_cap;
      hipLaunchKernelGGL(tile_kernel, dim3((unsigned)grid), dim3(tile_threads), lds, st, Dp, Fp, Gp, lv, Sdc, Soc, n, od, oo);
      Sdc = od; Soc = oo; p ^= 1;
      l = lf - 1;
      continue;
    }
    const int64_t n = L.ms[l], nb = level_blocks(n);
    T* od = (l == 0) ? Sd : bufs[p];
    T* oo = (l == 0) ? So : bufs[p] + cap * D * D;
    hipLaunchKernelGGL((cgps::inverse_level_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::LEVEL_THREADS), 0, st,
                       Dp + L.offD[l] * D * D, Fp + L.offF[l] * D * D, Gp + L.offG[l] * D * D, Sdc, Soc, n, od, oo);
    Sdc = od; Soc = oo; p ^= 1;
    --l;
  }
  return check_launch("inverse_blocks");
}
}  // namespace

extern "C" {

int cgps_inverse_blocks(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, void* Sd, void* So,
                        void* ws, size_t ws_bytes, void* stream) {
  if (bad_common(N, d) || !Dp || !Fp || !Gp || !Sd || (N > 1 && !So) || !ws)
    return fail(CGPS_ERR_ARG, "cgps_inverse_blocks: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    return run_inverse<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (T*)Sd, (T*)So, (char*)ws, ws_bytes,
                             (hipStream_t)stream);
  });
}

int cgps_mahal_logdet_adjoint(void* Sd, void* So, const void* w, int64_t N, int d, int dtype, const void* gm,
                              const void* gl, void* stream) {
  if (bad_common(N, d) || !Sd || (N > 1 && !So) || !w || !gm || !gl)
    return fail(CGPS_ERR_ARG, "cgps_mahal_logdet_adjoint: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    const int64_t total = (2 * N - 1) * D * D;
    int64_t nb = (total + cgps::ADJ_THREADS - 1) / cgps::ADJ_THREADS;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL((cgps::mahal_logdet_adjoint_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::ADJ_THREADS), 0,
                       (hipStream_t)stream, (T*)Sd, (T*)So, (const T*)w, N, (const T*)gm, (const T*)gl);
    return check_launch("mahal_logdet_adjoint");
  });
}

}  // extern "C"
