// cgps_decompose.hip -- factor-emitting decompose
// One translation unit of libcgps (include/cgps.h); host code only decides sizes/offsets and
// enqueues kernels on the caller's stream: nothing here allocates, copies to the host or synchronises.
#include "cgps_host.h"
#include "cgps_tile.h"
#include "cgps_decomp_tile.h"
#include "cgps_decomp_lds.h"

using namespace cgps_host;

namespace {
// ---- fused (tiled) factorisation: cgps_decomp_tile.h (bulk passes) + cgps_decomp_lds.h (tail) ----
// (8 x 8 blocks: the in-LDS passes, four lanes per elimination, beat the one-wave-per-tile register passes
// already at 2^18 rows -- config 3: 1 490 -> 1 447 us; 4 x 4 fp64: no difference between 2^15 and 2^18)
template <int D> constexpr int64_t dec_small_rows() { return D == 8 ? 262144 : 32768; }   // at or below this many rows a pass is latency-bound
template <typename T, int D>
int run_decompose_tile(const T* Rs, const T* Os, int64_t N, T* Dp, T* Fp, T* Gp, char* ws, size_t ws_bytes, int* info,
                       hipStream_t st, const T* y = nullptr, T* xcrr = nullptr, T* ynext = nullptr, T* owedy = nullptr,
                       int* rhs_levels = nullptr) {
  // y != nullptr (cgps_decompose_solve): the first pass, when it is a bulk pass of DEC_LP levels, also carries the
  // forward substitution of y through its levels (decomp_tile_kernel<.., RHS = true>); *rhs_levels = levels done
  using RL = cgps::RecordLayout<T, D>;
  LevelWs w = level_ws(N, D, sizeof(T), true, false);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  // persistent waves: as many workgroups (one wave each) as the chip holds at once
  const size_t lds = (size_t)64 * D * D * sizeof(T);      // staging of the coalesced factor stores
  struct Caps { int64_t c[2]; };
  static PerDevice<Caps> caps;                          // per device, filled once (thread-safe)
  const size_t lds_small = cgps::decomp_lds_tile_bytes<T, D>();
  const Caps& grid_caps = caps.get([&](int dev) {
    int nb0 = 4, nb1 = 4;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, cgps::decomp_tile_kernel<T, D, false>, cgps::DEC_NT, lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, cgps::decomp_tile_kernel<T, D, true>, cgps::DEC_NT, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::decomp_lds_kernel<T, D, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::decomp_lds_kernel<T, D, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    const int cus = device_cus(dev);
    return Caps{{(int64_t)cus * (nb0 > 0 ? nb0 : 1), (int64_t)cus * (nb1 > 0 ? nb1 : 1)}};
  });
  const int64_t* grid_cap = grid_caps.c;
  (void)hipMemsetAsync(info, 0, sizeof(int), st);
  T* recs[2] = {reinterpret_cast<T*>(ws + w.a_off), reinterpret_cast<T*>(ws + w.b_off)};   // records of a pass
  const T* rin = nullptr;
  int64_t n_rec = 0;
  int lvl = 0, p = 0, spt_in = 1;
  while (lvl < L.nlevels) {
    const int64_t rows = L.ms[lvl];
    const int remaining = L.nlevels - lvl;
    if (rows <= dec_small_rows<D>()) {
      // latency-bound tail (or a small system): 256-row tiles in LDS, 8 levels per launch, four
      // waves per elimination (cgps_decomp_lds.h); one record per tile
      const int64_t g = (rows + cgps::DECL_TS - 1) / cgps::DECL_TS;
      const int nl = (g == 1) ? remaining : cgps::DECL_LP;                 // <= DECL_LP + 1
      cgps::DecompLevelsL dl;
      dl.nlev = nl;
      for (int j = 0; j < cgps::DECL_MAXLEV; ++j) {
        const int l = lvl + j < L.nlevels ? lvl + j : L.nlevels - 1;
        dl.offD[j] = L.offD[l]; dl.offF[j] = L.offF[l]; dl.offG[j] = L.offG[l];
      }
      T* rout = recs[p & 1];
      if (p == 0)
        hipLaunchKernelGGL((cgps::decomp_lds_kernel<T, D, false>), dim3((unsigned)g), dim3(cgps::DECL_NT), lds_small, st,
                           Rs, Os, rows, (int64_t)0, 1, dl, lvl, Dp, Fp, Gp, rout, info);
      else
        hipLaunchKernelGGL((cgps::decomp_lds_kernel<T, D, true>), dim3((unsigned)g), dim3(cgps::DECL_NT), lds_small, st,
                           rin, (const T*)nullptr, rows, n_rec, spt_in, dl, lvl, Dp, Fp, Gp, rout, info);
      rin = rout;
      n_rec = g;
      spt_in = 1;
      lvl += nl;
      ++p;
      continue;
    }
    const int64_t g = (rows + cgps::DEC_TS - 1) / cgps::DEC_TS;
    const bool top = g == 1;                                               // one tile takes it to the end
    // many tiles: a few levels per pass keep the lanes busy; few tiles: all levels of a tile
    const int nl = top ? remaining : (g >= cgps::DEC_FEW_TILES ? cgps::DEC_LP : cgps::DEC_TS_LOG2);   // <= DEC_MAXLEV
    cgps::DecompLevels dl;
    dl.nlev = nl;
    for (int j = 0; j < cgps::DEC_MAXLEV; ++j) {
      const int l = lvl + j < L.nlevels ? lvl + j : L.nlevels - 1;
      dl.offD[j] = L.offD[l]; dl.offF[j] = L.offF[l]; dl.offG[j] = L.offG[l];
    }
    T* rout = top ? nullptr : recs[p & 1];
    const int64_t cap = grid_cap[p == 0 ? 0 : 1];
    const unsigned grid = (unsigned)(g < cap ? g : cap);
    if (p == 0 && y != nullptr && !top && nl == cgps::DEC_LP) {
      hipLaunchKernelGGL((cgps::decomp_tile_kernel<T, D, false, true>), dim3(grid), dim3(cgps::DEC_NT),
                         lds + (size_t)(D * D + D) * sizeof(T), st, Rs, Os,
                         rows, (int64_t)0, 1, dl, lvl, Dp, Fp, Gp, rout, info, y, xcrr, ynext, owedy);
      const int64_t pairs = (g - 1) * D;
      if (pairs > 0)
        hipLaunchKernelGGL((cgps::decomp_rhs_fixup_kernel<T, D>), dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, ynext,
                           (const T*)owedy, g, (int)(cgps::DEC_TS >> nl), L.ms[nl]);
      if (rhs_levels) *rhs_levels = nl;
    } else if (p == 0)
      hipLaunchKernelGGL((cgps::decomp_tile_kernel<T, D, false>), dim3(grid), dim3(cgps::DEC_NT), lds, st, Rs, Os,
                         rows, (int64_t)0, 1, dl, lvl, Dp, Fp, Gp, rout, info);
    else
      hipLaunchKernelGGL((cgps::decomp_tile_kernel<T, D, true>), dim3(grid), dim3(cgps::DEC_NT), lds, st, rin,
                         (const T*)nullptr, rows, n_rec, spt_in, dl, lvl, Dp, Fp, Gp, rout, info);
    rin = rout;
    // every tile leaves DEC_TS >> nl records, the last one what survives of it, at least one
    {
      const int64_t spt = cgps::DEC_TS >> nl;
      const int64_t last = (rows - (g - 1) * cgps::DEC_TS) >> nl;
      n_rec = (g - 1) * spt + (last > 0 ? last : 1);
      spt_in = (int)(spt > 0 ? spt : 1);
    }
    lvl += nl;
    ++p;
  }
  (void)RL::STRIDE;
  return check_launch("decompose (tiled)");
}

// ---- blocks whose 256-row tile does not fit the LDS (fp64 d = 6, 7, 8): passes of cgps_decomp_lds.h with 64-row
// tiles, six levels per launch (2^20 rows: 2^20 -> 2^14 -> 2^8 -> 4 -> done; level by level: 21 launches, every
// level's rows written and read back).  Measured (prof_case --op decompose, 2^20 rows, level by level -> this):
// d = 6 1 084 -> 836 us.  For d = 7, 8 the tile passes are bound by their two workgroups per CU (65 KB of LDS per
// 8 x 8 tile): 1 588 -> 1 931 us and 1 836 -> 2 278 us at 2^20 rows, but 263 -> 170 / 330 -> 164 us at 2^14 and
// 334 -> 269 / 403 -> 276 us at 2^16 (a tie at 2^18) -- so those sizes run their levels of more than 2^17 rows one
// launch per level (level_kernel, full occupancy) and hand over to the tile passes below that. --------------------
template <typename T, int D>
int run_decompose_lds_only(const T* Rs, const T* Os, int64_t N, T* Dp, T* Fp, T* Gp, char* ws, size_t ws_bytes, int* info,
                           hipStream_t st) {
  constexpr int LP = cgps::decomp_lds_lp<T, D>(), TS = 1 << LP;
  LevelWs w = level_ws(N, D, sizeof(T), true, false);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  const size_t lds = cgps::decomp_lds_tile_bytes<T, D, LP>();
  struct Done { int ok; };
  static PerDevice<Done> attr;
  (void)attr.get([&](int) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::decomp_lds_kernel<T, D, false, LP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::decomp_lds_kernel<T, D, true, LP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return Done{1};
  });
  (void)hipMemsetAsync(info, 0, sizeof(int), st);
  T* recs[2] = {reinterpret_cast<T*>(ws + w.a_off), reinterpret_cast<T*>(ws + w.b_off)};
  const T* rin = nullptr;
  int64_t n_rec = 0;
  int lvl = 0, p = 0;
  // the levels too large for the tile passes (d = 7, 8), one launch each; their output alternates between the two
  // level buffers, which are also the record buffers of the tile passes: the first tile pass writes the other one
  constexpr int64_t LDS_MAX_ROWS = (D >= 7) ? (int64_t)131072 : ((int64_t)1 << 62);
  LevelBuf<T> bufs[2] = {carve<T>(ws + w.a_off, w.capA, D, true, false), carve<T>(ws + w.b_off, w.capB, D, true, false)};
  double* partial = reinterpret_cast<double*>(ws + w.partial_off);
  const T *Rl = Rs, *Ol = Os;
  int64_t pb = 0;
  int out_idx = 0;
  while (lvl < L.nlevels - 1 && L.ms[lvl] > LDS_MAX_ROWS) {
    const int64_t n = L.ms[lvl], nb = level_blocks(n);
    LevelBuf<T>& nx = bufs[lvl & 1];
    hipLaunchKernelGGL((cgps::level_kernel<T, D, true, false>), dim3((unsigned)nb), dim3(cgps::LEVEL_THREADS), 0, st, Rl, Ol,
                       (const T*)nullptr, n, lvl, Dp + L.offD[lvl] * D * D, Fp + L.offF[lvl] * D * D, Gp + L.offG[lvl] * D * D,
                       (T*)nullptr, nx.R, nx.O, nx.y, partial + 2 * pb, info);
    pb += nb;
    Rl = nx.R;
    Ol = nx.O;
    out_idx = (lvl & 1) ^ 1;
    ++lvl;
  }
  while (lvl < L.nlevels) {
    const int64_t rows = L.ms[lvl];
    const int remaining = L.nlevels - lvl;
    const int64_t g = (rows + TS - 1) / TS;
    const int nl = (g == 1) ? remaining : LP;                              // <= LP + 1
    cgps::DecompLevelsL dl;
    dl.nlev = nl;
    for (int j = 0; j < cgps::DECL_MAXLEV; ++j) {
      const int l = lvl + j < L.nlevels ? lvl + j : L.nlevels - 1;
      dl.offD[j] = L.offD[l]; dl.offF[j] = L.offF[l]; dl.offG[j] = L.offG[l];
    }
    T* rout = recs[out_idx];
    out_idx ^= 1;
    if (p == 0)
      hipLaunchKernelGGL((cgps::decomp_lds_kernel<T, D, false, LP>), dim3((unsigned)g), dim3(cgps::DECL_NT), lds, st, Rl, Ol,
                         rows, (int64_t)0, 1, dl, lvl, Dp, Fp, Gp, rout, info);
    else
      hipLaunchKernelGGL((cgps::decomp_lds_kernel<T, D, true, LP>), dim3((unsigned)g), dim3(cgps::DECL_NT), lds, st, rin,
                         (const T*)nullptr, rows, n_rec, 1, dl, lvl, Dp, Fp, Gp, rout, info);
    rin = rout;
    n_rec = g;
    lvl += nl;
    ++p;
  }
  return check_launch("decompose (in-LDS passes)");
}
}  // namespace

extern "C" {

int cgps_decompose_solve(const void* Rs, const void* Os, const void* y, int64_t N, int d, int dtype, void* Dp, void* Fp,
                         void* Gp, void* xcrr, void* x, void* ws, size_t ws_bytes, int* info, void* stream) {
  if (bad_common(N, d) || !Rs || (N > 1 && !Os) || !y || !Dp || !Fp || !Gp || !xcrr || !x || !ws || !info)
    return fail(CGPS_ERR_ARG, "cgps_decompose_solve: null pointer or N < 1");
  size_t need = 0;
  if (int rc = cgps_workspace_bytes(N, d, dtype, CGPS_OP_DECOMPOSE_SOLVE, &need)) return rc;
  if (ws_bytes < need) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, need);
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    // workspace: [what decompose / the sweeps need, one after the other] [ynext: (N/8 + 16) rows] [owedy: N/128 + 2 rows]
    const size_t tail = decompose_solve_tail_bytes(N, D, sizeof(T));
    const size_t main_bytes = ws_bytes - tail;
    T* ynext = reinterpret_cast<T*>((char*)ws + main_bytes);
    T* owedy = ynext + (size_t)(N / 8 + 16) * D;
    int rhs_levels = 0, rc;
    if constexpr (cgps::tile_fits_256<T, D>()) {
      if (!levelwise_solve_requested())
        rc = run_decompose_tile<T, D>((const T*)Rs, (const T*)Os, N, (T*)Dp, (T*)Fp, (T*)Gp, (char*)ws, main_bytes, info,
                                      (hipStream_t)stream, (const T*)y, (T*)xcrr, ynext, owedy, &rhs_levels);
      else
        rc = cgps_decompose(Rs, Os, N, d, dtype, Dp, Fp, Gp, ws, main_bytes, info, stream);
    } else {
      rc = cgps_decompose(Rs, Os, N, d, dtype, Dp, Fp, Gp, ws, main_bytes, info, stream);
    }
    if (rc != CGPS_OK) return rc;
    if (rhs_levels > 0) {
      // the forward sweep goes on at level rhs_levels: the packed factor of levels >= l IS the packed factor of the
      // (N >> l)-row system those levels reduce (level sizes halve with the same rounding), so the stored-factor
      // sweep runs on that sub-system with the surviving rows' right-hand side
      Layout L;
      make_layout(N, L);
      const int l = rhs_levels;
      rc = cgps_halfsolve((const T*)Dp + L.offD[l] * D * D, (const T*)Fp + L.offF[l] * D * D, (const T*)Gp + L.offG[l] * D * D,
                          L.ms[l], d, dtype, 1, ynext, (T*)xcrr + L.offD[l] * D, ws, main_bytes, nullptr, stream);
    } else {
      rc = cgps_halfsolve(Dp, Fp, Gp, N, d, dtype, 1, y, xcrr, ws, main_bytes, nullptr, stream);
    }
    if (rc != CGPS_OK) return rc;
    return cgps_backsolve(Dp, Fp, Gp, N, d, dtype, 1, xcrr, x, ws, main_bytes, stream);
  });
}

int cgps_decompose(const void* Rs, const void* Os, int64_t N, int d, int dtype, void* Dp, void* Fp, void* Gp, void* ws,
                   size_t ws_bytes, int* info, void* stream) {
  if (bad_common(N, d) || !Rs || (N > 1 && !Os) || !Dp || !Fp || !Gp || !ws || !info)
    return fail(CGPS_ERR_ARG, "cgps_decompose: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    // tiled form for every block size whose 256-row tile fits the LDS (no register spills up to
    // 8x8 fp32 / 5x5 fp64); larger blocks go level by level
    if constexpr (cgps::tile_fits_256<T, D>()) {
      if (!levelwise_solve_requested())
        return run_decompose_tile<T, D>((const T*)Rs, (const T*)Os, N, (T*)Dp, (T*)Fp, (T*)Gp, (char*)ws, ws_bytes,
                                        info, (hipStream_t)stream);
    } else {
      if (!levelwise_solve_requested())
        return run_decompose_lds_only<T, D>((const T*)Rs, (const T*)Os, N, (T*)Dp, (T*)Fp, (T*)Gp, (char*)ws, ws_bytes,
                                            info, (hipStream_t)stream);
    }
    return run_levelwise<T, D>((const T*)Rs, (const T*)Os, nullptr, N, (T*)Dp, (T*)Fp, (T*)Gp, nullptr, (char*)ws,
                               ws_bytes, nullptr, info, (hipStream_t)stream);
  });
}

}  // extern "C"
