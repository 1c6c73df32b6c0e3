// cgps_solve.hip -- halfsolve / backhalfsolve / solve on a stored factor
// One translation unit of libcgps (include/cgps.h); host code only decides sizes/offsets and
// enqueues kernels on the caller's stream: nothing here allocates, copies to the host or synchronises.
#include "cgps_host.h"
#include "cgps_tile.h"
#include "cgps_solve_tile.h"
#include "cgps_solve_tile_m.h"

using namespace cgps_host;

namespace {
// ---- fused (tiled) substitution sweeps: cgps_solve_tile.h ---------------------------------------
struct SolvePasses {
  int np;
  int ts[8];                 // rows per tile of the pass
  bool deep[8];              // latency-bound form (every factor block requested up front)
  int first[8];
  cgps::PassLevels lv[8];
  int64_t rows[8];
};

// deep_tiles: passes of at most this many tiles take the latency-bound kernels (0: never): one
// 1024-row tile when the rows fit it, 512-row tiles (twice the CUs pulling the factor) otherwise
// ts_deep / lp_deep (panel sweeps): passes of at most `deep_tiles_m` such tiles take them, in the latency-bound form
void make_passes(const Layout& L, SolvePasses& P, int wide_lp, int ts_in = cgps::SOLVE_TS, int lp_in = cgps::SOLVE_LP,
                 int64_t deep_tiles = 0, int ts_deep = 0, int lp_deep = 0, int64_t deep_tiles_m = 0) {
  P.np = 0;
  int lvl = 0;
  while (lvl < L.nlevels) {
    const int64_t rows = L.ms[lvl];
    const int remaining = L.nlevels - lvl;
    int ts = ts_in, lp = lp_in;
    bool deep = false;
    if (deep_tiles > 0) {
      if (rows <= ts_in) deep = true;
      else if ((rows + ts_in / 2 - 1) / (ts_in / 2) <= deep_tiles) { deep = true; ts = ts_in / 2; lp = lp_in - 1; }
    }
    if (ts_deep > 0 && rows < cgps::SOLVE_WIDE_ROWS && (rows + ts_deep - 1) / ts_deep <= deep_tiles_m) {
      deep = true; ts = ts_deep; lp = lp_deep;
    }
    P.ts[P.np] = ts;
    P.deep[P.np] = deep;
    // many tiles: a few levels per pass (every lane busy, few barrier-separated latency
    // exposures, the factor still read once); few tiles: all ten levels of a tile
    const int nl = (rows <= ts) ? remaining : (rows >= cgps::SOLVE_WIDE_ROWS ? wide_lp : lp);   // <= lp + 1
    cgps::PassLevels& pl = P.lv[P.np];
    pl.nlev = nl;
    pl.endD = L.offD[lvl + nl];
    pl.endF = L.offF[lvl + nl < L.nlevels ? lvl + nl : L.nlevels - 1];
    pl.endG = L.offG[lvl + nl < L.nlevels ? lvl + nl : L.nlevels - 1];
    for (int j = 0; j < cgps::SOLVE_MAXLEV; ++j) {
      const int l = lvl + j < L.nlevels ? lvl + j : L.nlevels - 1;
      pl.offD[j] = L.offD[l]; pl.offF[j] = L.offF[l]; pl.offG[j] = L.offG[l];
      pl.m[j] = lvl + j < L.nlevels ? L.ms[l] : 0;
    }
    P.first[P.np] = lvl;
    P.rows[P.np] = rows;
    ++P.np;
    lvl += nl;
  }
}

// passes of at most one tile per CU take the "deep" kernels (CGPS_NO_DEEP_SOLVE=1: the level-by-level loads, for A/B timing)
inline int64_t deep_tiles_max() {
  static PerDevice<int64_t> cus;
  return cus.get([](int dev) { return (int64_t)device_cus(dev); });
}
inline bool fused_top_enabled() {           // CGPS_NO_FUSED_TOP=1: separate forward / backward top passes
  static const bool on = [] { const char* e = getenv("CGPS_NO_FUSED_TOP"); return !(e && e[0] == '1'); }();
  return on;
}
inline bool deep_solve_enabled() {
  static const bool on = [] { const char* e = getenv("CGPS_NO_DEEP_SOLVE"); return !(e && e[0] == '1'); }();
  return on;
}

template <typename T, int D>
int64_t deep_tiles_for() {
  if constexpr (cgps::solve_deep_supported<T, D>()) return deep_solve_enabled() ? deep_tiles_max() : 0;
  else return 0;
}

template <typename T, int D>
void solve_tile_attributes() {
  static PerDevice<int> done;
  done.get([](int) {
  const int lds = (int)cgps::solve_lds_bytes<T, D>();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::halfsolve_tile_kernel<T, D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::backsolve_tile_kernel<T, D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if constexpr (cgps::solve_deep_supported<T, D>()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::halfsolve_deep_kernel<T, D, cgps::SOLVE_LP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::backsolve_deep_kernel<T, D, cgps::SOLVE_LP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::halfsolve_deep_kernel<T, D, cgps::SOLVE_LP - 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::backsolve_deep_kernel<T, D, cgps::SOLVE_LP - 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::solve_top_kernel<T, D, cgps::SOLVE_LP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::solve_top_kernel<T, D, cgps::SOLVE_LP - 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  return 1;
  });
}

template <typename T, int D>
int run_halfsolve_tile(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* y0, T* xcrr, char* ws,
                       size_t ws_bytes, double* mahal_out, hipStream_t st, T* const* fused_top_bufs = nullptr,
                       bool* fused_top = nullptr) {
  LevelWs w = level_ws(N, D, sizeof(T), false, true);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  SolvePasses P;
  make_passes(L, P, cgps::SOLVE_LP_WIDE, cgps::SOLVE_TS, cgps::SOLVE_LP, deep_tiles_for<T, D>());
  solve_tile_attributes<T, D>();
  double* partial = reinterpret_cast<double*>(ws + w.partial_off);
  T* bufs[2] = {reinterpret_cast<T*>(ws + w.a_off), reinterpret_cast<T*>(ws + w.b_off)};
  const size_t lds = cgps::solve_lds_bytes<T, D>();
  const T* y = y0;
  const T* owed_in = nullptr;
  int64_t n_owed = 0, pb = 0;
  int spt_in = 1;
  for (int p = 0; p < P.np; ++p) {
    const int64_t n = P.rows[p], g = (n + P.ts[p] - 1) / P.ts[p];
    const bool more = (p + 1 < P.np);
    const int64_t nsurv = n >> P.lv[p].nlev;           // rows of the next pass
    T* yout = more ? bufs[p & 1] : nullptr;            // [nsurv][D] surviving rows, then [g][D] owed vectors
    T* owed_out = more ? bufs[p & 1] + (nsurv + 1) * D : nullptr;
    bool launched = false;
    if constexpr (cgps::solve_deep_supported<T, D>()) {
      // solve(): the single-tile top pass runs its forward and its backward sweep in ONE launch
      // (solve_top_kernel) and leaves the solution of its rows where the backward sweep expects it
      if (fused_top_bufs != nullptr && !more && g == 1 && P.deep[p] && fused_top_enabled()) {
        T* xtop = (p == 0) ? fused_top_bufs[2] : fused_top_bufs[p & 1];
        if (P.ts[p] == cgps::SOLVE_TS)
          hipLaunchKernelGGL((cgps::solve_top_kernel<T, D, cgps::SOLVE_LP>), dim3(1), dim3(cgps::SOLVE_NT), lds, st, Dp, Fp, Gp,
                             P.lv[p], owed_in, n_owed, spt_in, y, n, xcrr, xtop, partial + 2 * pb);
        else
          hipLaunchKernelGGL((cgps::solve_top_kernel<T, D, cgps::SOLVE_LP - 1>), dim3(1), dim3(cgps::SOLVE_NT / 2), lds, st, Dp,
                             Fp, Gp, P.lv[p], owed_in, n_owed, spt_in, y, n, xcrr, xtop, partial + 2 * pb);
        launched = true;
        *fused_top = true;
      }
    }
    if constexpr (cgps::solve_deep_supported<T, D>()) {
      // few tiles (at most one per CU): the latency-bound form with every factor block requested up front
      if (launched) {
      } else if (P.deep[p] && P.ts[p] == cgps::SOLVE_TS) {
        hipLaunchKernelGGL((cgps::halfsolve_deep_kernel<T, D, cgps::SOLVE_LP>), dim3((unsigned)g), dim3(cgps::SOLVE_NT), lds, st,
                           Dp, Fp, Gp, P.lv[p], owed_in, n_owed, spt_in, y, n, xcrr, yout, owed_out, partial + 2 * pb);
        launched = true;
      } else if (P.deep[p]) {
        hipLaunchKernelGGL((cgps::halfsolve_deep_kernel<T, D, cgps::SOLVE_LP - 1>), dim3((unsigned)g),
                           dim3(cgps::SOLVE_NT / 2), lds, st, Dp, Fp, Gp, P.lv[p], owed_in, n_owed, spt_in, y, n, xcrr, yout,
                           owed_out, partial + 2 * pb);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((cgps::halfsolve_tile_kernel<T, D>), dim3((unsigned)g), dim3(cgps::SOLVE_NT), lds, st, Dp, Fp, Gp,
                         P.lv[p], owed_in, n_owed, spt_in, y, n, xcrr, yout, owed_out, partial + 2 * pb);
    pb += g;
    y = yout;
    owed_in = owed_out;
    n_owed = g;
    spt_in = P.ts[p] >> P.lv[p].nlev;
    if (spt_in < 1) spt_in = 1;
  }
  if (mahal_out) {
    double* tmp = partial + 2 * pb;  // one spare slot was reserved
    hipLaunchKernelGGL(cgps::sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, pb, tmp);
    hipMemcpyAsync(mahal_out, tmp, sizeof(double), hipMemcpyDeviceToDevice, st);
  }
  return check_launch("halfsolve (tiled)");
}

template <typename T, int D>
int run_backsolve_tile(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* ycrr, T* x, char* ws,
                       size_t ws_bytes, hipStream_t st, bool top_done = false) {
  LevelWs w = level_ws(N, D, sizeof(T), false, true);
  const size_t need = w.partial_bytes + 2 * align_up((size_t)D * sizeof(T) * w.capA);
  if (ws_bytes < need) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, need);
  Layout L;
  make_layout(N, L);
  SolvePasses P;
  make_passes(L, P, cgps::SOLVE_LP_WIDE, cgps::SOLVE_TS, cgps::SOLVE_LP, deep_tiles_for<T, D>());   // (same passes as the forward sweep)
  solve_tile_attributes<T, D>();
  T* bufs[2] = {reinterpret_cast<T*>(ws + w.partial_bytes),
                reinterpret_cast<T*>(ws + w.partial_bytes + align_up((size_t)D * sizeof(T) * w.capA))};
  const size_t lds = cgps::solve_lds_bytes<T, D>();
  const T* xc = nullptr;
  for (int p = P.np - 1; p >= 0; --p) {
    const int64_t n = P.rows[p], g = (n + P.ts[p] - 1) / P.ts[p];
    T* X = (p == 0) ? x : bufs[p & 1];
    if (top_done && p == P.np - 1) {                     // solve_top_kernel has left this pass's solution in X
      xc = X;
      continue;
    }
    bool launched = false;
    if constexpr (cgps::solve_deep_supported<T, D>()) {
      if (P.deep[p] && P.ts[p] == cgps::SOLVE_TS) {
        hipLaunchKernelGGL((cgps::backsolve_deep_kernel<T, D, cgps::SOLVE_LP>), dim3((unsigned)g), dim3(cgps::SOLVE_NT), lds, st,
                           Dp, Fp, Gp, P.lv[p], ycrr, xc, n, X);
        launched = true;
      } else if (P.deep[p]) {
        hipLaunchKernelGGL((cgps::backsolve_deep_kernel<T, D, cgps::SOLVE_LP - 1>), dim3((unsigned)g),
                           dim3(cgps::SOLVE_NT / 2), lds, st, Dp, Fp, Gp, P.lv[p], ycrr, xc, n, X);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((cgps::backsolve_tile_kernel<T, D>), dim3((unsigned)g), dim3(cgps::SOLVE_NT), lds, st, Dp, Fp, Gp,
                         P.lv[p], ycrr, xc, n, X);
    xc = X;
  }
  return check_launch("backsolve (tiled)");
}

template <typename T, int D>
int run_halfsolve_levelwise(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* y0, T* xcrr, char* ws,
                            size_t ws_bytes, double* mahal_out, hipStream_t st);
template <typename T, int D>
int run_backsolve_levelwise(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* ycrr, T* x, char* ws,
                            size_t ws_bytes, hipStream_t st);

template <typename T, int D>
int run_halfsolve(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* y0, T* xcrr, char* ws, size_t ws_bytes,
                  double* mahal_out, hipStream_t st, T* const* fused_top_bufs = nullptr, bool* fused_top = nullptr) {
  if (levelwise_solve_requested()) return run_halfsolve_levelwise<T, D>(Dp, Fp, Gp, N, y0, xcrr, ws, ws_bytes, mahal_out, st);
  return run_halfsolve_tile<T, D>(Dp, Fp, Gp, N, y0, xcrr, ws, ws_bytes, mahal_out, st, fused_top_bufs, fused_top);
}
template <typename T, int D>
int run_backsolve(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* ycrr, T* x, char* ws, size_t ws_bytes,
                  hipStream_t st, bool top_done = false) {
  if (levelwise_solve_requested()) return run_backsolve_levelwise<T, D>(Dp, Fp, Gp, N, ycrr, x, ws, ws_bytes, st);
  return run_backsolve_tile<T, D>(Dp, Fp, Gp, N, ycrr, x, ws, ws_bytes, st, top_done);
}

template <typename T, int D>
int run_halfsolve_levelwise(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* y0, T* xcrr, char* ws,
                            size_t ws_bytes, double* mahal_out, hipStream_t st) {
  LevelWs w = level_ws(N, D, sizeof(T), false, true);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  double* partial = reinterpret_cast<double*>(ws + w.partial_off);
  T* bufs[2] = {reinterpret_cast<T*>(ws + w.a_off), reinterpret_cast<T*>(ws + w.b_off)};
  const T* y = y0;
  int64_t pb = 0;
  for (int l = 0; l < L.nlevels; ++l) {
    const int64_t n = L.ms[l], nb = level_blocks(n);
    T* yn = bufs[l & 1];
    hipLaunchKernelGGL((cgps::halfsolve_level_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::LEVEL_THREADS), 0, st,
                       Dp + L.offD[l] * D * D, Fp + L.offF[l] * D * D, Gp + L.offG[l] * D * D, y, n,
                       xcrr + L.offD[l] * D, yn, partial + 2 * pb);
    pb += nb;
    y = yn;
  }
  if (mahal_out) {
    double* tmp = partial + 2 * pb;  // one spare slot was reserved
    hipLaunchKernelGGL(cgps::sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, pb, tmp);
    hipMemcpyAsync(mahal_out, tmp, sizeof(double), hipMemcpyDeviceToDevice, st);
  }
  return check_launch("halfsolve");
}

template <typename T, int D>
int run_backsolve_levelwise(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* ycrr, T* x, char* ws,
                            size_t ws_bytes, hipStream_t st) {
  LevelWs w = level_ws(N, D, sizeof(T), false, true);
  // both ping-pong buffers must hold a level-1 vector here
  const size_t need = w.partial_bytes + 2 * align_up((size_t)D * sizeof(T) * w.capA);
  if (ws_bytes < need) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, need);
  Layout L;
  make_layout(N, L);
  T* bufs[2] = {reinterpret_cast<T*>(ws + w.partial_bytes),
                reinterpret_cast<T*>(ws + w.partial_bytes + align_up((size_t)D * sizeof(T) * w.capA))};
  const T* xo = nullptr;
  for (int l = L.nlevels - 1; l >= 0; --l) {
    const int64_t n = L.ms[l], nb = level_blocks(n);
    T* X = (l == 0) ? x : bufs[l & 1];
    hipLaunchKernelGGL((cgps::backsolve_level_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::LEVEL_THREADS), 0, st,
                       Dp + L.offD[l] * D * D, Fp + L.offF[l] * D * D, Gp + L.offG[l] * D * D,
                       ycrr + L.offD[l] * D, xo, n, X);
    xo = X;
  }
  return check_launch("backsolve");
}

// ---- several right-hand sides per sweep: cgps_solve_tile_m.h ----------------------------------------
// Workspace of one panel sweep (MC columns): partial sums | two ping-pong buffers of [N/2+1][D][MC].
struct PanelWs {
  size_t partial_bytes, buf_bytes, total;
};
inline PanelWs panel_ws(int64_t N, int d, size_t s, int mc, int chunks) {
  PanelWs w{};
  // grids of all passes of all chunks: < 1.2 N / 128 tiles per chunk (the smallest tile has 128 rows)
  const int64_t tiles = (N / 128 + 64) * 2;
  w.partial_bytes = align_up((size_t)(tiles * chunks + 2) * 16);
  w.buf_bytes = align_up((size_t)(N / 2 + 2) * d * mc * s);
  w.total = w.partial_bytes + 2 * w.buf_bytes;
  return w;
}
inline int panel_width(int nrhs) { return nrhs <= 2 ? 2 : (nrhs <= 4 ? 4 : 8); }
// passes of a panel sweep over at most this many 2^TSLD-row tiles (two per CU) take the latency-bound kernels
// (cgps_solve_tile_m.h: every factor block requested up front); CGPS_NO_DEEP_SOLVE=1: never
// Measured at 2^20 rows, d = 4 fp64 (tools/prof_case.py --op solve --nrhs m): two columns 285 -> 257 us; four columns
// 402 -> 451 us, eight 631 -> 740-775 us (two 64 KB tiles per CU, each a chain of eight dependent levels on wide panels,
// against the four or five smaller tiles per CU the regular kernels keep in flight; LDS bank conflicts are not it:
// padding the panel rows changed nothing, 626 against 635 us): two-column panels only.
template <int MC> inline int64_t panel_deep_tiles_for() { return (MC <= 2 && deep_solve_enabled()) ? 512 : 0; }

template <typename T, int D, int MC>
void solve_m_attributes() {
  static PerDevice<int> done;
  done.get([](int) {
    const int lds = (int)cgps::solve_m_lds_bytes<T, D, MC>();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::halfsolve_tile_m_kernel<T, D, MC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::backsolve_tile_m_kernel<T, D, MC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (cgps::solve_deep_supported<T, D>()) {
      const int ldsd = (int)cgps::solve_m_deep_lds_bytes<T, D, MC>();
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::halfsolve_deep_m_kernel<T, D, MC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, ldsd);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::backsolve_deep_m_kernel<T, D, MC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, ldsd);
    }
    return 1;
  });
}

// forward sweep of w <= MC columns: y [N][D][ld_y] -> xcrr [N][D][ld_x]; partial sums appended at *pb
template <typename T, int D, int MC>
int run_halfsolve_panel(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* y0, int ld_y, int w, T* xcrr, int ld_x,
                        double* partial, int64_t* pb, T* buf0, T* buf1, hipStream_t st) {
  constexpr int TSL = cgps::solve_m_tile_log2<MC>(), TS = 1 << TSL, NT = TS / 2, PW = D * MC;
  constexpr int TSLD = cgps::solve_m_deep_tile_log2<MC>(), TSD = 1 << TSLD, CS = cgps::solve_m_col_splits<MC>();
  constexpr bool DEEP = cgps::solve_deep_supported<T, D>();
  Layout L;
  make_layout(N, L);
  SolvePasses P;
  make_passes(L, P, cgps::SOLVE_LP_WIDE, TS, TSL, 0, DEEP ? TSD : 0, TSLD, panel_deep_tiles_for<MC>());
  solve_m_attributes<T, D, MC>();
  T* bufs[2] = {buf0, buf1};
  const size_t lds = cgps::solve_m_lds_bytes<T, D, MC>();
  const T* y = y0;
  int ld = ld_y;
  const T* owed_in = nullptr;
  int64_t n_owed = 0;
  int spt_in = 1;
  for (int p = 0; p < P.np; ++p) {
    const int ts = P.ts[p];
    const int64_t n = P.rows[p], g = (n + ts - 1) / ts;
    const bool more = (p + 1 < P.np);
    const int64_t nsurv = n >> P.lv[p].nlev;
    T* yout = more ? bufs[p & 1] : nullptr;            // [nsurv][D][MC] surviving rows, then [g][D][MC] owed panels
    T* owed_out = more ? bufs[p & 1] + (nsurv + 1) * PW : nullptr;
    bool launched = false;
    if constexpr (DEEP) {
      if (P.deep[p]) {
        const size_t ldsd = cgps::solve_m_deep_lds_bytes<T, D, MC>();
        hipLaunchKernelGGL((cgps::halfsolve_deep_m_kernel<T, D, MC>), dim3((unsigned)g), dim3(TSD / 2 * CS), ldsd, st, Dp, Fp, Gp,
                           P.lv[p], owed_in, n_owed, spt_in, y, ld, n, w, xcrr, ld_x, yout, owed_out, partial + 2 * *pb);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((cgps::halfsolve_tile_m_kernel<T, D, MC>), dim3((unsigned)g), dim3(NT * CS), lds, st, Dp, Fp, Gp, P.lv[p],
                         owed_in, n_owed, spt_in, y, ld, n, w, xcrr, ld_x, yout, owed_out, partial + 2 * *pb);
    *pb += g;
    y = yout;
    ld = MC;
    owed_in = owed_out;
    n_owed = g;
    spt_in = ts >> P.lv[p].nlev;
    if (spt_in < 1) spt_in = 1;
  }
  return CGPS_OK;
}

template <typename T, int D, int MC>
int run_backsolve_panel(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* b, int ld_b, int w, T* x, int ld_o,
                        T* buf0, T* buf1, hipStream_t st) {
  constexpr int TSL = cgps::solve_m_tile_log2<MC>(), TS = 1 << TSL, NT = TS / 2;
  constexpr int TSLD = cgps::solve_m_deep_tile_log2<MC>(), TSD = 1 << TSLD, CS = cgps::solve_m_col_splits<MC>();
  constexpr bool DEEP = cgps::solve_deep_supported<T, D>();
  Layout L;
  make_layout(N, L);
  SolvePasses P;
  make_passes(L, P, cgps::SOLVE_LP_WIDE, TS, TSL, 0, DEEP ? TSD : 0, TSLD, panel_deep_tiles_for<MC>());   // (same passes as the forward sweep)
  solve_m_attributes<T, D, MC>();
  T* bufs[2] = {buf0, buf1};
  const size_t lds = cgps::solve_m_lds_bytes<T, D, MC>();
  const T* xc = nullptr;
  for (int p = P.np - 1; p >= 0; --p) {
    const int ts = P.ts[p];
    const int64_t n = P.rows[p], g = (n + ts - 1) / ts;
    T* X = (p == 0) ? x : bufs[p & 1];
    bool launched = false;
    if constexpr (DEEP) {
      if (P.deep[p]) {
        const size_t ldsd = cgps::solve_m_deep_lds_bytes<T, D, MC>();
        hipLaunchKernelGGL((cgps::backsolve_deep_m_kernel<T, D, MC>), dim3((unsigned)g), dim3(TSD / 2 * CS), ldsd, st, Dp, Fp, Gp,
                           P.lv[p], b, ld_b, xc, n, w, X, (p == 0) ? ld_o : MC);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((cgps::backsolve_tile_m_kernel<T, D, MC>), dim3((unsigned)g), dim3(NT * CS), lds, st, Dp, Fp, Gp, P.lv[p],
                         b, ld_b, xc, n, w, X, (p == 0) ? ld_o : MC);
    xc = X;
  }
  return CGPS_OK;
}

enum class PanelOp { Half, Back, Solve };
// nrhs >= 2 columns, panels of up to eight: halfsolve (y -> xcrr, mahal), backsolve (y = CRR -> x), solve (y -> x)
template <typename T, int D, int MC>
int run_panels(PanelOp op, const T* Dp, const T* Fp, const T* Gp, int64_t N, int nrhs, const T* y, T* out, char* ws,
               size_t ws_bytes, double* mahal_out, hipStream_t st) {
  const int chunks = (nrhs + MC - 1) / MC;
  const PanelWs w = panel_ws(N, D, sizeof(T), MC, chunks);
  const size_t crr = (op == PanelOp::Solve) ? align_up((size_t)N * D * MC * sizeof(T)) : 0;
  if (ws_bytes < w.total + crr) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total + crr);
  double* partial = reinterpret_cast<double*>(ws);
  T* buf0 = reinterpret_cast<T*>(ws + w.partial_bytes);
  T* buf1 = reinterpret_cast<T*>(ws + w.partial_bytes + w.buf_bytes);
  T* xcrr_ws = reinterpret_cast<T*>(ws + w.total);
  int64_t pb = 0;
  for (int c0 = 0; c0 < nrhs; c0 += MC) {
    const int wd = nrhs - c0 < MC ? nrhs - c0 : MC;
    if (op == PanelOp::Half)
      run_halfsolve_panel<T, D, MC>(Dp, Fp, Gp, N, y + c0, nrhs, wd, out + c0, nrhs, partial, &pb, buf0, buf1, st);
    else if (op == PanelOp::Back)
      run_backsolve_panel<T, D, MC>(Dp, Fp, Gp, N, y + c0, nrhs, wd, out + c0, nrhs, buf0, buf1, st);
    else {
      run_halfsolve_panel<T, D, MC>(Dp, Fp, Gp, N, y + c0, nrhs, wd, xcrr_ws, MC, partial, &pb, buf0, buf1, st);
      run_backsolve_panel<T, D, MC>(Dp, Fp, Gp, N, xcrr_ws, MC, wd, out + c0, nrhs, buf0, buf1, st);
    }
  }
  if (mahal_out) {
    double* tmp = partial + 2 * pb;
    hipLaunchKernelGGL(cgps::sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, pb, tmp);
    (void)hipMemcpyAsync(mahal_out, tmp, sizeof(double), hipMemcpyDeviceToDevice, st);
  }
  return check_launch("panel substitution sweeps");
}

template <typename T, int D>
int run_panels_any(PanelOp op, const T* Dp, const T* Fp, const T* Gp, int64_t N, int nrhs, const T* y, T* out, char* ws,
                   size_t ws_bytes, double* mahal_out, hipStream_t st) {
  switch (panel_width(nrhs)) {
    case 2: return run_panels<T, D, 2>(op, Dp, Fp, Gp, N, nrhs, y, out, ws, ws_bytes, mahal_out, st);
    case 4: return run_panels<T, D, 4>(op, Dp, Fp, Gp, N, nrhs, y, out, ws, ws_bytes, mahal_out, st);
    default: return run_panels<T, D, 8>(op, Dp, Fp, Gp, N, nrhs, y, out, ws, ws_bytes, mahal_out, st);
  }
}
}  // namespace

extern "C" {

int cgps_solve_workspace_bytes(int64_t N, int d, int dtype, int op, int nrhs, size_t* bytes) {
  if (bad_common(N, d) || !bytes || nrhs < 1) return fail(CGPS_ERR_ARG, "cgps_solve_workspace_bytes: bad argument");
  if (nrhs == 1) return cgps_workspace_bytes(N, d, dtype, op, bytes);
  if (d > 8) return fail(CGPS_ERR_UNSUPPORTED, "block size d=%d outside 1..8", d);
  if (dtype != CGPS_F32 && dtype != CGPS_F64) return fail(CGPS_ERR_UNSUPPORTED, "dtype %d not supported", dtype);
  if (op != CGPS_OP_HALFSOLVE && op != CGPS_OP_BACKSOLVE && op != CGPS_OP_SOLVE)
    return fail(CGPS_ERR_ARG, "cgps_solve_workspace_bytes: op %d takes no right-hand sides", op);
  const size_t s = dtype == CGPS_F32 ? 4 : 8;
  const int mc = panel_width(nrhs);
  const PanelWs w = panel_ws(N, d, s, mc, (nrhs + mc - 1) / mc);
  *bytes = w.total + (op == CGPS_OP_SOLVE ? align_up((size_t)N * d * mc * s) : 0);
  return CGPS_OK;
}

int cgps_halfsolve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, int nrhs, const void* y,
                   void* xcrr, void* ws, size_t ws_bytes, double* mahal_out, void* stream) {
  if (bad_common(N, d) || nrhs < 1 || !Dp || !Fp || !Gp || !y || !xcrr || !ws)
    return fail(CGPS_ERR_ARG, "cgps_halfsolve: null pointer, N < 1 or nrhs < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if (nrhs > 1)
      return run_panels_any<T, D>(PanelOp::Half, (const T*)Dp, (const T*)Fp, (const T*)Gp, N, nrhs, (const T*)y, (T*)xcrr,
                                  (char*)ws, ws_bytes, mahal_out, (hipStream_t)stream);
    return run_halfsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (const T*)y, (T*)xcrr, (char*)ws, ws_bytes,
                               mahal_out, (hipStream_t)stream);
  });
}

int cgps_backsolve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, int nrhs,
                   const void* ycrr, void* x, void* ws, size_t ws_bytes, void* stream) {
  if (bad_common(N, d) || nrhs < 1 || !Dp || !Fp || !Gp || !ycrr || !x || !ws)
    return fail(CGPS_ERR_ARG, "cgps_backsolve: null pointer, N < 1 or nrhs < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if (nrhs > 1)
      return run_panels_any<T, D>(PanelOp::Back, (const T*)Dp, (const T*)Fp, (const T*)Gp, N, nrhs, (const T*)ycrr, (T*)x,
                                  (char*)ws, ws_bytes, nullptr, (hipStream_t)stream);
    return run_backsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (const T*)ycrr, (T*)x, (char*)ws, ws_bytes,
                               (hipStream_t)stream);
  });
}

int cgps_solve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, int nrhs, const void* y,
               void* x, void* ws, size_t ws_bytes, void* stream) {
  if (bad_common(N, d) || nrhs < 1 || !Dp || !Fp || !Gp || !y || !x || !ws)
    return fail(CGPS_ERR_ARG, "cgps_solve: null pointer, N < 1 or nrhs < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if (nrhs > 1)
      return run_panels_any<T, D>(PanelOp::Solve, (const T*)Dp, (const T*)Fp, (const T*)Gp, N, nrhs, (const T*)y, (T*)x,
                                  (char*)ws, ws_bytes, nullptr, (hipStream_t)stream);
    const size_t crr = align_up((size_t)N * D * sizeof(T));
    if (ws_bytes < crr) return fail(CGPS_ERR_ARG, "workspace too small");
    T* xcrr = (T*)ws;
    // where the backward sweep keeps the solution of pass p (run_backsolve_tile): bufs[p & 1], the
    // caller's x for pass 0 -- the fused top pass writes there
    const LevelWs w = level_ws(N, D, sizeof(T), false, true);
    T* top_bufs[3] = {reinterpret_cast<T*>((char*)ws + crr + w.partial_bytes),
                      reinterpret_cast<T*>((char*)ws + crr + w.partial_bytes + align_up((size_t)D * sizeof(T) * w.capA)), (T*)x};
    bool top_done = false;
    int rc = run_halfsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (const T*)y, xcrr, (char*)ws + crr,
                                 ws_bytes - crr, nullptr, (hipStream_t)stream, top_bufs, &top_done);
    if (rc != CGPS_OK) return rc;
    return run_backsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, xcrr, (T*)x, (char*)ws + crr,
                               ws_bytes - crr, (hipStream_t)stream, top_done);
  });
}

}  // extern "C"
