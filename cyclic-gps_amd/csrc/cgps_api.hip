// C-ABI entry points of libcgps (see include/cgps.h).  Host code only decides
// sizes/offsets and enqueues kernels on the caller's stream; nothing here
// allocates, copies to the host or synchronises.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <type_traits>

#include "../../include/cgps.h"
#include "cgps_level.h"
#include "cgps_tile.h"
#include "cgps_solve_tile.h"
#include "cgps_decomp_tile.h"
#include "cgps_decomp_lds.h"
#include "cgps_inverse_tile.h"
#include "cgps_leg.h"
#include <cstdlib>

namespace {

thread_local char g_err[512] = "";
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CGPS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return CGPS_OK;
}

template <typename Fn>
int dispatch(int dtype, int d, Fn&& fn) {
#define CGPS_CASE(DV)                                                        \
  case DV:                                                                   \
    return dtype == CGPS_F32 ? fn(float{}, std::integral_constant<int, DV>{}) \
                             : fn(double{}, std::integral_constant<int, DV>{});
  if (dtype != CGPS_F32 && dtype != CGPS_F64) return fail(CGPS_ERR_UNSUPPORTED, "dtype %d not supported", dtype);
  switch (d) {
    CGPS_CASE(1) CGPS_CASE(2) CGPS_CASE(3) CGPS_CASE(4) CGPS_CASE(5) CGPS_CASE(6) CGPS_CASE(7) CGPS_CASE(8)
    default:
      return fail(CGPS_ERR_UNSUPPORTED, "block size d=%d outside 1..8", d);
  }
#undef CGPS_CASE
}

struct Layout {
  int nlevels;
  int64_t ms[CGPS_MAX_LEVELS], offD[CGPS_MAX_LEVELS + 1], offF[CGPS_MAX_LEVELS + 1], offG[CGPS_MAX_LEVELS + 1];
};

void make_layout(int64_t N, Layout& L) {
  int l = 0;
  int64_t m = N, oD = 0, oF = 0, oG = 0;
  for (;;) {
    L.ms[l] = m;
    L.offD[l] = oD; L.offF[l] = oF; L.offG[l] = oG;
    oD += (m + 1) / 2; oF += m / 2; oG += (m - 1) / 2;
    ++l;
    if (m == 1) break;
    m /= 2;
  }
  L.nlevels = l;
  L.offD[l] = oD; L.offF[l] = oF; L.offG[l] = oG;
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
inline int64_t level_blocks(int64_t m) { return ((m + 1) / 2 + cgps::LEVEL_THREADS - 1) / cgps::LEVEL_THREADS; }

// workspace carve-up shared by the level-wise paths
struct LevelWs {
  size_t partial_off, partial_bytes;  // [total blocks][2] doubles
  size_t a_off, b_off;                // ping-pong level buffers
  int64_t capA, capB;                 // rows
  size_t total;
};

LevelWs level_ws(int64_t N, int d, size_t s, bool with_mats, bool with_vec) {
  LevelWs w{};
  Layout L;
  make_layout(N, L);
  int64_t nb = 0;
  for (int l = 0; l < L.nlevels; ++l) nb += level_blocks(L.ms[l]);
  w.partial_off = 0;
  w.partial_bytes = align_up((size_t)(nb + 1) * 16);
  w.capA = N / 2 + 1;
  w.capB = N / 4 + 1;
  size_t per_row = (with_mats ? 2 * (size_t)d * d : 0) + (with_vec ? (size_t)d : 0);
  w.a_off = w.partial_bytes;
  w.b_off = w.a_off + align_up(per_row * s * w.capA);
  w.total = w.b_off + align_up(per_row * s * w.capB);
  return w;
}

template <typename T>
struct LevelBuf {
  T *R, *O, *y;
};
template <typename T>
LevelBuf<T> carve(char* base, int64_t cap, int d, bool with_mats, bool with_vec) {
  LevelBuf<T> b{nullptr, nullptr, nullptr};
  T* p = reinterpret_cast<T*>(base);
  if (with_mats) { b.R = p; p += cap * d * d; b.O = p; p += cap * d * d; }
  if (with_vec) b.y = p;
  return b;
}

// ---------------------------------------------------------------------------------
template <typename T, int D>
int run_levelwise(const T* Rs, const T* Os, const T* x, int64_t N, T* Dp, T* Fp, T* Gp, T* xcrr,
                  char* ws, size_t ws_bytes, double* out2, int* info, hipStream_t st) {
  const bool rhs = (x != nullptr), emit = (Dp != nullptr);
  LevelWs w = level_ws(N, D, sizeof(T), true, rhs);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  double* partial = reinterpret_cast<double*>(ws + w.partial_off);
  LevelBuf<T> bufs[2] = {carve<T>(ws + w.a_off, w.capA, D, true, rhs), carve<T>(ws + w.b_off, w.capB, D, true, rhs)};
  hipMemsetAsync(info, 0, sizeof(int), st);
  const T *R = Rs, *O = Os, *y = x;
  int64_t pb = 0;
  for (int l = 0; l < L.nlevels; ++l) {
    const int64_t n = L.ms[l];
    const int64_t nb = level_blocks(n);
    LevelBuf<T>& nx = bufs[l & 1];
    T* Dk = emit ? Dp + L.offD[l] * D * D : nullptr;
    T* Fk = emit ? Fp + L.offF[l] * D * D : nullptr;
    T* Gk = emit ? Gp + L.offG[l] * D * D : nullptr;
    T* xk = (emit && rhs && xcrr) ? xcrr + L.offD[l] * D : nullptr;
    dim3 grid((unsigned)nb), block(cgps::LEVEL_THREADS);
    if (l == 0 && g_prof_start) hipEventRecord(g_prof_start, st);
    if (emit && rhs)
      hipLaunchKernelGGL((cgps::level_kernel<T, D, true, true>), grid, block, 0, st, R, O, y, n, l, Dk, Fk, Gk, xk,
                         nx.R, nx.O, nx.y, partial + 2 * pb, info);
    else if (emit)
      hipLaunchKernelGGL((cgps::level_kernel<T, D, true, false>), grid, block, 0, st, R, O, y, n, l, Dk, Fk, Gk, xk,
                         nx.R, nx.O, nx.y, partial + 2 * pb, info);
    else if (rhs)
      hipLaunchKernelGGL((cgps::level_kernel<T, D, false, true>), grid, block, 0, st, R, O, y, n, l, Dk, Fk, Gk, xk,
                         nx.R, nx.O, nx.y, partial + 2 * pb, info);
    else
      hipLaunchKernelGGL((cgps::level_kernel<T, D, false, false>), grid, block, 0, st, R, O, y, n, l, Dk, Fk, Gk, xk,
                         nx.R, nx.O, nx.y, partial + 2 * pb, info);
    if (l == 0 && g_prof_stop) {
      hipEventRecord(g_prof_stop, st);
      g_prof_start = g_prof_stop = nullptr;
    }
    pb += nb;
    R = nx.R; O = nx.O; y = nx.y;
  }
  if (out2) hipLaunchKernelGGL(cgps::sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, pb, out2);
  return check_launch("levelwise reduction");
}

// ---- fused (tiled) substitution sweeps: cgps_solve_tile.h ---------------------------------------
struct SolvePasses {
  int np;
  int first[8];
  cgps::PassLevels lv[8];
  int64_t rows[8];
};

void make_passes(const Layout& L, SolvePasses& P, int wide_lp) {
  P.np = 0;
  int lvl = 0;
  while (lvl < L.nlevels) {
    const int64_t rows = L.ms[lvl];
    const int remaining = L.nlevels - lvl;
    // many tiles: a few levels per pass (every lane busy, few barrier-separated latency
    // exposures, the factor still read once); few tiles: all ten levels of a tile
    const int nl = (rows <= cgps::SOLVE_TS) ? remaining
                   : (rows >= cgps::SOLVE_WIDE_ROWS ? wide_lp : cgps::SOLVE_LP);   // <= SOLVE_LP + 1
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

bool levelwise_solve_requested() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CGPS_LEVELWISE_SOLVE");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

template <typename T, int D>
void solve_tile_attributes() {
  static bool done = false;
  if (done) return;
  const int lds = (int)cgps::solve_lds_bytes<T, D>();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::halfsolve_tile_kernel<T, D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::backsolve_tile_kernel<T, D>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  done = true;
}

template <typename T, int D>
int run_halfsolve_tile(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* y0, T* xcrr, char* ws,
                       size_t ws_bytes, double* mahal_out, hipStream_t st) {
  LevelWs w = level_ws(N, D, sizeof(T), false, true);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  SolvePasses P;
  make_passes(L, P, cgps::SOLVE_LP_WIDE);
  solve_tile_attributes<T, D>();
  double* partial = reinterpret_cast<double*>(ws + w.partial_off);
  T* bufs[2] = {reinterpret_cast<T*>(ws + w.a_off), reinterpret_cast<T*>(ws + w.b_off)};
  const size_t lds = cgps::solve_lds_bytes<T, D>();
  const T* y = y0;
  const T* owed_in = nullptr;
  int64_t n_owed = 0, pb = 0;
  int spt_in = 1;
  for (int p = 0; p < P.np; ++p) {
    const int64_t n = P.rows[p], g = (n + cgps::SOLVE_TS - 1) / cgps::SOLVE_TS;
    const bool more = (p + 1 < P.np);
    const int64_t nsurv = n >> P.lv[p].nlev;           // rows of the next pass
    T* yout = more ? bufs[p & 1] : nullptr;            // [nsurv][D] surviving rows, then [g][D] owed vectors
    T* owed_out = more ? bufs[p & 1] + (nsurv + 1) * D : nullptr;
    hipLaunchKernelGGL((cgps::halfsolve_tile_kernel<T, D>), dim3((unsigned)g), dim3(cgps::SOLVE_NT), lds, st, Dp, Fp, Gp,
                       P.lv[p], owed_in, n_owed, spt_in, y, n, xcrr, yout, owed_out, partial + 2 * pb);
    pb += g;
    y = yout;
    owed_in = owed_out;
    n_owed = g;
    spt_in = cgps::SOLVE_TS >> P.lv[p].nlev;
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
                       size_t ws_bytes, hipStream_t st) {
  LevelWs w = level_ws(N, D, sizeof(T), false, true);
  const size_t need = w.partial_bytes + 2 * align_up((size_t)D * sizeof(T) * w.capA);
  if (ws_bytes < need) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, need);
  Layout L;
  make_layout(N, L);
  SolvePasses P;
  make_passes(L, P, cgps::SOLVE_LP_WIDE);      // (the two sweeps need not use the same passes; 3 / 3 measured best)
  solve_tile_attributes<T, D>();
  T* bufs[2] = {reinterpret_cast<T*>(ws + w.partial_bytes),
                reinterpret_cast<T*>(ws + w.partial_bytes + align_up((size_t)D * sizeof(T) * w.capA))};
  const size_t lds = cgps::solve_lds_bytes<T, D>();
  const T* xc = nullptr;
  for (int p = P.np - 1; p >= 0; --p) {
    const int64_t n = P.rows[p], g = (n + cgps::SOLVE_TS - 1) / cgps::SOLVE_TS;
    T* X = (p == 0) ? x : bufs[p & 1];
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
                  double* mahal_out, hipStream_t st) {
  if (levelwise_solve_requested()) return run_halfsolve_levelwise<T, D>(Dp, Fp, Gp, N, y0, xcrr, ws, ws_bytes, mahal_out, st);
  return run_halfsolve_tile<T, D>(Dp, Fp, Gp, N, y0, xcrr, ws, ws_bytes, mahal_out, st);
}
template <typename T, int D>
int run_backsolve(const T* Dp, const T* Fp, const T* Gp, int64_t N, const T* ycrr, T* x, char* ws, size_t ws_bytes,
                  hipStream_t st) {
  if (levelwise_solve_requested()) return run_backsolve_levelwise<T, D>(Dp, Fp, Gp, N, ycrr, x, ws, ws_bytes, st);
  return run_backsolve_tile<T, D>(Dp, Fp, Gp, N, ycrr, x, ws, ws_bytes, st);
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
  constexpr bool FUSED = (size_t)D * D * sizeof(T) <= 128;
  const size_t lds = (size_t)64 * D * D * sizeof(T);
  static int grid_cap = 0;
  if (FUSED && grid_cap == 0) {
    int dev = 0, cus = 256, nb = 2;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cgps::inverse_tile_kernel<T, D>, cgps::INV_NT, lds);
    grid_cap = cus * (nb > 0 ? nb : 1);
  }
  int p = 0;
  for (int l = L.nlevels - 1; l >= 0;) {
    const int have = l + 1;                               // Sdc / Soc hold Sigma of this level
    if (FUSED && Sdc != nullptr && have % cgps::INV_LP == 0 && L.ms[have] >= 1 &&
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
      hipLaunchKernelGGL((cgps::inverse_tile_kernel<T, D>), dim3((unsigned)grid), dim3(cgps::INV_NT), lds, st, Dp, Fp,
                         Gp, lv, Sdc, Soc, n, od, oo);
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

// ---- fused (tiled) factorisation: cgps_decomp_tile.h (bulk passes) + cgps_decomp_lds.h (tail) ----
constexpr int64_t DEC_SMALL_ROWS = 32768;   // at or below this many rows a pass is latency-bound
template <typename T, int D>
int run_decompose_tile(const T* Rs, const T* Os, int64_t N, T* Dp, T* Fp, T* Gp, char* ws, size_t ws_bytes, int* info,
                       hipStream_t st) {
  using RL = cgps::RecordLayout<T, D>;
  LevelWs w = level_ws(N, D, sizeof(T), true, false);
  if (ws_bytes < w.total) return fail(CGPS_ERR_ARG, "workspace too small: %zu < %zu", ws_bytes, w.total);
  Layout L;
  make_layout(N, L);
  // persistent waves: as many workgroups (one wave each) as the chip holds at once
  const size_t lds = (size_t)64 * D * D * sizeof(T);      // staging of the coalesced factor stores
  static int grid_cap[2] = {0, 0};
  if (grid_cap[0] == 0) {
    int dev = 0, cus = 256, nb0 = 4, nb1 = 4;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, cgps::decomp_tile_kernel<T, D, false>, cgps::DEC_NT, lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, cgps::decomp_tile_kernel<T, D, true>, cgps::DEC_NT, lds);
    grid_cap[0] = cus * (nb0 > 0 ? nb0 : 1);
    grid_cap[1] = cus * (nb1 > 0 ? nb1 : 1);
  }
  (void)hipMemsetAsync(info, 0, sizeof(int), st);
  T* recs[2] = {reinterpret_cast<T*>(ws + w.a_off), reinterpret_cast<T*>(ws + w.b_off)};   // records of a pass
  const T* rin = nullptr;
  int64_t n_rec = 0;
  int lvl = 0, p = 0, spt_in = 1;
  const size_t lds_small = cgps::decomp_lds_tile_bytes<T, D>();
  static bool small_attr_done = false;
  if (!small_attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::decomp_lds_kernel<T, D, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cgps::decomp_lds_kernel<T, D, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    small_attr_done = true;
  }
  while (lvl < L.nlevels) {
    const int64_t rows = L.ms[lvl];
    const int remaining = L.nlevels - lvl;
    if (rows <= DEC_SMALL_ROWS) {
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
    if (p == 0)
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

bool bad_common(int64_t N, int d) { return N < 1 || d < 1; }

}  // namespace

extern "C" {

int cgps_version(void) { return CGPS_VERSION; }
const char* cgps_last_error(void) { return g_err; }

int cgps_profile_next_call(void* start_event, void* stop_event) {
  g_prof_start = (hipEvent_t)start_event;
  g_prof_stop = (hipEvent_t)stop_event;
  return CGPS_OK;
}

int cgps_level_layout(int64_t N, int* nlevels, int64_t* ms, int64_t* offD, int64_t* offF, int64_t* offG) {
  if (N < 1 || !nlevels) return fail(CGPS_ERR_ARG, "cgps_level_layout: N must be >= 1");
  Layout L;
  make_layout(N, L);
  *nlevels = L.nlevels;
  for (int l = 0; l < L.nlevels; ++l)
    if (ms) ms[l] = L.ms[l];
  for (int l = 0; l <= L.nlevels; ++l) {
    if (offD) offD[l] = L.offD[l];
    if (offF) offF[l] = L.offF[l];
    if (offG) offG[l] = L.offG[l];
  }
  return CGPS_OK;
}

int cgps_workspace_bytes(int64_t N, int d, int dtype, int op, size_t* bytes) {
  if (bad_common(N, d) || !bytes) return fail(CGPS_ERR_ARG, "cgps_workspace_bytes: bad argument");
  if (d > 8) return fail(CGPS_ERR_UNSUPPORTED, "block size d=%d outside 1..8", d);
  if (dtype != CGPS_F32 && dtype != CGPS_F64) return fail(CGPS_ERR_UNSUPPORTED, "dtype %d not supported", dtype);
  const size_t s = dtype == CGPS_F32 ? 4 : 8;
  const int64_t capA = N / 2 + 1;
  switch (op) {
    case CGPS_OP_MAHAL_LOGDET_LEVELWISE:
      *bytes = level_ws(N, d, s, true, true).total;
      return CGPS_OK;
    case CGPS_OP_MAHAL_LOGDET: {
      size_t a = level_ws(N, d, s, true, true).total;
      size_t b = cgps::tile_ws_bytes(N, d, s);
      *bytes = a > b ? a : b;
      return CGPS_OK;
    }
    case CGPS_OP_DECOMPOSE:
      *bytes = level_ws(N, d, s, true, false).total;
      return CGPS_OK;
    case CGPS_OP_HALFSOLVE:
      *bytes = level_ws(N, d, s, false, true).total;
      return CGPS_OK;
    case CGPS_OP_BACKSOLVE:
    case CGPS_OP_SOLVE: {
      LevelWs w = level_ws(N, d, s, false, true);
      size_t back = w.partial_bytes + 2 * align_up((size_t)d * s * capA);
      size_t crr = (op == CGPS_OP_SOLVE) ? align_up((size_t)N * d * s) : 0;
      size_t m = w.total > back ? w.total : back;
      *bytes = crr + m;
      return CGPS_OK;
    }
    case CGPS_OP_LOGDET_FACTOR:
      *bytes = align_up((size_t)(1024 + 2) * 16);
      return CGPS_OK;
    case CGPS_OP_INVERSE_BLOCKS:
      *bytes = 2 * align_up((size_t)2 * d * d * s * capA);
      return CGPS_OK;
    default:
      return fail(CGPS_ERR_ARG, "cgps_workspace_bytes: unknown op %d", op);
  }
}

int cgps_mahal_logdet_levelwise(const void* Rs, const void* Os, const void* x, int64_t N, int d, int dtype, void* ws,
                                size_t ws_bytes, double* out2, int* info, void* stream) {
  if (bad_common(N, d) || !Rs || (N > 1 && !Os) || !x || !ws || !out2 || !info)
    return fail(CGPS_ERR_ARG, "cgps_mahal_logdet: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    return run_levelwise<T, D>((const T*)Rs, (const T*)Os, (const T*)x, N, nullptr, nullptr, nullptr, nullptr,
                               (char*)ws, ws_bytes, out2, info, (hipStream_t)stream);
  });
}

int cgps_mahal_logdet(const void* Rs, const void* Os, const void* x, int64_t N, int d, int dtype, void* ws,
                      size_t ws_bytes, double* out2, int* info, void* stream) {
  if (bad_common(N, d) || !Rs || (N > 1 && !Os) || !x || !ws || !out2 || !info)
    return fail(CGPS_ERR_ARG, "cgps_mahal_logdet: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if constexpr (!cgps::tile_supported<T, D>()) {
      return run_levelwise<T, D>((const T*)Rs, (const T*)Os, (const T*)x, N, nullptr, nullptr, nullptr, nullptr,
                                 (char*)ws, ws_bytes, out2, info, (hipStream_t)stream);
    } else {
      int rc = cgps::run_tile_mahal_logdet<T, D>((const T*)Rs, (const T*)Os, (const T*)x, N, (char*)ws, ws_bytes,
                                                 out2, info, (hipStream_t)stream, g_prof_start, g_prof_stop);
      g_prof_start = g_prof_stop = nullptr;
      if (rc == -1) return fail(CGPS_ERR_ARG, "workspace too small for cgps_mahal_logdet");
      return check_launch("tile reduction");
    }
  });
}

int cgps_record_elems(int d, int dtype, int64_t* elems) {
  if (!elems || d < 1) return fail(CGPS_ERR_ARG, "cgps_record_elems: bad argument");
  if (d > 8 || (dtype != CGPS_F32 && dtype != CGPS_F64)) return fail(CGPS_ERR_UNSUPPORTED, "unsupported d / dtype");
  *elems = ((3 * d * d + 2 * d + 3) / 4) * 4;
  return CGPS_OK;
}

int cgps_shard_reduce(const void* Rs, const void* Os, const void* x, const void* O_left, int64_t n_loc, int d,
                      int dtype, void* ws, size_t ws_bytes, void* record_out, double* partial_out, void* stream) {
  if (bad_common(n_loc, d) || !Rs || (n_loc > 1 && !Os) || !x || !ws || !record_out || !partial_out)
    return fail(CGPS_ERR_ARG, "cgps_shard_reduce: null pointer or n_loc < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if constexpr (!cgps::tile_supported<T, D>()) {
      return fail(CGPS_ERR_UNSUPPORTED, "sharded reduction is built for fp64 d<=5 and fp32 d<=8");
    } else {
      int rc = cgps::run_tile_mahal_logdet<T, D>((const T*)Rs, (const T*)Os, (const T*)x, n_loc, (char*)ws, ws_bytes,
                                                 nullptr, nullptr, (hipStream_t)stream, g_prof_start, g_prof_stop,
                                                 (const T*)O_left, (T*)record_out, partial_out);
      g_prof_start = g_prof_stop = nullptr;
      if (rc == -1) return fail(CGPS_ERR_ARG, "workspace too small for cgps_shard_reduce");
      return check_launch("shard reduction");
    }
  });
}

int cgps_finish_records(const void* records, size_t record_stride_bytes, const double* partials,
                        size_t partial_stride_bytes, int64_t P, int64_t rows_per_shard, int64_t N_total, int d,
                        int dtype, double* out2, int* info, void* stream) {
  if (P < 1 || d < 1 || !records || !partials || !out2 || !info)
    return fail(CGPS_ERR_ARG, "cgps_finish_records: null pointer or P < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if constexpr (!cgps::tile_supported<T, D>()) {
      return fail(CGPS_ERR_UNSUPPORTED, "sharded reduction is built for fp64 d<=5 and fp32 d<=8");
    } else {
      if (record_stride_bytes % sizeof(T) != 0 || partial_stride_bytes % sizeof(double) != 0 ||
          record_stride_bytes < cgps::RecordLayout<T, D>::STRIDE * sizeof(T) || partial_stride_bytes < 32)
        return fail(CGPS_ERR_ARG, "cgps_finish_records: bad record / partial stride");
      int rc = cgps::run_tile_finish<T, D>((const T*)records, (int64_t)(record_stride_bytes / sizeof(T)), partials,
                                           (int64_t)(partial_stride_bytes / sizeof(double)), P, rows_per_shard,
                                           N_total, out2, info, (hipStream_t)stream);
      if (rc == -1) return fail(CGPS_ERR_ARG, "cgps_finish_records: P outside 1..2048");
      return check_launch("finish records");
    }
  });
}

int cgps_decompose_step(const void* Rs, const void* Os, int64_t n, int d, int dtype, void* Dk, void* Fk, void* Gk,
                        void* Rn, void* On, int* info, void* stream) {
  if (n < 2 || d < 1 || !Rs || !Os || !Dk || !Fk || !Rn || !info || (n > 2 && (!Gk)) || (n > 3 && !On))
    return fail(CGPS_ERR_ARG, "cgps_decompose_step: null pointer or n < 2");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(info, 0, sizeof(int), st);
    const int64_t nb = level_blocks(n);
    hipLaunchKernelGGL((cgps::level_kernel<T, D, true, false>), dim3((unsigned)nb), dim3(cgps::LEVEL_THREADS), 0, st,
                       (const T*)Rs, (const T*)Os, (const T*)nullptr, n, 0, (T*)Dk, (T*)Fk, (T*)Gk, (T*)nullptr,
                       (T*)Rn, (T*)On, (T*)nullptr, (double*)nullptr, info);
    return check_launch("decompose_step");
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
    if constexpr (cgps::tile_supported<T, D>()) {
      if (!levelwise_solve_requested())
        return run_decompose_tile<T, D>((const T*)Rs, (const T*)Os, N, (T*)Dp, (T*)Fp, (T*)Gp, (char*)ws, ws_bytes,
                                        info, (hipStream_t)stream);
    }
    return run_levelwise<T, D>((const T*)Rs, (const T*)Os, nullptr, N, (T*)Dp, (T*)Fp, (T*)Gp, nullptr, (char*)ws,
                               ws_bytes, nullptr, info, (hipStream_t)stream);
  });
}

int cgps_halfsolve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, const void* y,
                   void* xcrr, void* ws, size_t ws_bytes, double* mahal_out, void* stream) {
  if (bad_common(N, d) || !Dp || !Fp || !Gp || !y || !xcrr || !ws)
    return fail(CGPS_ERR_ARG, "cgps_halfsolve: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    return run_halfsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (const T*)y, (T*)xcrr, (char*)ws, ws_bytes,
                               mahal_out, (hipStream_t)stream);
  });
}

int cgps_backsolve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, const void* ycrr,
                   void* x, void* ws, size_t ws_bytes, void* stream) {
  if (bad_common(N, d) || !Dp || !Fp || !Gp || !ycrr || !x || !ws)
    return fail(CGPS_ERR_ARG, "cgps_backsolve: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    return run_backsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (const T*)ycrr, (T*)x, (char*)ws, ws_bytes,
                               (hipStream_t)stream);
  });
}

int cgps_solve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, const void* y, void* x,
               void* ws, size_t ws_bytes, void* stream) {
  if (bad_common(N, d) || !Dp || !Fp || !Gp || !y || !x || !ws)
    return fail(CGPS_ERR_ARG, "cgps_solve: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    const size_t crr = align_up((size_t)N * D * sizeof(T));
    if (ws_bytes < crr) return fail(CGPS_ERR_ARG, "workspace too small");
    T* xcrr = (T*)ws;
    int rc = run_halfsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, (const T*)y, xcrr, (char*)ws + crr,
                                 ws_bytes - crr, nullptr, (hipStream_t)stream);
    if (rc != CGPS_OK) return rc;
    return run_backsolve<T, D>((const T*)Dp, (const T*)Fp, (const T*)Gp, N, xcrr, (T*)x, (char*)ws + crr,
                               ws_bytes - crr, (hipStream_t)stream);
  });
}

int cgps_logdet_factor(const void* Dp, int64_t N, int d, int dtype, void* ws, size_t ws_bytes, double* out,
                       void* stream) {
  if (bad_common(N, d) || !Dp || !ws || !out) return fail(CGPS_ERR_ARG, "cgps_logdet_factor: null pointer or N < 1");
  if (ws_bytes < (size_t)(1024 + 2) * 16) return fail(CGPS_ERR_ARG, "workspace too small");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)ws;
    int64_t nb = (N * D + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL((cgps::logdiag_kernel<T, D>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)Dp, N, partial);
    double* tmp = partial + 2 * nb;
    hipLaunchKernelGGL(cgps::sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, nb, tmp);
    hipMemcpyAsync(out, tmp + 1, sizeof(double), hipMemcpyDeviceToDevice, st);
    return check_launch("logdet_factor");
  });
}

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

int cgps_peg_precision(const void* ts, const void* G, int64_t N, int d, int dtype, void* Rs, void* Os, int* info,
                       void* stream) {
  if (bad_common(N, d) || !ts || !G || !Rs || (N > 1 && !Os) || !info)
    return fail(CGPS_ERR_ARG, "cgps_peg_precision: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(info, 0, sizeof(int), st);
    const int64_t nb = (N + cgps::LEG_THREADS - 1) / cgps::LEG_THREADS;
    hipLaunchKernelGGL((cgps::peg_precision_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::LEG_THREADS), 0, st,
                       (const T*)ts, (const T*)G, N, (T*)Rs, (T*)Os, info);
    return check_launch("peg_precision");
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
