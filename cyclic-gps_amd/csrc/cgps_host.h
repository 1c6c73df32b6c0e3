// Host-side plumbing shared by the translation units of libcgps (one .hip file per group of
// entry points, so that the library builds in parallel): error string, dtype / block-size
// dispatch, level layout, workspace carve-up, per-device one-time setup.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "../../include/cgps.h"
#include "cgps_level.h"

namespace cgps_host {

// defined once, in cgps_core.hip
extern thread_local char g_err[512];
extern thread_local hipEvent_t g_prof_start, g_prof_stop;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
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

// ---- per-device one-time setup -------------------------------------------------------------------
// Function attributes (dynamic-LDS limits) and occupancy-derived grid sizes belong to a device,
// not to the process: a process that drives several GPUs, or two host threads racing the first
// call, must each get them right.  One slot per (kernel family, device), filled under call_once.
constexpr int MAX_DEVICES = 64;
inline int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return (dev >= 0 && dev < MAX_DEVICES) ? dev : 0;
}
template <typename V>
struct PerDevice {
  std::once_flag once[MAX_DEVICES];
  V value[MAX_DEVICES];
  template <typename Fn>
  const V& get(Fn&& make) {            // make(int device) -> V, run once per device
    const int dev = current_device();
    std::call_once(once[dev], [&] { value[dev] = make(dev); });
    return value[dev];
  }
};
inline int device_cus(int dev) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus > 0 ? cus : 256;
}

// ---- level layout of the packed factor ----------------------------------------------------------
struct Layout {
  int nlevels;
  int64_t ms[CGPS_MAX_LEVELS], offD[CGPS_MAX_LEVELS + 1], offF[CGPS_MAX_LEVELS + 1], offG[CGPS_MAX_LEVELS + 1];
};

inline void make_layout(int64_t N, Layout& L) {
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

inline LevelWs level_ws(int64_t N, int d, size_t s, bool with_mats, bool with_vec, int nrhs = 1) {
  LevelWs w{};
  Layout L;
  make_layout(N, L);
  int64_t nb = 0;
  for (int l = 0; l < L.nlevels; ++l) nb += level_blocks(L.ms[l]);
  w.partial_off = 0;
  w.partial_bytes = align_up((size_t)(nb + 1) * 16);
  w.capA = N / 2 + 1;
  w.capB = N / 4 + 1;
  size_t per_row = (with_mats ? 2 * (size_t)d * d : 0) + (with_vec ? (size_t)d * nrhs : 0);
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

// cgps_decompose_solve keeps the right-hand side of the rows that survive its first pass, and what each tile owes the
// previous one, behind the workspace of the ops it runs one after the other
inline size_t decompose_solve_tail_bytes(int64_t N, int d, size_t s) {
  return align_up((size_t)(N / 8 + 16) * d * s) + align_up((size_t)(N / 128 + 2) * d * s);
}

inline bool bad_common(int64_t N, int d) { return N < 1 || d < 1; }

inline bool levelwise_solve_requested() {
  static const int v = [] {
    const char* e = getenv("CGPS_LEVELWISE_SOLVE");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return v == 1;
}

// ---- one launch per reduction level (the simple, always-correct form) -----------------------------
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
  (void)hipMemsetAsync(info, 0, sizeof(int), st);
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
    if (l == 0 && g_prof_start) (void)hipEventRecord(g_prof_start, st);
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
      (void)hipEventRecord(g_prof_stop, st);
      g_prof_start = g_prof_stop = nullptr;
    }
    pb += nb;
    R = nx.R; O = nx.O; y = nx.y;
  }
  if (out2) hipLaunchKernelGGL(cgps::sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, pb, out2);
  return check_launch("levelwise reduction");
}

}  // namespace cgps_host
