// cgps_core.hip -- version / errors / layout / workspace sizes, level-at-a-time entry points
// One translation unit of libcgps (include/cgps.h); host code only decides sizes/offsets and
// enqueues kernels on the caller's stream: nothing here allocates, copies to the host or synchronises.
#include "cgps_host.h"
#include "cgps_tile_sizes.h"

using namespace cgps_host;

namespace cgps_host {
thread_local char g_err[512] = "";
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
}  // namespace cgps_host

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
    case CGPS_OP_DECOMPOSE_SOLVE: {
      size_t a = 0, b = 0, c = 0;
      (void)cgps_workspace_bytes(N, d, dtype, CGPS_OP_DECOMPOSE, &a);
      (void)cgps_workspace_bytes(N, d, dtype, CGPS_OP_HALFSOLVE, &b);
      (void)cgps_workspace_bytes(N, d, dtype, CGPS_OP_BACKSOLVE, &c);
      const size_t m = a > b ? (a > c ? a : c) : (b > c ? b : c);
      *bytes = align_up(m) + decompose_solve_tail_bytes(N, d, s);
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

}  // extern "C"
