// cgps_mahal.hip -- fused solve + log-det (mahal_and_det), shard reduce / finish
// One translation unit of libcgps (include/cgps.h); host code only decides sizes/offsets and
// enqueues kernels on the caller's stream: nothing here allocates, copies to the host or synchronises.
#include "cgps_host.h"
#include "cgps_tile.h"
#include "cgps_boundary.h"

using namespace cgps_host;

extern "C" {

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
      if (record_stride_bytes % 16 != 0 || partial_stride_bytes % sizeof(double) != 0 ||
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

int cgps_leg_mahal_logdet(const void* ts, const void* G, const void* A, const void* v, int64_t N, int d, int dtype, void* ws,
                          size_t ws_bytes, double* out2, int* info, void* stream) {
  if (bad_common(N, d) || !ts || !G || !ws || !out2 || !info)
    return fail(CGPS_ERR_ARG, "cgps_leg_mahal_logdet: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    const int rc = cgps::run_tile_leg<T, D>((const T*)ts, (const T*)G, (const T*)A, (const T*)v, N, (char*)ws, ws_bytes, out2,
                                            info, (hipStream_t)stream);
    if (rc == -1) return fail(CGPS_ERR_ARG, "workspace too small for cgps_leg_mahal_logdet");
    if (rc == -2) return fail(CGPS_ERR_UNSUPPORTED, "cgps_leg_mahal_logdet: not built for this block size (d = 8, fp64 d = 6) or CGPS_NO_FOLD=1");
    return check_launch("LEG tile reduction");
  });
}

int cgps_leg_mahal_logdet_pair(const void* ts, const void* G, const void* A, const void* v, int64_t N, int d, int dtype,
                               void* ws, size_t ws_bytes, double* out4, int* info2, void* stream) {
  if (bad_common(N, d) || !ts || !G || !ws || !out4 || !info2)
    return fail(CGPS_ERR_ARG, "cgps_leg_mahal_logdet_pair: null pointer or N < 1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    const int rc = cgps::run_tile_leg<T, D>((const T*)ts, (const T*)G, (const T*)A, (const T*)v, N, (char*)ws, ws_bytes, out4,
                                            info2, (hipStream_t)stream, true);
    if (rc == -1) return fail(CGPS_ERR_ARG, "workspace too small for cgps_leg_mahal_logdet_pair (twice cgps_mahal_logdet's, each rounded up to 256 bytes)");
    if (rc == -2) return fail(CGPS_ERR_UNSUPPORTED, "cgps_leg_mahal_logdet_pair: not built for this block size (d = 8, fp64 d = 6) or CGPS_NO_FOLD=1");
    return check_launch("LEG tile reduction (pair)");
  });
}

int cgps_boundary_solve(const void* records, size_t record_stride_bytes, int64_t P, int d, int dtype, void* xsep, int* info,
                        void* stream) {
  if (P < 1 || d < 1 || !records || !xsep || !info) return fail(CGPS_ERR_ARG, "cgps_boundary_solve: null pointer or P < 1");
  if (P > cgps::BOUNDARY_MAX_P) return fail(CGPS_ERR_UNSUPPORTED, "cgps_boundary_solve: P = %lld records, at most %d", (long long)P, cgps::BOUNDARY_MAX_P);
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if (record_stride_bytes % sizeof(T) != 0 || record_stride_bytes < cgps::RecordLayout<T, D>::STRIDE * sizeof(T))
      return fail(CGPS_ERR_ARG, "cgps_boundary_solve: bad record stride");
    hipLaunchKernelGGL((cgps::boundary_solve_kernel<T, D>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const T*)records,
                       (int64_t)(record_stride_bytes / sizeof(T)), (int)P, (T*)xsep, info);
    return check_launch("boundary solve");
  });
}

int cgps_boundary_recursions(const void* records, size_t record_stride_bytes, int64_t P, int64_t rank, int d, int dtype,
                             void* out, int* info, void* stream) {
  if (P < 1 || rank < 0 || rank >= P || d < 1 || !records || !out || !info)
    return fail(CGPS_ERR_ARG, "cgps_boundary_recursions: null pointer, P < 1 or rank outside 0..P-1");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    if (record_stride_bytes % sizeof(T) != 0 || record_stride_bytes < cgps::RecordLayout<T, D>::STRIDE * sizeof(T))
      return fail(CGPS_ERR_ARG, "cgps_boundary_recursions: bad record stride");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(info, 0, sizeof(int), st);
    hipLaunchKernelGGL((cgps::boundary_recursions_kernel<T, D>), dim3(1), dim3(128), 0, st, (const T*)records,
                       (int64_t)(record_stride_bytes / sizeof(T)), (int)P, (int)rank, (T*)out, info);
    return check_launch("boundary recursions");
  });
}

int cgps_reset_counters(void* stream) {
  if (cgps::fold_reset_counters((hipStream_t)stream) != hipSuccess) return check_launch("cgps_reset_counters");
  return CGPS_OK;
}

}  // extern "C"
