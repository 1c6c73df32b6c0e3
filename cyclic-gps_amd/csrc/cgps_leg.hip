// cgps_leg.hip -- operand assembly for LEG models
// One translation unit of libcgps (include/cgps.h); host code only decides sizes/offsets and
// enqueues kernels on the caller's stream: nothing here allocates, copies to the host or synchronises.
#include "cgps_host.h"
#include "cgps_leg.h"

using namespace cgps_host;

extern "C" {

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

int cgps_peg_precision_adjoint(const void* ts, const void* G, int64_t N, int d, int dtype, const void* gRs,
                               const void* gOs, void* gG_partial, void* gtau, void* stream) {
  if (bad_common(N, d) || N < 2 || !ts || !G || !gRs || !gOs || !gG_partial)
    return fail(CGPS_ERR_ARG, "cgps_peg_precision_adjoint: null pointer or N < 2");
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    const int64_t nb = (N - 1 + cgps::LEG_THREADS - 1) / cgps::LEG_THREADS;
    hipLaunchKernelGGL((cgps::peg_precision_adjoint_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::LEG_THREADS), 0,
                       (hipStream_t)stream, (const T*)ts, (const T*)G, N, (const T*)gRs, (const T*)gOs, (T*)gG_partial,
                       (T*)gtau);
    return check_launch("peg_precision_adjoint");
  });
}

int cgps_leg_intercast(const void* ts, int64_t n, const void* target_ts, int64_t p, const void* G, int d, int dtype,
                       const void* ip_mean, const void* ip_cov_diag, const void* ip_cov_offdiag, void* out_mean,
                       void* out_cov, void* stream) {
  if (bad_common(n, d) || p < 0 || !ts || !G || !ip_mean || !ip_cov_diag || (n > 1 && !ip_cov_offdiag) ||
      (p > 0 && (!target_ts || !out_mean || !out_cov)))
    return fail(CGPS_ERR_ARG, "cgps_leg_intercast: null pointer, n < 1 or p < 0");
  if (p == 0) return CGPS_OK;
  return dispatch(dtype, d, [&](auto t, auto dc) {
    using T = decltype(t);
    constexpr int D = decltype(dc)::value;
    const int64_t nb = (p + cgps::LEG_THREADS - 1) / cgps::LEG_THREADS;
    hipLaunchKernelGGL((cgps::leg_intercast_kernel<T, D>), dim3((unsigned)nb), dim3(cgps::LEG_THREADS), 0,
                       (hipStream_t)stream, (const T*)ts, n, (const T*)target_ts, p, (const T*)G, (const T*)ip_mean,
                       (const T*)ip_cov_diag, (const T*)ip_cov_offdiag, (T*)out_mean, (T*)out_cov);
    return check_launch("leg_intercast");
  });
}

}  // extern "C"
