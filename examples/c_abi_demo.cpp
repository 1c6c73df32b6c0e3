// The C ABI of libcgps.so used from a plain C++ / HIP host program: no Python, no torch.
//
//   hipcc -O2 -I include examples/c_abi_demo.cpp -L cyclic-gps_amd/lib -lcgps \
//         -Wl,-rpath,$PWD/cyclic-gps_amd/lib -o /tmp/c_abi_demo && /tmp/c_abi_demo [N]
//
// Builds J = L L^T with L block lower-bidiagonal (so log|J| and the solution of J x = b are known
// in closed form), hands device pointers to cgps_mahal_logdet / cgps_decompose / cgps_solve / cgps_decompose_solve on a
// stream of its own, and checks the results; then the LEG reductions with the operands assembled inside the launch
// (cgps_leg_mahal_logdet_pair) against cgps_peg_precision + cgps_mahal_logdet.  Exit code 0 = all checks passed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cgps.h"

#define HIP_OK(x)                                                                 \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
#define CGPS_OK_(x)                                                               \
  do {                                                                            \
    int r_ = (x);                                                                 \
    if (r_ != CGPS_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #x, r_, cgps_last_error());               \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
  const int d = 4;
  const int dd = d * d;
  // L: diagonal blocks Ld_i = 1.5 I + small lower-triangular noise, sub-diagonal blocks Lo_i small
  std::vector<double> Ld(N * dd, 0.0), Lo((N - 1) * dd, 0.0), xt(N * d), Rs(N * dd, 0.0), Os((N - 1) * dd, 0.0), b(N * d, 0.0);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
  double logdet = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    for (int r = 0; r < d; ++r)
      for (int c = 0; c <= r; ++c) Ld[i * dd + r * d + c] = (r == c ? 1.5 : 0.0) + 0.2 * rnd();
    for (int r = 0; r < d; ++r) logdet += 2.0 * std::log(std::fabs(Ld[i * dd + r * d + r]));
    if (i + 1 < N)
      for (int k = 0; k < dd; ++k) Lo[i * dd + k] = 0.3 * rnd();
    for (int r = 0; r < d; ++r) xt[i * d + r] = 2.0 * rnd();
  }
  // J = L L^T: R_i = Ld_i Ld_i^T + Lo_{i-1} Lo_{i-1}^T,  O_i = J[i+1, i] = Lo_i Ld_i^T;  b = J x_true
  for (int64_t i = 0; i < N; ++i)
    for (int r = 0; r < d; ++r)
      for (int c = 0; c < d; ++c) {
        double v = 0.0;
        for (int k = 0; k < d; ++k) v += Ld[i * dd + r * d + k] * Ld[i * dd + c * d + k];
        if (i > 0)
          for (int k = 0; k < d; ++k) v += Lo[(i - 1) * dd + r * d + k] * Lo[(i - 1) * dd + c * d + k];
        Rs[i * dd + r * d + c] = v;
        if (i + 1 < N) {
          double o = 0.0;
          for (int k = 0; k < d; ++k) o += Lo[i * dd + r * d + k] * Ld[i * dd + c * d + k];
          Os[i * dd + r * d + c] = o;
        }
      }
  double mahal = 0.0;
  for (int64_t i = 0; i < N; ++i)
    for (int r = 0; r < d; ++r) {
      double v = 0.0;
      for (int c = 0; c < d; ++c) {
        v += Rs[i * dd + r * d + c] * xt[i * d + c];
        if (i > 0) v += Os[(i - 1) * dd + r * d + c] * xt[(i - 1) * d + c];
        if (i + 1 < N) v += Os[i * dd + c * d + r] * xt[(i + 1) * d + c];
      }
      b[i * d + r] = v;
      mahal += v * xt[i * d + r];
    }

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  double *dR, *dO, *db, *dx, *dD, *dF, *dG, *dout;
  int* dinfo;
  HIP_OK(hipMalloc(&dR, Rs.size() * 8));
  HIP_OK(hipMalloc(&dO, (Os.size() + 1) * 8));
  HIP_OK(hipMalloc(&db, b.size() * 8));
  HIP_OK(hipMalloc(&dx, b.size() * 8));
  HIP_OK(hipMalloc(&dout, 2 * 8));
  HIP_OK(hipMalloc(&dinfo, 4));
  HIP_OK(hipMemcpy(dR, Rs.data(), Rs.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dO, Os.data(), Os.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));

  // sizes of the packed factor and of the scratch: asked from the library, allocated by the caller
  int nlev = 0;
  int64_t ms[CGPS_MAX_LEVELS], oD[CGPS_MAX_LEVELS + 1], oF[CGPS_MAX_LEVELS + 1], oG[CGPS_MAX_LEVELS + 1];
  CGPS_OK_(cgps_level_layout(N, &nlev, ms, oD, oF, oG));
  HIP_OK(hipMalloc(&dD, (oD[nlev] + 1) * dd * 8));
  HIP_OK(hipMalloc(&dF, (oF[nlev] + 1) * dd * 8));
  HIP_OK(hipMalloc(&dG, (oG[nlev] + 1) * dd * 8));
  size_t ws_bytes = 0;
  for (int op = 0; op <= CGPS_OP_DECOMPOSE_SOLVE; ++op) {
    size_t wsz = 0;
    CGPS_OK_(cgps_workspace_bytes(N, d, CGPS_F64, op, &wsz));
    if (wsz > ws_bytes) ws_bytes = wsz;
  }
  void* ws;
  HIP_OK(hipMalloc(&ws, ws_bytes));

  int bad = 0, info = 0;
  double out[2];
  CGPS_OK_(cgps_mahal_logdet(dR, dO, db, N, d, CGPS_F64, ws, ws_bytes, dout, dinfo, st));
  HIP_OK(hipMemcpyAsync(out, dout, 16, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(&info, dinfo, 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  printf("mahal_and_det : mahal %.12e (true %.12e)  logdet %.12e (true %.12e)  info %d\n", out[0], mahal, out[1], logdet, info);
  if (info != 0 || std::fabs(out[0] - mahal) > 1e-9 * std::fabs(mahal) || std::fabs(out[1] - logdet) > 1e-9 * std::fabs(logdet)) ++bad;

  CGPS_OK_(cgps_decompose(dR, dO, N, d, CGPS_F64, dD, dF, dG, ws, ws_bytes, dinfo, st));
  CGPS_OK_(cgps_solve(dD, dF, dG, N, d, CGPS_F64, /*nrhs*/ 1, db, dx, ws, ws_bytes, st));
  CGPS_OK_(cgps_logdet_factor(dD, N, d, CGPS_F64, ws, ws_bytes, dout, st));
  std::vector<double> x(N * d);
  HIP_OK(hipMemcpyAsync(x.data(), dx, x.size() * 8, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(out, dout, 8, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(&info, dinfo, 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  double err = 0.0;
  for (size_t i = 0; i < x.size(); ++i) err = std::fmax(err, std::fabs(x[i] - xt[i]));
  printf("decompose+solve: max |x - x_true| %.3e  logdet(factor) %.12e  info %d  (%d levels)\n", err, out[0], info, nlev);
  if (info != 0 || err > 1e-9 || std::fabs(out[0] - logdet) > 1e-9 * std::fabs(logdet)) ++bad;

  // factor and solve in ONE call (cgps_decompose_solve): the same solution, the same factor
  {
    double *dD2, *dF2, *dG2, *dxcrr, *dx2;
    HIP_OK(hipMalloc(&dD2, (oD[nlev] + 1) * dd * 8));
    HIP_OK(hipMalloc(&dF2, (oF[nlev] + 1) * dd * 8));
    HIP_OK(hipMalloc(&dG2, (oG[nlev] + 1) * dd * 8));
    HIP_OK(hipMalloc(&dxcrr, b.size() * 8));
    HIP_OK(hipMalloc(&dx2, b.size() * 8));
    CGPS_OK_(cgps_decompose_solve(dR, dO, db, N, d, CGPS_F64, dD2, dF2, dG2, dxcrr, dx2, ws, ws_bytes, dinfo, st));
    std::vector<double> x2(N * d), D1(oD[nlev] * dd), D2(oD[nlev] * dd);
    HIP_OK(hipMemcpyAsync(x2.data(), dx2, x2.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(D1.data(), dD, D1.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(D2.data(), dD2, D2.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&info, dinfo, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    double err2 = 0.0;
    size_t differ = 0;
    for (size_t i = 0; i < x2.size(); ++i) err2 = std::fmax(err2, std::fabs(x2[i] - xt[i]));
    for (size_t i = 0; i < D1.size(); ++i) differ += (D1[i] != D2[i]);
    printf("decompose_solve: max |x - x_true| %.3e  factor entries that differ from cgps_decompose's: %zu  info %d\n", err2, differ, info);
    if (info != 0 || err2 > 1e-9 || differ != 0) ++bad;
  }

  // the two reductions of a LEG log-likelihood with the operands assembled inside the launch (cgps_leg_mahal_logdet_pair)
  // against cgps_peg_precision (blocks in memory) + cgps_mahal_logdet: the same numbers
  {
    std::vector<double> ts(N), G(dd, 0.0), A(dd, 0.0);
    double t = 0.0;
    for (int64_t i = 0; i < N; ++i) { t += 0.2 + 0.5 * (rnd() + 0.5); ts[i] = t; }
    for (int r = 0; r < d; ++r) {                       // G = N N^T + R - R^T: symmetric part well conditioned
      G[r * d + r] = 0.8 + 0.1 * r;
      for (int c = 0; c < r; ++c) { G[r * d + c] = 0.15 + 0.05 * c; G[c * d + r] = -0.05 * r; }
      A[r * d + r] = 0.5;
    }
    double *dts, *dGm, *dA, *dout4, *dR2, *dO2;
    int* dinfo2;
    HIP_OK(hipMalloc(&dts, N * 8));
    HIP_OK(hipMalloc(&dGm, dd * 8));
    HIP_OK(hipMalloc(&dA, dd * 8));
    HIP_OK(hipMalloc(&dout4, 4 * 8));
    HIP_OK(hipMalloc(&dinfo2, 8));
    HIP_OK(hipMalloc(&dR2, Rs.size() * 8));
    HIP_OK(hipMalloc(&dO2, (Os.size() + 1) * 8));
    HIP_OK(hipMemcpy(dts, ts.data(), N * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dGm, G.data(), dd * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dA, A.data(), dd * 8, hipMemcpyHostToDevice));
    size_t one = 0;
    CGPS_OK_(cgps_workspace_bytes(N, d, CGPS_F64, CGPS_OP_MAHAL_LOGDET, &one));
    const size_t half = (one + 255) / 256 * 256;
    void* ws2;
    HIP_OK(hipMalloc(&ws2, 2 * half));
    CGPS_OK_(cgps_leg_mahal_logdet_pair(dts, dGm, dA, db, N, d, CGPS_F64, ws2, 2 * half, dout4, dinfo2, st));
    double o4[4], ref[2], refp[2];
    HIP_OK(hipMemcpyAsync(o4, dout4, 32, hipMemcpyDeviceToHost, st));
    // blocks in memory: prior precision, then + A on every diagonal block (on the host, it is a check)
    CGPS_OK_(cgps_peg_precision(dts, dGm, N, d, CGPS_F64, dR2, dO2, dinfo, st));
    CGPS_OK_(cgps_mahal_logdet(dR2, dO2, db, N, d, CGPS_F64, ws, ws_bytes, dout, dinfo, st));
    HIP_OK(hipMemcpyAsync(refp, dout, 16, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> R2(Rs.size());
    HIP_OK(hipMemcpy(R2.data(), dR2, R2.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i)
      for (int r = 0; r < d; ++r) R2[i * dd + r * d + r] += 0.5;
    HIP_OK(hipMemcpy(dR2, R2.data(), R2.size() * 8, hipMemcpyHostToDevice));
    CGPS_OK_(cgps_mahal_logdet(dR2, dO2, db, N, d, CGPS_F64, ws, ws_bytes, dout, dinfo, st));
    HIP_OK(hipMemcpyAsync(ref, dout, 16, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    printf("leg pair      : mahal %.12e (blocks in memory %.12e)  log|K| %.12e (%.12e)  log|prior| %.12e (%.12e)\n", o4[0], ref[0],
           o4[1], ref[1], o4[3], refp[1]);
    auto close = [](double a, double c) { return std::fabs(a - c) <= 1e-9 * std::fmax(1.0, std::fabs(c)); };
    if (!close(o4[0], ref[0]) || !close(o4[1], ref[1]) || !close(o4[3], refp[1])) ++bad;
  }

  printf(bad ? "FAILED\n" : "OK (libcgps version %d)\n", cgps_version());
  return bad ? 1 : 0;
}
