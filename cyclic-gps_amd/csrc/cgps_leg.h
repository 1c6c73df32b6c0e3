// Operand assembly for LEG models (the step BEFORE the cyclic reduction):
// diagonal and lower off-diagonal blocks of the PEG prior precision from the time stamps and the
// d x d generator G (reference models.py:181-239, restated):
//     E_i  = exp(-1/2 (t_{i+1} - t_i) G)
//     a_i  = (I - E_i^T E_i)^-1 E_i^T,     b_i = (I - E_i E_i^T)^-1 E_i
//     Rs_i = I + E_i^T b_i + E_{i-1} a_{i-1},     Os_i = -b_i                     (J[i+1, i] = Os_i)
// Embarrassingly parallel over the time axis: one lane per block row, everything in registers.
// A lane evaluates its own gap (i) and the gap before it (i-1), so no lane waits for a neighbour
// and the result does not depend on the launch geometry.
//
// exp(A): scaling and squaring around a degree-18 Taylor polynomial in Horner form, ||A||_1 scaled
// below 1/2 (truncation < 2e-23, the same polynomial degree torch.matrix_exp uses in fp64); the
// two d x d systems are symmetric positive definite (||E||_2 < 1 because G + G^T is positive
// definite) and go through the Cholesky routines of cgps_math.h.  A gap of zero length makes them
// singular: reported through `info` (1 + row index), like a non-positive-definite block elsewhere.
#pragma once
#include "cgps_math.h"

namespace cgps {

constexpr int LEG_THREADS = 64;

template <typename T, int D>
__device__ __forceinline__ void mat_mul(T (&C)[D][D], const T (&A)[D][D], const T (&B)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T s = T(0);
#pragma unroll
      for (int m = 0; m < D; ++m) s = fmaT(A[i][m], B[m][j], s);
      C[i][j] = s;
    }
}

// E = exp(A); A is destroyed
template <typename T, int D>
__device__ __forceinline__ void mat_exp(T (&E)[D][D], T (&A)[D][D]) {
  T nrm = T(0);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T c = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) c += A[i][j] < T(0) ? -A[i][j] : A[i][j];
    nrm = c > nrm ? c : nrm;
  }
  int s = 0;
  T scale = T(1);
  while (nrm * scale > T(0.5) && s < 60) { scale *= T(0.5); ++s; }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] *= scale;
  // Horner: E = I + A (I + A/2 (I + A/3 (...)))
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) E[i][j] = (i == j) ? T(1) : T(0);
#pragma unroll 1
  for (int k = 18; k >= 1; --k) {
    T P[D][D];
    mat_mul<T, D>(P, A, E);
    const T rk = T(1) / T(k);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) E[i][j] = ((i == j) ? T(1) : T(0)) + P[i][j] * rk;
  }
#pragma unroll 1
  for (int q = 0; q < s; ++q) {
    T P[D][D];
    mat_mul<T, D>(P, E, E);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) E[i][j] = P[i][j];
  }
}

// X = (I - S)^-1 B for the symmetric S (lower triangle read); returns false when I - S is not
// positive definite
template <typename T, int D>
__device__ __forceinline__ bool spd_solve_i_minus(const T (&S)[D][D], const T (&B)[D][D], T (&X)[D][D]) {
  T M[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) M[i][j] = ((i == j) ? T(1) : T(0)) - S[i][j];
  Chol<T, D> c;
  bool f = false;
  chol_lower<T, D>(M, c, f);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T v[D];
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = B[i][j];
    fwd_subst<T, D>(c, v);
    bwd_subst<T, D>(c, v);
#pragma unroll
    for (int i = 0; i < D; ++i) X[i][j] = v[i];
  }
  return !f;
}

// the two contributions of one time gap: toRight = E a (goes to the row after the gap),
// toLeft = E^T b (to the row before it), b itself (the coupling is -b)
template <typename T, int D>
__device__ __forceinline__ bool gap_terms(T dt, const T (&G)[D][D], bool want_right, bool want_left, T (&toRight)[D][D],
                                          T (&toLeft)[D][D], T (&b)[D][D]) {
  T A[D][D], E[D][D], Et[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = T(-0.5) * dt * G[i][j];
  mat_exp<T, D>(E, A);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) Et[i][j] = E[j][i];
  bool ok = true;
  if (want_right) {
    T S[D][D], a[D][D];
    mat_mul<T, D>(S, Et, E);                       // E^T E
    ok = spd_solve_i_minus<T, D>(S, Et, a) && ok;  // a = (I - E^T E)^-1 E^T
    mat_mul<T, D>(toRight, E, a);
  }
  if (want_left) {
    T S[D][D];
    mat_mul<T, D>(S, E, Et);                       // E E^T
    ok = spd_solve_i_minus<T, D>(S, E, b) && ok;   // b = (I - E E^T)^-1 E
    mat_mul<T, D>(toLeft, Et, b);
  }
  return ok;
}

template <typename T, int D>
__global__ __launch_bounds__(LEG_THREADS) void peg_precision_kernel(const T* __restrict__ ts, const T* __restrict__ Gg,
                                                                    int64_t N, T* __restrict__ Rs, T* __restrict__ Os,
                                                                    int* __restrict__ info) {
  constexpr int DD = D * D;
  const int64_t i = (int64_t)blockIdx.x * LEG_THREADS + threadIdx.x;
  if (i >= N) return;
  T G[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) G[a][b] = Gg[a * D + b];
  T R[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) R[a][b] = (a == b) ? T(1) : T(0);
  bool ok = true;
  if (i + 1 < N) {                                 // the gap after this row
    T c1[D][D], c2[D][D], bb[D][D];
    ok = gap_terms<T, D>(ts[i + 1] - ts[i], G, false, true, c1, c2, bb) && ok;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) { R[a][b] += c2[a][b]; bb[a][b] = -bb[a][b]; }
    store_block<T, D>(Os + i * DD, bb);
  }
  if (i >= 1) {                                    // the gap before it
    T c1[D][D], c2[D][D], bb[D][D];
    ok = gap_terms<T, D>(ts[i] - ts[i - 1], G, true, false, c1, c2, bb) && ok;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) R[a][b] += c1[a][b];
  }
  store_block<T, D>(Rs + i * DD, R);
  if (!ok) report_fail(info, i);
}

}  // namespace cgps
