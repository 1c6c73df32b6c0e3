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


// ---- adjoint of the assembly (training through the path, reference models.py:374-381) -----------------
// Given gRs[N] = d loss / d Rs and gOs[N-1] = d loss / d Os, one lane per time gap i recomputes the
// gap's E, a, b and walks the expressions above backwards:
//   c2 = E^T b -> Rs_i  (upstream U2 = gRs_i),  c1 = E a -> Rs_{i+1}  (U1 = gRs_{i+1}),  Os_i = -b
//   bbar = E U2 - gOs_i ;          W = (I - E E^T)^-1 bbar ;  Ebar  = b U2^T + W + (W b^T + b W^T) E
//   abar = E^T U1 ;                V = (I - E^T E)^-1 abar ;  Ebar += U1 a^T + V^T + E (a V^T + V a^T)
//   E = exp(A), A = -1/2 tau G:    Abar = L_exp(A^T, Ebar)    (the adjoint of the Frechet derivative of
//                                   exp at A is the Frechet derivative at A^T), by the same scaling and
//                                   squaring, carried for the pair (exp, derivative)
//   Gbar += -1/2 tau Abar ;        taubar = -1/2 <Abar, G>
// Gbar is summed over the lanes of a workgroup in a fixed order and written per workgroup
// (gG_partial[block][d][d]: the caller adds the few partial sums -- deterministic, no atomics);
// gtau[i] = d loss / d (t_{i+1} - t_i).

// (E, L) <- (exp(A), L_exp(A, dA)); A and dA are destroyed
template <typename T, int D>
__device__ __forceinline__ void mat_exp_frechet(T (&E)[D][D], T (&L)[D][D], T (&A)[D][D], T (&dA)[D][D]) {
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
    for (int j = 0; j < D; ++j) { A[i][j] *= scale; dA[i][j] *= scale; }
  // Horner for the pair: P = I + (A / k) P,  dP = (dA / k) P + (A / k) dP
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) { E[i][j] = (i == j) ? T(1) : T(0); L[i][j] = T(0); }
#pragma unroll 1
  for (int k = 18; k >= 1; --k) {
    T P[D][D], Q[D][D], R[D][D];
    mat_mul<T, D>(P, A, E);
    mat_mul<T, D>(Q, dA, E);
    mat_mul<T, D>(R, A, L);
    const T rk = T(1) / T(k);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        E[i][j] = ((i == j) ? T(1) : T(0)) + P[i][j] * rk;
        L[i][j] = (Q[i][j] + R[i][j]) * rk;
      }
  }
#pragma unroll 1
  for (int q = 0; q < s; ++q) {                          // (E, L) <- (E E, E L + L E)
    T P[D][D], Q[D][D], R[D][D];
    mat_mul<T, D>(Q, E, L);
    mat_mul<T, D>(R, L, E);
    mat_mul<T, D>(P, E, E);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) { E[i][j] = P[i][j]; L[i][j] = Q[i][j] + R[i][j]; }
  }
}

template <typename T, int D>
__global__ __launch_bounds__(LEG_THREADS) void peg_precision_adjoint_kernel(
    const T* __restrict__ ts, const T* __restrict__ Gg, int64_t N, const T* __restrict__ gRs, const T* __restrict__ gOs,
    T* __restrict__ gG_partial, T* __restrict__ gtau) {
  constexpr int DD = D * D;
  __shared__ T red[DD];
  const int64_t i = (int64_t)blockIdx.x * LEG_THREADS + threadIdx.x;      // the gap between rows i and i + 1
  T Gbar[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) Gbar[a][b] = T(0);
  if (i + 1 < N) {
    T G[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) G[a][b] = Gg[a * D + b];
    const T tau = ts[i + 1] - ts[i];
    T E[D][D], Et[D][D];
    {
      T A[D][D];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) A[a][b] = T(-0.5) * tau * G[a][b];
      mat_exp<T, D>(E, A);
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) Et[a][b] = E[b][a];
    T Ebar[D][D];
    {   // through c2 = E^T b (to Rs_i) and Os_i = -b
      T S[D][D], bm[D][D], U2[D][D], gO[D][D], bbar[D][D], W[D][D], X[D][D], Y[D][D];
      mat_mul<T, D>(S, E, Et);
      (void)spd_solve_i_minus<T, D>(S, E, bm);                 // b = (I - E E^T)^-1 E
      load_block<T, D>(gRs + i * DD, U2);
      load_block<T, D>(gOs + i * DD, gO);
      mat_mul<T, D>(bbar, E, U2);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) bbar[a][b] -= gO[a][b];
      (void)spd_solve_i_minus<T, D>(S, bbar, W);               // W = (I - E E^T)^-1 bbar
      // Ebar = b U2^T + W + (W b^T + b W^T) E
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          T s = T(0), z = W[a][b];
#pragma unroll
          for (int m = 0; m < D; ++m) {
            s = fmaT(W[a][m], bm[b][m], s);
            s = fmaT(bm[a][m], W[b][m], s);
            z = fmaT(bm[a][m], U2[b][m], z);
          }
          X[a][b] = s;
          Ebar[a][b] = z;
        }
      mat_mul<T, D>(Y, X, E);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) Ebar[a][b] += Y[a][b];
    }
    {   // through c1 = E a (to Rs_{i+1})
      T S[D][D], am[D][D], U1[D][D], abar[D][D], V[D][D], X[D][D], Y[D][D];
      mat_mul<T, D>(S, Et, E);
      (void)spd_solve_i_minus<T, D>(S, Et, am);                // a = (I - E^T E)^-1 E^T
      load_block<T, D>(gRs + (i + 1) * DD, U1);
      mat_mul<T, D>(abar, Et, U1);
      (void)spd_solve_i_minus<T, D>(S, abar, V);               // V = (I - E^T E)^-1 abar
      // Ebar += U1 a^T + V^T + E (a V^T + V a^T)
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          T s = T(0), z = V[b][a];
#pragma unroll
          for (int m = 0; m < D; ++m) {
            s = fmaT(am[a][m], V[b][m], s);
            s = fmaT(V[a][m], am[b][m], s);
            z = fmaT(U1[a][m], am[b][m], z);
          }
          X[a][b] = s;
          Ebar[a][b] += z;
        }
      mat_mul<T, D>(Y, E, X);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) Ebar[a][b] += Y[a][b];
    }
    // Abar = L_exp(A^T, Ebar)
    T At[D][D], Ex[D][D], Abar[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) At[a][b] = T(-0.5) * tau * G[b][a];
    mat_exp_frechet<T, D>(Ex, Abar, At, Ebar);
    T tb = T(0);
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        Gbar[a][b] = T(-0.5) * tau * Abar[a][b];
        tb = fmaT(Abar[a][b], G[a][b], tb);
      }
    if (gtau != nullptr) gtau[i] = T(-0.5) * tb;
  }
  // sum over the wave (LEG_THREADS = 64: one wave per workgroup), fixed order
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      T v = Gbar[a][b];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (threadIdx.x == 0) red[a * D + b] = v;
    }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < DD; ++q) gG_partial[(size_t)blockIdx.x * DD + q] = red[q];
  }
}

}  // namespace cgps
