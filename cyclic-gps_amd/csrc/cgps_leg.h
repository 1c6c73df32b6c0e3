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
// exp(A): scaling and squaring around a Taylor polynomial in Horner form (degree 13 in fp64, 8 in fp32: with ||A||_1
// <= 1/2 the remainder 0.5^(m+1) / (m+1)! is 7e-16 / 5e-9, below the rounding of the format; rounds 1 and 2 ran
// degree 18 in both, the degree torch.matrix_exp uses in fp64 for norms up to 5.4), ||A||_1 scaled below 1/2; the
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

template <typename T> constexpr int exp_taylor_degree() { return sizeof(T) == 8 ? 13 : 8; }

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
  for (int k = exp_taylor_degree<T>(); k >= 1; --k) {
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


// ---- prediction glue (the step AFTER the cyclic reduction) ----------------------------
// Posterior of the latent at arbitrary target times from the in-sample posterior (mean[n][d], diagonal
// covariance blocks P[n][d][d], lower off-diagonal blocks C[n-1][d][d] = Cov(z_{i+1}, z_i)): the
// reference walks the targets in a Python loop (intercast, models.py:455-514: searchsorted, one or two
// matrix exponentials, one gaussian_stitch of a 2d x 2d or 3d x 3d Gaussian, model_utils.py:64-107).
// Here one lane per target, everything in registers, and the stitches reduced to d x d algebra:
//   forecast from a neighbour with posterior N(mu, P) over a gap with E = exp(-1/2 gap G)
//   (E transposed when the target lies BEFORE the first observation):
//       mean = E mu,   cov = I - E E^T + E P E^T
//   interpolation between neighbours j-1 (mu0, P0) and j (mu1, P1), C = Cov(z_j, z_j-1),
//   E1 = exp(-1/2 (t - t_j-1) G), E2 = exp(-1/2 (t_j - t) G), E3 = E1 E2: the prior of
//   (z_j-1, z_j) has covariance [[I, E3^T], [E3, I]], whose inverse through the Schur complement
//   S = I - E3 E3^T (symmetric positive definite) gives the 2d-wide mean transformer as two d x d blocks
//       W = (E2^T - E1 E3^T) S^-1,   Ma = E1 - W E3,   Mb = W
//       mean = Ma mu0 + Mb mu1
//       cov  = I - (Ma E1^T + Mb E2) + Ma P0 Ma^T + Ma C^T Mb^T + Mb C Ma^T + Mb P1 Mb^T
// Branches as the reference takes them: t before the first / after the last observation -> forecast;
// t at the first / last observation (torch.allclose defaults: |a - b| <= 1e-8 + 1e-5 |b|) -> the
// in-sample values; otherwise interpolation between searchsorted's neighbours.
template <typename T, int D>
__device__ __forceinline__ void leg_forecast(const T (&E)[D][D], const T* __restrict__ mu, const T* __restrict__ Pg,
                                             T (&mean)[D], T (&cov)[D][D]) {
  T P[D][D], EP[D][D];
  load_block<T, D>(Pg, P);
  mat_mul<T, D>(EP, E, P);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    T s = T(0);
#pragma unroll
    for (int m = 0; m < D; ++m) s = fmaT(E[i][m], mu[m], s);
    mean[i] = s;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T c = (i == j) ? T(1) : T(0);
#pragma unroll
      for (int m = 0; m < D; ++m) c = fmaT(EP[i][m] - E[i][m], E[j][m], c);   // + (E P - E) E^T
      cov[i][j] = c;
    }
  }
}

template <typename T, int D>
__global__ __launch_bounds__(LEG_THREADS) void leg_intercast_kernel(const T* __restrict__ ts, int64_t n,
                                                                    const T* __restrict__ tt, int64_t p,
                                                                    const T* __restrict__ Gg, const T* __restrict__ mu,
                                                                    const T* __restrict__ Pd, const T* __restrict__ Co,
                                                                    T* __restrict__ out_mean, T* __restrict__ out_cov) {
  constexpr int DD = D * D;
  const int64_t k = (int64_t)blockIdx.x * LEG_THREADS + threadIdx.x;
  if (k >= p) return;
  const T t = tt[k];
  int64_t lo = 0, hi = n;                          // first index with ts[idx] >= t (torch.searchsorted, right = False)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ts[mid] < t) lo = mid + 1; else hi = mid;
  }
  const int64_t idx = lo;
  auto close = [](T a, T b) {
    const T dlt = a > b ? a - b : b - a, ab = b < T(0) ? -b : b;
    return dlt <= T(1e-8) + T(1e-5) * ab;
  };
  const T t_first = ts[0], t_last = ts[n - 1];
  const bool at_first = idx == 0 && close(t, t_first);
  const bool at_last = idx > 0 && close(t, t_last);
  const bool back = idx == 0 && !at_first, fwd = idx == n && !at_last;
  T mean[D], cov[D][D];
  if (at_first || at_last) {
    const int64_t r = at_first ? 0 : n - 1;
    load_vec<T, D>(mu + r * D, mean);
    load_block<T, D>(Pd + r * DD, cov);
  } else {
    T G[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) G[a][b] = Gg[a * D + b];
    auto exp_gap = [&](T gap, T (&E)[D][D]) {
      T A[D][D];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) A[a][b] = T(-0.5) * gap * G[a][b];
      mat_exp<T, D>(E, A);
    };
    if (back || fwd) {
      T E[D][D];
      exp_gap(back ? t_first - t : t - t_last, E);
      if (back) {
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = a + 1; b < D; ++b) { const T v = E[a][b]; E[a][b] = E[b][a]; E[b][a] = v; }
      }
      const int64_t r = back ? 0 : n - 1;
      T m0[D];
      load_vec<T, D>(mu + r * D, m0);
      leg_forecast<T, D>(E, m0, Pd + r * DD, mean, cov);
    } else {
      int64_t j = idx < 1 ? 1 : idx;
      if (j > n - 1) j = n - 1;
      T E1[D][D], E2[D][D], E3[D][D], W[D][D], Ma[D][D];
      exp_gap(t - ts[j - 1], E1);
      exp_gap(ts[j] - t, E2);
      mat_mul<T, D>(E3, E1, E2);
      {
        T S[D][D], Xt[D][D], Wt[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) {
            T s = T(0), x = E2[a][b];              // X^T[a][b] = X[b][a] = E2[a][b] - (E1 E3^T)[b][a]
#pragma unroll
            for (int m = 0; m < D; ++m) {
              s = fmaT(E3[a][m], E3[b][m], s);     // (E3 E3^T)[a][b]
              x = fmaT(-E1[b][m], E3[a][m], x);
            }
            S[a][b] = s;
            Xt[a][b] = x;
          }
        (void)spd_solve_i_minus<T, D>(S, Xt, Wt);  // W^T = S^-1 X^T
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) W[a][b] = Wt[b][a];
      }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          T s = E1[a][b];
#pragma unroll
          for (int m = 0; m < D; ++m) s = fmaT(-W[a][m], E3[m][b], s);
          Ma[a][b] = s;
        }
      T m0[D], m1[D];
      load_vec<T, D>(mu + (j - 1) * D, m0);
      load_vec<T, D>(mu + j * D, m1);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        T s = T(0);
#pragma unroll
        for (int m = 0; m < D; ++m) s = fmaT(Ma[a][m], m0[m], fmaT(W[a][m], m1[m], s));
        mean[a] = s;
      }
      // cov = I - (Ma E1^T + W E2) + (Ma P0 + W C) Ma^T + (Ma C^T + W P1) W^T
      T Q[D][D], U[D][D], V[D][D];
      load_block<T, D>(Pd + (j - 1) * DD, Q);      // P0
      mat_mul<T, D>(U, Ma, Q);
      load_block<T, D>(Co + (j - 1) * DD, Q);      // C
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          T u = U[a][b], v = T(0);
#pragma unroll
          for (int m = 0; m < D; ++m) {
            u = fmaT(W[a][m], Q[m][b], u);         // + W C
            v = fmaT(Ma[a][m], Q[b][m], v);        // Ma C^T
          }
          U[a][b] = u;
          V[a][b] = v;
        }
      load_block<T, D>(Pd + j * DD, Q);            // P1
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          T v = V[a][b];
#pragma unroll
          for (int m = 0; m < D; ++m) v = fmaT(W[a][m], Q[m][b], v);
          V[a][b] = v;
        }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          T c = (a == b) ? T(1) : T(0);
#pragma unroll
          for (int m = 0; m < D; ++m) {
            c = fmaT(-Ma[a][m], E1[b][m], c);      // - Ma E1^T
            c = fmaT(-W[a][m], E2[m][b], c);       // - W E2
            c = fmaT(U[a][m], Ma[b][m], c);        // + U Ma^T
            c = fmaT(V[a][m], W[b][m], c);         // + V W^T
          }
          cov[a][b] = c;
        }
    }
  }
  store_vec<T, D>(out_mean + k * D, mean);
  store_block<T, D>(out_cov + k * DD, cov);
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
  for (int k = exp_taylor_degree<T>(); k >= 1; --k) {
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
