// Level-at-a-time kernels (one launch per reduction level).
//
// These are the general, always-correct forms (any 1 <= d <= 8, any n, fp32/fp64):
// they back decompose_step(), the factor-based solves and inverse_blocks(), and
// serve as the on-device cross-check for the tile-fused kernels in
// cgps_tile.h.  One lane owns one even (eliminated) block row 2k plus, when it
// exists, the odd (surviving) row 2k+1; the Cholesky of the right neighbour
// 2k+2 is recomputed by the lane instead of exchanged (d^3/3 extra flops, no
// cross-lane traffic).
#pragma once
#include "cgps_math.h"

namespace cgps {

constexpr int LEVEL_THREADS = 128;

// One cyclic-reduction level (reference decompose_step, cyclic_reduction.py:203-259,
// fused with the per-level pieces of mahal_and_det :412-427 when RHS).
//   in : R[n], O[n-1], y[n] (RHS)
//   out: EMIT -> Dk[ceil(n/2)], Fk[n/2], Gk[(n-1)/2], xk[ceil(n/2)] (RHS)
//        NEXT -> Rn[n/2], On[n/2-1], yn[n/2] (RHS)
//        partial[blockIdx][2] += {sum x^2, sum log(prod pivots)} over the block's even rows
template <typename T, int D, bool EMIT, bool RHS>
__global__ __launch_bounds__(LEVEL_THREADS) void level_kernel(
    const T* __restrict__ R, const T* __restrict__ O, const T* __restrict__ y, int64_t n, int lvl,
    T* __restrict__ Dk, T* __restrict__ Fk, T* __restrict__ Gk, T* __restrict__ xk,
    T* __restrict__ Rn, T* __restrict__ On, T* __restrict__ yn,
    double* __restrict__ partial, int* __restrict__ info) {
  constexpr int DD = D * D;
  __shared__ double red[2 * (LEVEL_THREADS / 64)];
  const int64_t k = (int64_t)blockIdx.x * LEVEL_THREADS + threadIdx.x;
  const int64_t e = 2 * k;
  double mah = 0.0, logp = 0.0;
  if (e < n) {
    bool fail = false;
    T A[D][D];
    Chol<T, D> c;
    load_block<T, D>(R + e * DD, A);
    double piv = chol_lower<T, D>(A, c, fail);
    if (fail) report_fail(info, ((e + 1) << lvl) - 1);
    logp = log(piv);
    T x[D];
    if constexpr (RHS) {
      load_vec<T, D>(y + e * D, x);
      fwd_subst<T, D>(c, x);
#pragma unroll
      for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
    }
    if constexpr (EMIT) {
      T L[D][D];
      chol_to_dense<T, D>(c, L);
      store_block<T, D>(Dk + k * DD, L);
      if constexpr (RHS) store_vec<T, D>(xk + k * D, x);
    }
    if (e + 1 < n) {
      T F[D][D];
      load_block<T, D>(O + e * DD, F);
      rsolve_lt<T, D>(c, F);                       // F_k = O_2k D_k^-T
      if constexpr (EMIT) store_block<T, D>(Fk + k * DD, F);
      T Ro[D][D];
      load_block<T, D>(R + (e + 1) * DD, Ro);
      syrk_sub_lower<T, D>(Ro, F);
      T yo[D];
      if constexpr (RHS) {
        load_vec<T, D>(y + (e + 1) * D, yo);
        gemv_sub<T, D>(yo, F, x);
      }
      if (e + 2 < n) {
        bool fail2 = false;
        Chol<T, D> c2;
        load_block<T, D>(R + (e + 2) * DD, A);
        chol_lower<T, D>(A, c2, fail2);            // its owner (lane k+1) reports failures
        T Oo[D][D], G[D][D];
        load_block<T, D>(O + (e + 1) * DD, Oo);
        rsolve_lt_transposed<T, D>(c2, Oo, G);     // G_k = O_2k+1^T D_k+1^-T
        if constexpr (EMIT) store_block<T, D>(Gk + k * DD, G);
        syrk_sub_lower<T, D>(Ro, G);
        if constexpr (RHS) {
          T x2[D];
          load_vec<T, D>(y + (e + 2) * D, x2);
          fwd_subst<T, D>(c2, x2);
          gemv_sub<T, D>(yo, G, x2);
        }
        if (e + 3 < n) {
          T F2[D][D], On_[D][D];
          load_block<T, D>(O + (e + 2) * DD, F2);
          rsolve_lt<T, D>(c2, F2);                 // F_k+1
          neg_abt<T, D>(On_, F2, G);               // O'_k = -F_k+1 G_k^T
          store_block<T, D>(On + k * DD, On_);
        }
      }
      mirror_lower<T, D>(Ro);
      store_block<T, D>(Rn + k * DD, Ro);
      if constexpr (RHS) store_vec<T, D>(yn + k * D, yo);
    }
  }
  if (partial == nullptr) return;   // uniform across the grid
  block_sum2<LEVEL_THREADS>(mah, logp, red);
  if (threadIdx.x == 0) {
    partial[2 * (int64_t)blockIdx.x] = mah;
    partial[2 * (int64_t)blockIdx.x + 1] = logp;
  }
}

// out[0] = sum partial[.][0] ; out[1] = sum partial[.][1]   (fixed order: deterministic)
static __global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int64_t count,
                                                           double* __restrict__ out) {
  __shared__ double red[2 * 4];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < count; i += 256) {
    a += partial[2 * i];
    b += partial[2 * i + 1];
  }
  block_sum2<256>(a, b, red);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

// One forward-substitution level with a stored factor (reference halfsolve,
// cyclic_reduction.py:318-336): x_k = D_k^-1 y_2k ; y'_k = y_2k+1 - F_k x_k - G_k x_k+1.
template <typename T, int D>
__global__ __launch_bounds__(LEVEL_THREADS) void halfsolve_level_kernel(
    const T* __restrict__ Dk, const T* __restrict__ Fk, const T* __restrict__ Gk,
    const T* __restrict__ y, int64_t n, T* __restrict__ xk, T* __restrict__ yn, double* __restrict__ partial) {
  constexpr int DD = D * D;
  __shared__ double red[2 * (LEVEL_THREADS / 64)];
  const int64_t k = (int64_t)blockIdx.x * LEVEL_THREADS + threadIdx.x;
  const int64_t e = 2 * k;
  double mah = 0.0, zero = 0.0;
  if (e < n) {
    T L[D][D], x[D];
    Chol<T, D> c;
    load_block<T, D>(Dk + k * DD, L);
    chol_from_dense<T, D>(L, c);
    load_vec<T, D>(y + e * D, x);
    fwd_subst<T, D>(c, x);
    store_vec<T, D>(xk + k * D, x);
#pragma unroll
    for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
    if (e + 1 < n) {
      T yo[D], M[D][D];
      load_vec<T, D>(y + (e + 1) * D, yo);
      load_block<T, D>(Fk + k * DD, M);
      gemv_sub<T, D>(yo, M, x);
      if (e + 2 < n) {
        T x2[D];
        Chol<T, D> c2;
        load_block<T, D>(Dk + (k + 1) * DD, L);
        chol_from_dense<T, D>(L, c2);
        load_vec<T, D>(y + (e + 2) * D, x2);
        fwd_subst<T, D>(c2, x2);
        load_block<T, D>(Gk + k * DD, M);
        gemv_sub<T, D>(yo, M, x2);
      }
      store_vec<T, D>(yn + k * D, yo);
    }
  }
  block_sum2<LEVEL_THREADS>(mah, zero, red);
  if (threadIdx.x == 0 && partial) {
    partial[2 * (int64_t)blockIdx.x] = mah;
    partial[2 * (int64_t)blockIdx.x + 1] = 0.0;
  }
}

// One back-substitution level (reference backhalfsolve, cyclic_reduction.py:362-373):
//   x_even[k] = D_k^-T ( b_k - F_k^T xo[k] - G_k-1^T xo[k-1] ),  X[2k] = x_even[k], X[2k+1] = xo[k]
// b: this level's CRR slice [ceil(n/2), D]; xo: solution of the coarser level [n/2, D]; X: [n, D].
template <typename T, int D>
__global__ __launch_bounds__(LEVEL_THREADS) void backsolve_level_kernel(
    const T* __restrict__ Dk, const T* __restrict__ Fk, const T* __restrict__ Gk,
    const T* __restrict__ b, const T* __restrict__ xo, int64_t n, T* __restrict__ X) {
  constexpr int DD = D * D;
  const int64_t k = (int64_t)blockIdx.x * LEVEL_THREADS + threadIdx.x;
  const int64_t e = 2 * k;
  if (e >= n) return;
  T r[D], M[D][D], L[D][D];
  load_vec<T, D>(b + k * D, r);
  if (e + 1 < n) {
    T xr[D];
    load_vec<T, D>(xo + k * D, xr);
    load_block<T, D>(Fk + k * DD, M);
    gemvT_sub<T, D>(r, M, xr);
    store_vec<T, D>(X + (e + 1) * D, xr);
  }
  if (k >= 1) {
    T xl[D];
    load_vec<T, D>(xo + (k - 1) * D, xl);
    load_block<T, D>(Gk + (k - 1) * DD, M);
    gemvT_sub<T, D>(r, M, xl);
  }
  Chol<T, D> c;
  load_block<T, D>(Dk + k * DD, L);
  chol_from_dense<T, D>(L, c);
  bwd_subst<T, D>(c, r);
  store_vec<T, D>(X + e * D, r);
}

// logdet from a stored factor (reference det, cyclic_reduction.py:447-458):
// partial[block][1] = sum over the block's D blocks of sum_j log D[j][j]  (x2 applied by the caller)
template <typename T, int D>
__global__ __launch_bounds__(256) void logdiag_kernel(const T* __restrict__ Dall, int64_t count,
                                                      double* __restrict__ partial) {
  __shared__ double red[2 * 4];
  double a = 0.0, zero = 0.0;
  const int64_t total = count * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t blk = i / D;
    const int j = (int)(i - blk * D);
    a += log((double)Dall[blk * D * D + j * D + j]);
  }
  block_sum2<256>(zero, a, red);
  if (threadIdx.x == 0) {
    partial[2 * (int64_t)blockIdx.x] = 0.0;
    partial[2 * (int64_t)blockIdx.x + 1] = 2.0 * a;
  }
}

// C = A B (all dense D x D, registers)
template <typename T, int D>
__device__ __forceinline__ void mm(T (&C)[D][D], const T (&A)[D][D], const T (&B)[D][D]) {
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
// C += A^T B
template <typename T, int D>
__device__ __forceinline__ void mm_tn_acc(T (&C)[D][D], const T (&A)[D][D], const T (&B)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T s = C[i][j];
#pragma unroll
      for (int m = 0; m < D; ++m) s = fmaT(A[m][i], B[m][j], s);
      C[i][j] = s;
    }
}
// C += A B
template <typename T, int D>
__device__ __forceinline__ void mm_acc(T (&C)[D][D], const T (&A)[D][D], const T (&B)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T s = C[i][j];
#pragma unroll
      for (int m = 0; m < D; ++m) s = fmaT(A[i][m], B[m][j], s);
      C[i][j] = s;
    }
}

// One level of the selected inverse (reference inverse_blocks, cyclic_reduction.py:478-501),
// coarse -> fine.  With A_k = F_k D_k^-1, B_k = G_k D_k+1^-1, S~ the coarser level's blocks
// (Sd_c[n/2], So_c[n/2-1] lower), M[k,k] = S~[k,k] A_k + S~[k,k-1] B_k-1,
// M[k-1,k] = S~[k-1,k-1] B_k-1 + S~[k,k-1]^T A_k:
//   Sig[2k,2k]   = D_k^-T D_k^-1 + A_k^T M[k,k] + B_k-1^T M[k-1,k]
//   Sig[2k+1,2k] = -M[k,k] ;  Sig[2k,2k-1] = -M[k-1,k]^T ;  Sig[2k+1,2k+1] = S~[k,k]
template <typename T, int D>
__device__ __forceinline__ void inverse_level_row(
    const T* __restrict__ Dk, const T* __restrict__ Fk, const T* __restrict__ Gk,
    const T* __restrict__ Sd_c, const T* __restrict__ So_c, int64_t n, T* __restrict__ Sd, T* __restrict__ So,
    int64_t k) {
  constexpr int DD = D * D;
  const int64_t e = 2 * k;
  if (e >= n) return;
  const int64_t nf = n / 2;
  T L[D][D], Di[D][D];
  Chol<T, D> c;
  load_block<T, D>(Dk + k * DD, L);
  chol_from_dense<T, D>(L, c);
  // Di = D^-1 (lower): column j of D^-1 is fwd_subst of e_j; build rows of Di^T then transpose
  T DiT[D][D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int i = 0; i < D; ++i) DiT[j][i] = (i == j) ? T(1) : T(0);
    fwd_subst<T, D>(c, DiT[j]);
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) Di[i][j] = DiT[j][i];
  T See[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) See[i][j] = T(0);
  mm_tn_acc<T, D>(See, Di, Di);                       // D^-T D^-1
  const bool has_odd = (k < nf);
  const bool has_left = (k >= 1);
  T Ak[D][D], Bk[D][D], Soc[D][D], Sc[D][D], M[D][D];
  if (has_odd) {
    load_block<T, D>(Fk + k * DD, L);
    mm<T, D>(Ak, L, Di);                               // A_k
  }
  if (has_left) {
    load_block<T, D>(Gk + (k - 1) * DD, L);
    mm<T, D>(Bk, L, Di);                               // B_k-1
  }
  if (has_odd && has_left) load_block<T, D>(So_c + (k - 1) * DD, Soc);  // S~[k,k-1]
  if (has_odd) {
    load_block<T, D>(Sd_c + k * DD, Sc);               // S~[k,k]
    store_block<T, D>(Sd + (e + 1) * DD, Sc);
    mm<T, D>(M, Sc, Ak);
    if (has_left) mm_acc<T, D>(M, Soc, Bk);
    mm_tn_acc<T, D>(See, Ak, M);
    T neg[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) neg[i][j] = -M[i][j];
    store_block<T, D>(So + e * DD, neg);                // Sig[2k+1,2k]
  }
  if (has_left) {
    load_block<T, D>(Sd_c + (k - 1) * DD, Sc);         // S~[k-1,k-1]
    mm<T, D>(M, Sc, Bk);
    if (has_odd) mm_tn_acc<T, D>(M, Soc, Ak);          // + S~[k,k-1]^T A_k
    mm_tn_acc<T, D>(See, Bk, M);
    T negT[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) negT[i][j] = -M[j][i];
    store_block<T, D>(So + (e - 1) * DD, negT);         // Sig[2k,2k-1]
  }
  store_block<T, D>(Sd + e * DD, See);
}

template <typename T, int D>
__global__ __launch_bounds__(LEVEL_THREADS) void inverse_level_kernel(
    const T* __restrict__ Dk, const T* __restrict__ Fk, const T* __restrict__ Gk,
    const T* __restrict__ Sd_c, const T* __restrict__ So_c, int64_t n,
    T* __restrict__ Sd, T* __restrict__ So) {
  inverse_level_row<T, D>(Dk, Fk, Gk, Sd_c, So_c, n, Sd, So, (int64_t)blockIdx.x * LEVEL_THREADS + threadIdx.x);
}

// Adjoint of mahal_and_det in the blocks, given Sigma's blocks (inverse_blocks) and w = J^-1 x:
//   d/dR_i = gl Sigma[i,i] - gm w_i w_i^T,   d/dO_i = 2 (gl Sigma[i+1,i] - gm w_i+1 w_i^T)
// in place over Sd / So, one pass (as batched torch ops: ten element-wise kernels over N d^2 arrays).
constexpr int ADJ_THREADS = 256;
template <typename T, int D>
__global__ __launch_bounds__(ADJ_THREADS) void mahal_logdet_adjoint_kernel(T* __restrict__ Sd, T* __restrict__ So,
                                                                           const T* __restrict__ w, int64_t N,
                                                                           const T* __restrict__ gm_p,
                                                                           const T* __restrict__ gl_p) {
  constexpr int DD = D * D;
  const T gm = *gm_p, gl = *gl_p;
  const int64_t nd = N * DD, total = nd + (N - 1) * DD;
  for (int64_t idx = (int64_t)blockIdx.x * ADJ_THREADS + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * ADJ_THREADS) {
    const bool diag = idx < nd;
    const int64_t j = diag ? idx : idx - nd;
    const int64_t i = j / DD;
    const int r = (int)(j - i * DD), a = r / D, b = r - a * D;
    if (diag) Sd[j] = gl * Sd[j] - gm * w[i * D + a] * w[i * D + b];
    else So[j] = T(2) * (gl * So[j] - gm * w[(i + 1) * D + a] * w[i * D + b]);
  }
}

}  // namespace cgps
