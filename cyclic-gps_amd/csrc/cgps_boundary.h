// The small systems of the sharded solve / posterior (cyclic_gps/sharded.py), from the gathered shard records, as ONE
// launch each instead of a dozen batched torch launches and a P-row decompose + solve on the critical path of every
// rank.  P = world size x records per rank (a handful): the work is a chain of P dependent d x d steps, one lane.
// Records: RecordLayout (cgps_tile.h): Rs | Cs | dRa | ys | dya; consecutive records rstride elements apart.
//   boundary system (rows = the shards' last rows, after the shards' interiors have been eliminated):
//       R_w = Rs_w + dRa_{w+1},   y_w = ys_w + dya_{w+1},   J[w+1, w] = Cs_{w+1}
#pragma once
#include "cgps_tile.h"

namespace cgps {

constexpr int BOUNDARY_MAX_P = 64;

template <typename T, int D>
__device__ __forceinline__ void boundary_row(const T* __restrict__ rec, int64_t rstride, int w, int P, T (&R)[D][D], T (&y)[D]) {
  using RL = RecordLayout<T, D>;
  const T* r = rec + (size_t)w * rstride;
  load_block<T, D>(r + RL::RS, R);
  load_vec<T, D>(r + RL::YS, y);
  if (w + 1 < P) {
    T nR[D][D], ny[D];
    load_block<T, D>(r + rstride + RL::DRA, nR);
    load_vec<T, D>(r + rstride + RL::DYA, ny);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      y[i] += ny[i];
#pragma unroll
      for (int j = 0; j < D; ++j) R[i][j] += nR[i][j];
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) R[i][j] = R[j][i] = T(0.5) * (R[i][j] + R[j][i]);   // (symmetric up to rounding)
}

// x at every separator: block Cholesky of the P-row boundary system, forward and backward substitution (one lane;
// the factors of the forward sweep wait in LDS).  xsep[P][D]; info: 0 or 1 + the first row whose pivot block fails.
template <typename T, int D>
__global__ __launch_bounds__(64) void boundary_solve_kernel(const T* __restrict__ rec, int64_t rstride, int P, T* __restrict__ xsep,
                                                            int* __restrict__ info) {
  using RL = RecordLayout<T, D>;
  __shared__ T sL[BOUNDARY_MAX_P][D * D + D];      // per row: the dense factor L_w (lower), then z_w = L_w^-1 t_w
  if (threadIdx.x != 0) return;
  bool fail = false;
  int bad = 0;
  Chol<T, D> c;
  T z[D];
  for (int w = 0; w < P; ++w) {
    T S[D][D], t[D];
    boundary_row<T, D>(rec, rstride, w, P, S, t);
    if (w > 0) {
      T M[D][D];
      load_block<T, D>(rec + (size_t)w * rstride + RL::CS, M);
      rsolve_lt<T, D>(c, M);                       // M = J[w, w-1] L_{w-1}^-T
      syrk_sub_lower<T, D>(S, M);
      gemv_sub<T, D>(t, M, z);
    }
    bool f = false;
    chol_lower<T, D>(S, c, f);
    if (f && !fail) { fail = true; bad = w + 1; }
#pragma unroll
    for (int i = 0; i < D; ++i) z[i] = t[i];
    fwd_subst<T, D>(c, z);
    T L[D][D];
    chol_to_dense<T, D>(c, L);
    store_block<T, D>(&sL[w][0], L);
    store_vec<T, D>(&sL[w][D * D], z);
  }
  T x[D];
  for (int w = P - 1; w >= 0; --w) {
    T L[D][D], v[D];
    load_block<T, D>(&sL[w][0], L);
    load_vec<T, D>(&sL[w][D * D], v);
    chol_from_dense<T, D>(L, c);
    if (w + 1 < P) {                               // z_w - M_{w+1}^T x_{w+1},  M_{w+1}^T x = L_w^-1 (J[w+1, w]^T x)
      T Cs[D][D], u[D];
      load_block<T, D>(rec + (size_t)(w + 1) * rstride + RL::CS, Cs);
      set_zero<T, D>(u);
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int m = 0; m < D; ++m) u[i] = fmaT(Cs[m][i], x[m], u[i]);
      fwd_subst<T, D>(c, u);
#pragma unroll
      for (int i = 0; i < D; ++i) v[i] -= u[i];
    }
    bwd_subst<T, D>(c, v);
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = v[i];
    store_vec<T, D>(xsep + (size_t)w * D, x);
  }
  *info = bad;
}

// X = A^-1 [B | b] for the symmetric positive definite A (lower triangle read): Cholesky + two substitutions per column
template <typename T, int D>
__device__ __forceinline__ bool spd_solve_cols(const T (&A)[D][D], T (&B)[D][D], T (&b)[D]) {
  Chol<T, D> c;
  bool f = false;
  chol_lower<T, D>(A, c, f);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T v[D];
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = B[i][j];
    fwd_subst<T, D>(c, v);
    bwd_subst<T, D>(c, v);
#pragma unroll
    for (int i = 0; i < D; ++i) B[i][j] = v[i];
  }
  fwd_subst<T, D>(c, b);
  bwd_subst<T, D>(c, b);
  return !f;
}

// What the rest of the system does to `rank`'s rows (sharded.boundary_recursions):
//   left to right  P_0 = Rs_0, p_0 = ys_0;  P_w = Rs_w - Cs_w (P_{w-1} + dRa_w)^-1 Cs_w^T,  p_w likewise  -> (P, p) of rank - 1
//   right to left  dR_{P-1} = 0;  dR_w = dRa_{w+1} - Cs_{w+1}^T (Rs_{w+1} + dR_{w+1})^-1 Cs_{w+1},  dy likewise -> (dR, dy) of rank
// out: [Pa (D*D) | pa (D) | dR (D*D) | dy (D)], symmetric blocks written in full.  Two lanes in two waves, one per chain.
template <typename T, int D>
__global__ __launch_bounds__(128) void boundary_recursions_kernel(const T* __restrict__ rec, int64_t rstride, int P, int rank,
                                                                   T* __restrict__ out, int* __restrict__ info) {
  using RL = RecordLayout<T, D>;
  constexpr int DD = D * D;
  auto sym_load = [&](const T* p, T (&A)[D][D]) {
    load_block<T, D>(p, A);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) A[i][j] = A[j][i] = T(0.5) * (A[i][j] + A[j][i]);
  };
  if (threadIdx.x == 0) {
    T Pw[D][D], pw[D];
    set_zero<T, D>(Pw);
    set_zero<T, D>(pw);
    bool ok = true;
    if (rank > 0) {
      sym_load(rec + RL::RS, Pw);
      load_vec<T, D>(rec + RL::YS, pw);
      for (int w = 1; w < rank; ++w) {
        const T* r = rec + (size_t)w * rstride;
        T A[D][D], Cs[D][D], Z[D][D], zb[D], u[D], dRa[D][D];
        sym_load(r + RL::DRA, dRa);
        load_vec<T, D>(r + RL::DYA, u);
        load_block<T, D>(r + RL::CS, Cs);
#pragma unroll
        for (int i = 0; i < D; ++i) {
          zb[i] = pw[i] + u[i];
#pragma unroll
          for (int j = 0; j < D; ++j) { A[i][j] = Pw[i][j] + dRa[i][j]; Z[i][j] = Cs[j][i]; }
        }
        ok = spd_solve_cols<T, D>(A, Z, zb) && ok;            // Z = A^-1 Cs^T, zb = A^-1 (p + dya)
        sym_load(r + RL::RS, Pw);
        load_vec<T, D>(r + RL::YS, pw);
#pragma unroll
        for (int i = 0; i < D; ++i) {
          T s = pw[i];
#pragma unroll
          for (int m = 0; m < D; ++m) s = fmaT(-Cs[i][m], zb[m], s);
          pw[i] = s;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T q = Pw[i][j];
#pragma unroll
            for (int m = 0; m < D; ++m) q = fmaT(-Cs[i][m], Z[m][j], q);
            Pw[i][j] = q;
          }
        }
      }
    }
    store_block<T, D>(out, Pw);
    store_vec<T, D>(out + DD, pw);
    if (!ok) atomicMax(info, 1);
  } else if (threadIdx.x == 64) {
    T dR[D][D], dy[D];
    set_zero<T, D>(dR);
    set_zero<T, D>(dy);
    bool ok = true;
    for (int w = P - 2; w >= rank; --w) {
      const T* r = rec + (size_t)(w + 1) * rstride;
      T A[D][D], Cs[D][D], Z[D][D], zb[D], Rn[D][D], yn[D];
      sym_load(r + RL::RS, Rn);
      load_vec<T, D>(r + RL::YS, yn);
      load_block<T, D>(r + RL::CS, Cs);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        zb[i] = yn[i] + dy[i];
#pragma unroll
        for (int j = 0; j < D; ++j) { A[i][j] = Rn[i][j] + dR[i][j]; Z[i][j] = Cs[i][j]; }
      }
      ok = spd_solve_cols<T, D>(A, Z, zb) && ok;              // Z = A^-1 Cs, zb = A^-1 (ys + dy)
      T dRa[D][D], dya[D];
      sym_load(r + RL::DRA, dRa);
      load_vec<T, D>(r + RL::DYA, dya);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        T s = dya[i];
#pragma unroll
        for (int m = 0; m < D; ++m) s = fmaT(-Cs[m][i], zb[m], s);
        dy[i] = s;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          T q = dRa[i][j];
#pragma unroll
          for (int m = 0; m < D; ++m) q = fmaT(-Cs[m][i], Z[m][j], q);
          dR[i][j] = q;
        }
      }
    }
    store_block<T, D>(out + DD + D, dR);
    store_vec<T, D>(out + 2 * DD + D, dy);
    if (!ok) atomicMax(info, 1);
  }
}

}  // namespace cgps
