// Substitution sweeps with SEVERAL right-hand sides at once: solve(decomp, Y[N, d, m]).
// The reference's einsums carry a trailing "..." (cyclic_reduction.py:52-57, 76-84), so its
// halfsolve / backhalfsolve / solve accept Y[N, d, m]; done column by column the 3 N d^2 s bytes of
// the factor are read once per column and sweep.  Here a "vector" is a d x MC panel (MC = 2, 4 or
// 8 columns, compile time; w <= MC of them real): one pass over the factor serves up to eight
// columns, and the factor's share of the traffic falls from 3 d / (3 d + 2) to 3 d / (3 d + 2 m).
//
// Same scheme as cgps_solve_tile.h (tile of right-hand sides in LDS, several levels per launch,
// rows stay in their level-0 slots, the reference's even/odd order and CRR layout), with the tile
// size chosen so that the panel tile takes the LDS of the single-column tile: TS = 2^TSL rows,
// TS / 2 threads.  Global vectors are [row][d][ld] with the caller's leading dimension ld (= nrhs
// for its own arrays, = MC for workspace buffers), columns [0, w) of a panel valid.
#pragma once
#include "cgps_solve_tile.h"

namespace cgps {

template <int MC> constexpr int solve_m_tile_log2() { return MC <= 2 ? 9 : (MC <= 4 ? 8 : 7); }
// Eight columns: the 128-row tile has 64 threads' worth of eliminations, ONE wave per workgroup, and the LDS tile lets
// four or five of them share a CU -- barely more than a wave per SIMD, every latency of a level exposed (2.7 TB/s).
// The panel's columns are therefore split over CS = 2 waves of the workgroup: wave c takes columns [4c, 4c + 4) of
// every row, both read the same factor blocks (the second read hits L1), twice the waves per SIMD for the same LDS.
template <int MC> constexpr int solve_m_col_splits() { return MC >= 8 ? 2 : 1; }   // measured: 4 splits 760 us, 2 splits for four columns 433 against 400 us

// A thread's panel: D x MC values, columns [c0, c0 + MC) of a dense [D][LDC] panel in LDS / in the workspace (LDC = MC:
// the whole panel).
template <typename T, int D, int MC, int LDC = MC>
struct Panel {
  T v[D][MC];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) v[i][c] = T(0);
  }
  // global [d][ld], columns < w
  __device__ __forceinline__ void load(const T* __restrict__ p, int ld, int w) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) v[i][c] = (c < w) ? p[(size_t)i * ld + c] : T(0);
  }
  __device__ __forceinline__ void store(T* __restrict__ p, int ld, int w) const {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c)
        if (c < w) p[(size_t)i * ld + c] = v[i][c];
  }
  // global, dense [d][MC] (ld == MC, every column real): 16-byte vectors when a panel row is a
  // whole number of them (then every row is 16-byte aligned too), contiguous scalars otherwise
  static constexpr bool VEC = (D * MC) % Vec16<T>::N == 0;
  static constexpr bool ROWVEC = MC % Vec16<T>::N == 0 && LDC % Vec16<T>::N == 0;     // sub-panel: whole vectors per row
  __device__ __forceinline__ void load_dense(const T* __restrict__ p) {
    T* flat = &v[0][0];
    if constexpr (LDC != MC) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        if constexpr (ROWVEC) {
          constexpr int VN = Vec16<T>::N;
          using V = typename Vec16<T>::type;
#pragma unroll
          for (int g = 0; g < MC / VN; ++g) {
            const V x = *reinterpret_cast<const V*>(p + (size_t)i * LDC + g * VN);
            const T* e = reinterpret_cast<const T*>(&x);
#pragma unroll
            for (int u = 0; u < VN; ++u) v[i][g * VN + u] = e[u];
          }
        } else {
#pragma unroll
          for (int c = 0; c < MC; ++c) v[i][c] = p[(size_t)i * LDC + c];
        }
      }
    } else
    if constexpr (VEC) {
      constexpr int VN = Vec16<T>::N;
      using V = typename Vec16<T>::type;
      const V* q = reinterpret_cast<const V*>(p);
#pragma unroll
      for (int g = 0; g < D * MC / VN; ++g) {
        const V x = q[g];
        const T* e = reinterpret_cast<const T*>(&x);
#pragma unroll
        for (int u = 0; u < VN; ++u) flat[g * VN + u] = e[u];
      }
    } else {
#pragma unroll
      for (int i = 0; i < D * MC; ++i) flat[i] = p[i];
    }
  }
  __device__ __forceinline__ void store_dense(T* __restrict__ p) const {
    const T* flat = &v[0][0];
    if constexpr (LDC != MC) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        if constexpr (ROWVEC) {
          constexpr int VN = Vec16<T>::N;
          using V = typename Vec16<T>::type;
#pragma unroll
          for (int g = 0; g < MC / VN; ++g) {
            V x;
            T* e = reinterpret_cast<T*>(&x);
#pragma unroll
            for (int u = 0; u < VN; ++u) e[u] = v[i][g * VN + u];
            *reinterpret_cast<V*>(p + (size_t)i * LDC + g * VN) = x;
          }
        } else {
#pragma unroll
          for (int c = 0; c < MC; ++c) p[(size_t)i * LDC + c] = v[i][c];
        }
      }
    } else
    if constexpr (VEC) {
      constexpr int VN = Vec16<T>::N;
      using V = typename Vec16<T>::type;
      V* q = reinterpret_cast<V*>(p);
#pragma unroll
      for (int g = 0; g < D * MC / VN; ++g) {
        V x;
        T* e = reinterpret_cast<T*>(&x);
#pragma unroll
        for (int u = 0; u < VN; ++u) e[u] = flat[g * VN + u];
        q[g] = x;
      }
    } else {
#pragma unroll
      for (int i = 0; i < D * MC; ++i) p[i] = flat[i];
    }
  }
  // either form, chosen per call (wave-uniform); p points at this thread's first column c0, w counts the valid
  // columns of the WHOLE panel
  __device__ __forceinline__ void load_any(const T* __restrict__ p, int ld, int w, int c0 = 0) {
    if (ld == LDC && w == LDC) load_dense(p); else load(p, ld, w - c0);
  }
  __device__ __forceinline__ void store_any(T* __restrict__ p, int ld, int w, int c0 = 0) const {
    if (ld == LDC && w == LDC) store_dense(p); else store(p, ld, w - c0);
  }
  // LDS, dense [d][LDC] (p points at the thread's first column)
  __device__ __forceinline__ void lds_load(const T* p) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) v[i][c] = p[i * LDC + c];
  }
  __device__ __forceinline__ void lds_store(T* p) const {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) p[i * LDC + c] = v[i][c];
  }
  __device__ __forceinline__ void sub(const Panel& o) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) v[i][c] -= o.v[i][c];
  }
  // v <- L^-1 v, every column
  __device__ __forceinline__ void fwd(const Chol<T, D>& ch) {
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        T s = v[j][c];
#pragma unroll
        for (int q = 0; q < j; ++q) s = fmaT(-ch.l[j][q], v[q][c], s);
        v[j][c] = s * ch.inv[j];
      }
  }
  // v <- L^-T v
  __device__ __forceinline__ void bwd(const Chol<T, D>& ch) {
#pragma unroll
    for (int j = D - 1; j >= 0; --j)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        T s = v[j][c];
#pragma unroll
        for (int q = j + 1; q < D; ++q) s = fmaT(-ch.l[q][j], v[q][c], s);
        v[j][c] = s * ch.inv[j];
      }
  }
  // v -= A x   /   v -= A^T x
  __device__ __forceinline__ void gemm_sub(const T (&A)[D][D], const Panel& x) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        T s = v[i][c];
#pragma unroll
        for (int q = 0; q < D; ++q) s = fmaT(-A[i][q], x.v[q][c], s);
        v[i][c] = s;
      }
  }
  __device__ __forceinline__ void gemmT_sub(const T (&A)[D][D], const Panel& x) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        T s = v[i][c];
#pragma unroll
        for (int q = 0; q < D; ++q) s = fmaT(-A[q][i], x.v[q][c], s);
        v[i][c] = s;
      }
  }
  // v += A x
  __device__ __forceinline__ void gemm_add(const T (&A)[D][D], const Panel& x) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        T s = v[i][c];
#pragma unroll
        for (int q = 0; q < D; ++q) s = fmaT(A[i][q], x.v[q][c], s);
        v[i][c] = s;
      }
  }
  __device__ __forceinline__ double sumsq() const {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MC; ++c) s += (double)v[i][c] * (double)v[i][c];
    return s;
  }
};

template <typename T, int D, int MC>
constexpr size_t solve_m_lds_bytes() {
  return ((size_t)(1 << solve_m_tile_log2<MC>()) + 1) * D * MC * sizeof(T) + 64 * sizeof(double);
}

// ---- forward sweep (cf. halfsolve_tile_kernel) ---------------------------------------------------
// y_in [n][d][ld_y]; xcrr [N][d][ld_x] (CRR layout); y_out / owed_in / owed_out: workspace, ld = MC.
template <typename T, int D, int MC>
__global__ __launch_bounds__((1 << solve_m_tile_log2<MC>()) / 2 * solve_m_col_splits<MC>()) void halfsolve_tile_m_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ owed_in, int64_t n_owed, int spt_in, const T* __restrict__ y_in, int ld_y, int64_t n, int w,
    T* __restrict__ xcrr, int ld_x, T* __restrict__ y_out, T* __restrict__ owed_out, double* __restrict__ partial) {
  constexpr int DD = D * D, TS = 1 << solve_m_tile_log2<MC>(), NT = TS / 2, PW = D * MC;
  constexpr int CS = solve_m_col_splits<MC>(), MS = MC / CS, NTT = NT * CS;      // threads of a column split x splits
  using P = Panel<T, D, MS, MC>;
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* ys = reinterpret_cast<T*>(solve_smem);                                   // [TS][D][MC]
  T* owed = ys + (size_t)TS * PW;                                             // [D][MC]: sum_j G_j x_j of the tile's first rows
  double* red = reinterpret_cast<double*>(owed + PW);
  const int tid = threadIdx.x % NT, c0 = (threadIdx.x / NT) * MS;             // elimination index base / first column
  const int64_t row0 = (int64_t)blockIdx.x * TS;
  const int n0 = (int)((n - row0) < TS ? (n - row0) : TS);
  ys += c0;
  owed += c0;
  for (int r = tid; r < n0; r += NT) {
    P v;
    v.load_any(y_in + (row0 + r) * (size_t)D * ld_y + c0, ld_y, w, c0);
    const int64_t wn = row0 + r + 1;
    if (owed_in != nullptr && wn % spt_in == 0 && wn / spt_in < n_owed) {     // the last survivor of a tile of the previous pass
      P o;
      o.load_dense(owed_in + (wn / spt_in) * (size_t)PW + c0);
      v.sub(o);
    }
    v.lds_store(ys + (size_t)r * PW);
  }
  for (int i = threadIdx.x; i < PW; i += NTT) (owed - c0)[i] = T(0);
  __syncthreads();
  double mah = 0.0, zero = 0.0;
  int nj = n0;
  auto eliminate = [&](int j, int k, P& x) {
    const int64_t g0 = row0 >> (j + 1);
    T L[D][D];
    Chol<T, D> c;
    load_block<T, D>(Dp + (lv.offD[j] + g0 + k) * DD, L);
    chol_from_dense<T, D>(L, c);
    x.fwd(c);
    x.lds_store(ys + (size_t)(((2 * k + 1) << j) - 1) * PW);
    x.store_any(xcrr + (lv.offD[j] + g0 + k) * (size_t)D * ld_x + c0, ld_x, w, c0);
    mah += x.sumsq();
    if (k == 0 && g0 >= 1) {                             // the previous tile's last row is this row's left neighbour
      T G[D][D];
      load_block<T, D>(Gp + (lv.offG[j] + g0 - 1) * DD, G);
      P o;
      o.lds_load(owed);
      o.gemm_add(G, x);
      o.lds_store(owed);
    }
  };
  for (int k = tid; k < ((nj + 1) >> 1); k += NT) {
    P x;
    x.lds_load(ys + (size_t)(2 * k) * PW);
    eliminate(0, k, x);
  }
  __syncthreads();
  for (int j = 0; j < lv.nlev && nj >= 1; ++j) {
    const int no = nj >> 1;
    const int64_t g0 = row0 >> (j + 1);
    const bool more = j + 1 < lv.nlev;
    for (int k = tid; k < no; k += NT) {                 // y'_k = y_2k+1 - F_k x_k - G_k x_k+1
      T M[D][D];
      P x, yo;
      T* slot = ys + (size_t)(((2 * k + 2) << j) - 1) * PW;
      yo.lds_load(slot);
      load_block<T, D>(Fp + (lv.offF[j] + g0 + k) * DD, M);
      x.lds_load(ys + (size_t)(((2 * k + 1) << j) - 1) * PW);
      yo.gemm_sub(M, x);
      if (2 * k + 2 < nj) {                              // right neighbour inside the tile
        load_block<T, D>(Gp + (lv.offG[j] + g0 + k) * DD, M);
        x.lds_load(ys + (size_t)(((2 * k + 3) << j) - 1) * PW);
        yo.gemm_sub(M, x);
      }
      if (more && (k & 1) == 0) eliminate(j + 1, k >> 1, yo);
      else yo.lds_store(slot);
    }
    __syncthreads();
    nj = no;
  }
  if (y_out != nullptr) {                                // the tile's surviving rows: nj = n0 >> nlev of them
    const int spt_out = TS >> lv.nlev;
    for (int r = tid; r < nj; r += NT) {
      P v;
      v.lds_load(ys + (size_t)(((r + 1) << lv.nlev) - 1) * PW);
      v.store_dense(y_out + ((size_t)blockIdx.x * spt_out + r) * PW + c0);
    }
  }
  if (owed_out != nullptr)
    for (int i = threadIdx.x; i < PW; i += NTT) owed_out[(size_t)blockIdx.x * PW + i] = (owed - c0)[i];
  block_sum2<NTT>(mah, zero, red);
  if (threadIdx.x == 0 && partial != nullptr) {
    partial[2 * (size_t)blockIdx.x] = mah;
    partial[2 * (size_t)blockIdx.x + 1] = 0.0;
  }
}

// ---- backward sweep (cf. backsolve_tile_kernel) --------------------------------------------------
// b [N][d][ld_b] in CRR layout; x_coarse: workspace (ld = MC), nullptr for the top pass; x_out [n][d][ld_o].
template <typename T, int D, int MC>
__global__ __launch_bounds__((1 << solve_m_tile_log2<MC>()) / 2 * solve_m_col_splits<MC>()) void backsolve_tile_m_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ b, int ld_b, const T* __restrict__ x_coarse, int64_t n, int w, T* __restrict__ x_out, int ld_o) {
  constexpr int DD = D * D, TS = 1 << solve_m_tile_log2<MC>(), NT = TS / 2, PW = D * MC;
  constexpr int CS = solve_m_col_splits<MC>(), MS = MC / CS;
  using P = Panel<T, D, MS, MC>;
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* xs = reinterpret_cast<T*>(solve_smem);                                   // [TS][D][MC]
  const int tid = threadIdx.x % NT, c0 = (threadIdx.x / NT) * MS;
  const int64_t row0 = (int64_t)blockIdx.x * TS;
  const int n0 = (int)((n - row0) < TS ? (n - row0) : TS);
  xs += c0;
  P xleft;                                               // x of the previous tile's last row
  xleft.zero();
  if (x_coarse != nullptr) {                             // solution of the rows that survived this pass's levels
    const int spt = TS >> lv.nlev;
    if (blockIdx.x > 0) xleft.load_dense(x_coarse + ((size_t)blockIdx.x * spt - 1) * PW + c0);
    for (int r = tid; r < (n0 >> lv.nlev); r += NT) {
      P v;
      v.load_dense(x_coarse + ((size_t)blockIdx.x * spt + r) * PW + c0);
      v.lds_store(xs + (size_t)(((r + 1) << lv.nlev) - 1) * PW);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = lv.nlev - 1; j >= 0; --j) {
    const int nj = n0 >> j;
    if (nj >= 1) {
      const int ne = (nj + 1) >> 1;
      const int64_t g0 = row0 >> (j + 1);
      for (int k = tid; k < ne; k += NT) {
        T M[D][D];
        P r, xo;
        r.load_any(b + (lv.offD[j] + g0 + k) * (size_t)D * ld_b + c0, ld_b, w, c0);
        if (2 * k + 1 < nj) {
          load_block<T, D>(Fp + (lv.offF[j] + g0 + k) * DD, M);
          xo.lds_load(xs + (size_t)(((2 * k + 2) << j) - 1) * PW);
          r.gemmT_sub(M, xo);
        }
        if (k >= 1) {
          load_block<T, D>(Gp + (lv.offG[j] + g0 + k - 1) * DD, M);
          xo.lds_load(xs + (size_t)(((2 * k) << j) - 1) * PW);
          r.gemmT_sub(M, xo);
        } else if (g0 >= 1) {                            // left neighbour = previous tile's last row
          load_block<T, D>(Gp + (lv.offG[j] + g0 - 1) * DD, M);
          r.gemmT_sub(M, xleft);
        }
        T L[D][D];
        Chol<T, D> c;
        load_block<T, D>(Dp + (lv.offD[j] + g0 + k) * DD, L);
        chol_from_dense<T, D>(L, c);
        r.bwd(c);
        r.lds_store(xs + (size_t)(((2 * k + 1) << j) - 1) * PW);
      }
    }
    __syncthreads();
  }
  for (int r = tid; r < n0; r += NT) {
    P v;
    v.lds_load(xs + (size_t)r * PW);
    v.store_any(x_out + (row0 + r) * (size_t)D * ld_o + c0, ld_o, w, c0);
  }
}

// ---- latency-bound passes of a panel sweep (at most two tiles per CU): every factor block up front -------------
// The panel counterpart of halfsolve_deep_kernel / backsolve_deep_kernel (cgps_solve_tile.h): the passes after the bulk
// pass walk ALL levels of their tiles, and with a level's blocks requested when the level starts every level exposes
// one memory round trip (eight columns at 2^20 rows: 91 + 64 us of the 632 in such passes).  WHICH blocks a tile needs
// depends on no data: lane t owns elimination t of level 0 and ONE deeper elimination (deep_owner), requests the D / F / G
// of both -- and, in the backward sweep, its two right-hand-side panels -- before anything else, and the levels then only
// touch LDS.  Tiles of 2^TSLD rows, 2^TSLD / 2 lanes per column split: 256 threads, one wave per SIMD (seven blocks
// and two panels in registers); blocks of at most 128 bytes, as for the single-column form.
template <int MC> constexpr int solve_m_deep_tile_log2() { return MC >= 8 ? 8 : 9; }
template <typename T, int D, int MC>
constexpr size_t solve_m_deep_lds_bytes() {
  return ((size_t)(1 << solve_m_deep_tile_log2<MC>()) + 2 + SOLVE_MAXLEV) * D * MC * sizeof(T) + 64 * sizeof(double);
}

template <typename T, int D, int MC>
__global__ __launch_bounds__((1 << solve_m_deep_tile_log2<MC>()) / 2 * solve_m_col_splits<MC>())
    __attribute__((amdgpu_waves_per_eu(1, 1))) void halfsolve_deep_m_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ owed_in, int64_t n_owed, int spt_in, const T* __restrict__ y_in, int ld_y, int64_t n, int w,
    T* __restrict__ xcrr, int ld_x, T* __restrict__ y_out, T* __restrict__ owed_out, double* __restrict__ partial) {
  constexpr int DD = D * D, TSL = solve_m_deep_tile_log2<MC>(), TS = 1 << TSL, NT = TS / 2, PW = D * MC;
  constexpr int CS = solve_m_col_splits<MC>(), MS = MC / CS, NTT = NT * CS;
  using P = Panel<T, D, MS, MC>;
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* ys = reinterpret_cast<T*>(solve_smem);                                   // [TS][D][MC]
  T* owed = ys + (size_t)TS * PW;                                             // [2 + MAXLEV][D][MC]: per level, G_left x
  double* red = reinterpret_cast<double*>(owed + (size_t)(2 + SOLVE_MAXLEV) * PW);
  const int tid = threadIdx.x % NT, c0 = (threadIdx.x / NT) * MS;
  const int64_t row0 = (int64_t)blockIdx.x * TS;
  const int n0 = (int)((n - row0) < TS ? (n - row0) : TS);
  ys += c0;
  owed += c0;
  // ---- every factor block of the pass ----------------------------------------------------------------------------
  T L0[D][D], F0[D][D], G0[D][D];
  const int ne0 = (n0 + 1) >> 1, no0 = n0 >> 1;
  const int64_t g00 = row0 >> 1;
  if (tid < ne0) load_block<T, D>(Dp + (lv.offD[0] + g00 + tid) * DD, L0); else set_zero_block(L0);
  const bool upd0 = tid < no0 && lv.nlev >= 1, rgt0 = upd0 && (2 * tid + 2 < n0);
  if (upd0) load_block<T, D>(Fp + (lv.offF[0] + g00 + tid) * DD, F0); else set_zero_block(F0);
  if (rgt0) load_block<T, D>(Gp + (lv.offG[0] + g00 + tid) * DD, G0); else set_zero_block(G0);
  const DeepOwner own = deep_owner<TSL>(tid);
  const int dj = own.j, dk = own.k;
  const int nj_d = (dj <= lv.nlev) ? (n0 >> dj) : 0;
  const bool elim_d = dj >= 1 && dk < ((nj_d + 1) >> 1) && (dj < lv.nlev);
  const bool upd_d = elim_d && dk < (nj_d >> 1);
  const bool rgt_d = upd_d && (2 * dk + 2 < nj_d);
  T Ld[D][D], Fd[D][D], Gd[D][D];
  const int64_t g0d = row0 >> (dj + 1);
  if (elim_d) load_block<T, D>(Dp + (lv.offD[dj] + g0d + dk) * DD, Ld); else set_zero_block(Ld);
  if (upd_d) load_block<T, D>(Fp + (lv.offF[dj] + g0d + dk) * DD, Fd); else set_zero_block(Fd);
  if (rgt_d) load_block<T, D>(Gp + (lv.offG[dj] + g0d + dk) * DD, Gd); else set_zero_block(Gd);
  T Gl[D][D];                                            // G of the block left of the tile, one level per lane
  const bool left = tid < lv.nlev && (row0 >> (tid + 1)) >= 1 && (n0 >> tid) >= 1;
  if (left) load_block<T, D>(Gp + (lv.offG[tid] + (row0 >> (tid + 1)) - 1) * DD, Gl); else set_zero_block(Gl);
  for (int r = tid; r < n0; r += NT) {
    P v;
    v.load_any(y_in + (row0 + r) * (size_t)D * ld_y + c0, ld_y, w, c0);
    const int64_t wn = row0 + r + 1;
    if (owed_in != nullptr && wn % spt_in == 0 && wn / spt_in < n_owed) {
      P o;
      o.load_dense(owed_in + (wn / spt_in) * (size_t)PW + c0);
      v.sub(o);
    }
    v.lds_store(ys + (size_t)r * PW);
  }
  __syncthreads();
  double mah = 0.0, zero = 0.0;
  auto eliminate = [&](int j, int k, const T (&L)[D][D]) {
    P x;
    T* slot = ys + (size_t)(((2 * k + 1) << j) - 1) * PW;
    x.lds_load(slot);
    Chol<T, D> c;
    chol_from_dense<T, D>(L, c);
    x.fwd(c);
    x.lds_store(slot);
    x.store_any(xcrr + (lv.offD[j] + (row0 >> (j + 1)) + k) * (size_t)D * ld_x + c0, ld_x, w, c0);
    mah += x.sumsq();
  };
  auto update = [&](int j, int k, bool right, const T (&F)[D][D], const T (&G)[D][D]) {
    P x, yo;
    T* slot = ys + (size_t)(((2 * k + 2) << j) - 1) * PW;
    yo.lds_load(slot);
    x.lds_load(ys + (size_t)(((2 * k + 1) << j) - 1) * PW);
    yo.gemm_sub(F, x);
    if (right) {
      x.lds_load(ys + (size_t)(((2 * k + 3) << j) - 1) * PW);
      yo.gemm_sub(G, x);
    }
    yo.lds_store(slot);
  };
  if (tid < ne0) eliminate(0, tid, L0);
  __syncthreads();
  int nj = n0;
  for (int j = 0; j < lv.nlev && nj >= 1; ++j) {
    if (j == 0) { if (upd0) update(0, tid, rgt0, F0, G0); }
    else if (dj == j && upd_d) update(j, dk, rgt_d, Fd, Gd);
    __syncthreads();
    if (j + 1 < lv.nlev && dj == j + 1 && elim_d) eliminate(j + 1, dk, Ld);
    __syncthreads();
    nj >>= 1;
  }
  if (left) {                                            // owed to the previous tile's last row: G_left x of level tid's first elimination
    P x, o;
    x.lds_load(ys + (size_t)((1 << tid) - 1) * PW);
    o.zero();
    o.gemm_add(Gl, x);
    o.lds_store(owed + (size_t)(1 + tid) * PW);
  }
  __syncthreads();
  if (y_out != nullptr) {
    const int spt_out = TS >> lv.nlev;
    for (int r = tid; r < (n0 >> lv.nlev); r += NT) {
      P v;
      v.lds_load(ys + (size_t)(((r + 1) << lv.nlev) - 1) * PW);
      v.store_dense(y_out + ((size_t)blockIdx.x * spt_out + r) * PW + c0);
    }
  }
  if (owed_out != nullptr && tid == 0) {
    P ow;
    ow.zero();
    for (int l = 0; l < lv.nlev; ++l) {
      if ((row0 >> (l + 1)) >= 1 && (n0 >> l) >= 1) {
        P o;
        o.lds_load(owed + (size_t)(1 + l) * PW);
        ow.sub(o);
      }
    }
    // (ow holds minus the sum: the regular panel kernel hands on +sum, so negate)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int c = 0; c < MS; ++c) ow.v[i][c] = -ow.v[i][c];
    ow.store_dense(owed_out + (size_t)blockIdx.x * PW + c0);
  }
  block_sum2<NTT>(mah, zero, red);
  if (threadIdx.x == 0 && partial != nullptr) {
    partial[2 * (size_t)blockIdx.x] = mah;
    partial[2 * (size_t)blockIdx.x + 1] = 0.0;
  }
}

// what elimination k of level j needs in the backward sweep: x = D^-T (b - F^T x_right - G^T x_left)
template <typename T, int D, int MS, int MC>
__device__ __forceinline__ BackFlags back_request_m(const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp,
                                                    const T* __restrict__ b, int ld_b, int w, int c0, const PassLevels& lv,
                                                    int64_t row0, int n0, int j, int k, bool on, T (&L)[D][D], T (&F)[D][D],
                                                    T (&G)[D][D], Panel<T, D, MS, MC>& r) {
  constexpr int DD = D * D;
  const int nj = n0 >> j;
  const int64_t g0 = row0 >> (j + 1);
  BackFlags f;
  f.on = on && k < ((nj + 1) >> 1);
  f.right = f.on && (2 * k + 1 < nj);
  f.left_in = f.on && k >= 1;
  f.left_out = f.on && k == 0 && g0 >= 1;
  if (f.on) {
    r.load_any(b + (lv.offD[j] + g0 + k) * (size_t)D * ld_b + c0, ld_b, w, c0);
    load_block<T, D>(Dp + (lv.offD[j] + g0 + k) * DD, L);
  } else {
    set_zero_block(L);
    r.zero();
  }
  if (f.right) load_block<T, D>(Fp + (lv.offF[j] + g0 + k) * DD, F); else set_zero_block(F);
  if (f.left_in || f.left_out) load_block<T, D>(Gp + (lv.offG[j] + g0 + k - 1) * DD, G); else set_zero_block(G);
  return f;
}
template <typename T, int D, int MS, int MC>
__device__ __forceinline__ void back_run_m(T* xs, int j, int k, const BackFlags& f, const T (&L)[D][D], const T (&F)[D][D],
                                           const T (&G)[D][D], Panel<T, D, MS, MC>& r, const Panel<T, D, MS, MC>& xleft) {
  if (!f.on) return;
  constexpr int PW = D * MC;
  Panel<T, D, MS, MC> xo;
  if (f.right) {
    xo.lds_load(xs + (size_t)(((2 * k + 2) << j) - 1) * PW);
    r.gemmT_sub(F, xo);
  }
  if (f.left_in) {
    xo.lds_load(xs + (size_t)(((2 * k) << j) - 1) * PW);
    r.gemmT_sub(G, xo);
  } else if (f.left_out) {
    r.gemmT_sub(G, xleft);
  }
  Chol<T, D> c;
  chol_from_dense<T, D>(L, c);
  r.bwd(c);
  r.lds_store(xs + (size_t)(((2 * k + 1) << j) - 1) * PW);
}

template <typename T, int D, int MC>
__global__ __launch_bounds__((1 << solve_m_deep_tile_log2<MC>()) / 2 * solve_m_col_splits<MC>())
    __attribute__((amdgpu_waves_per_eu(1, 1))) void backsolve_deep_m_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ b, int ld_b, const T* __restrict__ x_coarse, int64_t n, int w, T* __restrict__ x_out, int ld_o) {
  constexpr int TSL = solve_m_deep_tile_log2<MC>(), TS = 1 << TSL, NT = TS / 2, PW = D * MC;
  constexpr int CS = solve_m_col_splits<MC>(), MS = MC / CS;
  using P = Panel<T, D, MS, MC>;
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* xs = reinterpret_cast<T*>(solve_smem);                                   // [TS][D][MC]
  const int tid = threadIdx.x % NT, c0 = (threadIdx.x / NT) * MS;
  const int64_t row0 = (int64_t)blockIdx.x * TS;
  const int n0 = (int)((n - row0) < TS ? (n - row0) : TS);
  xs += c0;
  T L0[D][D], F0[D][D], G0[D][D], Ld[D][D], Fd[D][D], Gd[D][D];
  P r0, rd;
  const BackFlags f0 = back_request_m<T, D, MS, MC>(Dp, Fp, Gp, b, ld_b, w, c0, lv, row0, n0, 0, tid, lv.nlev >= 1, L0, F0, G0, r0);
  const DeepOwner own = deep_owner<TSL>(tid);
  const BackFlags fd = back_request_m<T, D, MS, MC>(Dp, Fp, Gp, b, ld_b, w, c0, lv, row0, n0, own.j, own.k,
                                                    own.j >= 1 && own.j < lv.nlev, Ld, Fd, Gd, rd);
  P xleft;
  xleft.zero();
  if (x_coarse != nullptr) {
    const int spt = TS >> lv.nlev;
    if (blockIdx.x > 0) xleft.load_dense(x_coarse + ((size_t)blockIdx.x * spt - 1) * PW + c0);
    for (int r = tid; r < (n0 >> lv.nlev); r += NT) {
      P v;
      v.load_dense(x_coarse + ((size_t)blockIdx.x * spt + r) * PW + c0);
      v.lds_store(xs + (size_t)(((r + 1) << lv.nlev) - 1) * PW);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = lv.nlev - 1; j >= 1; --j) {
    if (own.j == j) back_run_m<T, D, MS, MC>(xs, j, own.k, fd, Ld, Fd, Gd, rd, xleft);
    __syncthreads();
  }
  back_run_m<T, D, MS, MC>(xs, 0, tid, f0, L0, F0, G0, r0, xleft);
  __syncthreads();
  for (int r = tid; r < n0; r += NT) {
    P v;
    v.lds_load(xs + (size_t)r * PW);
    v.store_any(x_out + (row0 + r) * (size_t)D * ld_o + c0, ld_o, w, c0);
  }
}

}  // namespace cgps
