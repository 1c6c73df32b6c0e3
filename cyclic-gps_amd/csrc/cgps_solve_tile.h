// Fused forward / backward substitution with a stored cyclic-reduction factor
// (reference halfsolve cyclic_reduction.py:312-338, backhalfsolve :341-377, in the
// reference's exact even/odd order and CRR layout).
//
// With the factor stored, a row costs one d x d triangular solve and two d x d mat-vecs, and the
// only state that must be shared between rows is the right-hand side: 8 d bytes per row.  So a
// workgroup keeps the vectors of a TS = 1024-row tile in LDS and walks LP = 10 reduction levels
// over them, while the D / F / G blocks of each level stream from HBM exactly once (consecutive
// lanes read consecutive blocks of the level's packed array).  While the system is large
// (>= 2^20 rows) a pass runs only SOLVE_LP_WIDE = 3 levels -- 512 + 256 + 128 eliminations on
// 512 lanes, 6 barrier-separated phases instead of 20 -- and hands the 128 surviving rows of
// every tile to the next pass; smaller systems are latency-bound and run all ten levels.
// N = 2^20 needs three launches per sweep instead of 21.
//
// In-place indexing: row r of local level j lives in slot (r + 1) 2^j - 1 of the tile, i.e. every
// row stays where it was at local level 0; eliminated rows' slots are reused for x.
// Tile boundaries: a tile's last row is odd at every local level, so it only ever RECEIVES
// updates; the one it gets from the next tile's first row at each level is applied when the next
// pass loads the survivors (forward sweep), respectively comes from the coarser pass (backward
// sweep: the x of the previous tile's last row).
#pragma once
#include "cgps_math.h"

namespace cgps {

constexpr int SOLVE_LP = 10;            // levels per pass
constexpr int SOLVE_TS = 1 << SOLVE_LP; // rows per tile
constexpr int SOLVE_NT = 512;           // threads per workgroup (= eliminations of a tile's level 0)
constexpr int SOLVE_MAXLEV = SOLVE_LP + 1;
constexpr int SOLVE_LP_WIDE = 3;                  // levels per pass while the system is large ...
constexpr int64_t SOLVE_WIDE_ROWS = 1 << 20;      // ... i.e. has at least this many rows (2^18 .. 2^20 measured alike)
template <typename T, int D> constexpr size_t solve_lds_bytes() {
  return (size_t)SOLVE_TS * D * sizeof(T) + 2 * (SOLVE_NT / 64) * sizeof(double) + (size_t)(2 + SOLVE_MAXLEV) * D * sizeof(T);
}

// Blocks of <= 128 bytes: hold the forward sweep to 64 registers (it needs 65 otherwise), i.e. four
// resident workgroups per CU; larger blocks need more registers than that anyway.
template <typename T, int D> constexpr int solve_min_waves() { return (size_t)D * D * sizeof(T) <= 128 ? 8 : 1; }

// offsets (in blocks) of the levels one pass covers, and their sizes
struct PassLevels {
  int64_t offD[SOLVE_MAXLEV], offF[SOLVE_MAXLEV], offG[SOLVE_MAXLEV], m[SOLVE_MAXLEV];
  int nlev;       // levels this pass runs
  int64_t endD, endF, endG;   // one past the last block of this pass's levels in Dp / Fp / Gp
};

// A single-tile (top) pass is a chain of ~2 nlev dependent phases; touching its small,
// contiguous slice of the factor once up front turns every later load into an L2 hit.
template <typename T>
__device__ __forceinline__ void warm_l2(const T* p, int64_t elems) {
  const char* c = reinterpret_cast<const char*>(p);
  const int64_t bytes = elems * (int64_t)sizeof(T);
  int acc = 0;
  for (int64_t o = (int64_t)threadIdx.x * 128; o < bytes; o += (int64_t)blockDim.x * 128)
    acc += *reinterpret_cast<const volatile int*>(c + o);
  asm volatile("" ::"v"(acc));
}

template <typename T, int D>
__device__ __forceinline__ void set_zero_block(T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = T(0);
}
template <typename T, int D>
__device__ __forceinline__ void lds_load_vec(const T* p, T (&v)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = p[i];
}
template <typename T, int D>
__device__ __forceinline__ void lds_store_vec(T* p, const T (&v)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) p[i] = v[i];
}

// ---- forward sweep ---------------------------------------------------------------------------
// y_in  : the n rows of this pass (level lv.first); for pass > 0 these are the previous pass's
//         surviving rows, each still missing what the NEXT tile's first rows owe it: owed_in[t] =
//         sum over that pass's levels j of G_j x_j for tile t's first row (n_owed entries);
//         row i takes owed_in[i + 1].
// xcrr  : output, CRR layout (Dp offsets).   y_out : this pass's surviving rows (SOLVE_TS >> nlev per
//         full tile; tile t, survivor r -> y_out[t * (SOLVE_TS >> nlev) + r]).
// owed_out[tile] : what this tile's first rows owe the previous tile's LAST surviving row
//         (spt_in = survivors per tile of the pass that wrote owed_in).
template <typename T, int D>
__global__ __launch_bounds__(SOLVE_NT, (solve_min_waves<T, D>())) void halfsolve_tile_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ owed_in, int64_t n_owed, int spt_in, const T* __restrict__ y_in, int64_t n,
    T* __restrict__ xcrr, T* __restrict__ y_out, T* __restrict__ owed_out, double* __restrict__ partial) {
  constexpr int DD = D * D;
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* ys = reinterpret_cast<T*>(solve_smem);                                   // [SOLVE_TS][D]
  double* red = reinterpret_cast<double*>(solve_smem + (size_t)SOLVE_TS * D * sizeof(T));
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * SOLVE_TS;
  const int n0 = (int)((n - row0) < SOLVE_TS ? (n - row0) : SOLVE_TS);
  if (gridDim.x == 1) {
    warm_l2(Dp + lv.offD[0] * DD, (lv.endD - lv.offD[0]) * DD);
    warm_l2(Fp + lv.offF[0] * DD, (lv.endF - lv.offF[0]) * DD);
    warm_l2(Gp + lv.offG[0] * DD, (lv.endG - lv.offG[0]) * DD);
  }
  // The first level's D block of this lane does not depend on anything: request it now, so that
  // its latency passes under the load of the tile's rows instead of behind the barrier that follows.
  T L0[D][D];
  if (tid < ((n0 + 1) >> 1)) load_block<T, D>(Dp + (lv.offD[0] + (row0 >> 1) + tid) * DD, L0);
  else set_zero_block(L0);
  // load (and complete) the tile's rows
  for (int r = tid; r < n0; r += SOLVE_NT) {
    T v[D];
    load_vec<T, D>(y_in + (row0 + r) * D, v);
    // only the last survivor of a tile of the previous pass (spt_in survivors per tile) is owed
    const int64_t wn = row0 + r + 1;
    if (owed_in != nullptr && wn % spt_in == 0 && wn / spt_in < n_owed) {
      T w[D];
      load_vec<T, D>(owed_in + (wn / spt_in) * D, w);
#pragma unroll
      for (int i = 0; i < D; ++i) v[i] -= w[i];
    }
    lds_store_vec<T, D>(ys + r * D, v);
  }
  __syncthreads();
  double mah = 0.0, zero = 0.0;
  int nj = n0;
  // thread 0 only: sum_j G_j x_j of the tile's first rows.  In LDS rather than in registers: the
  // 2 D registers it would pin in every lane are what separates 69 from 64 VGPRs, i.e. three from
  // four resident workgroups per CU -- and 1024 tiles (N = 2^20) from fitting the chip in one round.
  T* owed = reinterpret_cast<T*>(red + 2 * (SOLVE_NT / 64));
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) owed[i] = T(0);
  }
  // elimination k of local level j: x_k = D_k^-1 y_2k (y handed in registers), x into the row's
  // slot and into xcrr; the tile's first elimination also owes G x to the previous tile's last row
  auto eliminate_with = [&](int j, int k, const T (&L)[D][D], T (&x)[D]) {
    const int64_t g0 = row0 >> (j + 1);
    Chol<T, D> c;
    chol_from_dense<T, D>(L, c);
    fwd_subst<T, D>(c, x);
    lds_store_vec<T, D>(ys + (size_t)(((2 * k + 1) << j) - 1) * D, x);
    store_vec<T, D>(xcrr + (lv.offD[j] + g0 + k) * D, x);
#pragma unroll
    for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
    if (k == 0 && g0 >= 1) {                             // the previous tile's last row is this row's left neighbour
      T G[D][D];
      load_block<T, D>(Gp + (lv.offG[j] + g0 - 1) * DD, G);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        T acc = owed[i];
#pragma unroll
        for (int m2 = 0; m2 < D; ++m2) acc = fmaT(G[i][m2], x[m2], acc);
        owed[i] = acc;
      }
    }
  };
  auto eliminate = [&](int j, int k, T (&x)[D]) {
    T L[D][D];
    load_block<T, D>(Dp + (lv.offD[j] + (row0 >> (j + 1)) + k) * DD, L);
    eliminate_with(j, k, L, x);
  };
  // One barrier per level: the lane that completes an odd row 2k+1 of level j (= row k of level
  // j+1) goes straight on to eliminate it when k is even, instead of parking it in LDS for a
  // separate phase behind another barrier.
  for (int k = tid; k < ((nj + 1) >> 1); k += SOLVE_NT) {
    T x[D];
    lds_load_vec<T, D>(ys + (size_t)(2 * k) * D, x);
    if (k == tid) eliminate_with(0, k, L0, x);           // requested before the tile's rows were loaded
    else eliminate(0, k, x);
  }
  __syncthreads();
  for (int j = 0; j < lv.nlev && nj >= 1; ++j) {
    const int no = nj >> 1;
    const int64_t g0 = row0 >> (j + 1);
    const bool more = j + 1 < lv.nlev;
    for (int k = tid; k < no; k += SOLVE_NT) {           // y'_k = y_2k+1 - F_k x_k - G_k x_k+1
      T M[D][D], x[D], yo[D];
      T* slot = ys + (size_t)(((2 * k + 2) << j) - 1) * D;
      lds_load_vec<T, D>(slot, yo);
      load_block<T, D>(Fp + (lv.offF[j] + g0 + k) * DD, M);
      lds_load_vec<T, D>(ys + (size_t)(((2 * k + 1) << j) - 1) * D, x);
      gemv_sub<T, D>(yo, M, x);
      if (2 * k + 2 < nj) {                              // right neighbour inside the tile
        load_block<T, D>(Gp + (lv.offG[j] + g0 + k) * DD, M);
        lds_load_vec<T, D>(ys + (size_t)(((2 * k + 3) << j) - 1) * D, x);
        gemv_sub<T, D>(yo, M, x);
      }
      if (more && (k & 1) == 0) {
        eliminate(j + 1, k >> 1, yo);
      } else {
        lds_store_vec<T, D>(slot, yo);
      }
    }
    __syncthreads();
    nj = no;
  }
  if (y_out != nullptr) {                                // the tile's surviving rows: nj = n0 >> nlev of them
    const int spt_out = SOLVE_TS >> lv.nlev;
    for (int r = tid; r < nj; r += SOLVE_NT) {
      T v[D];
      lds_load_vec<T, D>(ys + (size_t)(((r + 1) << lv.nlev) - 1) * D, v);
      store_vec<T, D>(y_out + ((size_t)blockIdx.x * spt_out + r) * D, v);
    }
  }
  if (owed_out != nullptr && tid == 0) {
    T ow[D];
    lds_load_vec<T, D>(owed, ow);
    store_vec<T, D>(owed_out + (size_t)blockIdx.x * D, ow);
  }
  block_sum2<SOLVE_NT>(mah, zero, red);
  if (tid == 0 && partial != nullptr) {
    partial[2 * (size_t)blockIdx.x] = mah;
    partial[2 * (size_t)blockIdx.x + 1] = 0.0;
  }
}


// ---- latency-bound passes (at most one tile per CU): every factor block up front ---------------------
// A pass over few tiles walks all levels of a tile; with the blocks of a level requested when the
// level starts, each level exposes one HBM round trip (3.5 us measured per level: 35 us for the
// 128-tile middle pass of N = 2^20, 12 us for the single-tile top pass).  But WHICH blocks a tile
// needs does not depend on any data: the 512 eliminations of level 0 belong to the 512 lanes, and
// the 512 eliminations of ALL deeper levels (256 + 128 + ... + 1 + 1) are dealt to the same 512
// lanes, one each -- level j >= 1, elimination k to lane (512 >> j) + k, the single elimination of
// level 10 to lane 0 (1024-row tiles; 512-row tiles, TSL = 9, likewise with 256 lanes).  A lane requests its D / F / G of level 0 and of its deep elimination (and
// its right-hand-side entry in the backward sweep) before anything else: ONE round trip for the
// whole pass; after it the levels only touch LDS (two barriers each).  Six blocks in registers:
// for blocks of at most 128 bytes (fp64 d <= 4, fp32 d <= 5), one workgroup per CU.
template <typename T, int D> constexpr bool solve_deep_supported() { return (size_t)D * D * sizeof(T) <= 128; }

struct DeepOwner {
  int j, k;           // the deep elimination (level >= 1) this lane owns, j = -1: none
};
template <int TSL>
__device__ __forceinline__ DeepOwner deep_owner(int tid) {
  DeepOwner o;
  if (tid == 0) { o.j = TSL; o.k = 0; return o; }
  const int hb = 31 - __clz(tid);          // tid in [2^hb, 2^(hb+1))
  o.j = (TSL - 1) - hb;                    // (TS / 2) >> j == 2^hb
  o.k = tid - (1 << hb);
  return o;
}

template <typename T, int D, int TSL>
__global__ __launch_bounds__((1 << TSL) / 2, 2) void halfsolve_deep_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ owed_in, int64_t n_owed, int spt_in, const T* __restrict__ y_in, int64_t n,
    T* __restrict__ xcrr, T* __restrict__ y_out, T* __restrict__ owed_out, double* __restrict__ partial) {
  constexpr int DD = D * D, SOLVE_TS = 1 << TSL, SOLVE_NT = SOLVE_TS / 2;     // (shadow the single-column constants)
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* ys = reinterpret_cast<T*>(solve_smem);                                   // [SOLVE_TS][D]
  double* red = reinterpret_cast<double*>(solve_smem + (size_t)SOLVE_TS * D * sizeof(T));
  T* owed = reinterpret_cast<T*>(red + 2 * (SOLVE_NT / 64));
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * SOLVE_TS;
  const int n0 = (int)((n - row0) < SOLVE_TS ? (n - row0) : SOLVE_TS);
  // ---- every load of the pass --------------------------------------------------------------------
  // level 0, elimination tid: row 2 tid, its D; the update of row 2 tid + 1 needs F and (when the
  // right neighbour 2 tid + 2 exists) G
  T L0[D][D], F0[D][D], G0[D][D];
  const int ne0 = (n0 + 1) >> 1, no0 = n0 >> 1;
  const int64_t g00 = row0 >> 1;
  if (tid < ne0) load_block<T, D>(Dp + (lv.offD[0] + g00 + tid) * DD, L0); else set_zero_block(L0);
  const bool upd0 = tid < no0 && lv.nlev >= 1, rgt0 = upd0 && (2 * tid + 2 < n0);
  if (upd0) load_block<T, D>(Fp + (lv.offF[0] + g00 + tid) * DD, F0); else set_zero_block(F0);
  if (rgt0) load_block<T, D>(Gp + (lv.offG[0] + g00 + tid) * DD, G0); else set_zero_block(G0);
  // the lane's deep elimination (level dj >= 1 of this pass, elimination dk)
  const DeepOwner own = deep_owner<TSL>(tid);
  const int dj = own.j, dk = own.k;
  const int nj_d = (dj <= lv.nlev) ? (n0 >> dj) : 0;                     // rows of the tile at level dj
  // level dj is eliminated by this pass when dj < nlev, or dj == nlev and the pass ends the system
  // there (a single surviving row that is eliminated too: the top pass)
  const bool elim_d = dj >= 1 && dk < ((nj_d + 1) >> 1) && (dj < lv.nlev);
  const bool upd_d = elim_d && dk < (nj_d >> 1) && (dj + 1 <= lv.nlev) && (dj < lv.nlev);
  const bool rgt_d = upd_d && (2 * dk + 2 < nj_d);
  T Ld[D][D], Fd[D][D], Gd[D][D];
  const int64_t g0d = row0 >> (dj + 1);
  if (elim_d) load_block<T, D>(Dp + (lv.offD[dj] + g0d + dk) * DD, Ld); else set_zero_block(Ld);
  if (upd_d) load_block<T, D>(Fp + (lv.offF[dj] + g0d + dk) * DD, Fd); else set_zero_block(Fd);
  if (rgt_d) load_block<T, D>(Gp + (lv.offG[dj] + g0d + dk) * DD, Gd); else set_zero_block(Gd);
  // what the tile's first elimination of every level owes the previous tile's last row: G of the
  // block left of the tile, one level per lane (lanes 0 .. nlev - 1)
  T Gl[D][D];
  const bool left = tid < lv.nlev && (row0 >> (tid + 1)) >= 1 && (n0 >> tid) >= 1;
  if (left) load_block<T, D>(Gp + (lv.offG[tid] + (row0 >> (tid + 1)) - 1) * DD, Gl); else set_zero_block(Gl);
  // the tile's rows
  for (int r = tid; r < n0; r += SOLVE_NT) {
    T v[D];
    load_vec<T, D>(y_in + (row0 + r) * D, v);
    const int64_t wn = row0 + r + 1;
    if (owed_in != nullptr && wn % spt_in == 0 && wn / spt_in < n_owed) {
      T w[D];
      load_vec<T, D>(owed_in + (wn / spt_in) * D, w);
#pragma unroll
      for (int i = 0; i < D; ++i) v[i] -= w[i];
    }
    lds_store_vec<T, D>(ys + r * D, v);
  }
  __syncthreads();
  double mah = 0.0, zero = 0.0;
  // x = D^-1 y of elimination k of level j -> the row's slot, xcrr
  auto eliminate = [&](int j, int k, const T (&L)[D][D]) {
    T x[D];
    T* slot = ys + (size_t)(((2 * k + 1) << j) - 1) * D;
    lds_load_vec<T, D>(slot, x);
    Chol<T, D> c;
    chol_from_dense<T, D>(L, c);
    fwd_subst<T, D>(c, x);
    lds_store_vec<T, D>(slot, x);
    store_vec<T, D>(xcrr + (lv.offD[j] + (row0 >> (j + 1)) + k) * D, x);
#pragma unroll
    for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
  };
  // y'_k = y_2k+1 - F x_k - G x_k+1 of level j -> the row's slot (row k of level j + 1)
  auto update = [&](int j, int k, bool right, const T (&F)[D][D], const T (&G)[D][D]) {
    T x[D], yo[D];
    T* slot = ys + (size_t)(((2 * k + 2) << j) - 1) * D;
    lds_load_vec<T, D>(slot, yo);
    lds_load_vec<T, D>(ys + (size_t)(((2 * k + 1) << j) - 1) * D, x);
    gemv_sub<T, D>(yo, F, x);
    if (right) {
      lds_load_vec<T, D>(ys + (size_t)(((2 * k + 3) << j) - 1) * D, x);
      gemv_sub<T, D>(yo, G, x);
    }
    lds_store_vec<T, D>(slot, yo);
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
  // owed to the previous tile's last row: sum over the levels of G_left x (first elimination of the level)
  if (left) {
    T x[D], w[D];
    lds_load_vec<T, D>(ys + (size_t)((1 << tid) - 1) * D, x);       // x of elimination 0 of level tid
#pragma unroll
    for (int i = 0; i < D; ++i) {
      T acc = T(0);
#pragma unroll
      for (int m2 = 0; m2 < D; ++m2) acc = fmaT(Gl[i][m2], x[m2], acc);
      w[i] = acc;
    }
    lds_store_vec<T, D>(owed + (size_t)(1 + tid) * D, w);
  }
  __syncthreads();
  if (y_out != nullptr) {                                // the tile's surviving rows: n0 >> nlev of them
    const int spt_out = SOLVE_TS >> lv.nlev;
    for (int r = tid; r < (n0 >> lv.nlev); r += SOLVE_NT) {
      T v[D];
      lds_load_vec<T, D>(ys + (size_t)(((r + 1) << lv.nlev) - 1) * D, v);
      store_vec<T, D>(y_out + ((size_t)blockIdx.x * spt_out + r) * D, v);
    }
  }
  if (owed_out != nullptr && tid == 0) {
    T ow[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ow[i] = T(0);
    for (int l = 0; l < lv.nlev; ++l) {
      if ((row0 >> (l + 1)) >= 1 && (n0 >> l) >= 1) {
        T w[D];
        lds_load_vec<T, D>(owed + (size_t)(1 + l) * D, w);
#pragma unroll
        for (int i = 0; i < D; ++i) ow[i] += w[i];
      }
    }
    store_vec<T, D>(owed_out + (size_t)blockIdx.x * D, ow);
  }
  block_sum2<SOLVE_NT>(mah, zero, red);
  if (tid == 0 && partial != nullptr) {
    partial[2 * (size_t)blockIdx.x] = mah;
    partial[2 * (size_t)blockIdx.x + 1] = 0.0;
  }
}

// ---- backward sweep --------------------------------------------------------------------------
// b      : right-hand side in CRR layout.   x_coarse : solution of this pass's surviving rows (one
//          per full tile; level first + LP, natural order), nullptr for the top pass.
// x_out  : solution of this pass's rows (level lv.first, natural order).
template <typename T, int D>
__global__ __launch_bounds__(SOLVE_NT) void backsolve_tile_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ b, const T* __restrict__ x_coarse, int64_t n, T* __restrict__ x_out) {
  constexpr int DD = D * D;
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* xs = reinterpret_cast<T*>(solve_smem);                                   // [SOLVE_TS][D]
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * SOLVE_TS;
  const int n0 = (int)((n - row0) < SOLVE_TS ? (n - row0) : SOLVE_TS);
  if (gridDim.x == 1) {
    warm_l2(Dp + lv.offD[0] * DD, (lv.endD - lv.offD[0]) * DD);
    warm_l2(Fp + lv.offF[0] * DD, (lv.endF - lv.offF[0]) * DD);
    warm_l2(Gp + lv.offG[0] * DD, (lv.endG - lv.offG[0]) * DD);
    warm_l2(b + lv.offD[0] * D, (lv.endD - lv.offD[0]) * D);
  }
  T xleft[D];                                            // x of the previous tile's last row
#pragma unroll
  for (int i = 0; i < D; ++i) xleft[i] = T(0);
  if (x_coarse != nullptr) {                             // solution of the rows that survived this pass's levels
    const int spt = SOLVE_TS >> lv.nlev;                 // per full tile, natural order
    if (blockIdx.x > 0) load_vec<T, D>(x_coarse + ((size_t)blockIdx.x * spt - 1) * D, xleft);
    for (int r = tid; r < (n0 >> lv.nlev); r += SOLVE_NT) {
      T v[D];
      load_vec<T, D>(x_coarse + ((size_t)blockIdx.x * spt + r) * D, v);
      lds_store_vec<T, D>(xs + (size_t)(((r + 1) << lv.nlev) - 1) * D, v);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = lv.nlev - 1; j >= 0; --j) {
    const int nj = n0 >> j;                              // floor at every level
    if (nj >= 1) {
      const int ne = (nj + 1) >> 1;
      const int64_t g0 = row0 >> (j + 1);
      for (int k = tid; k < ne; k += SOLVE_NT) {
        T r[D], M[D][D], xo[D];
        load_vec<T, D>(b + (lv.offD[j] + g0 + k) * D, r);
        if (2 * k + 1 < nj) {
          load_block<T, D>(Fp + (lv.offF[j] + g0 + k) * DD, M);
          lds_load_vec<T, D>(xs + (size_t)(((2 * k + 2) << j) - 1) * D, xo);
          gemvT_sub<T, D>(r, M, xo);
        }
        if (k >= 1) {
          load_block<T, D>(Gp + (lv.offG[j] + g0 + k - 1) * DD, M);
          lds_load_vec<T, D>(xs + (size_t)(((2 * k) << j) - 1) * D, xo);
          gemvT_sub<T, D>(r, M, xo);
        } else if (g0 >= 1) {                            // left neighbour = previous tile's last row
          load_block<T, D>(Gp + (lv.offG[j] + g0 - 1) * DD, M);
          gemvT_sub<T, D>(r, M, xleft);
        }
        T L[D][D];
        Chol<T, D> c;
        load_block<T, D>(Dp + (lv.offD[j] + g0 + k) * DD, L);
        chol_from_dense<T, D>(L, c);
        bwd_subst<T, D>(c, r);
        lds_store_vec<T, D>(xs + (size_t)(((2 * k + 1) << j) - 1) * D, r);
      }
    }
    __syncthreads();
  }
  for (int r = tid; r < n0; r += SOLVE_NT) {
    T v[D];
    lds_load_vec<T, D>(xs + (size_t)r * D, v);
    store_vec<T, D>(x_out + (row0 + r) * D, v);
  }
}


// backward counterpart of halfsolve_deep_kernel: same lane <-> elimination map, every block and
// right-hand-side entry requested before the first level runs
struct BackFlags {
  bool on, right, left_in, left_out;
};
// what elimination k of level j needs: x = D^-T (b - F^T x_right - G^T x_left)
template <typename T, int D>
__device__ __forceinline__ BackFlags back_request(const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp,
                                                  const T* __restrict__ b, const PassLevels& lv, int64_t row0, int n0, int j,
                                                  int k, bool on, T (&L)[D][D], T (&F)[D][D], T (&G)[D][D], T (&r)[D]) {
  constexpr int DD = D * D;
  const int nj = n0 >> j;
  const int64_t g0 = row0 >> (j + 1);
  BackFlags f;
  f.on = on && k < ((nj + 1) >> 1);
  f.right = f.on && (2 * k + 1 < nj);
  f.left_in = f.on && k >= 1;
  f.left_out = f.on && k == 0 && g0 >= 1;                // left neighbour = previous tile's last row
  if (f.on) {
    load_vec<T, D>(b + (lv.offD[j] + g0 + k) * D, r);
    load_block<T, D>(Dp + (lv.offD[j] + g0 + k) * DD, L);
  } else {
    set_zero_block(L);
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = T(0);
  }
  if (f.right) load_block<T, D>(Fp + (lv.offF[j] + g0 + k) * DD, F); else set_zero_block(F);
  if (f.left_in || f.left_out) load_block<T, D>(Gp + (lv.offG[j] + g0 + k - 1) * DD, G); else set_zero_block(G);
  return f;
}
template <typename T, int D>
__device__ __forceinline__ void back_run(T* xs, int j, int k, const BackFlags& f, const T (&L)[D][D], const T (&F)[D][D],
                                         const T (&G)[D][D], T (&r)[D], const T (&xleft)[D]) {
  if (!f.on) return;
  T xo[D];
  if (f.right) {
    lds_load_vec<T, D>(xs + (size_t)(((2 * k + 2) << j) - 1) * D, xo);
    gemvT_sub<T, D>(r, F, xo);
  }
  if (f.left_in) {
    lds_load_vec<T, D>(xs + (size_t)(((2 * k) << j) - 1) * D, xo);
    gemvT_sub<T, D>(r, G, xo);
  } else if (f.left_out) {
    gemvT_sub<T, D>(r, G, xleft);
  }
  Chol<T, D> c;
  chol_from_dense<T, D>(L, c);
  bwd_subst<T, D>(c, r);
  lds_store_vec<T, D>(xs + (size_t)(((2 * k + 1) << j) - 1) * D, r);
}

template <typename T, int D, int TSL>
__global__ __launch_bounds__((1 << TSL) / 2, 2) void backsolve_deep_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ b, const T* __restrict__ x_coarse, int64_t n, T* __restrict__ x_out) {
  constexpr int SOLVE_TS = 1 << TSL, SOLVE_NT = SOLVE_TS / 2;                 // (shadow the single-column constants)
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* xs = reinterpret_cast<T*>(solve_smem);                                   // [SOLVE_TS][D]
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * SOLVE_TS;
  const int n0 = (int)((n - row0) < SOLVE_TS ? (n - row0) : SOLVE_TS);
  T L0[D][D], F0[D][D], G0[D][D], r0[D], Ld[D][D], Fd[D][D], Gd[D][D], rd[D];
  const BackFlags f0 = back_request<T, D>(Dp, Fp, Gp, b, lv, row0, n0, 0, tid, lv.nlev >= 1, L0, F0, G0, r0);
  const DeepOwner own = deep_owner<TSL>(tid);
  const BackFlags fd = back_request<T, D>(Dp, Fp, Gp, b, lv, row0, n0, own.j, own.k, own.j >= 1 && own.j < lv.nlev, Ld, Fd,
                                          Gd, rd);
  T xleft[D];                                            // x of the previous tile's last row
#pragma unroll
  for (int i = 0; i < D; ++i) xleft[i] = T(0);
  if (x_coarse != nullptr) {                             // solution of the rows that survived this pass's levels
    const int spt = SOLVE_TS >> lv.nlev;
    if (blockIdx.x > 0) load_vec<T, D>(x_coarse + ((size_t)blockIdx.x * spt - 1) * D, xleft);
    for (int r = tid; r < (n0 >> lv.nlev); r += SOLVE_NT) {
      T v[D];
      load_vec<T, D>(x_coarse + ((size_t)blockIdx.x * spt + r) * D, v);
      lds_store_vec<T, D>(xs + (size_t)(((r + 1) << lv.nlev) - 1) * D, v);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = lv.nlev - 1; j >= 1; --j) {
    if (own.j == j) back_run<T, D>(xs, j, own.k, fd, Ld, Fd, Gd, rd, xleft);
    __syncthreads();
  }
  back_run<T, D>(xs, 0, tid, f0, L0, F0, G0, r0, xleft);
  __syncthreads();
  for (int r = tid; r < n0; r += SOLVE_NT) {
    T v[D];
    lds_load_vec<T, D>(xs + (size_t)r * D, v);
    store_vec<T, D>(x_out + (row0 + r) * D, v);
  }
}


// ---- solve(): forward AND backward sweep of the single-tile top pass in one launch -----------------
// cgps_solve runs halfsolve and backhalfsolve back to back, and at the top of the reduction both are
// one workgroup working on the same <= TS rows with the same factor blocks: the forward half of
// halfsolve_deep_kernel, then -- the blocks D and F still in registers, the right-hand side of every
// elimination being the x the forward half just produced in the same lane -- the backward half of
// backsolve_deep_kernel; only G_k-1 (the forward half held G_k) is requested again, from L2.  One
// launch, one prologue and one HBM round trip less.  x_top: solution of the pass's rows, natural order
// (what the next backward pass reads as its coarse solution, or the final x of a small system).
template <typename T, int D, int TSL>
__global__ __launch_bounds__((1 << TSL) / 2, 2) void solve_top_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, PassLevels lv,
    const T* __restrict__ owed_in, int64_t n_owed, int spt_in, const T* __restrict__ y_in, int64_t n,
    T* __restrict__ xcrr, T* __restrict__ x_top, double* __restrict__ partial) {
  constexpr int DD = D * D, SOLVE_TS = 1 << TSL, SOLVE_NT = SOLVE_TS / 2;     // (shadow the single-column constants)
  extern __shared__ __attribute__((aligned(16))) char solve_smem[];
  T* ys = reinterpret_cast<T*>(solve_smem);                                   // [SOLVE_TS][D]
  double* red = reinterpret_cast<double*>(solve_smem + (size_t)SOLVE_TS * D * sizeof(T));
  const int tid = threadIdx.x;
  const int n0 = (int)n;                                                      // one tile: n <= SOLVE_TS, row0 = 0
  // ---- every load of the forward half (as halfsolve_deep_kernel, tile 0) -------------------------
  T L0[D][D], F0[D][D], G0[D][D];
  const int ne0 = (n0 + 1) >> 1, no0 = n0 >> 1;
  const bool el0 = tid < ne0 && lv.nlev >= 1;
  if (el0) load_block<T, D>(Dp + (lv.offD[0] + tid) * DD, L0); else set_zero_block(L0);
  const bool upd0 = tid < no0 && lv.nlev >= 1, rgt0 = upd0 && (2 * tid + 2 < n0);
  if (upd0) load_block<T, D>(Fp + (lv.offF[0] + tid) * DD, F0); else set_zero_block(F0);
  if (rgt0) load_block<T, D>(Gp + (lv.offG[0] + tid) * DD, G0); else set_zero_block(G0);
  const DeepOwner own = deep_owner<TSL>(tid);
  const int dj = own.j, dk = own.k;
  const int nj_d = (dj <= lv.nlev) ? (n0 >> dj) : 0;
  const bool elim_d = dj >= 1 && dk < ((nj_d + 1) >> 1) && (dj < lv.nlev);
  const bool upd_d = elim_d && dk < (nj_d >> 1);
  const bool rgt_d = upd_d && (2 * dk + 2 < nj_d);
  T Ld[D][D], Fd[D][D], Gd[D][D];
  if (elim_d) load_block<T, D>(Dp + (lv.offD[dj] + dk) * DD, Ld); else set_zero_block(Ld);
  if (upd_d) load_block<T, D>(Fp + (lv.offF[dj] + dk) * DD, Fd); else set_zero_block(Fd);
  if (rgt_d) load_block<T, D>(Gp + (lv.offG[dj] + dk) * DD, Gd); else set_zero_block(Gd);
  for (int r = tid; r < n0; r += SOLVE_NT) {
    T v[D];
    load_vec<T, D>(y_in + (size_t)r * D, v);
    const int64_t wn = r + 1;
    if (owed_in != nullptr && wn % spt_in == 0 && wn / spt_in < n_owed) {
      T w[D];
      load_vec<T, D>(owed_in + (wn / spt_in) * D, w);
#pragma unroll
      for (int i = 0; i < D; ++i) v[i] -= w[i];
    }
    lds_store_vec<T, D>(ys + r * D, v);
  }
  __syncthreads();
  double mah = 0.0, zero = 0.0;
  T x0[D], xd[D];                                        // x of this lane's two eliminations (the backward half's right-hand sides)
#pragma unroll
  for (int i = 0; i < D; ++i) { x0[i] = T(0); xd[i] = T(0); }
  auto eliminate = [&](int j, int k, const T (&L)[D][D], T (&x)[D]) {
    T* slot = ys + (size_t)(((2 * k + 1) << j) - 1) * D;
    lds_load_vec<T, D>(slot, x);
    Chol<T, D> c;
    chol_from_dense<T, D>(L, c);
    fwd_subst<T, D>(c, x);
    lds_store_vec<T, D>(slot, x);
    if (xcrr != nullptr) store_vec<T, D>(xcrr + (lv.offD[j] + k) * D, x);
#pragma unroll
    for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
  };
  auto update = [&](int j, int k, bool right, const T (&F)[D][D], const T (&G)[D][D]) {
    T x[D], yo[D];
    T* slot = ys + (size_t)(((2 * k + 2) << j) - 1) * D;
    lds_load_vec<T, D>(slot, yo);
    lds_load_vec<T, D>(ys + (size_t)(((2 * k + 1) << j) - 1) * D, x);
    gemv_sub<T, D>(yo, F, x);
    if (right) {
      lds_load_vec<T, D>(ys + (size_t)(((2 * k + 3) << j) - 1) * D, x);
      gemv_sub<T, D>(yo, G, x);
    }
    lds_store_vec<T, D>(slot, yo);
  };
  if (el0) eliminate(0, tid, L0, x0);
  __syncthreads();
  for (int j = 0; j < lv.nlev; ++j) {
    if (j == 0) { if (upd0) update(0, tid, rgt0, F0, G0); }
    else if (dj == j && upd_d) update(j, dk, rgt_d, Fd, Gd);
    __syncthreads();
    if (j + 1 < lv.nlev && dj == j + 1 && elim_d) eliminate(j + 1, dk, Ld, xd);
    __syncthreads();
  }
  // ---- backward half: G_k-1 of the lane's two eliminations (the forward half held G_k) -------------
  const bool lft0 = el0 && tid >= 1, lftd = elim_d && dk >= 1;
  if (lft0) load_block<T, D>(Gp + (lv.offG[0] + tid - 1) * DD, G0);
  if (lftd) load_block<T, D>(Gp + (lv.offG[dj] + dk - 1) * DD, Gd);
  // (the tile is the whole system: after the forward half every slot of an eliminated row holds its
  // forward x, which the backward half overwrites with the solution, coarse levels first)
  auto back = [&](int j, int k, bool right, bool left, const T (&L)[D][D], const T (&F)[D][D], const T (&G)[D][D], T (&r)[D]) {
    T xo[D];
    if (right) {
      lds_load_vec<T, D>(ys + (size_t)(((2 * k + 2) << j) - 1) * D, xo);
      gemvT_sub<T, D>(r, F, xo);
    }
    if (left) {
      lds_load_vec<T, D>(ys + (size_t)(((2 * k) << j) - 1) * D, xo);
      gemvT_sub<T, D>(r, G, xo);
    }
    Chol<T, D> c;
    chol_from_dense<T, D>(L, c);
    bwd_subst<T, D>(c, r);
    lds_store_vec<T, D>(ys + (size_t)(((2 * k + 1) << j) - 1) * D, r);
  };
#pragma unroll 1
  for (int j = lv.nlev - 1; j >= 1; --j) {
    if (elim_d && dj == j) back(j, dk, upd_d, lftd, Ld, Fd, Gd, xd);      // 2k+1 < nj  <=>  k < nj / 2
    __syncthreads();
  }
  if (el0) back(0, tid, upd0, lft0, L0, F0, G0, x0);
  __syncthreads();
  for (int r = tid; r < n0; r += SOLVE_NT) {
    T v[D];
    lds_load_vec<T, D>(ys + (size_t)r * D, v);
    store_vec<T, D>(x_top + (size_t)r * D, v);
  }
  block_sum2<SOLVE_NT>(mah, zero, red);
  if (tid == 0 && partial != nullptr) {
    partial[2 * (size_t)blockIdx.x] = mah;
    partial[2 * (size_t)blockIdx.x + 1] = 0.0;
  }
}

}  // namespace cgps
