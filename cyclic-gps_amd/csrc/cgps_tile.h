// Fused solve + log-det (mahal_and_det, reference cyclic_reduction.py:380-438):
// the whole reduction in two or three launches, every input byte read once.
//
// x^T J^-1 x and log|J| do not depend on the elimination order, so this path
// orders the block Gaussian elimination for the hardware (the factor-emitting
// decompose keeps the reference's even/odd order, its factors being API output):
//
//  stage 1, chunk_reduce_kernel: the time axis is cut into chunks of C block
//    rows, one chunk per lane.  A lane streams its rows from HBM (16-byte loads,
//    next row prefetched while the current one is eliminated) and eliminates
//    rows c0 .. c0+C-2 left to right in registers.  The Schur complement of the
//    chunk interior lands on its two boundary rows: the chunk's own last row
//    (kept: R_s, y_s, and C_s = its new coupling to the previous chunk's last
//    row) and that previous row (additive update dRa, dya).  No lane idles and
//    nothing but the inputs crosses HBM.
//  stage 2, tile_cr: the NT boundary rows of a workgroup form a block
//    tridiagonal system in LDS that is reduced by even/odd cyclic reduction
//    (log2 NT levels, one barrier each) to ONE boundary row + the update for
//    the previous workgroup's boundary row: a "record".
//  stage 3, record_reduce_kernel: records are rows of a (N / (C NT))-row system;
//    the same tile_cr reduces them (recursively for very large N) and the last
//    launch eliminates the final row and sums the partial log-det / mahal.
//
// Rows past the end of the system are padded with identity blocks (R = I,
// O = 0, y = 0): they contribute log 1 = 0 and 0 to the sums, so every tile is
// full and the reduction code has no ragged cases.
#pragma once
#include "cgps_level.h"

namespace cgps {

// ---- LDS tile of NT block rows: R[NT][DD], O[NT][DD] (O[i] couples row i and the next
// active row), y[NT][D].  16-byte granules of a block are XOR-swizzled by a fold of the
// row index so that the strided row access of every reduction level spreads over banks.
template <typename T, int D>
struct LdsTile {
  static constexpr int DD = D * D;
  static constexpr int VN = Vec16<T>::N;
  static constexpr int G = (DD % VN == 0) ? DD / VN : 0;          // granules per block
  static constexpr bool SWZ = G >= 2 && (G & (G - 1)) == 0;
  T* R;
  T* O;
  T* y;
  static __device__ __forceinline__ int key(int row) { return (row ^ (row >> 3) ^ (row >> 6) ^ (row >> 9)) & (G - 1); }

  static __device__ __forceinline__ void load_blk(const T* base, int row, T (&A)[D][D]) {
    if constexpr (SWZ) {
      using V = typename Vec16<T>::type;
      const V* q = reinterpret_cast<const V*>(base + (size_t)row * DD);
      const int kk = key(row);
      T flat[DD];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        V v = q[g ^ kk];
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) flat[g * VN + t] = e[t];
      }
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) A[i][j] = flat[i * D + j];
    } else {
      load_block<T, D>(base + (size_t)row * DD, A);
    }
  }
  static __device__ __forceinline__ void store_blk(T* base, int row, const T (&A)[D][D]) {
    if constexpr (SWZ) {
      using V = typename Vec16<T>::type;
      V* q = reinterpret_cast<V*>(base + (size_t)row * DD);
      const int kk = key(row);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        V v;
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) e[t] = A[(g * VN + t) / D][(g * VN + t) % D];
        q[g ^ kk] = v;
      }
    } else {
      store_block<T, D>(base + (size_t)row * DD, A);
    }
  }
};

template <typename T, int D>
__host__ __device__ constexpr size_t tile_lds_bytes(int nt) {
  return (size_t)nt * (2 * D * D + D) * sizeof(T);
}

// Running product of pivots with a rare fold into a log sum: one log() per lane
// instead of one per eliminated row.
struct PivotLog {
  double prod = 1.0, logsum = 0.0;
  __device__ __forceinline__ void mul(double p) {
    prod *= p;
    if (!(prod > 1e-150 && prod < 1e150)) { logsum += log(prod); prod = 1.0; }
  }
  __device__ __forceinline__ double value() const { return logsum + log(prod); }
};

// One elimination: current row (Rc, yc) with coupling Cc to the left boundary row and
// coupling On to the next row (Rn, yn).  Updates the left boundary accumulators, turns
// (Rn, yn) into the next current row and Cc into its coupling to the left boundary.
template <typename T, int D>
__device__ __forceinline__ void eliminate_forward(T (&Rc)[D][D], T (&yc)[D], T (&Cc)[D][D], T (&dRa)[D][D],
                                                  T (&dya)[D], T (&On)[D][D], T (&Rn)[D][D], T (&yn)[D],
                                                  PivotLog& pl, double& mah, bool& fail) {
  Chol<T, D> c;
  pl.mul(chol_lower<T, D>(Rc, c, fail));
  fwd_subst<T, D>(c, yc);                      // x = D^-1 y
#pragma unroll
  for (int i = 0; i < D; ++i) mah += (double)yc[i] * (double)yc[i];
  T G[D][D];
  rsolve_lt_transposed<T, D>(c, Cc, G);        // G = Cc^T D^-T   (coupling to the left boundary)
  syrk_sub_lower<T, D>(dRa, G);
  gemv_sub<T, D>(dya, G, yc);
  rsolve_lt<T, D>(c, On);                      // F = On D^-T     (coupling to the next row)
  syrk_sub_lower<T, D>(Rn, On);
  gemv_sub<T, D>(yn, On, yc);
  neg_abt<T, D>(Cc, On, G);                    // next row <-> left boundary: -F G^T
#pragma unroll
  for (int i = 0; i < D; ++i) {
    yc[i] = yn[i];
#pragma unroll
    for (int j = 0; j <= i; ++j) Rc[i][j] = Rn[i][j];
  }
}

// Even/odd cyclic reduction of the NT-row system held in the LDS tile, in place, rows
// at level l living at slots (m+1) 2^l - 1.  Thread k < NT / 2^(l+1) owns the even row
// e = (2k+1) 2^l - 1 (eliminated) and the odd row o = e + 2^l (kept); the Cholesky of
// the right even neighbour is recomputed instead of exchanged.  Thread 0 also carries
// the coupling of the tile's first active row to the row left of the tile
// (Cleft) and the additive update for that row (dRl, dyl).  On return slot NT-1 holds
// the tile's boundary row.  (cf. decompose_step, cyclic_reduction.py:225-254.)
template <typename T, int D, int NT>
__device__ __forceinline__ void tile_cr(LdsTile<T, D>& t, T (&Cleft)[D][D], T (&dRl)[D][D], T (&dyl)[D],
                                        PivotLog& pl, double& mah, bool& fail) {
  using LT = LdsTile<T, D>;
  const int k = threadIdx.x;
#pragma unroll 1
  for (int s = 1; s < NT; s <<= 1) {
    if (k < NT / (2 * s)) {
      const int e = (2 * k + 1) * s - 1, o = e + s, eR = o + s;
      T A[D][D], x[D];
      Chol<T, D> c;
      LT::load_blk(t.R, e, A);
      pl.mul(chol_lower<T, D>(A, c, fail));
      load_vec<T, D>(t.y + e * D, x);
      fwd_subst<T, D>(c, x);
#pragma unroll
      for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
      T F[D][D], Ro[D][D], yo[D];
      LT::load_blk(t.O, e, F);
      rsolve_lt<T, D>(c, F);
      LT::load_blk(t.R, o, Ro);
      syrk_sub_lower<T, D>(Ro, F);
      load_vec<T, D>(t.y + o * D, yo);
      gemv_sub<T, D>(yo, F, x);
      if (k == 0) {
        T G0[D][D];
        rsolve_lt_transposed<T, D>(c, Cleft, G0);
        syrk_sub_lower<T, D>(dRl, G0);
        gemv_sub<T, D>(dyl, G0, x);
        neg_abt<T, D>(Cleft, F, G0);
      }
      if (eR < NT) {
        bool f2 = false;
        Chol<T, D> c2;
        LT::load_blk(t.R, eR, A);
        chol_lower<T, D>(A, c2, f2);
        T Oo[D][D], G[D][D], x2[D];
        LT::load_blk(t.O, o, Oo);
        rsolve_lt_transposed<T, D>(c2, Oo, G);
        syrk_sub_lower<T, D>(Ro, G);
        load_vec<T, D>(t.y + eR * D, x2);
        fwd_subst<T, D>(c2, x2);
        gemv_sub<T, D>(yo, G, x2);
        if (eR + s < NT) {
          T F2[D][D], On_[D][D];
          LT::load_blk(t.O, eR, F2);
          rsolve_lt<T, D>(c2, F2);
          neg_abt<T, D>(On_, F2, G);
          LT::store_blk(t.O, o, On_);
        }
      }
      mirror_lower<T, D>(Ro);
      LT::store_blk(t.R, o, Ro);
      store_vec<T, D>(t.y + o * D, yo);
    }
    __syncthreads();
  }
}

// A record = what a tile leaves behind: its boundary row (Rs, ys), that row's coupling
// to the previous tile's boundary row (Cs = J[this, previous]) and the additive update
// (dRa, dya) for the previous tile's boundary row.
template <typename T, int D>
struct RecordLayout {
  static constexpr int DD = D * D;
  static constexpr int STRIDE = ((3 * DD + 2 * D + 3) / 4) * 4;   // elements, 16/32-byte aligned
  static constexpr int RS = 0, CS = DD, DRA = 2 * DD, YS = 3 * DD, DYA = 3 * DD + D;
};

template <typename T, int D>
__device__ __forceinline__ void set_identity(T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = (i == j) ? T(1) : T(0);
}
template <typename T, int D>
__device__ __forceinline__ void set_zero(T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = T(0);
}
template <typename T, int D>
__device__ __forceinline__ void set_zero(T (&v)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = T(0);
}

// Write the tile's record and its partial sums.  Called by all threads after tile_cr.
template <typename T, int D, int NT>
__device__ __forceinline__ void emit_record(LdsTile<T, D>& t, const T (&Cleft)[D][D], T (&dRl)[D][D],
                                            const T (&dyl)[D], T* __restrict__ rec, double mah, double logp,
                                            double* __restrict__ partial, double* red) {
  using RL = RecordLayout<T, D>;
  if (threadIdx.x == 0) {
    T Rs_[D][D], ys_[D];
    LdsTile<T, D>::load_blk(t.R, NT - 1, Rs_);
    load_vec<T, D>(t.y + (NT - 1) * D, ys_);
    T* r = rec + (size_t)blockIdx.x * RL::STRIDE;
    store_block<T, D>(r + RL::RS, Rs_);
    store_block<T, D>(r + RL::CS, Cleft);
    mirror_lower<T, D>(dRl);
    store_block<T, D>(r + RL::DRA, dRl);
    store_vec<T, D>(r + RL::YS, ys_);
    store_vec<T, D>(r + RL::DYA, dyl);
  }
  block_sum2<NT>(mah, logp, red);
  if (threadIdx.x == 0) {
    partial[2 * (size_t)blockIdx.x] = mah;
    partial[2 * (size_t)blockIdx.x + 1] = logp;
  }
}

// ---- stage 1 -----------------------------------------------------------------------------
template <typename T, int D, int C, int NT>
__global__ __launch_bounds__(NT) void chunk_reduce_kernel(const T* __restrict__ Rg, const T* __restrict__ Og,
                                                          const T* __restrict__ yg, int64_t N,
                                                          T* __restrict__ rec, double* __restrict__ partial,
                                                          int* __restrict__ info) {
  constexpr int DD = D * D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LdsTile<T, D> t;
  t.R = reinterpret_cast<T*>(smem);
  t.O = t.R + NT * DD;
  t.y = t.O + NT * DD;
  double* red = reinterpret_cast<double*>(t.y + NT * D);
  T* xch = reinterpret_cast<T*>(red + 2 * (NT / 64));        // [NT/64][DD + D] wave-boundary exchange

  const int tid = threadIdx.x;
  const int64_t r0 = ((int64_t)blockIdx.x * NT + tid) * C;
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;

  T Rc[D][D], yc[D], Cc[D][D], dRa[D][D], dya[D];
  set_zero<T, D>(dRa);
  set_zero<T, D>(dya);
  if (r0 < N) {
    load_block<T, D>(Rg + r0 * DD, Rc);
    load_vec<T, D>(yg + r0 * D, yc);
  } else {
    set_identity<T, D>(Rc);
    set_zero<T, D>(yc);
  }
  if (r0 >= 1 && r0 < N) load_block<T, D>(Og + (r0 - 1) * DD, Cc);
  else set_zero<T, D>(Cc);

#pragma unroll 1
  for (int j = 0; j < C - 1; ++j) {
    const int64_t rn = r0 + j + 1;
    T Rn[D][D], On[D][D], yn[D];
    if (rn < N) {
      load_block<T, D>(Rg + rn * DD, Rn);
      load_block<T, D>(Og + (rn - 1) * DD, On);
      load_vec<T, D>(yg + rn * D, yn);
    } else {
      set_identity<T, D>(Rn);
      set_zero<T, D>(On);
      set_zero<T, D>(yn);
    }
    eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, On, Rn, yn, pl, mah, fail);
  }
  if (fail) report_fail(info, r0 < N ? r0 : N - 1);

  // the update a lane computed for the row left of its chunk belongs to the previous lane's
  // kept row: fetch it from lane+1 (wave-local shuffle; LDS across the wave boundary)
  {
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0 && w > 0) {
      T* p = xch + (w - 1) * (DD + D);
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j2 = 0; j2 <= i; ++j2) p[i * D + j2] = dRa[i][j2];
        p[DD + i] = dya[i];
      }
    }
    __syncthreads();
    T nR[D][D], ny[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j2 = 0; j2 <= i; ++j2) nR[i][j2] = __shfl_down(dRa[i][j2], 1, 64);
      ny[i] = __shfl_down(dya[i], 1, 64);
    }
    if (lane == 63 && w < NT / 64 - 1) {
      const T* p = xch + w * (DD + D);
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j2 = 0; j2 <= i; ++j2) nR[i][j2] = p[i * D + j2];
        ny[i] = p[DD + i];
      }
    }
    if (tid < NT - 1) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j2 = 0; j2 <= i; ++j2) Rc[i][j2] += nR[i][j2];
        yc[i] += ny[i];
      }
    }
  }
  mirror_lower<T, D>(Rc);
  LdsTile<T, D>::store_blk(t.R, tid, Rc);
  store_vec<T, D>(t.y + tid * D, yc);
  if (tid > 0) LdsTile<T, D>::store_blk(t.O, tid - 1, Cc);
  if (tid != 0) {                       // only thread 0 carries the tile's left boundary
    set_zero<T, D>(Cc);
    set_zero<T, D>(dRa);
    set_zero<T, D>(dya);
  }
  __syncthreads();
  bool fail2 = false;
  tile_cr<T, D, NT>(t, Cc, dRa, dya, pl, mah, fail2);
  if (fail2) report_fail(info, r0 < N ? r0 : N - 1);
  emit_record<T, D, NT>(t, Cc, dRa, dya, rec, mah, pl.value(), partial, red);
}

// ---- stage 3 -----------------------------------------------------------------------------
// Records in -> records out (FINAL = false), or -> out2 = {mahal, logdet} (FINAL = true, one
// workgroup).  Row w of this stage: R = Rs[w] + dRa[w+1], y = ys[w] + dya[w+1], coupling to
// row w+1: Cs[w+1].
template <typename T, int D, int NT, bool FINAL>
__global__ __launch_bounds__(NT) void record_reduce_kernel(const T* __restrict__ rin, int64_t n,
                                                           T* __restrict__ rout, double* __restrict__ partial_out,
                                                           const double* __restrict__ partial_in, int64_t n_partial,
                                                           double* __restrict__ out2, int* __restrict__ info,
                                                           int64_t rows_per_record, int64_t N) {
  constexpr int DD = D * D;
  using RL = RecordLayout<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LdsTile<T, D> t;
  t.R = reinterpret_cast<T*>(smem);
  t.O = t.R + NT * DD;
  t.y = t.O + NT * DD;
  double* red = reinterpret_cast<double*>(t.y + NT * D);

  const int tid = threadIdx.x;
  const int64_t w = (int64_t)blockIdx.x * NT + tid;
  T Rc[D][D], yc[D], Cc[D][D], dRa[D][D], dya[D];
  if (w < n) {
    const T* r = rin + (size_t)w * RL::STRIDE;
    load_block<T, D>(r + RL::RS, Rc);
    load_vec<T, D>(r + RL::YS, yc);
    load_block<T, D>(r + RL::CS, Cc);
    load_block<T, D>(r + RL::DRA, dRa);
    load_vec<T, D>(r + RL::DYA, dya);
  } else {
    set_identity<T, D>(Rc);
    set_zero<T, D>(yc);
    set_zero<T, D>(Cc);
    set_zero<T, D>(dRa);
    set_zero<T, D>(dya);
  }
  if (tid < NT - 1 && w + 1 < n) {
    const T* r = rin + (size_t)(w + 1) * RL::STRIDE;
    T nR[D][D], ny[D];
    load_block<T, D>(r + RL::DRA, nR);
    load_vec<T, D>(r + RL::DYA, ny);
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j = 0; j < D; ++j) Rc[i][j] += nR[i][j];
      yc[i] += ny[i];
    }
  }
  LdsTile<T, D>::store_blk(t.R, tid, Rc);
  store_vec<T, D>(t.y + tid * D, yc);
  if (tid > 0) LdsTile<T, D>::store_blk(t.O, tid - 1, Cc);
  if (tid != 0) {
    set_zero<T, D>(Cc);
    set_zero<T, D>(dRa);
    set_zero<T, D>(dya);
  }
  __syncthreads();
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;
  tile_cr<T, D, NT>(t, Cc, dRa, dya, pl, mah, fail);
  if constexpr (!FINAL) {
    if (fail) {
      const int64_t last = (w + 1) * rows_per_record;
      report_fail(info, (last < N ? last : N) - 1);
    }
    emit_record<T, D, NT>(t, Cc, dRa, dya, rout, mah, pl.value(), partial_out, red);
  } else {
    if (tid == 0) {                        // the very last row of the whole system
      T A[D][D], x[D];
      Chol<T, D> c;
      LdsTile<T, D>::load_blk(t.R, NT - 1, A);
      pl.mul(chol_lower<T, D>(A, c, fail));
      load_vec<T, D>(t.y + (NT - 1) * D, x);
      fwd_subst<T, D>(c, x);
#pragma unroll
      for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
    }
    if (fail) report_fail(info, N - 1);
    double logp = pl.value();
    for (int64_t i = tid; i < n_partial; i += NT) {       // partial sums of every earlier launch
      mah += partial_in[2 * i];
      logp += partial_in[2 * i + 1];
    }
    block_sum2<NT>(mah, logp, red);
    if (tid == 0) { out2[0] = mah; out2[1] = logp; }
  }
}

// ---- host side ------------------------------------------------------------------------------
template <typename T, int D> constexpr bool tile_supported() {
  return (sizeof(T) == 8 && D <= 4) || (sizeof(T) == 4 && D <= 5);
}
template <typename T, int D> struct TileCfg {
  static constexpr int C = 8;        // rows per lane in stage 1
  static constexpr int NT1 = 256;    // lanes per workgroup in stage 1
  static constexpr int NT3 = 256;    // records per workgroup in stage 3
};

inline size_t tile_ws_bytes(int64_t N, int d, size_t s) {
  // records of every stage + partial sums; generous closed form (stage 1 has N / (C NT) tiles)
  const int64_t tiles = N / (8 * 256) + 2;
  const size_t stride = (size_t)(((3 * d * d + 2 * d + 3) / 4) * 4) * s;
  return ((size_t)(2 * tiles + 4) * stride + (size_t)(2 * tiles + 8) * 16 + 1024 + 255) & ~(size_t)255;
}

template <typename T, int D>
size_t stage_lds_bytes(int nt) {
  return tile_lds_bytes<T, D>(nt) + 2 * (nt / 64) * sizeof(double) + (size_t)(nt / 64) * (D * D + D) * sizeof(T) + 64;
}

// returns 0 on success, -1 when the workspace is too small
template <typename T, int D>
int run_tile_mahal_logdet(const T* Rs, const T* Os, const T* x, int64_t N, char* ws, size_t ws_bytes, double* out2,
                          int* info, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  using Cfg = TileCfg<T, D>;
  using RL = RecordLayout<T, D>;
  if (ws_bytes < tile_ws_bytes(N, D, sizeof(T))) return -1;
  const int64_t rows_per_tile = (int64_t)Cfg::C * Cfg::NT1;
  const int64_t tiles = (N + rows_per_tile - 1) / rows_per_tile;
  // workspace: [partials: 2*tiles+8 pairs][records A: tiles+2][records B: tiles+2]
  double* partial = reinterpret_cast<double*>(ws);
  const size_t pbytes = ((size_t)(2 * (N / (8 * 256) + 2) + 8) * 16 + 255) & ~(size_t)255;
  T* recA = reinterpret_cast<T*>(ws + pbytes);
  T* recB = recA + (size_t)(tiles + 2) * RL::STRIDE;
  (void)hipMemsetAsync(info, 0, sizeof(int), st);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_lds_bytes<T, D>(Cfg::NT1));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&record_reduce_kernel<T, D, Cfg::NT3, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_lds_bytes<T, D>(Cfg::NT3));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&record_reduce_kernel<T, D, Cfg::NT3, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_lds_bytes<T, D>(Cfg::NT3));
    attr_done = true;
  }
  const size_t lds1 = stage_lds_bytes<T, D>(Cfg::NT1), lds3 = stage_lds_bytes<T, D>(Cfg::NT3);
  if (ev_start) (void)hipEventRecord(ev_start, st);
  hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1>), dim3((unsigned)tiles), dim3(Cfg::NT1),
                     lds1, st, Rs, Os, x, N, recA, partial, info);
  if (ev_stop) (void)hipEventRecord(ev_stop, st);
  int64_t n = tiles, npart = tiles, rows_per_record = rows_per_tile;
  T *rin = recA, *rout = recB;
  while (n > Cfg::NT3) {
    const int64_t g = (n + Cfg::NT3 - 1) / Cfg::NT3;
    hipLaunchKernelGGL((record_reduce_kernel<T, D, Cfg::NT3, false>), dim3((unsigned)g), dim3(Cfg::NT3),
                       lds3, st, rin, n, rout, partial + 2 * npart,
                       (const double*)nullptr, (int64_t)0, (double*)nullptr, info, rows_per_record, N);
    npart += g;
    n = g;
    rows_per_record *= Cfg::NT3;
    T* tmp = rin; rin = rout; rout = tmp;
  }
  hipLaunchKernelGGL((record_reduce_kernel<T, D, Cfg::NT3, true>), dim3(1), dim3(Cfg::NT3),
                     lds3, st, rin, n, (T*)nullptr, (double*)nullptr,
                     (const double*)partial, npart, out2, info, rows_per_record, N);
  return 0;
}

}  // namespace cgps
