// Factor-emitting cyclic reduction (decompose) for the SMALL passes: the latency-bound tail of a
// large factorisation (<= 2^15 rows left) and small systems as a whole.  The bulk passes are
// cgps_decomp_tile.h (one wave per tile, registers, three levels per pass: built for throughput);
// here a pass is a handful of tiles, what counts is the time of ONE tile, and a tile runs 8
// levels per launch with each elimination split over four waves (~1.2 us per level).
//
// A workgroup holds a TS = 256-row tile (R and the couplings) in LDS and runs the role-split
// reduction of cgps_tile.h on it (four waves share one elimination: left products / right
// update / two halves of the new coupling), except that
//   * it starts at level 0 of the pass (no streaming stage: the order is prescribed),
//   * role 0 also writes D (dense lower factor) and G, role 1 writes F, straight into the
//     packed per-level arrays (index = global elimination index of that level),
//   * the LAST tile follows the reference's size rule (its last row IS eliminated when it is
//     even, cyclic_reduction.py:240-248); every other tile is full and keeps its last row as
//     the boundary, leaving a record (row, coupling to the previous tile's row, update owed to
//     it) that the next pass assembles.
// Record format and the DRA rule are those of cgps_decomp_tile.h (spt_in = survivors per tile of
// the pass that wrote the input records; this kernel leaves one per tile).
#pragma once
#include "cgps_tile.h"

namespace cgps {

#ifndef CGPS_DECOMP_QUAD
#define CGPS_DECOMP_QUAD 1     // 4 x 4 / 8 x 8 blocks: four lanes per elimination (0: role split, for A/B builds)
#endif
constexpr int DECL_LP = 8;
constexpr int DECL_TS = 1 << DECL_LP;     // 256 rows per tile
constexpr int DECL_NT = 256;             // four waves
constexpr int DECL_MAXLEV = DECL_LP + 1;

struct DecompLevelsL {
  int64_t offD[DECL_MAXLEV], offF[DECL_MAXLEV], offG[DECL_MAXLEV];
  int nlev;
};

// Blocks whose 256-row tile does not fit the LDS (fp64 d = 6, 7, 8) take tiles of 2^6 rows, six levels per launch,
// for EVERY pass of their factorisation (cgps_decompose.hip: run_decompose_lds_only) -- 2^20 rows in four launches
// instead of one per level; two to four such tiles share a CU.
template <typename T, int D> constexpr int decomp_lds_lp() {
  return (((size_t)DECL_TS * 2 * D * D + D * D) * sizeof(T) + 4096 <= 160 * 1024) ? DECL_LP : 6;
}
template <typename T, int D, int LP = DECL_LP>
constexpr size_t decomp_lds_tile_bytes() {
  return (((size_t)(1 << LP) * 2 * D * D + D * D) * sizeof(T) + 15 & ~(size_t)15) + 256;
}

// Reduction of the n0-row tile with factor emission.  keep_last: the tile's last row is a
// boundary (never eliminated).  row0 = index of the tile's first row at the pass's first level.
// Returns the number of executed levels.  LDS: R[TS][DD], Oc[TS+1][DD] (as LdsTile, y unused).
template <typename T, int D>
__device__ __forceinline__ int tile_cr_factor(LdsTile<T, D>& t, int n0, bool keep_last, int64_t row0,
                                              const DecompLevelsL& lv, T* __restrict__ Dp, T* __restrict__ Fp,
                                              T* __restrict__ Gp, int lvl_first, int* info, bool& fail) {
  using LT = LdsTile<T, D>;
  constexpr int DD = D * D;
  constexpr int DH = (D + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int role = wave & 3;
  const int K = n0 - 1;
  const int kl = keep_last ? 1 : 0;
  int levels = 0;
#pragma unroll 1
  for (int s = 1, j = 0; j < lv.nlev && (keep_last ? (s - 1) < K : (n0 >> j) >= 1); s <<= 1, ++j, ++levels) {
    const int M = n0 >> j, h = s >> 1;
    const int n_elim = (M + 1) / 2;
    const int64_t g0 = row0 >> (j + 1);
#pragma unroll 1
    for (int k0 = 0; k0 < n_elim; k0 += 64) {
      const int k = k0 + lane;
      const int e = (2 * k + 1) * s - 1;
      const bool act = (2 * k < M) && (e != K || !keep_last);
      const bool has_o = act && ((2 * k + 1 < M) || (keep_last && e < K));
      const int o = (2 * k + 1 < M) ? e + s : K;
      const int64_t ge = g0 + k;
      T W[D][D];
      if (act) {
        T A[D][D];
        LT::load_blk(t.R, e, A);
        if ((s > 1) && (e + h < K + 1 - kl)) {
          T P[D][D];
          LT::load_blk(t.R, e + h, P);
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int jj = 0; jj <= i; ++jj) A[i][jj] -= P[i][jj];
        }
        Chol<T, D> c;
        bool f = false;
        chol_lower<T, D>(A, c, f);
        if (role == 0) {
          if (f) {
            fail = true;
            report_fail(info, ((row0 + e + 1) << lvl_first) - 1);
          }
          T L[D][D];
          chol_to_dense<T, D>(c, L);
          store_block<T, D>(Dp + (lv.offD[j] + ge) * DD, L);
          T Ol[D][D], G[D][D];
          LT::load_blk(t.Oc, e - s + 1, Ol);
          rsolve_lt_transposed<T, D>(c, Ol, G);
          if (ge >= 1) store_block<T, D>(Gp + (lv.offG[j] + ge - 1) * DD, G);
          syrk_lower<T, D>(W, G);
        } else if (role == 1) {
          if (has_o) {
            T F[D][D];
            LT::load_blk(t.Oc, e + 1, F);
            rsolve_lt<T, D>(c, F);
            store_block<T, D>(Fp + (lv.offF[j] + ge) * DD, F);
            LT::load_blk(t.R, o, W);
            if ((s > 1) && (o + h < K + 1 - kl)) {
              T P[D][D];
              LT::load_blk(t.R, o + h, P);
#pragma unroll
              for (int i = 0; i < D; ++i)
#pragma unroll
                for (int jj = 0; jj <= i; ++jj) W[i][jj] -= P[i][jj];
            }
            syrk_sub_lower<T, D>(W, F);
            mirror_lower<T, D>(W);
          }
        } else if (has_o) {
          T Ol[D][D], G[D][D], F[D][D];
          LT::load_blk(t.Oc, e - s + 1, Ol);
          rsolve_lt_transposed<T, D>(c, Ol, G);
          LT::load_blk(t.Oc, e + 1, F);
          const int i0 = (role == 2) ? 0 : DH, i1 = (role == 2) ? DH : D;
#pragma unroll
          for (int i = 0; i < D; ++i) {
            if (i >= i0 && i < i1) {
              fwd_subst<T, D>(c, F[i]);
#pragma unroll
              for (int jj = 0; jj < D; ++jj) {
                T sacc = T(0);
#pragma unroll
                for (int m = 0; m < D; ++m) sacc = __builtin_fma(-F[i][m], G[jj][m], sacc);
                W[i][jj] = sacc;
              }
            }
          }
        }
      }
      __syncthreads();
      if (act) {
        if (role == 0) {
          LT::store_blk(t.R, e, W);
        } else if (has_o) {
          if (role == 1) LT::store_blk(t.R, o, W);
          else if (role == 2) LT::template store_rows<0, DH>(t.Oc, e - s + 1, W);
          else LT::template store_rows<DH, D>(t.Oc, e - s + 1, W);
        }
      }
      __syncthreads();
    }
  }
  return levels;
}

// The same with FOUR LANES PER ELIMINATION for 4 x 4 and 8 x 8 blocks (QuadTile, cgps_tile_quad.h): a DPP
// quad takes an elimination, lane q owning matrix rows RP q .. RP q + RP - 1 of every block; the Cholesky
// redundantly on the four lanes (lane 0 writes the dense factor D), the lane's rows of G and F solved, stored
// and quad-gathered, the lane's rows of the parked G G^T, of the new coupling -F G^T and of the right
// neighbour's R_o - F F^T.  Eliminations of a level touch disjoint slots and a quad reads before it writes:
// one barrier per level instead of two per 64 eliminations, no role redundancy (and at 8 x 8 no spills).
template <typename T, int D>
__device__ __forceinline__ int tile_cr_factor_quad(LdsTile<T, D>& t, int n0, bool keep_last, int64_t row0,
                                                   const DecompLevelsL& lv, T* __restrict__ Dp, T* __restrict__ Fp,
                                                   T* __restrict__ Gp, int lvl_first, int* info, bool& fail) {
  using QT = QuadTile<T, D>;
  constexpr int DD = D * D, RP = QT::RP, NQ = DECL_NT / 4;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                        // (see tile_cr: keeps this addressing out of the caller's code above)
  const int q = tid & 3, Q = tid >> 2;
  const int K = n0 - 1;
  const int kl = keep_last ? 1 : 0;
  int levels = 0;
#pragma unroll 1
  for (int s = 1, j = 0; j < lv.nlev && (keep_last ? (s - 1) < K : (n0 >> j) >= 1); s <<= 1, ++j, ++levels) {
    const int M = n0 >> j, h = s >> 1;
    const int n_elim = (M + 1) / 2;
    const int64_t g0 = row0 >> (j + 1);
#pragma unroll 1
    for (int k0 = 0; k0 < n_elim; k0 += NQ) {
      const int k = k0 + Q;
      const int e = (2 * k + 1) * s - 1;
      const bool act = (2 * k < M) && (e != K || !keep_last);
      if (act) {
        const bool has_o = (2 * k + 1 < M) || (keep_last && e < K);
        const int o = (2 * k + 1 < M) ? e + s : K;
        const int64_t ge = g0 + k;
        T A[RP][D];
        QT::load_rows(t.R, e, q, A);
        if ((s > 1) && (e + h < K + 1 - kl)) {
          T P[RP][D];
          QT::load_rows(t.R, e + h, q, P);
#pragma unroll
          for (int a = 0; a < RP; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) A[a][b] -= P[a][b];
        }
        Chol<T, D> c;
        {
          T Af[D][D];
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int jj = 0; jj <= i; ++jj) Af[i][jj] = quad_from<T>(A[i % RP][jj], i / RP);
          bool f = false;
          chol_lower<T, D>(Af, c, f);
          if (q == 0) {
            if (f) {
              fail = true;
              report_fail(info, ((row0 + e + 1) << lvl_first) - 1);
            }
            T L[D][D];
            chol_to_dense<T, D>(c, L);
            store_block<T, D>(Dp + (lv.offD[j] + ge) * DD, L);
          }
        }
        T G[RP][D], W[RP][D], Cn[RP][D], F[RP][D];
        QT::load_colpairs(t.Oc, e - s + 1, q, G);
#pragma unroll
        for (int a = 0; a < RP; ++a) fwd_subst<T, D>(c, G[a]);
        if (ge >= 1) QT::store_rows_global(Gp + (lv.offG[j] + ge - 1) * DD, q, G);
        if (has_o) {
          QT::load_rows(t.Oc, e + 1, q, F);
#pragma unroll
          for (int a = 0; a < RP; ++a) fwd_subst<T, D>(c, F[a]);
          QT::store_rows_global(Fp + (lv.offF[j] + ge) * DD, q, F);
        } else {
#pragma unroll
          for (int a = 0; a < RP; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) F[a][b] = T(0);
        }
        {
          T Gf[D][D];
          QT::gather(G, Gf);
#pragma unroll
          for (int a = 0; a < RP; ++a)
#pragma unroll
            for (int jj = 0; jj < D; ++jj) {
              T sw = T(0), sc = T(0);
#pragma unroll
              for (int m = 0; m < D; ++m) {
                sw = fmaT(G[a][m], Gf[jj][m], sw);
                sc = fmaT(-F[a][m], Gf[jj][m], sc);
              }
              W[a][jj] = sw;
              Cn[a][jj] = sc;
            }
        }
        T Ro[RP][D];
        if (has_o) {
          QT::load_rows(t.R, o, q, Ro);
          if ((s > 1) && (o + h < K + 1 - kl)) {
            T P[RP][D];
            QT::load_rows(t.R, o + h, q, P);
#pragma unroll
            for (int a = 0; a < RP; ++a)
#pragma unroll
              for (int b = 0; b < D; ++b) Ro[a][b] -= P[a][b];
          }
          T Ff[D][D];
          QT::gather(F, Ff);
#pragma unroll
          for (int a = 0; a < RP; ++a)
#pragma unroll
            for (int jj = 0; jj < D; ++jj) {
              T sr = Ro[a][jj];
#pragma unroll
              for (int m = 0; m < D; ++m) sr = fmaT(-F[a][m], Ff[jj][m], sr);
              Ro[a][jj] = sr;
            }
        }
        // everything this elimination reads has been read (the quad runs in lockstep): write
        QT::store_rows(t.R, e, q, W);
        if (has_o) {
          QT::store_rows(t.Oc, e - s + 1, q, Cn);
          QT::store_rows(t.R, o, q, Ro);
        }
      }
    }
    __syncthreads();                                  // the level's results are visible
  }
  return levels;
}

// One pass of the factorisation.  FROM_RECORDS = false: rows are the caller's Rs / Os (level 0).
// FROM_RECORDS = true: rows are the previous pass's records (RecordLayout without the vector
// parts): R = Rs[w] + dRa[w+1], coupling to the previous row Cs[w].
template <typename T, int D, bool FROM_RECORDS, int LP = DECL_LP>
__global__ __launch_bounds__(DECL_NT) void decomp_lds_kernel(const T* __restrict__ Rin, const T* __restrict__ Oin,
                                                             int64_t n, int64_t n_rec, int spt_in, DecompLevelsL lv,
                                                             int lvl_first,
                                                             T* __restrict__ Dp, T* __restrict__ Fp,
                                                             T* __restrict__ Gp, T* __restrict__ rec_out,
                                                             int* __restrict__ info) {
  constexpr int DD = D * D, TS = 1 << LP;
  static_assert(TS <= DECL_NT, "one thread loads one row of the tile");
  using RL = RecordLayout<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LdsTile<T, D> t;
  t.R = reinterpret_cast<T*>(smem);
  t.Oc = t.R + (size_t)TS * DD;
  t.y = nullptr;
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * TS;
  const int n0 = (int)((n - row0) < TS ? (n - row0) : TS);
  // A full tile running exactly LP levels ends with its last row still alive (odd at every one of
  // those levels): that row is handed on as a record.  A ragged tile (only the last one can be)
  // and the single tile of the top pass are reduced to nothing, by the reference's size rule.
  const bool keep_last = (n0 == TS) && (lv.nlev == LP);
  // rows of the tile -> LDS
  if (tid < n0) {
    const int64_t w = row0 + tid;
    T R[D][D], C[D][D];
    if constexpr (!FROM_RECORDS) {
      load_block<T, D>(Rin + w * DD, R);
      if (w >= 1) load_block<T, D>(Oin + (w - 1) * DD, C);
      else set_zero<T, D>(C);
    } else {
      const T* r = Rin + (size_t)w * RL::STRIDE;
      load_block<T, D>(r + RL::RS, R);
      load_block<T, D>(r + RL::CS, C);
      // the update the next tile of the previous pass owes this row: only the last survivor of a
      // tile is owed one, and only a tile's first record carries it (n_rec >= n: a ragged last
      // tile may have no row to hand on but still owes one)
      if (w + 1 < n_rec && (w + 1) % spt_in == 0) {
        T nR[D][D];
        load_block<T, D>(Rin + (size_t)(w + 1) * RL::STRIDE + RL::DRA, nR);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int jj = 0; jj < D; ++jj) R[i][jj] += nR[i][jj];
      }
    }
    LdsTile<T, D>::store_blk(t.R, tid, R);
    LdsTile<T, D>::store_blk(t.Oc, tid, C);        // Oc[i] = J[row i, row i-1]; Oc[0]: row left of the tile
  }
  __syncthreads();
  bool fail = false;
  int levels;
  if constexpr ((D == 4 || D == 8) && CGPS_DECOMP_QUAD)
    levels = tile_cr_factor_quad<T, D>(t, n0, keep_last, row0, lv, Dp, Fp, Gp, lvl_first, info, fail);
  else
    levels = tile_cr_factor<T, D>(t, n0, keep_last, row0, lv, Dp, Fp, Gp, lvl_first, info, fail);
  if (tid == 0 && rec_out != nullptr) {
    // record: boundary row and its coupling to the previous tile's row (full tiles only), and what
    // the previous tile's row is owed (every tile)
    T Rs_[D][D], Cs_[D][D], dRa[D][D];
    set_zero<T, D>(dRa);
    for (int l = 0; l < levels; ++l) {
      T P[D][D];
      LdsTile<T, D>::load_blk(t.R, (1 << l) - 1, P);
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int jj = 0; jj <= i; ++jj) dRa[i][jj] -= P[i][jj];
    }
    mirror_lower<T, D>(dRa);
    set_zero<T, D>(Rs_);
    set_zero<T, D>(Cs_);
    if (keep_last) {
      LdsTile<T, D>::load_blk(t.R, n0 - 1, Rs_);
      LdsTile<T, D>::load_blk(t.Oc, 0, Cs_);
    }
    T* r = rec_out + (size_t)blockIdx.x * RL::STRIDE;
    store_block<T, D>(r + RL::RS, Rs_);
    store_block<T, D>(r + RL::CS, Cs_);
    store_block<T, D>(r + RL::DRA, dRa);
  }
}

}  // namespace cgps
