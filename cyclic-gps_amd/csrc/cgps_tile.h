// Fused solve + log-det (mahal_and_det, reference cyclic_reduction.py:380-438):
// the whole reduction in ONE launch up to 2^22 rows (record launches only beyond 1024 stage-1
// workgroups), every input byte read from HBM exactly once.
//
// x^T J^-1 x and log|J| do not depend on the elimination order, so this path
// orders the block Gaussian elimination for the hardware (the factor-emitting
// decompose keeps the reference's even/odd order, its factors being API output):
//
//  stage 1, chunk_reduce_kernel: the time axis is cut into chunks of C block
//    rows, one chunk per lane.  A lane streams its rows from HBM (16-byte loads;
//    the right-hand-side line it shares with its next rows goes through LDS) and
//    eliminates rows c0 .. c0+C-2 left to right in registers.  The Schur
//    complement of the chunk interior lands on its two boundary rows: the
//    chunk's own last row (kept: R_s, y_s and C_s = its new coupling to the
//    previous chunk's last row) and that previous row (additive update dRa,
//    dya).  No lane idles and nothing but the inputs crosses HBM.
//  stage 2, tile_cr: the kept rows of a workgroup's lanes form a block
//    tridiagonal system in LDS that is reduced by even/odd cyclic reduction to ONE
//    boundary row + the update for the previous workgroup's boundary row: a
//    "record".  The levels are latency-bound (a lone wave issues one fp64 op per
//    ~8 cycles), so one elimination's ~480 fp64 instructions are split over FOUR
//    waves by role (left-neighbour products / right-neighbour update / two halves
//    of the new coupling block): the operands are in LDS anyway, no shuffles.
//    Levels with few eliminations (4 x 4 fp64 blocks) run on the matrix cores
//    instead, sixteen lanes per elimination: cgps_tile_mfma.h.
//  record stages: records are rows of a (N / (C NT))-row system.  Inside the stage-1 launch
//    (fold_record_stages): the workgroup of a group of 16 tiles that arrives last reduces the
//    group's records with the same tile_cr, the group leader that arrives last reduces the
//    group records, eliminates the final row, sums the partial log-det / mahal in a fixed order
//    and writes the info word (or leaves a shard's single record).  Beyond 1024 stage-1
//    workgroups: record_reduce_kernel launches (recursively for very large N).
//
// Ragged sizes are exact, no padding rows: a short chunk / tile simply keeps its
// last REAL row, which is what lets a shard of a larger system (one per GPU) be
// reduced by the same code to a single record the next shard couples to.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>

#include "cgps_level.h"
#include "cgps_tile_sizes.h"
#include "cgps_leg.h"

namespace cgps {
#include "cgps_tile_leg.h"

// ---- LDS tile of NTILE block rows ------------------------------------------------------
//   R[NTILE][DD], y[NTILE][D] : the rows; a slot whose row has been eliminated is reused for
//                               the update that elimination owes its LEFT neighbour
//   Oc[NTILE+1][DD]           : Oc[i+1] = J[next active row, row i]; Oc[0] = J[first active
//                               row, row left of the tile]
// 16-byte granules of a block are XOR-swizzled by a fold of the row index so that the
// strided row access of every reduction level spreads over the LDS banks.
template <typename T, int D>
struct LdsTile {
  static constexpr int DD = D * D;
  static constexpr int VN = Vec16<T>::N;
  static constexpr int G = (DD % VN == 0) ? DD / VN : 0;          // granules per block
  static constexpr bool SWZ = G >= 2 && (G & (G - 1)) == 0;
  T* R;
  T* Oc;
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
  // rows [r0, r1) of a block, element by element (two waves may write disjoint rows of one block)
  template <int R0, int R1>
  static __device__ __forceinline__ void store_rows(T* base, int row, const T (&A)[D][D]) {
    T* p = base + (size_t)row * DD;
    const int kk = SWZ ? key(row) : 0;
#pragma unroll
    for (int i = R0; i < R1; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int idx = i * D + j;
        if constexpr (SWZ) p[((idx / VN) ^ kk) * VN + (idx % VN)] = A[i][j];
        else p[idx] = A[i][j];
      }
  }
  __device__ __forceinline__ void carve(char* smem, int nt) {
    R = reinterpret_cast<T*>(smem);
    Oc = R + (size_t)nt * DD;
    y = Oc + (size_t)(nt + 1) * DD;
  }
};

// bytes of the LDS tile + block-reduction scratch + wave-exchange scratch + fail word
template <typename T, int D>
constexpr size_t stage_lds_bytes(int ntile, int nthr) {
  size_t tile = ((size_t)ntile * (2 * D * D + D) + D * D) * sizeof(T);
  tile = (tile + 15) & ~(size_t)15;
  return tile + 2 * (nthr / 64) * sizeof(double) + (size_t)(nthr / 64) * (D * D + D) * sizeof(T) + 64;
}

// Running product of pivots with a rare fold into a log sum: one log() per lane
// instead of one per eliminated row.
struct PivotLog {
  double prod = 1.0, logsum = 0.0;
  __device__ __forceinline__ void mul(double p) {
    prod *= p;
    if (!(prod > 1e-150 && prod < 1e150)) { logsum += log(prod); prod = 1.0; }
  }
  // (most lanes of the narrow record stages never saw a pivot: no log for them)
  __device__ __forceinline__ double value() const { return prod == 1.0 ? logsum : logsum + log(prod); }
};

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
// lower(S) = A A^T (fresh), upper part zero
template <typename T, int D>
__device__ __forceinline__ void syrk_lower(T (&S)[D][D], const T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T s = T(0);
      if (j <= i) {
#pragma unroll
        for (int m = 0; m < D; ++m) s = fmaT(A[i][m], A[j][m], s);
      }
      S[i][j] = s;
    }
}

// One elimination of the streaming stage: current row (Rc, yc) with coupling Cc to the left
// boundary row and coupling On to the next row (Rn, yn).  Updates the left boundary
// accumulators (dRa -= G G^T, dya -= G x), turns (Rn, yn) into the next current row and Cc
// into its coupling to the left boundary.
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

// Even/odd cyclic reduction of the n_real-row system in the LDS tile, in place; rows of level
// l live at slots (m+1) 2^l - 1 (cf. decompose_step, cyclic_reduction.py:225-254).
// Elimination k of a level removes the even row e = (2k+1) s - 1, s = 2^l, between its left
// neighbour l = e - s (or the row left of the tile for k = 0) and right neighbour o:
//   D = chol(R_e), x = D^-1 y_e, G = Oc[l]^T D^-T, F = Oc[e] D^-T
//   right neighbour:  R_o -= F F^T, y_o -= F x            (applied now)
//   left neighbour :  owes G G^T, G x                      (parked in the dead slot e and taken
//                      off row l by whoever touches that row at the next level, where its
//                      parking slot is exactly l + (2s)/2 = e)
//   new coupling   :  Oc[l] = -F G^T  (J[o, l])
// Ragged tiles are exact: the LAST real row K = n_real - 1 is never eliminated (it is the
// boundary the next tile / shard couples to) -- the reference's odd/even size rule
// (cyclic_reduction.py:240-248, 262-280) with the last row always kept: at a level with
// M = (K+1)/s regular rows, regular row 2k is eliminated unless it is K, and its right
// neighbour is regular row 2k+1, or K when 2k is the last regular row.
//
// Work split: wave w plays role w % 4 for eliminations 64 * (w / 4 + (NTHR/256) * iter) + lane:
//   role 0  x, G, parks G G^T and G x, accumulates log-det and mahal
//   role 1  x, F, updates the right neighbour
//   role 2  G, F rows [0, D/2): rows [0, D/2) of the new coupling
//   role 3  G, F rows [D/2, D): the other rows
// (every role factors R_e itself: 1/8 of the instructions, no exchange).  All roles read, then
// a barrier, then they write, then a barrier.
// On return slot K holds the tile's boundary row, Oc[0] its coupling to the row left of the
// tile and the slots 2^l - 1 of the executed levels what is owed to that row; returns the
// number of executed levels.
// dev-only: clock stamps per (level pass, wave, phase); no stamp executes in the library build
#ifdef CGPS_TILE_STAMPS
// stamps live in LDS (a global store here would sit in the vm queue the next barrier drains);
// slot layout [pass][wave][4 phases][2 clocks: shader cycles, 100 MHz wall]
#define CGPS_STAMP(slot)                                                                         \
  do {                                                                                           \
    if (stamps && lane == 0) {                                                                   \
      unsigned* sp_ = reinterpret_cast<unsigned*>(stamps) + (((size_t)stamp_pass * (NTHR / 64) + wave) * 4 + (slot)) * 2; \
      sp_[0] = (unsigned)clock64();                                                              \
      sp_[1] = (unsigned)wall_clock64();                                                         \
    }                                                                                            \
  } while (0)
#else
#define CGPS_STAMP(slot) do { } while (0)
#endif

#include "cgps_tile_mfma.h"
#include "cgps_tile_quad.h"
// A level goes to the 16-lanes-per-elimination form when one pass of the workgroup's waves covers
// it (16 eliminations for 256 threads, 32 for 512): measured 0.9-1.2 us against 1.9-2.6 us for a
// role-split pass, while two such passes are no faster than one role-split pass.

// MW: a level runs on the matrix cores when it has at most MW * NTHR / 16 eliminations (MW = 1, 2
// or 4 of them carried side by side per 16-lane group), role-split otherwise.
#ifndef CGPS_MFMA_WIDE
#define CGPS_MFMA_WIDE 1
#endif
#ifndef CGPS_TILE_QUAD
#define CGPS_TILE_QUAD 1      // 8 x 8 blocks: four lanes per elimination in the in-LDS levels (0: role-split, for A/B builds)
#endif
#ifndef CGPS_QUAD4_MIN_ELIM
#define CGPS_QUAD4_MIN_ELIM(nthr) ((nthr) / 16)   // fp64 4 x 4: levels with more eliminations than this go to the quads, the rest to the matrix cores
#endif
#ifndef CGPS_TILE_QUAD4
#define CGPS_TILE_QUAD4 1     // 4 x 4 blocks: the same for the wide levels (fp64) / all levels (fp32)
#endif
template <typename T, int D, int NTHR, int MW = CGPS_MFMA_WIDE>
__device__ __forceinline__ int tile_cr(LdsTile<T, D>& t, int n_real, PivotLog& pl, double& mah, bool& fail,
                                       long long* stamps = nullptr) {
  int stamp_pass = 0;
  (void)stamp_pass;
  static_assert(NTHR % 256 == 0, "tile_cr wants a multiple of four waves");
  using LT = LdsTile<T, D>;
  constexpr int NGRP = NTHR / 256;              // 64-elimination groups handled per pass
  constexpr int DH = (D + 1) / 2;               // rows of the new coupling done by role 2
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int role = wave & 3, grp = wave >> 2;
  const int K = n_real - 1;
  int levels = 0;
  int opaque_tid = threadIdx.x;                 // (cgps_tile_quad.h: keeps the quad levels' addressing out of the caller's loops)
  if constexpr (D == 8 || D == 4) asm volatile("" : "+v"(opaque_tid));
#pragma unroll 1
  for (int s = 1; (s - 1) < K; s <<= 1, ++levels) {
    const int M = (K + 1) / s, h = s >> 1;
    const int n_elim = (M + 1) / 2;             // upper bound on this level's eliminations
    if constexpr (D == 8 && CGPS_TILE_QUAD) {
      // four lanes per elimination (cgps_tile_quad.h), one barrier per level
      tile_cr_level_quad<T, D, NTHR>(t, opaque_tid, K, M, s, pl, mah, fail);
      continue;
    }
    if constexpr (D == 4 && CGPS_TILE_QUAD4) {
      // 4 x 4 blocks: the same for the levels too wide for the matrix cores (fp64), for every level (fp32)
      if (!std::is_same<T, double>::value || n_elim > CGPS_QUAD4_MIN_ELIM(NTHR)) {
        tile_cr_level_quad<T, D, NTHR>(t, opaque_tid, K, M, s, pl, mah, fail);
        continue;
      }
    }
    if constexpr (std::is_same<T, double>::value && D == 4) {
      // sixteen lanes per elimination on the matrix cores (cgps_tile_mfma.h); a level with more
      // eliminations than the workgroup has 16-lane groups carries 2 or 4 of them per group, side
      // by side (independent dependency chains that fill each other's latency slots)
      constexpr int PER = NTHR / 16;
      if (n_elim <= PER) { tile_cr_level_mfma<NTHR, 1>(t, K, M, s, pl, mah, fail); continue; }
      if constexpr (MW >= 2) {
        if (n_elim <= 2 * PER) { tile_cr_level_mfma<NTHR, 2>(t, K, M, s, pl, mah, fail); continue; }
      }
      if constexpr (MW >= 4) {
        if (n_elim <= 4 * PER) { tile_cr_level_mfma<NTHR, 4>(t, K, M, s, pl, mah, fail); continue; }
      }
    }
#pragma unroll 1
    for (int k0 = 0; k0 < n_elim; k0 += 64 * NGRP) {
      const int k = k0 + 64 * grp + lane;
      const int e = (2 * k + 1) * s - 1;
      const bool act = (2 * k < M) && (e != K);
      const int o = (2 * k + 1 < M) ? e + s : K;
      // results a role carries across the barrier
      T W[D][D], wv[D];                          // role 0: parked update; role 1: new R_o, y_o; roles 2/3: new coupling
      CGPS_STAMP(0);
      if (act) {
        T A[D][D];
        LT::load_blk(t.R, e, A);
        const bool pend_e = (s > 1) && (e + h < K);
        if (pend_e) {
          T P[D][D];
          LT::load_blk(t.R, e + h, P);
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) A[i][j] -= P[i][j];
        }
        Chol<T, D> c;
        bool f = false;
        const double piv = chol_lower<T, D>(A, c, f);
        T x[D];
        if (role < 2) {
          load_vec<T, D>(t.y + e * D, x);
          if (pend_e) {
            T p[D];
            load_vec<T, D>(t.y + (e + h) * D, p);
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] -= p[i];
          }
          fwd_subst<T, D>(c, x);
        }
        if (role == 0) {
          pl.mul(piv);
          fail = fail || f;
#pragma unroll
          for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
          T Ol[D][D], G[D][D];
          LT::load_blk(t.Oc, e - s + 1, Ol);
          rsolve_lt_transposed<T, D>(c, Ol, G);
          syrk_lower<T, D>(W, G);
          mirror_lower<T, D>(W);                   // parked blocks are stored symmetric (cgps_tile_mfma.h reads all of it)
          set_zero<T, D>(wv);
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int m = 0; m < D; ++m) wv[i] = fmaT(G[i][m], x[m], wv[i]);
        } else if (role == 1) {
          T F[D][D];
          LT::load_blk(t.Oc, e + 1, F);
          rsolve_lt<T, D>(c, F);
          LT::load_blk(t.R, o, W);
          load_vec<T, D>(t.y + o * D, wv);
          if ((s > 1) && (o + h < K)) {
            T P[D][D], p[D];
            LT::load_blk(t.R, o + h, P);
            load_vec<T, D>(t.y + (o + h) * D, p);
#pragma unroll
            for (int i = 0; i < D; ++i) {
              wv[i] -= p[i];
#pragma unroll
              for (int j = 0; j <= i; ++j) W[i][j] -= P[i][j];
            }
          }
          syrk_sub_lower<T, D>(W, F);
          gemv_sub<T, D>(wv, F, x);
          mirror_lower<T, D>(W);
        } else {
          T Ol[D][D], G[D][D], F[D][D];
          LT::load_blk(t.Oc, e - s + 1, Ol);
          rsolve_lt_transposed<T, D>(c, Ol, G);
          LT::load_blk(t.Oc, e + 1, F);
          const int i0 = (role == 2) ? 0 : DH, i1 = (role == 2) ? DH : D;
#pragma unroll
          for (int i = 0; i < D; ++i) {
            if (i >= i0 && i < i1) {
              fwd_subst<T, D>(c, F[i]);           // row i of F = Oc[e] D^-T
#pragma unroll
              for (int j = 0; j < D; ++j) {
                T sacc = T(0);
#pragma unroll
                for (int m = 0; m < D; ++m) sacc = fmaT(-F[i][m], G[j][m], sacc);
                W[i][j] = sacc;
              }
            }
          }
        }
      }
      CGPS_STAMP(1);
      __syncthreads();                            // every role has read its operands
      CGPS_STAMP(2);
      if (act) {
        if (role == 0) {
          LT::store_blk(t.R, e, W);
          store_vec<T, D>(t.y + e * D, wv);
        } else if (role == 1) {
          LT::store_blk(t.R, o, W);
          store_vec<T, D>(t.y + o * D, wv);
        } else if (role == 2) {
          LT::template store_rows<0, DH>(t.Oc, e - s + 1, W);
        } else {
          LT::template store_rows<DH, D>(t.Oc, e - s + 1, W);
        }
      }
      __syncthreads();                            // the level's results are visible
      CGPS_STAMP(3);
      ++stamp_pass;
    }
  }
  return levels;
}

// ---- in-launch hand-off of the stage-1 records to the workgroup that arrives last -------------
// Records and partial results are stored WRITE-THROUGH (agent-scope relaxed atomic stores =
// global_store ... sc1): they reach the memory side without a release fence, so that the last
// stage-1 workgroup of a launch can pick them up inside the same launch (fold_final below) as
// well as a later kernel can.
template <typename T>
__device__ __forceinline__ void store_wt(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The matching loads: agent-scope relaxed atomic loads (global_load ... sc1) are served past this
// CU's L1 from the coherent memory side, so a workgroup that learnt from the value its arrival
// atomic returned that every producer has stored (and drained) may read the handed-off bytes
// with them without an acquire fence -- provided EVERY load of those bytes is such a load
// (MI355X guide, inter-workgroup visibility: hand-offs measured with sc1 loads, first row:
// one lane per storing workgroup adds to ONE counter, the last adder told by the returned value,
// hipMalloc memory, one workgroup per CU, 8-byte stores and loads).  COH = false: plain loads.
template <bool COH, typename T>
__device__ __forceinline__ T load_coh(const T* p) {
  if constexpr (COH) return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
// 16 bytes as a Vec16<T>::type; coherent form: two 8-byte loads
template <bool COH, typename T>
__device__ __forceinline__ typename Vec16<T>::type load16_coh(const T* p) {
  using V = typename Vec16<T>::type;
  if constexpr (COH) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<T*>(p));
    unsigned long long w[2];
    w[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    V v;
    __builtin_memcpy(&v, w, 16);
    return v;
  } else {
    return *reinterpret_cast<const V*>(p);
  }
}

// Arrival counters of the folded final stage.  They live in the library's own device data (zero
// when the code object is loaded, touched by nothing else) rather than in the caller's
// workspace, whose contents are arbitrary before a call and which other entry points use as
// scratch (one copy per translation unit that launches the fused pipeline: cgps_mahal.hip).  A counter is incremented with a wrapping atomic (atomicInc: old >= limit ? 0 : old + 1),
// so the arrival that completes the count also puts it back to 0: no per-call memset, no reset
// store, and a replayed graph launch finds it clean.  The host maps each workspace it has seen to
// one slot (fold_slot_for): two launches may be in flight at the same time only with different
// workspaces (include/cgps.h), hence with different counters.
constexpr int FOLD_SLOTS = 1024;
#ifndef CGPS_FOLD_GROUP
#define CGPS_FOLD_GROUP 16
#endif
constexpr int FOLD_GROUP = CGPS_FOLD_GROUP; // stage-1 records per group (fold_final, below)
constexpr int FOLD_MAX_GROUPS = 64;         // <= 1024 stage-1 workgroups per launch
// [slot][0]: arrivals of the group leaders; [slot][1 + g]: arrivals of group g's workgroups
static __device__ unsigned int g_fold_counter[FOLD_SLOTS][1 + FOLD_MAX_GROUPS];

// records a record-stage lane eliminates sequentially before the LDS reduction (8 x 8 blocks: that
// one-lane code spills; one record per lane and more workgroups instead)
template <typename T, int D> constexpr int record_rcmax() { return (D == 8 || (sizeof(T) == 8 && D == 6)) ? 1 : 4; }

// A record = what a tile leaves behind: its boundary row (Rs, ys), that row's coupling
// to the previous tile's boundary row (Cs = J[this, previous]) and the additive update
// (dRa, dya) for the previous tile's boundary row.
template <typename T, int D>
struct RecordLayout {
  static constexpr int DD = D * D;
  static constexpr int STRIDE = ((3 * DD + 2 * D + 3) / 4) * 4;   // elements, 16/32-byte aligned
  static constexpr int RS = 0, CS = DD, DRA = 2 * DD, YS = 3 * DD, DYA = 3 * DD + D;
};


// Thread 0: add what the executed levels owe the row left of the tile to (dRa, dya) (which
// already hold the streaming stage's share, as negative sums).
template <typename T, int D>
__device__ __forceinline__ void collect_left_updates(LdsTile<T, D>& t, int levels, T (&dRa)[D][D], T (&dya)[D]) {
  for (int l = 0; l < levels; ++l) {
    const int slot = (1 << l) - 1;
    T P[D][D], p[D];
    LdsTile<T, D>::load_blk(t.R, slot, P);
    load_vec<T, D>(t.y + slot * D, p);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      dya[i] -= p[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) dRa[i][j] -= P[i][j];
    }
  }
}

// Fail codes: 1 + a row near the block that was not positive definite.  A failure poisons everything downstream (NaN),
// so the in-LDS levels and the record stages after it fail as well, at rows that have nothing to do with the cause:
// their codes carry FAIL_LATE, the smallest code wins, so a failure of the streaming stage (the lane's own chunk) is
// what `info` reports whenever there is one; the flag is stripped when `info` is written.
constexpr int FAIL_LATE = 0x40000000;
__device__ __forceinline__ int fail_code(bool early, bool any, int64_t row) {
  const int r = (int)(row < (int64_t)(FAIL_LATE - 2) ? row : (int64_t)(FAIL_LATE - 2)) + 1;
  return early ? r : (any ? (r | FAIL_LATE) : 0);
}
template <int NT>
__device__ __forceinline__ void write_partial(double mah, double logp, int failrow_plus1, double* __restrict__ partial,
                                              double* red, int* sfail) {
  if (failrow_plus1) atomicMin(sfail, failrow_plus1);
  block_sum2<NT>(mah, logp, red);      // contains a barrier when NT > 64
  if constexpr (NT <= 64) __syncthreads();
  if (threadIdx.x == 0) {
    double* p = partial;                 // this workgroup's slot
    const int f = *sfail;
    store_wt(p + 0, mah);
    store_wt(p + 1, logp);
    store_wt(p + 2, (f == 0x7fffffff) ? 0.0 : (double)f);
    store_wt(p + 3, 0.0);
  }
}

// dev-only wall-clock stamps of the final reduction (dev_bench.hip defines CGPS_FIN_STAMPS; no stamp
// executes in the library build)
#ifdef CGPS_FIN_STAMPS
static __device__ long long g_fin_stamps[16];
static __device__ long long g_k_stamps[512][12];
static __device__ int g_k_xcc[512];
#define CGPS_KSTAMP(k) do { if (threadIdx.x == 0) { g_k_stamps[blockIdx.x][k] = wall_clock64(); \
  if ((k) == 0) g_k_xcc[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11))); } } while (0)
#define CGPS_FSTAMP(k) do { if (FINAL && threadIdx.x == 0) g_fin_stamps[k] = wall_clock64(); } while (0)
#else
#define CGPS_FSTAMP(k) do { } while (0)
#define CGPS_KSTAMP(k) do { } while (0)
#endif

// The update a lane computed for the row left of its chunk belongs to the previous lane's
// kept row: fetch it from lane+1 (wave-local shuffle; LDS across the wave boundary) and add it.
// The tile's last real lane keeps its row as is (its update arrives with the next tile's record).
template <typename T, int D, int NT>
__device__ __forceinline__ void absorb_right_neighbour_update(T (&Rc)[D][D], T (&yc)[D], const T (&dRa)[D][D],
                                                              const T (&dya)[D], T* xch, int n_real) {
  constexpr int DD = D * D;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if constexpr (NT > 64) {
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
  }
  T nR[D][D], ny[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j2 = 0; j2 <= i; ++j2) nR[i][j2] = __shfl_down(dRa[i][j2], 1, 64);
    ny[i] = __shfl_down(dya[i], 1, 64);
  }
  if constexpr (NT > 64) {
    if (lane == 63 && w < NT / 64 - 1) {
      const T* p = xch + w * (DD + D);
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j2 = 0; j2 <= i; ++j2) nR[i][j2] = p[i * D + j2];
        ny[i] = p[DD + i];
      }
    }
  }
  if (tid < n_real - 1) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j2 = 0; j2 <= i; ++j2) Rc[i][j2] += nR[i][j2];
      yc[i] += ny[i];
    }
  }
}

// Common second half of both kernels: the lanes' kept rows -> LDS tile -> cyclic reduction ->
// this tile's record (thread 0).  Lanes >= n_real carry nothing (zero updates, not stored).
template <typename T, int D, int NT>
__device__ __forceinline__ void reduce_staged_tile_and_emit(LdsTile<T, D>& t, int n_real, T* xch, T* __restrict__ rec_out,
                                                            PivotLog& pl, double& mah, bool& fail);
template <typename T, int D, int NT>
__device__ __forceinline__ void reduce_tile_and_emit(LdsTile<T, D>& t, T (&Rc)[D][D], T (&yc)[D], T (&Cc)[D][D],
                                                     T (&dRa)[D][D], T (&dya)[D], int n_real, T* xch,
                                                     T* __restrict__ rec_out, PivotLog& pl, double& mah, bool& fail) {
  using RL = RecordLayout<T, D>;
  const int tid = threadIdx.x;
  absorb_right_neighbour_update<T, D, NT>(Rc, yc, dRa, dya, xch, n_real);
  if (tid < n_real) {
    mirror_lower<T, D>(Rc);
    LdsTile<T, D>::store_blk(t.R, tid, Rc);
    store_vec<T, D>(t.y + tid * D, yc);
    LdsTile<T, D>::store_blk(t.Oc, tid, Cc);     // Oc[tid] = J[row tid, row tid-1]; Oc[0]: left of the tile
  }
  // thread 0's share for the row left of the tile waits in LDS while the reduction runs, so it
  // does not occupy registers across it
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) xch[i * D + j] = dRa[i][j];
      xch[D * D + i] = dya[i];
    }
  }
  reduce_staged_tile_and_emit<T, D, NT>(t, n_real, xch, rec_out, pl, mah, fail);
}

// Second half: the rows are in the LDS tile (slot i = row i, Oc[i] = its coupling to row i-1, Oc[0]
// to the row left of the tile) and xch holds the tile's own share for that row (lower triangle at
// xch[i*D+j], vector at xch[D*D+i]).  Barrier, cyclic reduction, then thread 0 writes the record.
template <typename T, int D, int NT>
__device__ __forceinline__ void reduce_staged_tile_and_emit(LdsTile<T, D>& t, int n_real, T* xch, T* __restrict__ rec_out,
                                                            PivotLog& pl, double& mah, bool& fail) {
  using RL = RecordLayout<T, D>;
  const int tid = threadIdx.x;
  __syncthreads();
  CGPS_KSTAMP(8);
  const int levels = tile_cr<T, D, NT>(t, n_real, pl, mah, fail);
  CGPS_KSTAMP(9);
  // the record, one block element per lane (every parked block is stored symmetric): D*D lanes of
  // wave 0 instead of thread 0 walking the levels one after the other
  if (rec_out != nullptr && tid < D * D) {
    using LT = LdsTile<T, D>;
    constexpr int VN = LT::VN;
    const int i = tid / D, j = tid % D;
    auto elem = [&](const T* base, int slot) {
      const int idx = i * D + j;
      const T* p = base + (size_t)slot * D * D;
      return LT::SWZ ? p[((idx / VN) ^ LT::key(slot)) * VN + (idx % VN)] : p[idx];
    };
    T dra = (i >= j) ? xch[i * D + j] : xch[j * D + i];
    T dyv = xch[D * D + i];
    for (int l = 0; l < levels; ++l) {
      const int slot = (1 << l) - 1;
      dra -= elem(t.R, slot);
      dyv -= t.y[slot * D + i];
    }
    T* r = rec_out;                      // this tile's record
    store_wt(r + RL::RS + tid, elem(t.R, n_real - 1));
    store_wt(r + RL::CS + tid, elem(t.Oc, 0));
    store_wt(r + RL::DRA + tid, dra);
    if (j == 0) {
      store_wt(r + RL::YS + i, t.y[(n_real - 1) * D + i]);
      store_wt(r + RL::DYA + i, dyv);
    }
  }
}

template <typename T, int D, int NTILE, int NT>
struct StageSmem {
  static constexpr int DD = D * D;
  LdsTile<T, D> t;
  double* red;
  T* xch;
  int* sfail;
  __device__ __forceinline__ StageSmem(char* smem) {
    t.carve(smem, NTILE);
    char* tail = smem + ((((size_t)NTILE * (2 * DD + D) + DD) * sizeof(T) + 15) & ~(size_t)15);
    red = reinterpret_cast<double*>(tail);
    xch = reinterpret_cast<T*>(red + 2 * (NT / 64));        // [NT/64][DD + D] wave-boundary exchange
    sfail = reinterpret_cast<int*>(xch + (NT / 64) * (DD + D));
  }
};


// ---- stage 1 -----------------------------------------------------------------------------
// (two workgroups per CU = two waves per SIMD: the streaming phase needs the second wave to
// cover HBM latency when the grid is larger than the chip, so registers are capped at 256)
// (blocks up to 4x4 fp64 / 5x5 fp32 fit that budget; larger ones get the whole register file)
// rows a lane of stage 1 takes per group of vector loads (see chunk_reduce_kernel): blocks that are not a
// multiple of 16 bytes, where the 2 or 4 operand sets still fit the register budget
template <typename T, int D> constexpr int stage1_row_group() {
  if (sizeof(T) == 8 && (D == 3 || D == 5)) return 2;     // 110 -> 104 us (d = 3), 209 -> 204 us (d = 5) at 2^20 rows
  return 1;                                                 // fp32 d = 3 in groups of 4: no faster (87 us), one wave per SIMD fewer
}
template <typename T, int D> constexpr int stage1_min_waves() {
  return ((sizeof(T) == 8 && D <= 4) || (sizeof(T) == 4 && D <= 5)) ? 2 : 1;
}
// NT lanes stream (one chunk each); NW >= NT threads run the workgroup.  NW = 2 NT is for grids of
// at most one workgroup per CU: the extra four waves stream nothing, they are a second set of
// role waves, so that the 128 eliminations of the tile's first level take one pass instead of two
// (and the 32 of its third go to the matrix cores).
// FOLD: the workgroup whose record arrives last also runs the final reduction of
// all the launch's records and writes out2 / info: no second launch, no cold start of a lone final
// workgroup.  For grids of at most NT workgroups (one final tile) with NW = 2 NT threads, which is
// what the final reduction wants.
struct FoldArgs {
  int slot;                        // this launch's arrival counters: g_fold_counter[slot][..]
  void* group_records;             // [groups] records of the second level (workspace)
  double* out2;                    // whole system: {mahal, logdet} and info ...
  int* info;
  void* shard_record;              // ... or one shard of a larger system (non-null): its single record
  double* shard_partial;           //     and its {sum of squares, sum of log pivots, fail, 0}
  int rows_per_lane;               // chunk_reduce_kernel<.., C = 0, ..>: rows per lane given at run time (a multiple of 4)
  int pair_slot;                   // SRC = 1, gridDim.y = 2: counters and workspace offset (bytes) of the second system
  size_t pair_ws_stride;
};
template <typename T, int D, int NTILE, int NT, bool FINAL, bool INL = false>
__device__ __forceinline__ void record_reduce_body(char* smem, unsigned tile_index, const T* __restrict__ rin, int64_t n,
                                                   int rc, T* __restrict__ rout, double* __restrict__ partial_out,
                                                   const double* __restrict__ partial_in, int64_t n_partial,
                                                   double* __restrict__ out2, int* __restrict__ info,
                                                   int64_t rows_per_record, int64_t N, int64_t rstride, int64_t pstride);
// The record stages inside the launch (FOLD variants of the stage-1 kernels call this last).
// rows_per_tile: block rows one stage-1 workgroup covers; NW: threads of the workgroup; ONE_PER_CU:
// the launch holds at most one workgroup per CU (see COH below).
template <typename T, int D, int NW, bool ONE_PER_CU>
__device__ __forceinline__ void fold_record_stages(char* smem, int* last_flag, T* __restrict__ rec, double* __restrict__ partial,
                                                   const FoldArgs& fold, int64_t rows_per_tile, int64_t N) {
  const int tid = threadIdx.x;
    // Two levels inside the launch.  The workgroups of a GROUP of FOLD_GROUP consecutive tiles
    // arrive on the group's counter; the one that arrives last reduces the group's records to
    // one (four narrow levels, ~7 KB pulled through one CU) and arrives on the launch's counter;
    // the group leader that arrives last there reduces the group records (<= 64), and either
    // eliminates the last row and writes out2 / info (whole system) or leaves the shard's single
    // record and partial result (one shard of a larger system).  Against ONE workgroup taking all
    // 256 records in a second launch: the 115 KB copy through a single CU (~3 us) becomes sixteen
    // parallel 7 KB copies, and the two widest levels (128 and 64 eliminations, 2.3 + 1.3 us)
    // become narrow ones (0.85 us).
    // Hand-off (MI355X guide, inter-workgroup communication): every store of a record or
    // partial result is a write-through store issued by wave 0; wave 0 drains them, ONE lane
    // arrives with one returning agent-scope atomic.  With one workgroup per CU (NW = 2 NT) and a
    // vectorised copy the last arriver reads the handed-off bytes with coherent (sc1) loads ONLY
    // (COH: the form the guide lists as measured for exactly this shape); otherwise it takes an
    // agent-scope acquire and uses plain loads.  The workgroup barrier holds the other waves until
    // the arrival has returned (and the invalidate has completed).
    constexpr bool COH = ONE_PER_CU && (D * D) % Vec16<T>::N == 0;
    using RL = RecordLayout<T, D>;
    const unsigned grp = blockIdx.x / FOLD_GROUP, ngrp = (gridDim.x + FOLD_GROUP - 1) / FOLD_GROUP;
    const unsigned gsize = (grp + 1 < ngrp) ? (unsigned)FOLD_GROUP : gridDim.x - grp * FOLD_GROUP;
    auto arrive = [&](unsigned int* ctr, unsigned expected) {
      if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
          const bool last = atomicInc(ctr, expected - 1u) == expected - 1u;
          *last_flag = last ? 1 : 0;
          if (!COH && last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
      }
      __syncthreads();
      return *last_flag != 0;                    // workgroup-uniform
    };
    T* grec = reinterpret_cast<T*>(fold.group_records);
    double* gpartial = partial + PARTIAL_STRIDE * (size_t)gridDim.x;       // the groups' partial results
    const bool lead = arrive(&g_fold_counter[fold.slot][1 + grp], gsize);
    CGPS_KSTAMP(4);
    if (gridDim.x == 1) {
      // a single tile: its one record is what the top stage would be handed
      // (whole systems only: the host keeps a one-tile SHARD on the two-launch path)
      record_reduce_body<T, D, FOLD_MAX_GROUPS, NW, true, COH>(smem, 0u, rec, (int64_t)1, 1, (T*)nullptr, (double*)nullptr,
                                                               partial, (int64_t)1, fold.out2, fold.info, rows_per_tile, N,
                                                               (int64_t)RL::STRIDE, (int64_t)PARTIAL_STRIDE);
      return;
    }
    if (lead) {
      record_reduce_body<T, D, FOLD_GROUP, NW, false, COH>(smem, grp, rec, (int64_t)gridDim.x, 1, grec, gpartial,
                                                           (const double*)nullptr, (int64_t)0, (double*)nullptr,
                                                           (int*)nullptr, rows_per_tile, N, (int64_t)RL::STRIDE,
                                                           (int64_t)PARTIAL_STRIDE);
      CGPS_KSTAMP(5);
      const bool fin = arrive(&g_fold_counter[fold.slot][0], ngrp);
      CGPS_KSTAMP(6);
      if (fin) {
        if (fold.shard_record != nullptr)
          record_reduce_body<T, D, FOLD_MAX_GROUPS, NW, false, COH>(smem, 0u, grec, (int64_t)ngrp, 1,
                                                                    reinterpret_cast<T*>(fold.shard_record), fold.shard_partial,
                                                                    partial, (int64_t)gridDim.x + ngrp, (double*)nullptr,
                                                                    (int*)nullptr, rows_per_tile * FOLD_GROUP, N,
                                                                    (int64_t)RL::STRIDE, (int64_t)PARTIAL_STRIDE);
        else
          record_reduce_body<T, D, FOLD_MAX_GROUPS, NW, true, COH>(smem, 0u, grec, (int64_t)ngrp, 1, (T*)nullptr,
                                                                   (double*)nullptr, partial, (int64_t)gridDim.x + ngrp,
                                                                   fold.out2, fold.info, rows_per_tile * FOLD_GROUP, N,
                                                                   (int64_t)RL::STRIDE, (int64_t)PARTIAL_STRIDE);
        CGPS_KSTAMP(7);
      }
    }
}

// C = 0: the rows per lane come with the launch (fold.rows_per_lane, a multiple of 4).  A system of more rows than
// one round of the chip at the compiled C is still ONE round then -- every lane simply walks a longer chunk -- so
// the serial tail (in-LDS levels, record stages) is paid once per launch instead of once per round, and nothing
// but the last round's tail of a multi-round grid ever had the memory system to itself anyway.
// SRC = 1: the rows are those of a LEG model, assembled in registers (cgps_tile_leg.h): Rg = time stamps [N], Og = the
// generator G [d][d], Oleft = the block A added to every diagonal block (or nullptr), yg = right-hand side (or nullptr).
template <typename T, int D, int CT, int NT, int NW = NT, bool FOLD = false, int SRC = 0>
__global__ __launch_bounds__(NW, (NW > NT ? 1 : stage1_min_waves<T, D>())) void chunk_reduce_kernel(const T* __restrict__ Rg, const T* __restrict__ Og,
                                                          const T* __restrict__ yg, int64_t N,
                                                          const T* __restrict__ Oleft,
                                                          T* __restrict__ rec, double* __restrict__ partial,
                                                          FoldArgs fold) {
  // Oleft: J[row 0 of this shard, last row of the previous shard], or nullptr when row 0 is the
  // first row of the whole system.
  constexpr int DD = D * D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  StageSmem<T, D, NT, NW> sm(smem);
  const int tid = threadIdx.x;
  CGPS_KSTAMP(0);
  if (tid == 0) *sm.sfail = 0x7fffffff;
  const int C = CT > 0 ? CT : fold.rows_per_lane;
  if constexpr (SRC == 1) {
    // a pair launch (gridDim.y = 2: the two reductions of a LEG log-likelihood, models.py:349-367, side by side):
    // blockIdx.y = 1 is the prior precision itself -- no diagonal term, no right-hand side -- with its own records,
    // partial results, counters and outputs (out2 + 2, info + 1)
    if (blockIdx.y == 1) {
      Oleft = nullptr;
      yg = nullptr;
      rec = reinterpret_cast<T*>(reinterpret_cast<char*>(rec) + fold.pair_ws_stride);
      partial = reinterpret_cast<double*>(reinterpret_cast<char*>(partial) + fold.pair_ws_stride);
      fold.group_records = reinterpret_cast<char*>(fold.group_records) + fold.pair_ws_stride;
      fold.slot = fold.pair_slot;
      fold.out2 += 2;
      fold.info += 1;
    }
  }
  const int64_t lane0 = (int64_t)blockIdx.x * NT;
  const int64_t gl = lane0 + tid;
  int64_t r0 = gl * C, rE = gl * C + C;                         // this lane's chunk [r0, rE)
  if (tid >= NT) r0 = N;                                        // threads past the lanes hold no rows
  const bool whole_chunk = rE <= N;                             // the lane's chunk is complete
  const bool whole_chunk_and_next = rE <= N - 1;                // ... and the row after it exists
  if (rE > N) rE = N;
  const int L = r0 < N ? (int)(rE - r0) : 0;                    // rows of this lane
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;

  T Rc[D][D], yc[D], Cc[D][D], dRa[D][D], dya[D];
  set_zero<T, D>(dRa);
  set_zero<T, D>(dya);
  // a lane past the end of the shard holds nothing; a short last chunk keeps its last real row
  set_zero<T, D>(Rc);
  set_zero<T, D>(yc);
  set_zero<T, D>(Cc);
  // The right-hand side of YR = 4 consecutive rows of a 4 x 4 fp64 system shares one 128-byte
  // line that a lane consumes over four steps (~13 us): by then L1 and L2 have dropped it and it
  // is fetched again (PMC: 1.15 x the algorithmic bytes leave L2 that way).  Such lanes copy the
  // whole line into their 128 bytes of the (still idle) LDS tile when they reach it.
  constexpr int YR = 4;
  constexpr bool YSTAGE = SRC == 0 && std::is_same<T, double>::value && D == 4 && (CT == 0 || (CT >= YR && CT % YR == 0));
  T* ylds = reinterpret_cast<T*>(smem) + (size_t)tid * (YR * D);
  const bool yfull = YSTAGE && whole_chunk && r0 < N;      // the lane's chunk is complete: whole lines exist
  auto stage_y_line = [&](int64_t row) {                   // rows row .. row+YR-1 -> this lane's LDS line
    if constexpr (YSTAGE) {
      using V = typename Vec16<T>::type;
      const V* src = reinterpret_cast<const V*>(yg + row * D);
      V* dst = reinterpret_cast<V*>(ylds);
#pragma unroll
      for (int g = 0; g < YR * D / Vec16<T>::N; ++g) dst[g] = src[g];
    }
  };
  constexpr int RG = stage1_row_group<T, D>();
  constexpr bool GROUPED = SRC == 0 && RG > 1 && CT % RG == 0 && !YSTAGE;
  const bool grouped = GROUPED && whole_chunk_and_next && r0 < N;   // every row of the chunk, and O[last row], exist
  T cR[D][D], cB[D][D];                          // SRC = 1: what the gap before the next row leaves it (cgps_tile_leg.h)
  if constexpr (SRC == 1) {
    if (r0 < N) {
      if (r0 >= 1) {
        T tl[D][D];
        if (!leg_gap<T, D>(Rg, Og, r0 - 1, cR, tl, cB)) fail = true;
      } else {
        set_zero<T, D>(cR);
        set_zero<T, D>(cB);
      }
      leg_row<T, D>(Rg, Og, Oleft, yg, r0, N, cR, cB, Rc, Cc, yc, fail);
    }
  } else
  if (r0 < N) {
    if (!grouped) {
      load_block<T, D>(Rg + r0 * DD, Rc);
      if (yfull) {
        stage_y_line(r0);
        load_vec<T, D>(ylds, yc);
      } else {
        load_vec<T, D>(yg + r0 * D, yc);
      }
    }
    if (r0 >= 1) load_block<T, D>(Og + (r0 - 1) * DD, Cc);
    else if (Oleft != nullptr) load_block<T, D>(Oleft, Cc);
  }
  // Blocks whose size is not a multiple of 16 bytes (odd d) are read with scalar loads above and below:
  // 55 load instructions per row at fp64 d = 5.  RG = 2 consecutive rows of such an fp64 system DO fill
  // whole 16-byte granules and start on a 16-byte boundary when the first of them has an even index,
  // which a lane's chunk guarantees (C % RG == 0).  A lane whose chunk is complete therefore takes its
  // rows RG at a time with vector loads (half the instructions): R[a .. a+RG), y[a .. a+RG) and
  // O[a .. a+RG), O[a+i] being the coupling of row a+i to row a+i+1 -- the last of the group is carried
  // into the next group.  (A few per cent only: at one wave per SIMD the stage is bound by the waves'
  // own chains, and 70 of the 177 us at d = 5 are the record stages -- dev_bench -DDEVB_D=5 20 2 1.)
  if constexpr (GROUPED) {
    if (grouped) {
      T Oc[D][D];                              // O[a - 1], carried
#pragma unroll 1
      for (int p = 0; p < L / RG; ++p) {
        const int64_t a = r0 + (int64_t)RG * p;
        T Rq[RG][D][D], Oq[RG][D][D], yq[RG][D];
        load_rows<T, D, RG>(Rg + a * DD, Rq);
        load_rows<T, D, RG>(Og + a * DD, Oq);
        load_row_vecs<T, D, RG>(yg + a * D, yq);
#pragma unroll
        for (int i = 0; i < RG; ++i) {
          if (i == 0) {
            if (p > 0) {
              eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, Oc, Rq[0], yq[0], pl, mah, fail);
            } else {                           // the chunk's first row
#pragma unroll
              for (int u = 0; u < D; ++u) {
                yc[u] = yq[0][u];
#pragma unroll
                for (int v = 0; v < D; ++v) Rc[u][v] = Rq[0][u][v];
              }
            }
          } else {
            eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, Oq[i - 1], Rq[i], yq[i], pl, mah, fail);
          }
        }
#pragma unroll
        for (int u = 0; u < D; ++u)
#pragma unroll
          for (int v = 0; v < D; ++v) Oc[u][v] = Oq[RG - 1][u][v];
      }
    }
  }
  if (!grouped) {
#pragma unroll 1
  for (int j = 0; j < L - 1; ++j) {
    const int64_t rn = r0 + j + 1;
    T Rn[D][D], On[D][D], yn[D];
    if constexpr (SRC == 1) {
      leg_row<T, D>(Rg, Og, Oleft, yg, rn, N, cR, cB, Rn, On, yn, fail);
      eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, On, Rn, yn, pl, mah, fail);
      continue;
    }
    load_block<T, D>(Rg + rn * DD, Rn);
    load_block<T, D>(Og + (rn - 1) * DD, On);
    if (yfull) {
      if (((j + 1) & (YR - 1)) == 0) stage_y_line(rn);
      load_vec<T, D>(ylds + ((j + 1) & (YR - 1)) * D, yn);
    } else {
      load_vec<T, D>(yg + rn * D, yn);
    }
    eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, On, Rn, yn, pl, mah, fail);
  }
  }
  // (the tile is written only after the workgroup-wide barrier inside reduce_tile_and_emit: no lane
  // is still reading its y line then)

  CGPS_KSTAMP(1);
  const bool fail_stream = fail;                 // (what fails later may be a consequence of this)
  int64_t nreal64 = (N + C - 1) / C - lane0;     // lanes of this tile that hold real rows
  const int n_real = nreal64 > NT ? NT : (int)nreal64;
  reduce_tile_and_emit<T, D, NW>(sm.t, Rc, yc, Cc, dRa, dya, n_real, sm.xch, rec + (size_t)blockIdx.x * RecordLayout<T, D>::STRIDE,
                                 pl, mah, fail);
  CGPS_KSTAMP(2);
  int64_t frow = r0 < N ? r0 : N - 1;
  write_partial<NW>(mah, pl.value(), fail_code(fail_stream, fail, frow), partial + PARTIAL_STRIDE * (size_t)blockIdx.x, sm.red, sm.sfail);
  CGPS_KSTAMP(3);
  if constexpr (FOLD)
    fold_record_stages<T, D, NW, (NW == 2 * NT)>(smem, sm.sfail + 1, rec, partial, fold, (int64_t)C * NT, N);
}

// ---- stage 3 -----------------------------------------------------------------------------
// Records in -> records out (FINAL = false), or -> out2 = {mahal, logdet} and info (FINAL =
// true, one workgroup).  Row w of this stage: R = Rs[w] + dRa[w+1], y = ys[w] + dya[w+1],
// coupling to row w-1: Cs[w].  Lane t < NTILE first eliminates its rc consecutive rows left to
// right (like stage 1); then all NT lanes (NT >= NTILE: the extra waves only add hands to
// tile_cr) reduce the NTILE kept rows.  rstride / pstride: distance between consecutive
// records (elements of T) / partials (doubles).
template <typename T, int D>
__device__ __forceinline__ void load_record_row(const T* __restrict__ rin, int64_t rstride, int64_t w, int64_t n,
                                                bool add_next, T (&R)[D][D], T (&y)[D], T (&Cs)[D][D]) {
  using RL = RecordLayout<T, D>;
  if (w < n) {
    const T* r = rin + (size_t)w * rstride;
    load_block<T, D>(r + RL::RS, R);
    load_vec<T, D>(r + RL::YS, y);
    load_block<T, D>(r + RL::CS, Cs);
    if (add_next && w + 1 < n) {
      const T* q = rin + (size_t)(w + 1) * rstride;
      T nR[D][D], ny[D];
      load_block<T, D>(q + RL::DRA, nR);
      load_vec<T, D>(q + RL::DYA, ny);
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) R[i][j] += nR[i][j];
        y[i] += ny[i];
      }
    }
  } else {                      // a lane past the last record holds nothing
    set_zero<T, D>(R);
    set_zero<T, D>(y);
    set_zero<T, D>(Cs);
  }
}

template <typename T, int D, int NTILE, int NT, bool FINAL, bool INL>
__device__ __forceinline__ void record_reduce_body(char* smem, unsigned tile_index, const T* __restrict__ rin, int64_t n,
                                                   int rc, T* __restrict__ rout, double* __restrict__ partial_out,
                                                   const double* __restrict__ partial_in, int64_t n_partial,
                                                   double* __restrict__ out2, int* __restrict__ info,
                                                   int64_t rows_per_record, int64_t N, int64_t rstride, int64_t pstride);

template <typename T, int D, int NTILE, int NT, bool FINAL>
__global__ __launch_bounds__(NT) void record_reduce_kernel(const T* __restrict__ rin, int64_t n, int rc,
                                                           T* __restrict__ rout, double* __restrict__ partial_out,
                                                           const double* __restrict__ partial_in, int64_t n_partial,
                                                           double* __restrict__ out2, int* __restrict__ info,
                                                           int64_t rows_per_record, int64_t N, int64_t rstride,
                                                           int64_t pstride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  record_reduce_body<T, D, NTILE, NT, FINAL>(smem, blockIdx.x, rin, n, rc, rout, partial_out, partial_in, n_partial, out2,
                                             info, rows_per_record, N, rstride, pstride);
}

// tile_index: which NTILE * rc records this workgroup takes (blockIdx.x of record_reduce_kernel; 0
// when the last stage-1 workgroup of a launch runs the final reduction itself, fold_final)
template <typename T, int D, int NTILE, int NT, bool FINAL, bool INL>
__device__ __forceinline__ void record_reduce_body(char* smem, unsigned tile_index, const T* __restrict__ rin, int64_t n,
                                                   int rc, T* __restrict__ rout, double* __restrict__ partial_out,
                                                   const double* __restrict__ partial_in, int64_t n_partial,
                                                   double* __restrict__ out2, int* __restrict__ info,
                                                   int64_t rows_per_record, int64_t N, int64_t rstride, int64_t pstride) {
  using RL = RecordLayout<T, D>;
  StageSmem<T, D, NTILE, NT> sm(smem);
  const int tid = threadIdx.x;
  CGPS_FSTAMP(0);
  if (tid == 0) *sm.sfail = 0x7fffffff;
  const int64_t w0 = (int64_t)tile_index * NTILE * rc;    // first record of this tile
  int64_t wend = w0 + (int64_t)NTILE * rc;
  if (wend > n) wend = n;
  const int64_t wlast = wend - 1;                         // its last one (kept; its update is deferred)
  const int64_t wb = (tid < NTILE) ? w0 + (int64_t)tid * rc : n;
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;
  const int64_t nthreads_real = (n - w0 + rc - 1) / rc;
  const int n_real = nthreads_real > NTILE ? NTILE : (int)nthreads_real;
  // the partial results of the earlier launches: requested now, together with the records, and
  // added at the end (another exposed memory latency otherwise)
  double pre_mah = 0.0, pre_logp = 0.0;
  int pre_fail = 0x7fffffff;
  if (FINAL) {
    for (int64_t i = tid; i < n_partial; i += NT) {
      const double* p = partial_in + pstride * i;
      const double p0 = load_coh<INL>(p), p1 = load_coh<INL>(p + 1), p2 = load_coh<INL>(p + 2);
      pre_mah += p0;
      pre_logp += p1;
      if (p2 != 0.0 && (int)p2 < pre_fail) pre_fail = (int)p2;
    }
  }
  constexpr int VN = Vec16<T>::N;
  static_assert(!INL || (D * D) % VN == 0, "the in-launch form reads every handed-off byte with coherent loads: vector path only");
  bool staged = false;
  if constexpr ((D * D) % VN == 0) {
  if (INL || (rc == 1 && rstride % VN == 0)) {
    staged = true;
    // One record per row: no sequential eliminations first, so the rows go straight into the LDS
    // tile, all NT threads copying 16-byte granules (consecutive threads = consecutive granules of
    // a record) instead of NTILE lanes walking one record each with 64 lines per instruction.
    // Row w = Rs[w] + dRa[w+1], y = ys[w] + dya[w+1] (not for the tile's last row), coupling Cs[w].
    // Every load of the copy is requested before the first one is waited for (a loop that loads,
    // waits and stores per iteration exposes one memory round trip per iteration: six of them
    // measured ~5 us of the lone final workgroup's 16).
    using V = typename Vec16<T>::type;
    using LT = LdsTile<T, D>;
    constexpr int G = (D * D) / VN;
    constexpr int ITG = (NTILE * G + NT - 1) / NT, ITY = (NTILE * D + NT - 1) / NT;
    V ga[ITG], gc[ITG], gu[ITG];
    T yv[ITY], yu[ITY];
    const V vzero = V{};
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int gi = tid + it * NT;
      const bool ok = gi < n_real * G;
      const int slot = ok ? gi / G : 0, g = ok ? gi % G : 0;
      const int64_t w = w0 + slot;
      const T* r = rin + (size_t)w * rstride;
      ga[it] = load16_coh<INL>(r + RL::RS + g * VN);
      gc[it] = load16_coh<INL>(r + RL::CS + g * VN);
      gu[it] = (ok && w != wlast) ? load16_coh<INL>(r + rstride + RL::DRA + g * VN) : vzero;
    }
#pragma unroll
    for (int it = 0; it < ITY; ++it) {
      const int vi = tid + it * NT;
      const bool ok = vi < n_real * D;
      const int slot = ok ? vi / D : 0, i = ok ? vi % D : 0;
      const int64_t w = w0 + slot;
      const T* r = rin + (size_t)w * rstride;
      yv[it] = load_coh<INL>(r + RL::YS + i);
      yu[it] = (ok && w != wlast) ? load_coh<INL>(r + rstride + RL::DYA + i) : T(0);
    }
    // the tile's own share for the row left of it: one element per lane of the last wave
    constexpr int OWN_IT = (D * D + D + 63) / 64;
    T own[OWN_IT];
    const int oi = tid - (NT - 64);
#pragma unroll
    for (int it = 0; it < OWN_IT; ++it) {
      const int q = oi + 64 * it;
      own[it] = (oi >= 0 && q < D * D + D)
                    ? load_coh<INL>(rin + (size_t)w0 * rstride + (q < D * D ? RL::DRA + q : RL::DYA + (q - D * D))) : T(0);
    }
    CGPS_FSTAMP(1);
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int gi = tid + it * NT;
      if (gi < n_real * G) {
        const int slot = gi / G, g = gi % G;
        T* ae = reinterpret_cast<T*>(&ga[it]);
        const T* ue = reinterpret_cast<const T*>(&gu[it]);
#pragma unroll
        for (int q = 0; q < VN; ++q) ae[q] += ue[q];
        const int pg = LT::SWZ ? (g ^ LT::key(slot)) : g;
        reinterpret_cast<V*>(sm.t.R + (size_t)slot * D * D)[pg] = ga[it];
        reinterpret_cast<V*>(sm.t.Oc + (size_t)slot * D * D)[pg] = gc[it];
      }
    }
#pragma unroll
    for (int it = 0; it < ITY; ++it) {
      const int vi = tid + it * NT;
      if (vi < n_real * D) sm.t.y[vi] = yv[it] + yu[it];
    }
#pragma unroll
    for (int it = 0; it < OWN_IT; ++it) {
      const int q = oi + 64 * it;
      if (oi >= 0 && q < D * D + D) sm.xch[q] = own[it];  // xch[i*D+j] (lower triangle read), xch[D*D+i]
    }
    CGPS_FSTAMP(2);
    reduce_staged_tile_and_emit<T, D, NT>(sm.t, n_real, sm.xch, FINAL ? (T*)nullptr : rout + (size_t)tile_index * RL::STRIDE, pl,
                                          mah, fail);
  }
  }
  if constexpr (!INL && !(record_rcmax<T, D>() == 1 && (D * D) % VN == 0)) {
  if (!staged) {
  T Rc[D][D], yc[D], Cc[D][D], dRa[D][D], dya[D];
  set_zero<T, D>(dRa);
  set_zero<T, D>(dya);
  load_record_row<T, D>(rin, rstride, wb, n, wb != wlast, Rc, yc, Cc);
  if (tid == 0 && wb < n) {                               // the tile's own share for the row left of it
    const T* r = rin + (size_t)wb * rstride;
    load_block<T, D>(r + RL::DRA, dRa);
    load_vec<T, D>(r + RL::DYA, dya);
  }
#pragma unroll 1
  for (int j = 1; j < rc; ++j) {
    if (wb + j >= n) break;
    T Rn[D][D], On[D][D], yn[D];
    load_record_row<T, D>(rin, rstride, wb + j, n, wb + j != wlast, Rn, yn, On);
    eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, On, Rn, yn, pl, mah, fail);
  }
  reduce_tile_and_emit<T, D, NT>(sm.t, Rc, yc, Cc, dRa, dya, n_real, sm.xch,
                                 FINAL ? (T*)nullptr : rout + (size_t)tile_index * RL::STRIDE, pl, mah, fail);
  }
  }
  CGPS_FSTAMP(3);
  int64_t frow = (wb + rc) * rows_per_record;
  frow = (frow < N ? frow : N) - 1;
  if constexpr (!FINAL) {
    double logp = pl.value();
    int fcode = fail_code(false, fail, frow);
    if (partial_in != nullptr) {           // last launch of a shard: fold in the partial results of the earlier ones
      for (int64_t i = tid; i < n_partial; i += NT) {
        const double* p = partial_in + pstride * i;
        const double p0 = load_coh<INL>(p), p1 = load_coh<INL>(p + 1), p2 = load_coh<INL>(p + 2);
        mah += p0;
        logp += p1;
        if (p2 != 0.0 && (fcode == 0 || (int)p2 < fcode)) fcode = (int)p2;
      }
    }
    write_partial<NT>(mah, logp, fcode, partial_out + PARTIAL_STRIDE * (size_t)tile_index, sm.red, sm.sfail);
  } else {
    if (tid == 0) {                        // the very last row of the whole system
      T A[D][D], x[D];
      Chol<T, D> c;
      LdsTile<T, D>::load_blk(sm.t.R, n_real - 1, A);
      pl.mul(chol_lower<T, D>(A, c, fail));
      load_vec<T, D>(sm.t.y + (n_real - 1) * D, x);
      fwd_subst<T, D>(c, x);
#pragma unroll
      for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
    }
    CGPS_FSTAMP(4);
    if (fail) atomicMin(sm.sfail, fail_code(false, true, frow));
    double logp = pl.value() + pre_logp;
    mah += pre_mah;
    if (pre_fail != 0x7fffffff) atomicMin(sm.sfail, pre_fail);
    block_sum2<NT>(mah, logp, sm.red);
    if (tid == 0) {
      const int f = *sm.sfail;
      const bool ok = (f == 0x7fffffff);
      // a caller that does not read `info` must not get a plausible number out of a system that is
      // not positive definite (two negative pivots multiply to a positive "determinant")
      const double poison = __builtin_nan("");
      out2[0] = ok ? mah : poison;
      out2[1] = ok ? logp : poison;
      *info = ok ? 0 : (f & ~FAIL_LATE);
    }
    CGPS_FSTAMP(5);
  }
}

// {sum, sum, smallest non-zero, 0} over per-block partial results
static __global__ __launch_bounds__(256) void sum_partials4_kernel(const double* __restrict__ partial, int64_t count,
                                                            double* __restrict__ out4) {
  __shared__ double red[2 * 4];
  __shared__ int sf;
  if (threadIdx.x == 0) sf = 0x7fffffff;
  __syncthreads();
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < count; i += 256) {
    const double* p = partial + PARTIAL_STRIDE * i;
    a += p[0];
    b += p[1];
    if (p[2] != 0.0) atomicMin(&sf, (int)p[2]);
  }
  block_sum2<256>(a, b, red);
  if (threadIdx.x == 0) {
    out4[0] = a;
    out4[1] = b;
    out4[2] = (sf == 0x7fffffff) ? 0.0 : (double)sf;
    out4[3] = 0.0;
  }
}

// ---- stage 1 for large blocks: one block row over four lanes --------------------------------------
#include "cgps_tile_ml.h"

// ---- host side ------------------------------------------------------------------------------
// does the 256-row LDS tile of the record stages fit the 160 KB of LDS?  (fp64 d <= 5, fp32 d <= 8)
template <typename T, int D> constexpr bool tile_fits_256() {
  return ((size_t)256 * (2 * D * D + D) + D * D) * sizeof(T) + 4096 <= 160 * 1024;
}
// The fused pipeline is built for every block size 1 <= d <= 8 in both precisions.  (Above fp64
// d=4 / fp32 d=5 the one-lane-per-row code spills registers; it still reads the inputs once
// instead of three times, which is what counts for an HBM-bound path.)  The factor-emitting
// decompose has its own, narrower condition: decomp_tile_supported().
template <typename T, int D> constexpr bool tile_supported() { return D >= 1 && D <= 8; }
template <typename T, int D> constexpr bool decomp_tile_supported() { return tile_fits_256<T, D>(); }
template <typename T, int D> struct TileCfg {
  // BIG blocks (fp64 d = 6, 7, 8): a 256-row tile does not fit the LDS; tiles of 64 / 128 kept rows
  static constexpr bool BIG = !tile_fits_256<T, D>();
  // lanes sharing one block row in stage 1 (cgps_tile_ml.h): 8 x 8 blocks (and 6 x 6 fp64) do not fit
  // one lane's registers
  static constexpr int LPR = (D == 8) ? 4 : ((BIG && D == 6) ? 2 : 1);
  // lanes (= streaming threads) per workgroup in stage 1.  BIG blocks with one lane per row (7 x 7
  // fp64): 128 lanes, and the workgroup always runs with 256 threads (the extra waves are role waves)
  static constexpr int NT1 = (BIG && LPR == 1) ? 128 : 256;
  static constexpr bool ALWAYS_WIDE = NT1 < 256;
  // rows per lane (per lane group) in stage 1; 8 x 8 fp32 blocks: 128 rows per group of four lanes =
  // 8192 rows per workgroup, so that 2^22 rows are 512 workgroups = ONE round of the chip (two per CU)
  static constexpr int C = (D == 8) ? (sizeof(T) == 4 ? 128 : 64) : (LPR == 2 ? 32 : 16);
  static constexpr int NTILE3 = BIG ? 64 : 256;   // kept rows per workgroup in stage 3
  static constexpr int NT3 = 512;      // threads per workgroup in stage 3 (extra waves = extra hands)
  // records a stage-3 lane eliminates sequentially before the LDS reduction (8 x 8 blocks: that
  // one-lane code spills, more workgroups with one record per lane are faster)
  static constexpr int RCMAX = record_rcmax<T, D>();
  static constexpr int NG1 = NT1 / LPR;                 // kept rows (= LDS tile slots) per stage-1 workgroup
  static constexpr int64_t ROWS1 = (int64_t)C * NG1;    // rows per stage-1 workgroup
};
constexpr int TILE_MAX_DEVICES = 64;
template <typename T, int D>
void tile_set_attributes() {
  using Cfg = TileCfg<T, D>;
  static std::once_flag once[TILE_MAX_DEVICES];     // attributes belong to a device, not to the process
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= TILE_MAX_DEVICES) dev = 0;
  std::call_once(once[dev], [] {
  const int lds1 = (int)stage_lds_bytes<T, D>(Cfg::NG1, Cfg::NT1), lds3 = (int)stage_lds_bytes<T, D>(Cfg::NTILE3, Cfg::NT3);
  if constexpr (Cfg::LPR > 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_ml_kernel<T, D, Cfg::C, Cfg::NT1, Cfg::LPR>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_ml_kernel<T, D, Cfg::C, Cfg::NT1, Cfg::LPR, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
  }
  else if constexpr (Cfg::ALWAYS_WIDE) {
    const int ldsw = (int)stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 0, Cfg::NT1, 2 * Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 0, Cfg::NT1, 2 * Cfg::NT1, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 0, Cfg::NT1, Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 0, Cfg::NT1, Cfg::NT1, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    if constexpr (stage1_min_waves<T, D>() == 2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 0, Cfg::NT1, 2 * Cfg::NT1, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1));
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 8, Cfg::NT1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 8, Cfg::NT1, Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 4, Cfg::NT1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 4, Cfg::NT1, Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 1, Cfg::NT1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_reduce_kernel<T, D, 1, Cfg::NT1, Cfg::NT1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&record_reduce_kernel<T, D, Cfg::NTILE3, Cfg::NT3, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&record_reduce_kernel<T, D, Cfg::NTILE3, Cfg::NT3, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
  });
}

// Slot of the arrival counters that belongs to this (device, workspace).  The counters are device data of THIS
// translation unit (static __device__ above), so the map is too (internal linkage, not `inline`).  A slot's
// counters are zero whenever no launch that uses it is in flight (the arrival that completes a count wraps it to
// zero), so a slot can be handed to another workspace: when all FOLD_SLOTS slots are taken, the slot whose last
// launch is the longest ago is reused -- a caller whose workspace address changes from call to call (a C program that
// allocates per call, torch after empty_cache(), graph-private pools) keeps the one-launch path.  What this
// assumes: fewer than FOLD_SLOTS launches with different workspaces in flight at the same time.
// cgps_reset_counters() zeroes the counters of the current device (after a launch that died half-way).
static int fold_slot_for(const void* ws) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> slots;
  static std::pair<int, const void*> owner[FOLD_SLOTS];
  static unsigned long long last_use[FOLD_SLOTS];
  static unsigned long long tick = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::pair<int, const void*> key{dev, ws};
  std::lock_guard<std::mutex> lock(mu);
  auto it = slots.find(key);
  int s;
  if (it != slots.end()) {
    s = it->second;
  } else if ((int)slots.size() < FOLD_SLOTS) {
    s = (int)slots.size();
    slots.emplace(key, s);
    owner[s] = key;
  } else {
    s = 0;
    for (int k = 1; k < FOLD_SLOTS; ++k)
      if (last_use[k] < last_use[s]) s = k;
    slots.erase(owner[s]);
    slots.emplace(key, s);
    owner[s] = key;
  }
  last_use[s] = ++tick;
  return s;
}
// zero every arrival counter of the current device, on `st` (cgps_reset_counters)
static hipError_t fold_reset_counters(hipStream_t st) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_fold_counter));
  if (e != hipSuccess) return e;
  return hipMemsetAsync(p, 0, sizeof(unsigned int) * FOLD_SLOTS * (1 + FOLD_MAX_GROUPS), st);
}
inline bool fold_final_enabled() {          // CGPS_NO_FOLD=1: two launches as before (cross-check / A-B timing)
  static const bool on = [] { const char* e = getenv("CGPS_NO_FOLD"); return !(e && e[0] == '1'); }();
  return on;
}

// Systems of more rows than one round of the chip at the compiled rows-per-lane: rows per lane chosen at launch so
// that the grid is still ONE round (long_chunk_tiles() workgroups at most; CGPS_S1_TILES=<n> for A/B timing,
// CGPS_S1_LONG=0: rounds of the compiled-C kernel as before).  Returns 0 when the compiled C is to be used.
inline int long_chunk_target_tiles() {
  static const int v = [] {
    const char* e = getenv("CGPS_S1_LONG");
    if (e && e[0] == '0') return 0;
    const char* t = getenv("CGPS_S1_TILES");
    const int n = t ? atoi(t) : 0;
    return n > 0 ? n : (int)STAGE1_SMALL_TILES;
  }();
  return v;
}
inline bool long_chunk_narrow() {          // CGPS_S1_NARROW=1: the long chunks on the two-workgroups-per-CU kernel (A/B)
  static const bool on = [] { const char* e = getenv("CGPS_S1_NARROW"); return e && e[0] == '1'; }();
  return on;
}
// block_bytes: bytes of one d x d block.  The lanes of a wave read addresses C * block_bytes apart, all lanes of the
// chip at the same offset inside their chunks; measured (profiles/r03_long_chunk_sweep.txt, d = 4 fp64): strides of
// 4 KB and 8 KB stream at 4.3-5.9 TB/s depending on the box (the physical placement of the operands: L1->L2 read
// latency 1 480 cycles against 1 200 and twice the DRAM credit stalls in the slow case), 2 KB and >= 16 KB at
// 6.1-6.5 TB/s on every box seen.  Long chunks are therefore taken only outside (2 KB, 16 KB); in between the
// compiled C runs in rounds as before (CGPS_S1_C=<c> forces a value for A/B timing).
inline int long_chunk_rows_per_lane(int64_t N, int lanes, int c_full, int block_bytes) {
  const int tiles = long_chunk_target_tiles();
  if (tiles <= 0) return 0;
  static const int forced = [] { const char* e = getenv("CGPS_S1_C"); return e ? atoi(e) : 0; }();   // A/B timing
  if (forced > 0 && forced % 4 == 0 && (N + (int64_t)forced * lanes - 1) / ((int64_t)forced * lanes) <= 1024 &&
      N > (int64_t)forced * lanes)
    return forced;
  const int64_t per_round = (int64_t)tiles * lanes;
  int64_t c = (N + per_round - 1) / per_round;
  c = (c + 3) & ~(int64_t)3;
  if (c <= c_full || c > (1 << 20)) return 0;
  if (c * block_bytes > 2048 && c * block_bytes < 16384) return 0;
  return (int)c;
}

// The fused pipeline.  Whole system: shard_record == nullptr, results in out2 / info.
// One shard of a sharded system: shard_record / shard_partial receive the shard's single record
// and its {mahal, logdet, fail, 0} partial; Oleft = J[first row of the shard, last row of the
// previous shard] (nullptr for the first shard).
// returns 0 on success, -1 when the workspace is too small
template <typename T, int D>
int run_tile_mahal_logdet(const T* Rs, const T* Os, const T* x, int64_t N, char* ws, size_t ws_bytes, double* out2,
                          int* info, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                          const T* Oleft = nullptr, T* shard_record = nullptr, double* shard_partial = nullptr) {
  using Cfg = TileCfg<T, D>;
  using RL = RecordLayout<T, D>;
  if (ws_bytes < tile_ws_bytes(N, D, sizeof(T))) return -1;
  if (Cfg::ROWS1 != tile_rows1(D, sizeof(T))) return -1;   // the two definitions of ROWS1 must agree
  const int clong = (Cfg::LPR > 1 || !fold_final_enabled()) ? 0 : long_chunk_rows_per_lane(N, Cfg::NG1, Cfg::C, (int)(D * D * sizeof(T)));
  const int csel = clong > 0 ? clong : ((Cfg::LPR > 1 || Cfg::ALWAYS_WIDE) ? Cfg::C : stage1_rows_per_lane(N, Cfg::C, Cfg::NT1));
  const int64_t rows_per_tile = (int64_t)csel * Cfg::NG1;
  const int64_t tiles = (N + rows_per_tile - 1) / rows_per_tile;
  const int64_t tiles_cap = tile_cap(N, D, sizeof(T));
  double* partial = reinterpret_cast<double*>(ws);
  const size_t pbytes = ((size_t)(2 * tiles_cap + 8) * PARTIAL_STRIDE * sizeof(double) + 255) & ~(size_t)255;
  T* recA = reinterpret_cast<T*>(ws + pbytes);
  T* recB = recA + (size_t)(tiles_cap + 2) * RL::STRIDE;
  const size_t lds1 = stage_lds_bytes<T, D>(Cfg::NG1, Cfg::NT1), lds3 = stage_lds_bytes<T, D>(Cfg::NTILE3, Cfg::NT3);
  tile_set_attributes<T, D>();
  if (ev_start) (void)hipEventRecord(ev_start, st);
  if constexpr (Cfg::LPR > 1) {
    const int slot = (tiles > 1 && tiles <= (int64_t)FOLD_GROUP * FOLD_MAX_GROUPS && fold_final_enabled()) ? fold_slot_for(ws) : -1;
    if (slot >= 0) {
      const FoldArgs fa{slot, recB, out2, info, shard_record, shard_partial};
      hipLaunchKernelGGL((chunk_reduce_ml_kernel<T, D, Cfg::C, Cfg::NT1, Cfg::LPR, true>), dim3((unsigned)tiles),
                         dim3(Cfg::NT1), lds1, st, Rs, Os, x, N, Oleft, recA, partial, fa);
      if (ev_stop) (void)hipEventRecord(ev_stop, st);
      return 0;
    }
    hipLaunchKernelGGL((chunk_reduce_ml_kernel<T, D, Cfg::C, Cfg::NT1, Cfg::LPR>), dim3((unsigned)tiles), dim3(Cfg::NT1),
                       lds1, st, Rs, Os, x, N, Oleft, recA, partial, FoldArgs{});
  }
  else if constexpr (Cfg::ALWAYS_WIDE) {
    const int slot = (tiles > 1 && tiles <= (int64_t)FOLD_GROUP * FOLD_MAX_GROUPS && fold_final_enabled()) ? fold_slot_for(ws) : -1;
    const size_t ldsw = stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1);
    if (slot >= 0 && clong > 0) {
      const FoldArgs fa{slot, recB, out2, info, shard_record, shard_partial, clong};
      hipLaunchKernelGGL((chunk_reduce_kernel<T, D, 0, Cfg::NT1, 2 * Cfg::NT1, true>), dim3((unsigned)tiles),
                         dim3(2 * Cfg::NT1), ldsw, st, Rs, Os, x, N, Oleft, recA, partial, fa);
      if (ev_stop) (void)hipEventRecord(ev_stop, st);
      return 0;
    }
    if (slot >= 0) {
      const FoldArgs fa{slot, recB, out2, info, shard_record, shard_partial};
      hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1, true>), dim3((unsigned)tiles),
                         dim3(2 * Cfg::NT1), ldsw, st, Rs, Os, x, N, Oleft, recA, partial, fa);
      if (ev_stop) (void)hipEventRecord(ev_stop, st);
      return 0;
    }
    hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1>), dim3((unsigned)tiles),
                       dim3(2 * Cfg::NT1), ldsw, st, Rs, Os, x, N, Oleft, recA, partial, FoldArgs{});
  } else if (clong > 0) {
    // more rows than one round of the chip: one round of longer chunks (see long_chunk_target_tiles)
    const int slot = fold_slot_for(ws);
    const FoldArgs fa{slot, recB, out2, info, shard_record, shard_partial, clong};
    bool wide = false;
    if constexpr (stage1_min_waves<T, D>() == 2) {
      if (!long_chunk_narrow()) {
        wide = true;
        const size_t ldsw = stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1);
        hipLaunchKernelGGL((chunk_reduce_kernel<T, D, 0, Cfg::NT1, 2 * Cfg::NT1, true>), dim3((unsigned)tiles),
                           dim3(2 * Cfg::NT1), ldsw, st, Rs, Os, x, N, Oleft, recA, partial, fa);
      }
    }
    if (!wide)
      hipLaunchKernelGGL((chunk_reduce_kernel<T, D, 0, Cfg::NT1, Cfg::NT1, true>), dim3((unsigned)tiles), dim3(Cfg::NT1),
                         lds1, st, Rs, Os, x, N, Oleft, recA, partial, fa);
    if (ev_stop) (void)hipEventRecord(ev_stop, st);
    return 0;
  } else if (csel != Cfg::C) {
    // small systems (fewer rows per lane, at most one workgroup per CU): the same one-launch form --
    // no faster on the GPU's clock than two launches (measured 2^14 .. 2^19 rows), but one node in a
    // caller's HIP graph and one launch on the host; a single tile (N <= 2048) finishes in place
    const int slot = (fold_final_enabled() && !(shard_record != nullptr && tiles == 1)) ? fold_slot_for(ws) : -1;
    const FoldArgs fa{slot, recB, out2, info, shard_record, shard_partial};
    auto launch = [&](auto cs) {
      constexpr int CS = decltype(cs)::value;
      if (slot >= 0)
        hipLaunchKernelGGL((chunk_reduce_kernel<T, D, CS, Cfg::NT1, Cfg::NT1, true>), dim3((unsigned)tiles), dim3(Cfg::NT1),
                           lds1, st, Rs, Os, x, N, Oleft, recA, partial, fa);
      else
        hipLaunchKernelGGL((chunk_reduce_kernel<T, D, CS, Cfg::NT1>), dim3((unsigned)tiles), dim3(Cfg::NT1), lds1, st,
                           Rs, Os, x, N, Oleft, recA, partial, FoldArgs{});
    };
    if (csel == 1) launch(std::integral_constant<int, 1>{});
    else if (csel == 4) launch(std::integral_constant<int, 4>{});
    else launch(std::integral_constant<int, 8>{});
    if (slot >= 0) {
      if (ev_stop) (void)hipEventRecord(ev_stop, st);
      return 0;
    }
  }
  else {
    // up to FOLD_GROUP * FOLD_MAX_GROUPS stage-1 workgroups: the record stages run inside the
    // launch (fold_final), for a whole system and for one shard of a larger one alike
    const int slot = (tiles > 1 && tiles <= (int64_t)FOLD_GROUP * FOLD_MAX_GROUPS && fold_final_enabled()) ? fold_slot_for(ws) : -1;
    const FoldArgs fa{slot, recB, out2, info, shard_record, shard_partial};
    bool wide = false;
    if constexpr (stage1_min_waves<T, D>() == 2) {
      // at most one workgroup per CU: give each a second set of role waves for its LDS reduction
      if (tiles <= STAGE1_SMALL_TILES) {
        wide = true;
        const size_t ldsw = stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1);
        if (slot >= 0)
          hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1, true>), dim3((unsigned)tiles),
                             dim3(2 * Cfg::NT1), ldsw, st, Rs, Os, x, N, Oleft, recA, partial, fa);
        else
          hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, 2 * Cfg::NT1>), dim3((unsigned)tiles),
                             dim3(2 * Cfg::NT1), ldsw, st, Rs, Os, x, N, Oleft, recA, partial, FoldArgs{});
      }
    }
    if (!wide) {
      if (slot >= 0)
        hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1, Cfg::NT1, true>), dim3((unsigned)tiles), dim3(Cfg::NT1),
                           lds1, st, Rs, Os, x, N, Oleft, recA, partial, fa);
      else
        hipLaunchKernelGGL((chunk_reduce_kernel<T, D, Cfg::C, Cfg::NT1>), dim3((unsigned)tiles), dim3(Cfg::NT1), lds1, st,
                           Rs, Os, x, N, Oleft, recA, partial, FoldArgs{});
    }
    if (slot >= 0) {
      if (ev_stop) (void)hipEventRecord(ev_stop, st);
      return 0;
    }
  }
  if (ev_stop) (void)hipEventRecord(ev_stop, st);
  int64_t n = tiles, npart = tiles, rows_per_record = rows_per_tile;
  T *rin = recA, *rout = recB;
  // a shard is reduced all the way to ONE record; the whole system stops as soon as the final
  // workgroup can take what is left
  while (shard_record ? n > 1 : n > (int64_t)Cfg::NTILE3 * Cfg::RCMAX) {
    int rc = (int)((n + Cfg::NTILE3 - 1) / Cfg::NTILE3);
    if (rc > Cfg::RCMAX) rc = Cfg::RCMAX;
    const int64_t per = (int64_t)Cfg::NTILE3 * rc;
    const int64_t g = (n + per - 1) / per;
    const bool shard_last = shard_record && g == 1;
    // the shard's single record and the sum of all partial results go straight to the caller
    if (shard_last) rout = shard_record;
    hipLaunchKernelGGL((record_reduce_kernel<T, D, Cfg::NTILE3, Cfg::NT3, false>), dim3((unsigned)g), dim3(Cfg::NT3),
                       lds3, st, (const T*)rin, n, rc, rout, shard_last ? shard_partial : partial + PARTIAL_STRIDE * npart,
                       shard_last ? (const double*)partial : (const double*)nullptr, shard_last ? npart : (int64_t)0,
                       (double*)nullptr, (int*)nullptr, rows_per_record, N, (int64_t)RL::STRIDE,
                       (int64_t)PARTIAL_STRIDE);
    if (shard_last) return 0;
    npart += g;
    n = g;
    rows_per_record *= per;
    T* tmp = rin; rin = rout; rout = tmp;
  }
  if (shard_record) {
    if (rin != shard_record)                              // a shard of a single stage-1 tile: its record is still in the workspace
      (void)hipMemcpyAsync(shard_record, rin, RL::STRIDE * sizeof(T), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(sum_partials4_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, npart, shard_partial);
    return 0;
  }
  const int rc = (int)((n + Cfg::NTILE3 - 1) / Cfg::NTILE3);
  hipLaunchKernelGGL((record_reduce_kernel<T, D, Cfg::NTILE3, Cfg::NT3, true>), dim3(1), dim3(Cfg::NT3), lds3, st,
                     (const T*)rin, n, rc, (T*)nullptr, (double*)nullptr, (const double*)partial, npart, out2, info,
                     rows_per_record, N, (int64_t)RL::STRIDE, (int64_t)PARTIAL_STRIDE);
  return 0;
}

// The same pipeline with the operands of a LEG model assembled in registers (chunk_reduce_kernel<.., SRC = 1>,
// cgps_tile_leg.h): J = PEG precision(ts, G) + blockdiag(A), right-hand side v (nullptr: zeros).  ONE launch at any
// size: the rows per lane are chosen at launch so that the grid is one round of at most 256 workgroups -- one row per
// lane up to 65 536 rows, where what counts is the length of a lane's chain of matrix exponentials.
// Built for the block sizes whose stage 1 runs one lane per row (every d <= 7 but fp64 d = 6).
// returns 0 on success, -1 when the workspace is too small, -2 when this (dtype, d) is not built
template <typename T, int D> constexpr bool leg_source_supported() { return TileCfg<T, D>::LPR == 1; }
template <typename T, int D>
int run_tile_leg(const T* ts, const T* G, const T* A, const T* v, int64_t N, char* ws, size_t ws_bytes, double* out2,
                 int* info, hipStream_t st, bool pair = false) {
  if constexpr (!leg_source_supported<T, D>()) {
    return -2;
  } else {
    using Cfg = TileCfg<T, D>;
    using RL = RecordLayout<T, D>;
    if (!fold_final_enabled()) return -2;
    const size_t ws1 = (tile_ws_bytes(N, D, sizeof(T)) + 255) & ~(size_t)255;
    if (ws_bytes < (pair ? 2 * ws1 : ws1)) return -1;
    const int64_t per_round = (int64_t)STAGE1_SMALL_TILES * Cfg::NG1;
    const int64_t c = (N + per_round - 1) / per_round;
    if (c > (1 << 20)) return -2;
    const int64_t rows_per_tile = c * Cfg::NG1;
    const int64_t tiles = (N + rows_per_tile - 1) / rows_per_tile;
    const int64_t tiles_cap = tile_cap(N, D, sizeof(T));
    double* partial = reinterpret_cast<double*>(ws);
    const size_t pbytes = ((size_t)(2 * tiles_cap + 8) * PARTIAL_STRIDE * sizeof(double) + 255) & ~(size_t)255;
    T* recA = reinterpret_cast<T*>(ws + pbytes);
    T* recB = recA + (size_t)(tiles_cap + 2) * RL::STRIDE;
    tile_set_attributes<T, D>();
    const FoldArgs fa{fold_slot_for(ws), recB, out2, info, nullptr, nullptr, (int)c, pair ? fold_slot_for(ws + ws1) : 0, ws1};
    const dim3 grid((unsigned)tiles, pair ? 2u : 1u);
    if constexpr (Cfg::ALWAYS_WIDE) {
      const size_t ldsw = stage_lds_bytes<T, D>(Cfg::NG1, 2 * Cfg::NT1);
      hipLaunchKernelGGL((chunk_reduce_kernel<T, D, 0, Cfg::NT1, 2 * Cfg::NT1, true, 1>), grid, dim3(2 * Cfg::NT1), ldsw, st, ts,
                         G, v, N, A, recA, partial, fa);
    } else {
      const size_t lds1 = stage_lds_bytes<T, D>(Cfg::NG1, Cfg::NT1);
      hipLaunchKernelGGL((chunk_reduce_kernel<T, D, 0, Cfg::NT1, Cfg::NT1, true, 1>), grid, dim3(Cfg::NT1), lds1, st, ts, G, v,
                         N, A, recA, partial, fa);
    }
    return 0;
  }
}

// Finish a sharded reduction: P shard records (in shard order) + their partial results ->
// out2 = {mahal, logdet}, info.  One workgroup; P <= NTILE3 * RCMAX.  rstride / pstride: distance
// between consecutive records (elements of T) / partials (doubles), so both can be read in place
// from an all-gather receive buffer of [record | partial] messages.
template <typename T, int D>
int run_tile_finish(const T* records, int64_t rstride, const double* partials, int64_t pstride, int64_t P,
                    int64_t rows_per_shard, int64_t N, double* out2, int* info, hipStream_t st) {
  using Cfg = TileCfg<T, D>;
  if (P < 1 || P > (int64_t)Cfg::NTILE3 * Cfg::RCMAX) return -1;
  const size_t lds3 = stage_lds_bytes<T, D>(Cfg::NTILE3, Cfg::NT3);
  tile_set_attributes<T, D>();
  const int rc = (int)((P + Cfg::NTILE3 - 1) / Cfg::NTILE3);
  hipLaunchKernelGGL((record_reduce_kernel<T, D, Cfg::NTILE3, Cfg::NT3, true>), dim3(1), dim3(Cfg::NT3), lds3, st,
                     records, P, rc, (T*)nullptr, (double*)nullptr, partials, P, out2, info, rows_per_shard, N,
                     rstride, pstride);
  return 0;
}

}  // namespace cgps
