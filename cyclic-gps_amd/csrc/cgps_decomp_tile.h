// Fused factor-emitting cyclic reduction (decompose, reference cyclic_reduction.py:287-309):
// the reference's exact even/odd elimination order, so the emitted Ds / Fs / Gs are the
// reference's blocks, but several levels per launch instead of one.
//
// ONE WAVE per 128-row tile, the tile in REGISTERS, no LDS: lane k loads rows 2k and 2k+1, and a
// level's neighbour exchange (what an eliminated row owes the odd rows left and right of it, and
// the new coupling between those two) goes through wave shuffles.  Why:
//   * this kernel runs in the throughput regime (thousands of tiles); with the tile in LDS
//     (256 B per row) only ~640 rows fit a CU, one wave per SIMD, and every latency of a level
//     (LDS round trips, the dependent fp64 chain, the global stores) was exposed: 182-218 us for
//     the first pass at N = 2^20;
//   * waves are persistent (grid = what the chip holds); the factor blocks go out through a small
//     LDS staging buffer so that every store instruction writes whole 128-byte lines;
//   * a level with fewer eliminations than lanes wastes the idle lanes, so a pass over many
//     tiles stops after DEC_LP = 3 levels (64 + 32 + 16 eliminations on 64 lanes) and hands the
//     16 surviving rows of every tile to the next pass as records; a pass over few tiles is
//     latency-bound anyway and runs all 7 levels of its tiles (one survivor each).
// Level j >= 1 of a pass: its rows m sit in lanes (m+1) st - 1, st = 2^(j-1); lane e of an even
// row takes the coupling J[m+1, m] from lane e + st, computes D, G, F (written straight into the
// packed per-level arrays; index = global elimination index of that level) and sends G G^T to
// lane e - st, F F^T and the new coupling -F G^T to lane e + st.
//   * the reference's size rule decides what a level eliminates (the last row too when its index
//     is even, cyclic_reduction.py:240-248); tiles start at multiples of 128, so local and global
//     parities agree at every level of a pass;
//   * what a tile's first eliminations owe the previous tile's last row travels as the DRA part
//     of the tile's first record and is added when the next pass loads that row.
// The host runs these passes while more than 2^15 rows are left and hands the latency-bound tail
// to cgps_decomp_lds.h.  N = 2^20: 2^20 -> 2^17 -> 2^14 here, then 2^14 -> 64 -> done there; every
// input block is read once, survivors are written and re-read once per pass.
#pragma once
#include "cgps_tile.h"

namespace cgps {

constexpr int DEC_LP = 3;               // levels per pass over many tiles
constexpr int DEC_TS = 128;             // rows per tile: two per lane
constexpr int DEC_TS_LOG2 = 7;          // levels per pass over few tiles (one survivor per tile)
constexpr int DEC_NT = 64;              // one wave
constexpr int DEC_MAXLEV = 8;           // the last pass takes a system of <= DEC_TS rows to the end: log2(128) + 1
constexpr int64_t DEC_FEW_TILES = 512;  // below this a pass is latency-bound: run all levels of a tile

struct DecompLevels {
  int64_t offD[DEC_MAXLEV], offF[DEC_MAXLEV], offG[DEC_MAXLEV];
  int nlev;
};

// a block / the lower triangle of a symmetric block as held by lane `src` (every lane of the wave
// executes this; lanes whose src is out of range get something they must not use)
template <typename T, int D>
__device__ __forceinline__ void shfl_block(T (&dst)[D][D], const T (&v)[D][D], int src) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) dst[a][b] = __shfl(v[a][b], src, 64);
}
template <typename T, int D>
__device__ __forceinline__ void shfl_lower(T (&dst)[D][D], const T (&v)[D][D], int src) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) dst[a][b] = __shfl(v[a][b], src, 64);
}
template <typename T, int D>
__device__ __forceinline__ void sub_lower(T (&S)[D][D], const T (&U)[D][D]) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) S[a][b] -= U[a][b];
}

// 16 bytes LDS -> global with a NON-TEMPORAL store.  The factor (and Sigma in cgps_inverse_tile.h)
// is written once and not read again by the kernel: written with ordinary stores, every line first
// lands in the memory-side cache and has to push an older dirty line out; back to back the first
// pass then runs at 4.5 TB/s (142 us) instead of the 5.3 TB/s (124 us) a device copy reaches.
template <typename V>
__device__ __forceinline__ void store_streaming16(V* dst, const V* src) {
  static_assert(sizeof(V) == 16, "16-byte granules");
  typedef float nv4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(*reinterpret_cast<const nv4*>(src), reinterpret_cast<nv4*>(dst));
}

// The blocks of a wave's eliminations k = 0 .. cnt-1 (lane holds block k if `has`) -> the
// contiguous array dst[cnt][D*D], through LDS so that every store instruction writes 64 x 16
// consecutive bytes.  Stored straight from the lanes, each instruction would touch 64 different
// 128-byte lines 16 bytes at a time: the write path then runs at a fraction of its rate (first
// pass at N = 2^20: 170 us with such stores, 79 us with none).  `first` skips leading blocks
// (G of the system's very first row does not exist).  `stage` holds 64 blocks.
// ONE wave only: a wave's LDS instructions execute in issue order, so the write and the read
// phase need no barrier, just the compiler kept from reordering them -- a __syncthreads() here
// would also wait for every global store and prefetch load in flight (vmcnt(0)), which is what
// made the earlier versions of this kernel slow.
template <typename T, int D>
__device__ __forceinline__ void store_blocks_coalesced(T* stage, T* __restrict__ dst, const T (&A)[D][D], bool has, int k,
                                                       int cnt, int first) {
  constexpr int DD = D * D, VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  if constexpr (DD % VN == 0) {
    using V = typename Vec16<T>::type;
    constexpr int G = DD / VN;                                    // 16-byte granules per block
    constexpr bool SWZ = (G & (G - 1)) == 0 && G >= 2;            // spread a column of granules over the banks
    V* sv = reinterpret_cast<V*>(stage);
    if (has) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        V v;
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) e[t] = A[(g * VN + t) / D][(g * VN + t) % D];
        sv[k * G + (SWZ ? (g ^ (k & (G - 1))) : g)] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    V* dv = reinterpret_cast<V*>(dst);
#pragma unroll 1
    for (int v = first * G + lane; v < cnt * G; v += 64) {
      const int kk = v / G, g = v % G;
      store_streaming16(&dv[v], &sv[kk * G + (SWZ ? (g ^ (kk & (G - 1))) : g)]);
    }
    __builtin_amdgcn_wave_barrier();
  } else {
    if (has) {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) stage[k * DD + a * D + b] = A[a][b];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int v = first * DD + lane; v < cnt * DD; v += 64) dst[v] = stage[v];
    __builtin_amdgcn_wave_barrier();
  }
}

// One pass of the factorisation.  FROM_RECORDS = false: rows are the caller's Rs / Os (level 0).
// FROM_RECORDS = true: rows are the previous pass's records (RecordLayout without the vector
// parts): R = Rs[w] (+ dRa[w+1] for the last survivor of a tile, spt_in survivors per tile),
// coupling to the previous row Cs[w].
// The single tile of the last pass (rec_out == nullptr) runs lv.nlev <= DEC_MAXLEV levels until
// nothing is left; otherwise lv.nlev <= 7 levels, then the surviving rows go to rec_out
// (tile t, survivor m -> record t * (DEC_TS >> lv.nlev) + m; a tile's first record also carries DRA).
// RHS (first pass of cgps_decompose_solve, FROM_RECORDS = false): the forward substitution of a right-hand side rides
// along -- x = D^-1 y of every eliminated row goes to xcrr (the layout of halfsolve, cyclic_reduction.py:312-338: level
// by level, elimination index within the level), the odd rows take y -= F x + G x' exactly where their blocks take
// -F F^T - G G^T, the surviving rows' y goes to ynext[tile * survivors + m] and what the tile's first eliminations owe
// the previous tile's last row to owedy[tile] (decomp_rhs_fixup_kernel adds it).  The forward sweep of a later solve
// then starts at the level this pass stops at: it never reads the factor blocks of the levels done here.
template <typename T>
__device__ __forceinline__ T shfl_one(T v, int src) { return __shfl(v, src, 64); }
template <typename T, int D>
__global__ __launch_bounds__(256) void decomp_rhs_fixup_kernel(T* __restrict__ ynext, const T* __restrict__ owedy, int64_t ntiles,
                                                               int spt_out, int64_t n_next) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (tile t >= 1, component) pairs
  const int64_t t = i / D + 1;
  const int c = (int)(i % D);
  if (t >= ntiles) return;
  const int64_t row = t * spt_out - 1;                              // last survivor of tile t - 1
  if (row < n_next) ynext[row * D + c] += owedy[t * D + c];
}
template <typename T, int D, bool FROM_RECORDS, bool RHS = false>
__global__ __launch_bounds__(DEC_NT, (stage1_min_waves<T, D>())) void decomp_tile_kernel(const T* __restrict__ Rin, const T* __restrict__ Oin,
                                                             int64_t n, int64_t n_rec, int spt_in, DecompLevels lv,
                                                             int lvl_first,
                                                             T* __restrict__ Dp, T* __restrict__ Fp,
                                                             T* __restrict__ Gp, T* __restrict__ rec_out,
                                                             int* __restrict__ info, const T* __restrict__ yin = nullptr,
                                                             T* __restrict__ xcrr = nullptr, T* __restrict__ ynext = nullptr,
                                                             T* __restrict__ owedy = nullptr) {
  static_assert(!RHS || !FROM_RECORDS, "the right-hand side rides along in the first pass only");
  constexpr int DD = D * D;
  using RL = RecordLayout<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* stage = reinterpret_cast<T*>(smem);          // 64 blocks: staging of the coalesced factor stores
  const int lane = threadIdx.x;
  const int64_t ntiles = (n + DEC_TS - 1) / DEC_TS;

  auto load_row = [&](int64_t w, T (&R)[D][D], T (&C)[D][D]) {     // row w of this pass and J[w, w-1]
    if constexpr (!FROM_RECORDS) {
      load_block<T, D>(Rin + w * DD, R);
      if (w >= 1) load_block<T, D>(Oin + (w - 1) * DD, C);
      else set_zero<T, D>(C);
    } else {
      const T* r = Rin + (size_t)w * RL::STRIDE;
      load_block<T, D>(r + RL::RS, R);
      load_block<T, D>(r + RL::CS, C);
      // only the last survivor of a tile of the previous pass is owed an update by the next
      // tile, whose first record carries it (n_rec >= n: a tile without survivors still leaves one)
      if (w + 1 < n_rec && (w + 1) % spt_in == 0) {
        T nR[D][D];
        load_block<T, D>(Rin + (size_t)(w + 1) * RL::STRIDE + RL::DRA, nR);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) R[a][b] += nR[a][b];
      }
    }
  };
  auto process = [&](int64_t tile) {
    const int64_t row0 = tile * DEC_TS;
    // The even row now, the odd row only once the even row's D and G are out: with all four
    // blocks loaded up front the kernel needs 330 registers (one wave per SIMD, every latency of
    // the wave exposed); this way 238, two waves per SIMD cover each other (first pass 156 -> 132 us).
    T Re[D][D], Cl[D][D], Rr[D][D], Cm[D][D];
    set_zero<T, D>(Re); set_zero<T, D>(Cl); set_zero<T, D>(Rr); set_zero<T, D>(Cm);   // lanes past the end of a
    if (row0 + 2 * lane < n) load_row(row0 + 2 * lane, Re, Cl);                       // ragged tile load nothing
    const int n0 = (int)((n - row0) < DEC_TS ? (n - row0) : DEC_TS);
    auto report = [&](int slot_row) { report_fail(info, ((row0 + slot_row + 1) << lvl_first) - 1); };
    T Cc[D][D];                    // coupling of the row this lane carries (Rr) to the previous such row
    // lane 0: minus what this tile owes the previous tile's last row.  The plain kernel keeps it in registers (238, no
    // scratch); with a right-hand side riding along those registers are what spills, so there lane 0 keeps it in LDS
    // behind the staging buffer (one wave per workgroup: its LDS instructions execute in order)
    T owed[RHS ? 1 : D][RHS ? 1 : D];
    T* const owed_l = stage + 64 * DD;     // RHS: [DD] lower triangle, then [D] the vector part
    set_zero<T, D>(Cc);
    if constexpr (RHS) {
      if (lane < DD + D) owed_l[lane] = T(0);
      if (DD + D > 64 && lane + 64 < DD + D) owed_l[lane + 64] = T(0);
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) owed[a][b] = T(0);
    }
    auto owe = [&](const T (&U)[D][D]) {   // lane 0 only
      if constexpr (RHS) {
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) owed_l[a * D + b] -= U[a][b];
      } else {
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) owed[a][b] -= U[a][b];
      }
    };
    int levels = 0;
    T yr[D];                       // RHS: right-hand side of the row this lane carries
    set_zero<T, D>(yr);

    // ---- level 0: lane k eliminates row 2k; row 2k+1 (if any) is the lane's level-1 row k -----
    {
      const bool act = 2 * lane < n0, has_o = 2 * lane + 1 < n0;
      T UL[D][D], G[D][D];
      set_zero<T, D>(UL);
      set_zero<T, D>(G);
      Chol<T, D> c;
      const int64_t ge0 = row0 >> 1;
      const int cntD = (n0 + 1) >> 1, cntF = n0 >> 1;
      {
        T L[D][D];
        set_zero<T, D>(L);
        if (act) {
          bool f = false;
          chol_lower<T, D>(Re, c, f);
          if (f) report(2 * lane);
          chol_to_dense<T, D>(c, L);
        }
        store_blocks_coalesced<T, D>(stage, Dp + (lv.offD[0] + ge0) * DD, L, act, lane, cntD, 0);
      }
      T x[D], gy[D];
      set_zero<T, D>(x);
      set_zero<T, D>(gy);
      if (act) {
        rsolve_lt_transposed<T, D>(c, Cl, G);                      // G = J[2k, 2k-1]^T D^-T
        syrk_lower<T, D>(UL, G);                                   // owed to row 2k-1
        if constexpr (RHS) {
          load_vec<T, D>(yin + (row0 + 2 * lane) * D, x);
          fwd_subst<T, D>(c, x);                                   // x = D^-1 y
          store_vec<T, D>(xcrr + (lv.offD[0] + ge0 + lane) * D, x);
#pragma unroll
          for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) gy[a] = fmaT(G[a][b], x[b], gy[a]);
        }
      }
      store_blocks_coalesced<T, D>(stage, Gp + (lv.offG[0] + ge0 - 1) * DD, G, act, lane, cntD, ge0 == 0 ? 1 : 0);
      if (has_o) load_row(row0 + 2 * lane + 1, Rr, Cm);
      if constexpr (RHS) {
        if (has_o) load_vec<T, D>(yin + (row0 + 2 * lane + 1) * D, yr);
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const T v = shfl_one<T>(gy[a], lane + 1);                // G x of row 2k+2, owed to row 2k+1
          if (has_o && 2 * lane + 2 < n0) yr[a] -= v;
          if (lane == 0) owed_l[DD + a] -= gy[a];
        }
      }
      {
        T U[D][D];
        shfl_lower<T, D>(U, UL, lane + 1);                         // what row 2k+2 owes row 2k+1
        if (has_o && 2 * lane + 2 < n0) sub_lower<T, D>(Rr, U);
      }
      if (lane == 0) owe(UL);
      if (has_o) rsolve_lt<T, D>(c, Cm);                           // F = J[2k+1, 2k] D^-T
      store_blocks_coalesced<T, D>(stage, Fp + (lv.offF[0] + ge0) * DD, Cm, has_o, lane, cntF, 0);
      if (has_o) {
        syrk_sub_lower<T, D>(Rr, Cm);
        neg_abt<T, D>(Cc, Cm, G);                                  // J'[2k+1, 2k-1] = -F G^T
        if constexpr (RHS) gemv_sub<T, D>(yr, Cm, x);              // y_{2k+1} -= F x
      }
      levels = 1;
    }

    // ---- levels 1 .. nlev-1: row m of level j sits in lane (m+1) st - 1, st = 2^(j-1) ---------
#pragma unroll 1
    for (int j = 1; j < lv.nlev; ++j) {
      const int M = n0 >> j;
      if (M < 1) break;
      const int st = 1 << (j - 1);
      const bool on_grid = ((lane + 1) & (st - 1)) == 0;           // this lane holds a row of level j
      const int m = ((lane + 1) >> (j - 1)) - 1;                   // its index
      const bool exists = on_grid && m < M;
      const bool even = exists && (m & 1) == 0, odd = exists && (m & 1) == 1;
      const bool has_o = even && (m + 1 < M);
      const int64_t ge0 = row0 >> (j + 1);                         // global index of the tile's first elimination
      const int cntD = (M + 1) >> 1, cntF = M >> 1;
      T G[D][D], F[D][D], U[D][D];
      Chol<T, D> c;
      set_zero<T, D>(G);
      set_zero<T, D>(U);
      shfl_block<T, D>(F, Cc, lane + st);                          // J[m+1, m] lives with row m+1
      {
        T L[D][D];
        set_zero<T, D>(L);
        if (even) {
          bool f = false;
          chol_lower<T, D>(Rr, c, f);
          if (f) report(((lane + 1) << 1) - 1);
          chol_to_dense<T, D>(c, L);
        }
        store_blocks_coalesced<T, D>(stage, Dp + (lv.offD[j] + ge0) * DD, L, even, m >> 1, cntD, 0);
      }
      T x[D], gy[D];
      set_zero<T, D>(x);
      set_zero<T, D>(gy);
      if (even) {
        rsolve_lt_transposed<T, D>(c, Cc, G);
        syrk_lower<T, D>(U, G);                                    // owed to row m-1
        if constexpr (RHS) {
#pragma unroll
          for (int a = 0; a < D; ++a) x[a] = yr[a];
          fwd_subst<T, D>(c, x);
          store_vec<T, D>(xcrr + (lv.offD[j] + ge0 + (m >> 1)) * D, x);
#pragma unroll
          for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) gy[a] = fmaT(G[a][b], x[b], gy[a]);
        }
      }
      if constexpr (RHS) {
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const T v = shfl_one<T>(gy[a], lane + st);               // G x of row m+1, owed to row m
          if (odd && m + 1 < M) yr[a] -= v;
          const T v0 = shfl_one<T>(gy[a], st - 1);                 // the tile's first elimination: owed to the previous tile
          if (lane == 0) owed_l[DD + a] -= v0;
        }
      }
      store_blocks_coalesced<T, D>(stage, Gp + (lv.offG[j] + ge0 - 1) * DD, G, even, m >> 1, cntD, ge0 == 0 ? 1 : 0);
      {
        T V[D][D];
        shfl_lower<T, D>(V, U, lane + st);
        if (odd && m + 1 < M) sub_lower<T, D>(Rr, V);
        shfl_lower<T, D>(V, U, st - 1);                            // the tile's first elimination: owed to the previous tile
        if (lane == 0) owe(V);
      }
      set_zero<T, D>(U);
      if (has_o) rsolve_lt<T, D>(c, F);
      store_blocks_coalesced<T, D>(stage, Fp + (lv.offF[j] + ge0) * DD, F, has_o, m >> 1, cntF, 0);
      if (has_o) syrk_lower<T, D>(U, F);                           // owed to row m+1
      if constexpr (RHS) {
        T fy[D];
        set_zero<T, D>(fy);
        if (has_o) {
#pragma unroll
          for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) fy[a] = fmaT(F[a][b], x[b], fy[a]);
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const T v = shfl_one<T>(fy[a], lane - st);               // F x of row m-1, owed to row m
          if (odd) yr[a] -= v;
        }
      }
      {
        T V[D][D];
        shfl_lower<T, D>(V, U, lane - st);
        if (odd) sub_lower<T, D>(Rr, V);
      }
      if (has_o) neg_abt<T, D>(U, F, G);                           // J'[m+1, m-1] = -F G^T
      {
        T V[D][D];
        shfl_block<T, D>(V, U, lane - st);
        if (odd) {
#pragma unroll
          for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) Cc[a][b] = V[a][b];
        }
      }
      levels = j + 1;
    }

    if (rec_out == nullptr) return;
    // ---- survivors -> records: after `levels` >= 1 levels the rows of level `levels` sit in
    // lanes (m+1) 2^(levels-1) - 1
    const int spt_out = DEC_TS >> lv.nlev;
    const int nsurv = n0 >> levels;
    const int st = 1 << (levels - 1);
    const bool on_grid = ((lane + 1) & (st - 1)) == 0;
    const int m = ((lane + 1) >> (levels - 1)) - 1;
    if (on_grid && m < nsurv) {
      mirror_lower<T, D>(Rr);
      T* r = rec_out + ((size_t)tile * spt_out + m) * RL::STRIDE;
      store_block<T, D>(r + RL::RS, Rr);
      store_block<T, D>(r + RL::CS, Cc);
      if constexpr (RHS) store_vec<T, D>(ynext + ((size_t)tile * spt_out + m) * D, yr);
    }
    if (lane == 0) {
      if constexpr (RHS) {
        T W[D][D], wy[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
          wy[a] = owed_l[DD + a];
#pragma unroll
          for (int b = 0; b <= a; ++b) W[a][b] = W[b][a] = owed_l[a * D + b];
        }
        store_block<T, D>(rec_out + (size_t)tile * spt_out * RL::STRIDE + RL::DRA, W);
        store_vec<T, D>(owedy + (size_t)tile * D, wy);
      } else {
        T W[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) W[a][b] = W[b][a] = owed[a][b];
        store_block<T, D>(rec_out + (size_t)tile * spt_out * RL::STRIDE + RL::DRA, W);
      }
    }
  };

  // persistent wave: tiles blockIdx.x, + gridDim.x, ...  (a second register buffer that keeps the
  // next tile's rows in flight was tried and is slower: it costs the registers that a second
  // wave per SIMD needs, and that second wave is what hides the latencies)
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) process(tile);
}

}  // namespace cgps
