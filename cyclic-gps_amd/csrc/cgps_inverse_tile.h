// Fused selected inverse (inverse_blocks, reference cyclic_reduction.py:470-503), coarse -> fine:
// INV_LP = 3 levels of the recurrence per launch instead of one.
//
// Same shape as cgps_decomp_tile.h run backwards: ONE WAVE per 128-row tile of the pass's finest
// level, the tile's blocks of Sigma in REGISTERS, neighbours through wave shuffles.  Relative
// level t = 3 is the input (16 rows per tile, read from the previous pass's output), t = 2, 1, 0
// are computed; row m of level t >= 1 sits in lane (m+1) 2^(t-1) - 1, lane k of level 0 holds
// rows 2k and 2k+1.  Every lane keeps, for the row it holds, Sigma[row,row] and
// Sigma[row, previous row of the current level].
// An even row 2k of a level needs Sigma of its two odd neighbours (lanes +-st) and their mutual
// coupling (held by the right one), computes its own diagonal block and the two couplings to its
// neighbours (inverse_even_row below = the level-wise kernel's algebra), keeps the left coupling
// and hands the right one to the right neighbour.  A tile's first even row takes its left
// neighbour -- the previous tile's last row, an odd row at every level of the pass, so its
// Sigma block is the input level's -- from global memory.
// The level-wise form reads and writes every level's Sigma once each way (2/3 of its traffic at
// the fine levels is this ping-pong); here a pass reads 1/8 of what it writes.
#pragma once
#include "cgps_decomp_tile.h"
#include "cgps_level.h"
#include "cgps_solve_tile.h"

namespace cgps {

constexpr int INV_LP = 3;
constexpr int INV_TS = 128;
constexpr int INV_NT = 64;
// fp64 d = 4 needs ~335 registers: one wave per SIMD.  Forcing two (256 registers) spills 83 of
// them to scratch and the big pass goes from 138 to 255 us; at one wave the chain of a tile
// (input load -> three dependent levels -> stores) is what bounds the pass.
#ifndef INV_MIN_WAVES
#define INV_MIN_WAVES 1
#endif

struct InverseLevels {
  int64_t offD[INV_LP], offF[INV_LP], offG[INV_LP];   // packed-array offsets of levels L, L+1, L+2
};

template <typename T, int D>
__device__ __forceinline__ void set_identity(T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = (i == j) ? T(1) : T(0);
}

// Even row 2k of one level.  Dl = D_k (dense lower Cholesky factor), F = F_k (if has_odd),
// G = G_k-1 (if has_left); SdR / SdL = Sigma~ of the right / left odd neighbour (coarse rows k and
// k-1), SoR = Sigma~[k, k-1].  Out: See = Sigma[2k,2k], oR = Sigma[2k+1,2k], oL = Sigma[2k,2k-1].
template <typename T, int D>
__device__ __forceinline__ void inverse_even_row(const T (&Dl)[D][D], const T (&F)[D][D], const T (&G)[D][D],
                                                 const T (&SdR)[D][D], const T (&SoR)[D][D], const T (&SdL)[D][D],
                                                 bool has_odd, bool has_left, T (&See)[D][D], T (&oR)[D][D],
                                                 T (&oL)[D][D]) {
  // D^-1 is lower triangular and Sigma[2k,2k] symmetric: only those halves are computed
  // (446 instead of 720 multiply-adds at d = 4; at one wave per SIMD the pass is bound by this chain).
  T inv[D], Di[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i) inv[i] = rcp_fast(Dl[i][i]);
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (i < j) Di[i][j] = T(0);
      else if (i == j) Di[i][j] = inv[i];
      else {
        T sacc = T(0);
#pragma unroll
        for (int m = j; m < i; ++m) sacc = fmaT(-Dl[i][m], Di[m][j], sacc);
        Di[i][j] = sacc * inv[i];
      }
    }
  }
  set_zero<T, D>(oR);
  set_zero<T, D>(oL);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {                       // lower(D^-T D^-1)
      T sacc = T(0);
#pragma unroll
      for (int m = i; m < D; ++m) sacc = fmaT(Di[m][i], Di[m][j], sacc);
      See[i][j] = sacc;
    }
  T Ak[D][D], Bk[D][D], M[D][D];
  auto times_di = [&](T (&C)[D][D], const T (&X)[D][D]) {   // C = X D^-1
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        T sacc = T(0);
#pragma unroll
        for (int m = j; m < D; ++m) sacc = fmaT(X[i][m], Di[m][j], sacc);
        C[i][j] = sacc;
      }
  };
  auto acc_lower_tn = [&](const T (&X)[D][D], const T (&Y)[D][D]) {   // lower(See) += lower(X^T Y)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        T sacc = See[i][j];
#pragma unroll
        for (int m = 0; m < D; ++m) sacc = fmaT(X[m][i], Y[m][j], sacc);
        See[i][j] = sacc;
      }
  };
  // C (+)= S X with S symmetric, only its lower triangle read: the callers then never need (nor
  // shuffle, nor keep in registers) the upper triangles of the Sigma diagonal blocks
  auto sym_times = [&](T (&C)[D][D], const T (&S)[D][D], const T (&X)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        T sacc = T(0);
#pragma unroll
        for (int m = 0; m < D; ++m) sacc = fmaT(m <= i ? S[i][m] : S[m][i], X[m][j], sacc);
        C[i][j] = sacc;
      }
  };
  set_zero<T, D>(Ak);
  set_zero<T, D>(Bk);
  if (has_odd) times_di(Ak, F);                          // A_k = F_k D_k^-1
  if (has_left) times_di(Bk, G);                         // B_k-1 = G_k-1 D_k^-1
  if (has_odd) {
    sym_times(M, SdR, Ak);
    if (has_left) mm_acc<T, D>(M, SoR, Bk);
    acc_lower_tn(Ak, M);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) oR[i][j] = -M[i][j];
  }
  if (has_left) {
    sym_times(M, SdL, Bk);
    if (has_odd) mm_tn_acc<T, D>(M, SoR, Ak);
    acc_lower_tn(Bk, M);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) oL[i][j] = -M[j][i];
  }
  mirror_lower<T, D>(See);
}

// The same algebra with its operands fetched when they are first needed and its results handed over as
// soon as they are final (the order of the level-wise kernel, cgps_level.h): for blocks of 64 scalars
// and more the six operands and three results of inverse_even_row do not fit in registers together.
// load*(A) fill a block; emit*(A) take a finished one.  loadSdR / loadSdL may leave the upper triangle
// unset (only the lower one is read).
template <typename T, int D, class LD, class LF, class LG, class LSdR, class LSoR, class LSdL, class EoR, class EoL,
          class ESee>
__device__ __forceinline__ void inverse_even_row_stream(bool has_odd, bool has_left, LD&& loadD, LF&& loadF, LG&& loadG,
                                                        LSdR&& loadSdR, LSoR&& loadSoR, LSdL&& loadSdL, EoR&& emit_oR,
                                                        EoL&& emit_oL, ESee&& emit_See) {
  // the three factor blocks are requested together: one HBM round trip per level, not three
  T L[D][D], Di[D][D], See[D][D], Fb[D][D], Gb[D][D];
  loadD(L);
  if (has_odd) loadF(Fb);
  if (has_left) loadG(Gb);
  {
    T inv[D];
#pragma unroll
    for (int i = 0; i < D; ++i) inv[i] = rcp_fast(L[i][i]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        if (i < j) Di[i][j] = T(0);
        else if (i == j) Di[i][j] = inv[i];
        else {
          T sacc = T(0);
#pragma unroll
          for (int m = j; m < i; ++m) sacc = fmaT(-L[i][m], Di[m][j], sacc);
          Di[i][j] = sacc * inv[i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {                       // lower(D^-T D^-1)
      T sacc = T(0);
#pragma unroll
      for (int m = i; m < D; ++m) sacc = fmaT(Di[m][i], Di[m][j], sacc);
      See[i][j] = sacc;
    }
  auto times_di = [&](T (&C)[D][D], const T (&X)[D][D]) {   // C = X D^-1
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        T sacc = T(0);
#pragma unroll
        for (int m = j; m < D; ++m) sacc = fmaT(X[i][m], Di[m][j], sacc);
        C[i][j] = sacc;
      }
  };
  auto acc_lower_tn = [&](const T (&X)[D][D], const T (&Y)[D][D]) {   // lower(See) += lower(X^T Y)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        T sacc = See[i][j];
#pragma unroll
        for (int m = 0; m < D; ++m) sacc = fmaT(X[m][i], Y[m][j], sacc);
        See[i][j] = sacc;
      }
  };
  auto sym_times = [&](T (&C)[D][D], const T (&S)[D][D], const T (&X)[D][D]) {   // C = S X, lower(S) read
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        T sacc = T(0);
#pragma unroll
        for (int m = 0; m < D; ++m) sacc = fmaT(m <= i ? S[i][m] : S[m][i], X[m][j], sacc);
        C[i][j] = sacc;
      }
  };
  T Ak[D][D], Bk[D][D], Soc[D][D], M[D][D];
  set_zero<T, D>(Ak);
  set_zero<T, D>(Bk);
  set_zero<T, D>(Soc);
  if (has_odd) times_di(Ak, Fb);                         // A_k = F_k D_k^-1
  if (has_left) times_di(Bk, Gb);                        // B_k-1 = G_k-1 D_k^-1
  if (has_odd && has_left) loadSoR(Soc);
  if (has_odd) {
    loadSdR(L);
    sym_times(M, L, Ak);
    if (has_left) mm_acc<T, D>(M, Soc, Bk);
    acc_lower_tn(Ak, M);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) L[i][j] = -M[i][j];
    emit_oR(L);
  }
  if (has_left) {
    loadSdL(L);
    sym_times(M, L, Bk);
    if (has_odd) mm_tn_acc<T, D>(M, Soc, Ak);
    acc_lower_tn(Bk, M);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) L[i][j] = -M[j][i];
    emit_oL(L);
  }
  mirror_lower<T, D>(See);
  emit_See(See);
}

// Lane k holds blocks 2k (A0, if has0) and 2k+1 (A1, if has1) of dst[cnt][D*D]; blocks below
// `first` are not written.  Through LDS (`stage`: 64 blocks), half a wave's pairs at a time, so
// that every store instruction writes 64 x 16 consecutive bytes (see store_blocks_coalesced).
template <typename T, int D>
__device__ __forceinline__ void store_pairs_coalesced(T* stage, T* __restrict__ dst, const T (&A0)[D][D], bool has0,
                                                      const T (&A1)[D][D], bool has1, int cnt, int first) {
  constexpr int DD = D * D, VN = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int b0 = 64 * h;                                       // first block of this half
    if (b0 >= cnt) break;
    const bool mine = (lane >> 5) == h;
    const int k = 2 * (lane & 31);
    const int hi = (cnt - b0) < 64 ? (cnt - b0) : 64;
    const int lo = first > b0 ? first - b0 : 0;
    if constexpr (DD % VN == 0) {
      using V = typename Vec16<T>::type;
      constexpr int G = DD / VN;
      constexpr bool SWZ = (G & (G - 1)) == 0 && G >= 2;
      V* sv = reinterpret_cast<V*>(stage);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        V v0, v1;
        T* e0 = reinterpret_cast<T*>(&v0);
        T* e1 = reinterpret_cast<T*>(&v1);
#pragma unroll
        for (int t = 0; t < VN; ++t) {
          e0[t] = A0[(g * VN + t) / D][(g * VN + t) % D];
          e1[t] = A1[(g * VN + t) / D][(g * VN + t) % D];
        }
        if (mine && has0) sv[k * G + (SWZ ? (g ^ (k & (G - 1))) : g)] = v0;
        if (mine && has1) sv[(k + 1) * G + (SWZ ? (g ^ ((k + 1) & (G - 1))) : g)] = v1;
      }
      __builtin_amdgcn_wave_barrier();
      V* dv = reinterpret_cast<V*>(dst) + (size_t)b0 * G;
      if (lo == 0 && hi == 64) {
        // full half: all LDS reads in flight before the first store (the rolled loop below pays
        // an LDS round trip per 1 KB stored)
        V tmp[G];
#pragma unroll
        for (int it = 0; it < G; ++it) {
          const int v = it * 64 + lane, kk = v / G, g = v % G;
          tmp[it] = sv[kk * G + (SWZ ? (g ^ (kk & (G - 1))) : g)];
        }
#pragma unroll
        for (int it = 0; it < G; ++it) store_streaming16(&dv[it * 64 + lane], &tmp[it]);
      } else {
#pragma unroll 1
        for (int v = lo * G + lane; v < hi * G; v += 64) {
          const int kk = v / G, g = v % G;
          store_streaming16(&dv[v], &sv[kk * G + (SWZ ? (g ^ (kk & (G - 1))) : g)]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      if (mine) {
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) {
            if (has0) stage[k * DD + a * D + b] = A0[a][b];
            if (has1) stage[(k + 1) * DD + a * D + b] = A1[a][b];
          }
      }
      __builtin_amdgcn_wave_barrier();
      T* dd = dst + (size_t)b0 * DD;
#pragma unroll 1
      for (int v = lo * DD + lane; v < hi * DD; v += 64) dd[v] = stage[v];
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// One pass: Sigma of level L+3 (Sd_in[n >> 3], So_in[(n >> 3) - 1]) -> Sigma of level L
// (Sd_out[n], So_out[n-1]); n = rows of level L, n >> 3 >= 1.  Dp/Fp/Gp: the packed factor.
template <typename T, int D>
__global__ __launch_bounds__(INV_NT, INV_MIN_WAVES) void inverse_tile_kernel(const T* __restrict__ Dp, const T* __restrict__ Fp,
                                                              const T* __restrict__ Gp, InverseLevels lv,
                                                              const T* __restrict__ Sd_in, const T* __restrict__ So_in,
                                                              int64_t n, T* __restrict__ Sd_out,
                                                              T* __restrict__ So_out) {
  constexpr int DD = D * D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* stage = reinterpret_cast<T*>(smem);
  const int lane = threadIdx.x;
  const int64_t ntiles = (n + INV_TS - 1) / INV_TS;

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * INV_TS;
    const int n0 = (int)((n - row0) < INV_TS ? (n - row0) : INV_TS);
    const int64_t g3 = row0 >> INV_LP;
    const int n3 = n0 >> INV_LP;
    T Sdv[D][D], Sov[D][D];        // of Sdv only the lower triangle is kept up to date
    set_zero<T, D>(Sdv);
    set_zero<T, D>(Sov);
    {
      const int m = ((lane + 1) >> 2) - 1;
      if (((lane + 1) & 3) == 0 && m < n3) {
        load_block<T, D>(Sd_in + (g3 + m) * DD, Sdv);
        if (g3 + m >= 1) load_block<T, D>(So_in + (g3 + m - 1) * DD, Sov);
      }
    }
    // the tile's first even row of every level: its left neighbour is the previous tile's last
    // row (fetched when needed rather than kept: registers)
    auto left_of_tile = [&](T (&SdL)[D][D]) {
      if (row0 > 0) load_block<T, D>(Sd_in + (g3 - 1) * DD, SdL);
    };

    auto geom = [&](int t, bool& even, bool& odd, bool& has_odd, bool& has_left, int& m, int64_t& kg) {
      if (t == 0) {
        even = 2 * lane < n0; odd = false; has_odd = 2 * lane + 1 < n0; m = 2 * lane;
        kg = (row0 >> 1) + lane;
      } else {
        const int st = 1 << (t - 1);
        const int M = n0 >> t;
        m = ((lane + 1) >> (t - 1)) - 1;
        const bool exists = (((lane + 1) & (st - 1)) == 0) && m < M;
        even = exists && (m & 1) == 0; odd = exists && (m & 1) == 1;
        kg = (row0 >> (t + 1)) + (m >> 1);
        has_odd = even && (m + 1 < M);
      }
      has_left = even && kg >= 1;
    };
    auto load_factors = [&](int t, T (&Dl)[D][D], T (&F)[D][D], T (&G)[D][D]) {
      bool even, odd, has_odd, has_left; int m; int64_t kg;
      geom(t, even, odd, has_odd, has_left, m, kg);
      set_zero<T, D>(F);
      set_zero<T, D>(G);
      if (even) load_block<T, D>(Dp + (lv.offD[t] + kg) * DD, Dl);
      else set_identity<T, D>(Dl);
      if (has_odd) load_block<T, D>(Fp + (lv.offF[t] + kg) * DD, F);
      if (has_left) load_block<T, D>(Gp + (lv.offG[t] + kg - 1) * DD, G);
    };

    // ---- relative levels 2 and 1 -------------------------------------------------------------
#pragma unroll
    for (int t = INV_LP - 1; t >= 1; --t) {
      T Dl[D][D], F[D][D], G[D][D];
      load_factors(t, Dl, F, G);
      const int st = 1 << (t - 1);
      bool even, odd, has_odd, has_left; int m; int64_t kg;
      geom(t, even, odd, has_odd, has_left, m, kg);
      T SdR[D][D], SoR[D][D], SdL[D][D];
      shfl_block<T, D>(SdR, Sdv, lane + st);
      shfl_block<T, D>(SoR, Sov, lane + st);
      shfl_block<T, D>(SdL, Sdv, lane - st);
      if (even && m == 0) left_of_tile(SdL);
      T See[D][D], oR[D][D], oL[D][D];
      inverse_even_row<T, D>(Dl, F, G, SdR, SoR, SdL, has_odd, has_left, See, oR, oL);
      T V[D][D];
      shfl_block<T, D>(V, oR, lane - st);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
          if (even) { Sdv[a][b] = See[a][b]; Sov[a][b] = oL[a][b]; }
          if (odd) Sov[a][b] = V[a][b];
        }
    }

    // ---- relative level 0: lane k computes row 2k, holds row 2k+1 ------------------------------
    {
      bool even, odd, has_odd, has_left; int m; int64_t kg;
      geom(0, even, odd, has_odd, has_left, m, kg);
      T Dl[D][D], F[D][D], G[D][D];
      load_factors(0, Dl, F, G);
      T SdL[D][D];
      shfl_block<T, D>(SdL, Sdv, lane - 1);
      if (lane == 0) left_of_tile(SdL);
      T See[D][D], oR[D][D], oL[D][D];
      inverse_even_row<T, D>(Dl, F, G, Sdv, Sov, SdL, has_odd, has_left, See, oR, oL);
      mirror_lower<T, D>(Sdv);
      store_pairs_coalesced<T, D>(stage, Sd_out + row0 * DD, See, even, Sdv, has_odd, n0, 0);
      // couplings So[row0 - 1 .. row0 + n0 - 2]: pair index 2k <-> Sigma[2k, 2k-1], 2k+1 <-> Sigma[2k+1, 2k]
      store_pairs_coalesced<T, D>(stage, So_out + (row0 - 1) * DD, oL, even, oR, has_odd, n0, row0 == 0 ? 1 : 0);
    }
  }
}


// ---- the same pass for LARGE blocks (> 200 bytes: fp32 d = 8, fp64 d = 6..8) ------------------------
// With 64 or more scalars per block the register form above spills (920 bytes per lane at fp32
// d = 8: Sigma of the held row, its shuffled neighbours and the even row's algebra do not fit in 512
// registers).  Here Sigma of the tile's level-1 .. level-3 rows lives in LDS instead -- slot u = the
// level-1 row u = level-0 row 2u + 1; row m of relative level t >= 1 sits in slot (m + 1) 2^(t-1) - 1,
// which is also the lane that computes it -- with its diagonal block and its coupling to the PREVIOUS
// row of the current level (the scheme of inverse_deep_kernel).  An even row reads its two odd
// neighbours' slots, writes its own diagonal block and the two new couplings: disjoint slots per
// elimination, one barrier per level.  The registers then hold one even row's algebra only, exactly
// what the level-wise kernel holds.  Level 0 leaves through the coalescing stage of
// store_pairs_coalesced, which reuses the slots once every lane has read its neighbours.
template <typename T, int D> constexpr size_t inverse_tile_lds_bytes() { return (size_t)(2 * 64 + 1) * D * D * sizeof(T); }

// A slot array in LDS, one block per slot.  Lanes read and write whole blocks of neighbouring slots:
// at a block stride of 256 or 512 bytes every lane of a 16-byte access would land on the same four
// banks.  The 16-byte granules of slot u are therefore stored rotated by (u ^ u/4) -- distinct for the
// 16 consecutive slots of level 0, the 16 slots 4j + 1 of level 1 and the 8 slots 8j + 3 of level 2.
template <typename T, int D>
struct SlotIO {
  static constexpr int DD = D * D, VN = Vec16<T>::N;
  static constexpr bool VEC = DD % VN == 0;
  static constexpr int GR = VEC ? DD / VN : 1;
  static constexpr bool SWZ = VEC && (GR & (GR - 1)) == 0 && GR >= 4;
  static __device__ __forceinline__ int pos(int u, int g) { return u * GR + (SWZ ? (g ^ ((u ^ (u >> 2)) & (GR - 1))) : g); }
  static __device__ __forceinline__ void load(const T* base, int u, T (&A)[D][D]) {
    if constexpr (VEC) {
      using V = typename Vec16<T>::type;
      const V* b = reinterpret_cast<const V*>(base);
#pragma unroll
      for (int g = 0; g < GR; ++g) {
        const V v = b[pos(u, g)];
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) A[(g * VN + t) / D][(g * VN + t) % D] = e[t];
      }
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b2 = 0; b2 < D; ++b2) A[a][b2] = base[u * DD + a * D + b2];
    }
  }
  static __device__ __forceinline__ void store(T* base, int u, const T (&A)[D][D]) {
    if constexpr (VEC) {
      using V = typename Vec16<T>::type;
      V* b = reinterpret_cast<V*>(base);
#pragma unroll
      for (int g = 0; g < GR; ++g) {
        V v;
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int t = 0; t < VN; ++t) e[t] = A[(g * VN + t) / D][(g * VN + t) % D];
        b[pos(u, g)] = v;
      }
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b2 = 0; b2 < D; ++b2) base[u * DD + a * D + b2] = A[a][b2];
    }
  }
};

template <typename T, int D>
__global__ __launch_bounds__(INV_NT, 1) void inverse_tile_lds_kernel(const T* __restrict__ Dp, const T* __restrict__ Fp,
                                                                   const T* __restrict__ Gp, InverseLevels lv,
                                                                   const T* __restrict__ Sd_in, const T* __restrict__ So_in,
                                                                   int64_t n, T* __restrict__ Sd_out, T* __restrict__ So_out) {
  constexpr int DD = D * D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sd = reinterpret_cast<T*>(smem);                 // [64][DD]  Sigma[row, row]
  T* so = sd + 64 * DD;                               // [64][DD]  Sigma[row, previous row of the current level]
  T* halo = so + 64 * DD;                             // Sigma[row, row] of the previous tile's last row (an input row at every level)
  const int lane = threadIdx.x;
  const int64_t ntiles = (n + INV_TS - 1) / INV_TS;

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * INV_TS;
    const int n0 = (int)((n - row0) < INV_TS ? (n - row0) : INV_TS);
    const int64_t g3 = row0 >> INV_LP;
    const int n3 = n0 >> INV_LP;
    // input level: row m -> slot 4 (m + 1) - 1; the rows are contiguous in Sd_in / So_in
    {
      constexpr int VN = Vec16<T>::N;
      const int first_o = (g3 >= 1) ? 0 : 1;          // So_in[g3 + m - 1] exists from m = first_o on
      if constexpr (DD % VN == 0) {
        using V = typename Vec16<T>::type;
        constexpr int GR = DD / VN;
        const V* gd = reinterpret_cast<const V*>(Sd_in + g3 * DD);
        const V* go = reinterpret_cast<const V*>(So_in + (g3 - 1) * DD);
        V* sdv = reinterpret_cast<V*>(sd);
        V* sov = reinterpret_cast<V*>(so);
        for (int v = lane; v < n3 * GR; v += INV_NT) {
          const int m = v / GR, g = v % GR;
          sdv[SlotIO<T, D>::pos(4 * (m + 1) - 1, g)] = gd[v];
          if (m >= first_o) sov[SlotIO<T, D>::pos(4 * (m + 1) - 1, g)] = go[v];
        }
        if (g3 >= 1 && lane < GR) reinterpret_cast<V*>(halo)[lane] = (gd - GR)[lane];
      } else {
        for (int v = lane; v < n3 * DD; v += INV_NT) {
          const int m = v / DD, e = v % DD;
          sd[(4 * (m + 1) - 1) * DD + e] = Sd_in[g3 * DD + v];
          if (m >= first_o) so[(4 * (m + 1) - 1) * DD + e] = So_in[(g3 - 1) * DD + v];
        }
        if (g3 >= 1 && lane < DD) halo[lane] = Sd_in[(g3 - 1) * DD + lane];
      }
    }
    __syncthreads();

    auto geom = [&](int t, bool& even, bool& has_odd, bool& has_left, int& m, int64_t& kg) {
      if (t == 0) {
        even = 2 * lane < n0; has_odd = 2 * lane + 1 < n0; m = 2 * lane;
        kg = (row0 >> 1) + lane;
      } else {
        const int st = 1 << (t - 1);
        const int M = n0 >> t;
        m = ((lane + 1) >> (t - 1)) - 1;
        const bool exists = (((lane + 1) & (st - 1)) == 0) && m < M;
        even = exists && (m & 1) == 0;
        kg = (row0 >> (t + 1)) + (m >> 1);
        has_odd = even && (m + 1 < M);
      }
      has_left = even && kg >= 1;
    };
    // ---- relative levels 2 and 1: lane = slot of the row it computes ---------------------------
#pragma unroll 1
    for (int t = INV_LP - 1; t >= 1; --t) {
      bool even, has_odd, has_left; int m; int64_t kg;
      geom(t, even, has_odd, has_left, m, kg);
      const int st = 1 << (t - 1);
      if (even) {
        inverse_even_row_stream<T, D>(
            has_odd, has_left,
            [&](T (&A)[D][D]) { load_block<T, D>(Dp + (lv.offD[t] + kg) * DD, A); },
            [&](T (&A)[D][D]) { load_block<T, D>(Fp + (lv.offF[t] + kg) * DD, A); },
            [&](T (&A)[D][D]) { load_block<T, D>(Gp + (lv.offG[t] + kg - 1) * DD, A); },
            [&](T (&A)[D][D]) { SlotIO<T, D>::load(sd, lane + st, A); },
            [&](T (&A)[D][D]) { SlotIO<T, D>::load(so, lane + st, A); },
            [&](T (&A)[D][D]) {
              if (m == 0) load_block<T, D>(halo, A);                       // the previous tile's last row
              else SlotIO<T, D>::load(sd, lane - st, A);
            },
            [&](const T (&A)[D][D]) { SlotIO<T, D>::store(so, lane + st, A); },   // the right neighbour's previous row is now this one
            [&](const T (&A)[D][D]) { SlotIO<T, D>::store(so, lane, A); },
            [&](const T (&A)[D][D]) { SlotIO<T, D>::store(sd, lane, A); });
      }
      __syncthreads();
    }

    // ---- relative level 0: lane k computes row 2k; row 2k+1 is slot k ---------------------------
    // Results go straight to global memory as they become final, like the level-wise kernel's (a block
    // is two or more whole 128-byte lines): holding them for a coalescing stage costs the registers
    // the algebra needs.
    {
      bool even, has_odd, has_left; int m; int64_t kg;
      geom(0, even, has_odd, has_left, m, kg);
      const int64_t r = row0 + 2 * lane;
      if (even) {
        inverse_even_row_stream<T, D>(
            has_odd, has_left,
            [&](T (&A)[D][D]) { load_block<T, D>(Dp + (lv.offD[0] + kg) * DD, A); },
            [&](T (&A)[D][D]) { load_block<T, D>(Fp + (lv.offF[0] + kg) * DD, A); },
            [&](T (&A)[D][D]) { load_block<T, D>(Gp + (lv.offG[0] + kg - 1) * DD, A); },
            [&](T (&A)[D][D]) { SlotIO<T, D>::load(sd, lane, A); },
            [&](T (&A)[D][D]) { SlotIO<T, D>::load(so, lane, A); },
            [&](T (&A)[D][D]) {
              if (lane == 0) load_block<T, D>(halo, A);
              else SlotIO<T, D>::load(sd, lane - 1, A);
            },
            [&](const T (&A)[D][D]) { store_block<T, D>(So_out + r * DD, A); },          // Sigma[2k+1, 2k]
            [&](const T (&A)[D][D]) { store_block<T, D>(So_out + (r - 1) * DD, A); },    // Sigma[2k, 2k-1]
            [&](const T (&A)[D][D]) { store_block<T, D>(Sd_out + r * DD, A); });
      }
      // Sigma[2u+1, 2u+1] = slot u, unchanged: consecutive lanes copy consecutive 16-byte granules
      const int nodd = n0 >> 1;
      constexpr int VN = Vec16<T>::N;
      if constexpr (DD % VN == 0) {
        using V = typename Vec16<T>::type;
        constexpr int GR = DD / VN;
        const V* sdv = reinterpret_cast<const V*>(sd);
        V* od = reinterpret_cast<V*>(Sd_out + row0 * DD);
        for (int v = lane; v < nodd * GR; v += INV_NT) {
          const int u = v / GR, g = v % GR;
          od[(size_t)(2 * u + 1) * GR + g] = sdv[SlotIO<T, D>::pos(u, g)];
        }
      } else {
        for (int v = lane; v < nodd * DD; v += INV_NT) {
          const int u = v / DD, e = v % DD;
          Sd_out[(row0 + 2 * u + 1) * DD + e] = sd[u * DD + e];
        }
      }
      __syncthreads();
    }
  }
}


// ---- the coarse end of the recurrence in ONE launch --------------------------------------------------
// The recurrence starts at the single row of the coarsest level and doubles the rows per level; while a
// level has a few hundred rows a launch per level is pure latency (nine launches, 48 us, at
// N = 2^20).  Same idea as the latency-bound passes of the solve (cgps_solve_tile.h): WHICH factor
// blocks the levels need depends on no data, so every lane requests the D / F / G of its
// elimination of the finest level of this kernel and of ONE deeper elimination (lane <-> elimination
// map of deep_owner) before anything else -- one HBM round trip -- and then the levels only touch
// LDS: Sigma of the <= INVD_TS rows stays there, row m of local level j in slot (m + 1) 2^j - 1 with
// its diagonal block and its coupling to the PREVIOUS row of the current level.  An even row reads its
// two odd neighbours' diagonal blocks and their coupling, writes its own diagonal block and the two
// new couplings (inverse_even_row): disjoint slots per elimination, one barrier per level.
// Blocks <= 256 bytes (the tile is 512 rows x 2 blocks of <= 128 bytes, or 256 rows x 2 larger blocks), one workgroup.
// rows of the finest level this kernel takes: 512 for blocks <= 128 bytes, 256 up to 256 bytes
template <typename T, int D> constexpr int invd_tsl() { return (size_t)D * D * sizeof(T) <= 128 ? 9 : 8; }
constexpr int INVD_MAXLEV = 10;
struct InverseDeepLevels {
  int64_t offD[INVD_MAXLEV], offF[INVD_MAXLEV], offG[INVD_MAXLEV];   // packed-array offsets of the kernel's levels, finest first
  int nlev;
};
// (larger blocks spill in this kernel -- 376 B per lane at fp64 d = 5 -- and gain nothing: they stay one launch per coarse level)
template <typename T, int D> constexpr bool inverse_deep_supported() { return (size_t)D * D * sizeof(T) <= 128; }
template <typename T, int D> constexpr size_t inverse_deep_lds_bytes() { return ((size_t)2 << invd_tsl<T, D>()) * D * D * sizeof(T); }

// n: rows of the finest level (<= INVD_TS); output in natural order: Sd_out[n], So_out[n - 1]
template <typename T, int D>
__global__ __launch_bounds__((1 << invd_tsl<T, D>()) / 2, 1) void inverse_deep_kernel(const T* __restrict__ Dp, const T* __restrict__ Fp,
                                                                 const T* __restrict__ Gp, InverseDeepLevels lv, int n,
                                                                 T* __restrict__ Sd_out, T* __restrict__ So_out) {
  constexpr int DD = D * D, INVD_TSL = invd_tsl<T, D>(), INVD_TS = 1 << INVD_TSL, INVD_NT = INVD_TS / 2;
  extern __shared__ __attribute__((aligned(16))) char inv_smem[];
  T* sd = reinterpret_cast<T*>(inv_smem);                 // [INVD_TS][DD]  Sigma[row, row]
  T* so = sd + (size_t)INVD_TS * DD;                       // [INVD_TS][DD]  Sigma[row, previous row of the current level]
  const int tid = threadIdx.x;
  // what elimination k of local level j needs
  auto request = [&](int j, int k, bool on, T (&L)[D][D], T (&F)[D][D], T (&G)[D][D], bool& has_odd, bool& has_left) {
    const int nj = n >> j;
    on = on && j < lv.nlev && k < ((nj + 1) >> 1);
    has_odd = on && (2 * k + 1 < nj);
    has_left = on && k >= 1;
    if (on) load_block<T, D>(Dp + (lv.offD[j] + k) * DD, L); else set_identity<T, D>(L);
    if (has_odd) load_block<T, D>(Fp + (lv.offF[j] + k) * DD, F); else set_zero<T, D>(F);
    if (has_left) load_block<T, D>(Gp + (lv.offG[j] + k - 1) * DD, G); else set_zero<T, D>(G);
    return on;
  };
  T L0[D][D], F0[D][D], G0[D][D], Ld[D][D], Fd[D][D], Gd[D][D];
  bool odd0, left0, oddd, leftd;
  const bool on0 = request(0, tid, true, L0, F0, G0, odd0, left0);
  const DeepOwner own = deep_owner<INVD_TSL>(tid);
  const bool ond = request(own.j, own.k, own.j >= 1, Ld, Fd, Gd, oddd, leftd);
  auto run = [&](int j, int k, const T (&L)[D][D], const T (&F)[D][D], const T (&G)[D][D], bool has_odd, bool has_left) {
    const int se = ((2 * k + 1) << j) - 1, st = 1 << j;   // the even row's slot, distance to its odd neighbours
    T SdR[D][D], SoR[D][D], SdL[D][D], See[D][D], oR[D][D], oL[D][D];
    if (has_odd) load_block<T, D>(sd + (size_t)(se + st) * DD, SdR); else set_zero<T, D>(SdR);
    if (has_odd && has_left) load_block<T, D>(so + (size_t)(se + st) * DD, SoR); else set_zero<T, D>(SoR);
    if (has_left) load_block<T, D>(sd + (size_t)(se - st) * DD, SdL); else set_zero<T, D>(SdL);
    inverse_even_row<T, D>(L, F, G, SdR, SoR, SdL, has_odd, has_left, See, oR, oL);
    store_block<T, D>(sd + (size_t)se * DD, See);
    if (has_odd) store_block<T, D>(so + (size_t)(se + st) * DD, oR);     // the right neighbour's previous row is now this one
    if (has_left) store_block<T, D>(so + (size_t)se * DD, oL);
  };
#pragma unroll 1
  for (int j = lv.nlev - 1; j >= 1; --j) {
    if (ond && own.j == j) run(j, own.k, Ld, Fd, Gd, oddd, leftd);
    __syncthreads();
  }
  if (on0) run(0, tid, L0, F0, G0, odd0, left0);
  __syncthreads();
  // natural order out: 16-byte granules, consecutive threads = consecutive granules
  constexpr int VN = Vec16<T>::N;
  if constexpr (DD % VN == 0) {
    using V = typename Vec16<T>::type;
    constexpr int GR = DD / VN;
    const V* sdv = reinterpret_cast<const V*>(sd);
    const V* sov = reinterpret_cast<const V*>(so);
    V* od = reinterpret_cast<V*>(Sd_out);
    V* oo = reinterpret_cast<V*>(So_out);
    for (int v = tid; v < n * GR; v += INVD_NT) od[v] = sdv[v];
    for (int v = tid; v < (n - 1) * GR; v += INVD_NT) oo[v] = sov[v + GR];         // So[i] = Sigma[i+1, i]: slot i + 1
  } else {
    for (int v = tid; v < n * DD; v += INVD_NT) Sd_out[v] = sd[v];
    for (int v = tid; v < (n - 1) * DD; v += INVD_NT) So_out[v] = so[v + DD];
  }
}

}  // namespace cgps
