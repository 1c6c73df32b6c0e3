// Three levels of the selected inverse per launch for 8 x 8 blocks, FOUR LANES PER BLOCK ROW.
//
// inverse_tile_lds_kernel (cgps_inverse_tile.h) runs one lane per even row: at 64 scalars per block that is
// ~470 registers -- one wave per SIMD, nothing to cover a level's HBM round trip with -- and its levels 2 and
// 1 keep a quarter and a half of the lanes busy.  Here a DPP quad shares a row: lane q owns matrix rows
// 2q and 2q + 1 of every block (a quad loads one contiguous block together); one wave = 16 quads takes a
// 128-row tile (level 2 has 16 even rows, level 1 32, level 0 64: one, two, four rounds of the wave, all
// lanes busy in every round).  Sigma of the tile's level-1 .. level-3 rows lives in LDS exactly as in
// inverse_tile_lds_kernel (slot = level-1 row, SlotIO's rotated granules).
//
// One even row on a quad (the algebra of inverse_even_row):
//   D^-1 (lower) from the gathered factor block, redundantly on the four lanes;
//   own rows of A = F D^-1 and B = G D^-1;
//   gather A (quad broadcasts):  M1_own  = SdR_own A,       M2_own  = SoR^T_own A
//   gather B:                    M1_own += SoR_own B,       M2_own += SdL_own B
//     (own rows of SdR / SoR / SdL and the column pairs (2q, 2q+1) of SoR come straight from the slots)
//   Sigma[2k+1, 2k] = -M1 leaves by rows, Sigma[2k, 2k-1] = -M2^T by column pairs;
//   Sigma[2k, 2k] = D^-T D^-1 + A^T M1 + B^T M2: every lane sums the outer products of ITS two rows of
//   (A, M1) and (B, M2) into a full lower triangle, two quad exchanges add the four partial sums, the
//   lane keeps its two rows.
#pragma once
#include "cgps_inverse_tile.h"

namespace cgps {

#ifndef CGPS_INVQ_AHEAD
#define CGPS_INVQ_AHEAD 1
#endif
template <typename T> constexpr int inverse_quad_threads() { return sizeof(T) == 4 ? 64 : 256; }
template <typename T> constexpr size_t inverse_quad_lds_bytes() {
  return (size_t)(2 * 64 + 1 + inverse_quad_threads<T>() / 4) * 64 * sizeof(T);   // slots, halo, one transposing block per quad
}

template <typename T>
struct QuadRow {
  static constexpr int D = 8, DD = 64, RP = 2, VN = Vec16<T>::N, GO = 16 / VN;   // GO: granules of a lane's two rows
  using V = typename Vec16<T>::type;
  using SIO = SlotIO<T, D>;

  // own rows (2q, 2q+1) of a contiguous block in global memory
  static __device__ __forceinline__ void load_rows_global(const T* __restrict__ blk, int q, T (&own)[RP][D]) {
    const V* p = reinterpret_cast<const V*>(blk) + q * GO;
#pragma unroll
    for (int k = 0; k < GO; ++k) {
      const V v = p[k];
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int i = 0; i < VN; ++i) own[(k * VN + i) / D][(k * VN + i) % D] = e[i];
    }
  }
  static __device__ __forceinline__ void store_rows_global(T* __restrict__ blk, int q, const T (&own)[RP][D]) {
    V* p = reinterpret_cast<V*>(blk) + q * GO;
#pragma unroll
    for (int k = 0; k < GO; ++k) {
      V v;
      T* e = reinterpret_cast<T*>(&v);
#pragma unroll
      for (int i = 0; i < VN; ++i) e[i] = own[(k * VN + i) / D][(k * VN + i) % D];
      p[k] = v;
    }
  }
  // own rows of slot u (rotated granules)
  static __device__ __forceinline__ void load_rows_slot(const T* base, int u, int q, T (&own)[RP][D]) {
    const V* b = reinterpret_cast<const V*>(base);
#pragma unroll
    for (int k = 0; k < GO; ++k) {
      const V v = b[SIO::pos(u, q * GO + k)];
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int i = 0; i < VN; ++i) own[(k * VN + i) / D][(k * VN + i) % D] = e[i];
    }
  }
  static __device__ __forceinline__ void store_rows_slot(T* base, int u, int q, const T (&own)[RP][D]) {
    V* b = reinterpret_cast<V*>(base);
#pragma unroll
    for (int k = 0; k < GO; ++k) {
      V v;
      T* e = reinterpret_cast<T*>(&v);
#pragma unroll
      for (int i = 0; i < VN; ++i) e[i] = own[(k * VN + i) / D][(k * VN + i) % D];
      b[SIO::pos(u, q * GO + k)] = v;
    }
  }
  // element pair (m, 2q), (m, 2q+1) of slot u: a lane's two COLUMNS, row by row
  static __device__ __forceinline__ T* pair_ptr(T* base, int u, int q, int m) {
    const int e = m * D + 2 * q;
    return base + (size_t)SIO::pos(u, e / VN) * VN + (e % VN);
  }
  static __device__ __forceinline__ void load_colpairs_slot(const T* base, int u, int q, T (&ownT)[RP][D]) {
#pragma unroll
    for (int m = 0; m < D; ++m) {
      const T* p = pair_ptr(const_cast<T*>(base), u, q, m);
      ownT[0][m] = p[0];
      ownT[1][m] = p[1];
    }
  }
  // block^T given by rows: element (j, 2q + t) = ownT[t][j]
  static __device__ __forceinline__ void store_colpairs_slot(T* base, int u, int q, const T (&ownT)[RP][D]) {
#pragma unroll
    for (int m = 0; m < D; ++m) {
      T* p = pair_ptr(base, u, q, m);
      p[0] = ownT[0][m];
      p[1] = ownT[1][m];
    }
  }
  static __device__ __forceinline__ void gather(const T (&own)[RP][D], T (&full)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) full[i][j] = quad_from<T>(own[i % RP][j], i / RP);
  }

  // One even row.  Dl / F / G: the lane's rows of D_k, F_k (zero unless has_odd), G_k-1 (zero unless
  // has_left).  load*(own) fetch the lane's rows / column pairs of the neighbours' Sigma blocks (called only
  // when the neighbour exists); emit_oR(rows), emit_oL(column pairs of Sigma[2k, 2k-1]), emit_See(rows).
  template <class LSdR, class LSoR, class LSoRT, class LSdL, class EoR, class EoL, class ESee>
  static __device__ __forceinline__ void even_row(int q, bool has_odd, bool has_left, const T (&Dl)[RP][D],
                                                  const T (&F)[RP][D], const T (&G)[RP][D], LSdR&& loadSdR,
                                                  LSoR&& loadSoR, LSoRT&& loadSoRT, LSdL&& loadSdL, EoR&& emit_oR,
                                                  EoL&& emit_oL, ESee&& emit_See) {
    T Di[D][D];
    {
      T L[D][D], inv[D];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i][j] = quad_from<T>(Dl[i % RP][j], i / RP);
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
    T Ak[RP][D], Bk[RP][D];                        // own rows of F D^-1, G D^-1
#pragma unroll
    for (int t = 0; t < RP; ++t)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        T a = T(0), b = T(0);
#pragma unroll
        for (int m = j; m < D; ++m) {
          a = fmaT(F[t][m], Di[m][j], a);
          b = fmaT(G[t][m], Di[m][j], b);
        }
        Ak[t][j] = a;
        Bk[t][j] = b;
      }
    T M1[RP][D], M2[RP][D];
#pragma unroll
    for (int t = 0; t < RP; ++t)
#pragma unroll
      for (int j = 0; j < D; ++j) { M1[t][j] = T(0); M2[t][j] = T(0); }
    {
      T X[D][D], S[RP][D];
      gather(Ak, X);
      if (has_odd) {
        loadSdR(S);                                // M1 = SdR A
#pragma unroll
        for (int t = 0; t < RP; ++t)
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T s = M1[t][j];
#pragma unroll
            for (int m = 0; m < D; ++m) s = fmaT(S[t][m], X[m][j], s);
            M1[t][j] = s;
          }
      }
      if (has_odd && has_left) {
        loadSoRT(S);                               // M2 = SoR^T A
#pragma unroll
        for (int t = 0; t < RP; ++t)
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T s = M2[t][j];
#pragma unroll
            for (int m = 0; m < D; ++m) s = fmaT(S[t][m], X[m][j], s);
            M2[t][j] = s;
          }
      }
      gather(Bk, X);
      if (has_odd && has_left) {
        loadSoR(S);                                // M1 += SoR B
#pragma unroll
        for (int t = 0; t < RP; ++t)
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T s = M1[t][j];
#pragma unroll
            for (int m = 0; m < D; ++m) s = fmaT(S[t][m], X[m][j], s);
            M1[t][j] = s;
          }
      }
      if (has_left) {
        loadSdL(S);                                // M2 += SdL B
#pragma unroll
        for (int t = 0; t < RP; ++t)
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T s = M2[t][j];
#pragma unroll
            for (int m = 0; m < D; ++m) s = fmaT(S[t][m], X[m][j], s);
            M2[t][j] = s;
          }
      }
    }
    {
      T N1[RP][D];
      if (has_odd) {
#pragma unroll
        for (int t = 0; t < RP; ++t)
#pragma unroll
          for (int j = 0; j < D; ++j) N1[t][j] = -M1[t][j];
        emit_oR(N1);                               // Sigma[2k+1, 2k] = -M1, by rows
      }
      if (has_left) {
#pragma unroll
        for (int t = 0; t < RP; ++t)
#pragma unroll
          for (int j = 0; j < D; ++j) N1[t][j] = -M2[t][j];
        emit_oL(N1);                               // Sigma[2k, 2k-1] = -M2^T: element (j, 2q+t) = -M2[2q+t][j]
      }
    }
    // lower(Sigma[2k, 2k]) = lower(D^-T D^-1) + sum over the quad of the lanes' outer products
    T See[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        T p = T(0);
#pragma unroll
        for (int t = 0; t < RP; ++t) p = fmaT(Ak[t][i], M1[t][j], fmaT(Bk[t][i], M2[t][j], p));
        p = quad_sum<T>(p);
#pragma unroll
        for (int m = i; m < D; ++m) p = fmaT(Di[m][i], Di[m][j], p);
        See[i][j] = p;
      }
    T own[RP][D];                                  // rows 2q, 2q+1 of the symmetric block
#pragma unroll
    for (int t = 0; t < RP; ++t)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        auto at = [&](int i) { return i >= j ? See[i][j] : See[j][i]; };
        const T v01 = (q & 1) ? at(2 + t) : at(0 + t);
        const T v23 = (q & 1) ? at(6 + t) : at(4 + t);
        own[t][j] = (q & 2) ? v23 : v01;
      }
    emit_See(own);
  }
};

// One pass: Sigma of level L+3 -> Sigma of level L; same contract as inverse_tile_kernel.
// ONE WAVE per 128-row tile (NT = 64, 16 quads): level 2's 16 even rows are one round of the wave, level 1's
// 32 two, level 0's 64 four -- every round with all lanes busy, every wave on its own (four tiles per CU, one
// per SIMD), and the next round's factor blocks (of the next level too) requested before the current round's
// algebra.  Big pass of config 3 (fp32): 1 250 us; a 256-thread workgroup per tile 1 698 us (three of its four
// waves idle at level 2, two at level 1); two waves per tile at 256 registers 2 400 us (344 bytes of scratch per
// lane); one lane per row (inverse_tile_lds_kernel) 1 556 us.
template <typename T, int NT>
__global__ __launch_bounds__(NT, 1) void inverse_tile_quad_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, InverseLevels lv,
    const T* __restrict__ Sd_in, const T* __restrict__ So_in, int64_t n, T* __restrict__ Sd_out, T* __restrict__ So_out) {
  using QR = QuadRow<T>;
  using SIO = SlotIO<T, 8>;
  using V = typename Vec16<T>::type;
  constexpr int D = 8, DD = 64, RP = 2, VN = Vec16<T>::N, GR = DD / VN, NQ = NT / 4;
  constexpr bool AHEAD = CGPS_INVQ_AHEAD;            // a second set of factor registers: the next round's blocks are on their way
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sd = reinterpret_cast<T*>(smem);                 // [64][DD]  Sigma[row, row]
  T* so = sd + 64 * DD;                               // [64][DD]  Sigma[row, previous row of the current level]
  T* halo = so + 64 * DD;                             // Sigma[row, row] of the previous tile's last row
  T* xpose = halo + DD;                               // [NQ quads][DD]: level 0 turns Sigma[2k, 2k-1] from column pairs into rows
  const int tid = threadIdx.x, q = tid & 3, Q = tid >> 2;
  const int64_t ntiles = (n + INV_TS - 1) / INV_TS;

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * INV_TS;
    const int n0 = (int)((n - row0) < INV_TS ? (n - row0) : INV_TS);
    const int64_t g3 = row0 >> INV_LP;
    const int n3 = n0 >> INV_LP;
    // the lane's rows of the factor blocks of elimination kq of relative level t (identity / zeros where a block does not exist)
    auto request = [&](int t, int kq, T (&Dl)[RP][D], T (&F)[RP][D], T (&G)[RP][D]) {
      const int M = n0 >> t, m = 2 * kq;
      const bool even = m < M, has_odd = even && (m + 1 < M);
      const int64_t kg = (row0 >> (t + 1)) + kq;
      const bool has_left = even && kg >= 1;
#pragma unroll
      for (int a = 0; a < RP; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) { Dl[a][b] = (2 * q + a == b) ? T(1) : T(0); F[a][b] = T(0); G[a][b] = T(0); }
      if (even) QR::load_rows_global(Dp + (lv.offD[t] + kg) * DD, q, Dl);
      if (has_odd) QR::load_rows_global(Fp + (lv.offF[t] + kg) * DD, q, F);
      if (has_left) QR::load_rows_global(Gp + (lv.offG[t] + kg - 1) * DD, q, G);
    };
    T Dp_[RP][D], Fp_[RP][D], Gp_[RP][D];               // AHEAD: the current round's blocks, requested one round earlier
    if constexpr (AHEAD) request(INV_LP - 1, Q, Dp_, Fp_, Gp_);     // on their way while the input level is staged
    {                                                   // input level: row m -> slot 4 (m + 1) - 1
      const int first_o = (g3 >= 1) ? 0 : 1;
      const V* gd = reinterpret_cast<const V*>(Sd_in + g3 * DD);
      const V* go = reinterpret_cast<const V*>(So_in + (g3 - 1) * DD);
      V* sdv = reinterpret_cast<V*>(sd);
      V* sov = reinterpret_cast<V*>(so);
      for (int v = tid; v < n3 * GR; v += NT) {
        const int m = v / GR, g = v % GR;
        sdv[SIO::pos(4 * (m + 1) - 1, g)] = gd[v];
        if (m >= first_o) sov[SIO::pos(4 * (m + 1) - 1, g)] = go[v];
      }
      if (g3 >= 1 && tid < GR) reinterpret_cast<V*>(halo)[tid] = (gd - GR)[tid];
    }
    __syncthreads();

    // ---- relative levels 2, 1, 0 as one sequence of rounds: the level's even row 2 kq goes to quad kq % NQ in
    // round kq / NQ; a barrier where the level changes ------------------------------------------------------------
    int t = INV_LP - 1, k0 = 0;
#pragma unroll 1
    while (t >= 0) {
      const int M = n0 >> t;                            // rows of the level in this tile
      const int E = (M + 1) >> 1;                       // its even rows
      const int st = t >= 1 ? (1 << (t - 1)) : 0;
      const int kq = k0 + Q, m = 2 * kq;
      const bool even = m < M, has_odd = even && (m + 1 < M);
      const int64_t kg = (row0 >> (t + 1)) + kq;
      const bool has_left = even && kg >= 1;
      const bool level_ends = k0 + NQ >= E;
      const int tn = level_ends ? t - 1 : t, kn = level_ends ? 0 : k0 + NQ;     // the next round
      T Dl[RP][D], F[RP][D], G[RP][D];
      if constexpr (AHEAD) {
#pragma unroll
        for (int a = 0; a < RP; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) { Dl[a][b] = Dp_[a][b]; F[a][b] = Fp_[a][b]; G[a][b] = Gp_[a][b]; }
        if (tn >= 0) request(tn, kn + Q, Dp_, Fp_, Gp_);
      } else {
        if (even) request(t, kq, Dl, F, G);
      }
      if (even) {
        if (t >= 1) {
          const int u = (m + 1) * st - 1;               // the row's slot
          QR::even_row(
              q, has_odd, has_left, Dl, F, G,
              [&](T (&A)[RP][D]) { QR::load_rows_slot(sd, u + st, q, A); },
              [&](T (&A)[RP][D]) { QR::load_rows_slot(so, u + st, q, A); },
              [&](T (&A)[RP][D]) { QR::load_colpairs_slot(so, u + st, q, A); },
              [&](T (&A)[RP][D]) {
                if (m == 0) QR::load_rows_global(halo, q, A);               // (LDS, plain layout) the previous tile's last row
                else QR::load_rows_slot(sd, u - st, q, A);
              },
              [&](const T (&A)[RP][D]) { QR::store_rows_slot(so, u + st, q, A); },   // the right neighbour's previous row is now this one
              [&](const T (&A)[RP][D]) { QR::store_colpairs_slot(so, u, q, A); },
              [&](const T (&A)[RP][D]) { QR::store_rows_slot(sd, u, q, A); });
        } else {
          // level 0: row 2 kq of the tile; its odd right neighbour is slot kq.  Results go straight to global memory.
          const int64_t r = row0 + m;
          T* xq = xpose + (size_t)Q * DD;
          QR::even_row(
              q, has_odd, has_left, Dl, F, G,
              [&](T (&A)[RP][D]) { QR::load_rows_slot(sd, kq, q, A); },
              [&](T (&A)[RP][D]) { QR::load_rows_slot(so, kq, q, A); },
              [&](T (&A)[RP][D]) { QR::load_colpairs_slot(so, kq, q, A); },
              [&](T (&A)[RP][D]) {
                if (kq == 0) QR::load_rows_global(halo, q, A);
                else QR::load_rows_slot(sd, kq - 1, q, A);
              },
              [&](const T (&A)[RP][D]) { QR::store_rows_global(So_out + r * DD, q, A); },          // Sigma[2k+1, 2k]
              [&](const T (&A)[RP][D]) {                                                          // Sigma[2k, 2k-1]
                // column pairs -> the quad's transposing block -> rows (the four lanes are one wave: in order)
#pragma unroll
                for (int mm = 0; mm < D; ++mm) {
                  xq[mm * D + 2 * q] = A[0][mm];
                  xq[mm * D + 2 * q + 1] = A[1][mm];
                }
                __builtin_amdgcn_wave_barrier();
                T rows[RP][D];
                QR::load_rows_global(xq, q, rows);
                __builtin_amdgcn_wave_barrier();
                QR::store_rows_global(So_out + (r - 1) * DD, q, rows);
              },
              [&](const T (&A)[RP][D]) { QR::store_rows_global(Sd_out + r * DD, q, A); });
        }
      }
      if (level_ends) __syncthreads();                  // the level's slots are written
      t = tn;
      k0 = kn;
    }
    {
      // Sigma[2u+1, 2u+1] = slot u, unchanged: consecutive lanes copy consecutive 16-byte granules
      const int nodd = n0 >> 1;
      const V* sdv = reinterpret_cast<const V*>(sd);
      V* od = reinterpret_cast<V*>(Sd_out + row0 * DD);
      for (int v = tid; v < nodd * GR; v += NT) {
        const int u = v / GR, g = v % GR;
        od[(size_t)(2 * u + 1) * GR + g] = sdv[SIO::pos(u, g)];
      }
      __syncthreads();
    }
  }
}

// The same pass with a 256-thread workgroup per tile (64 quads: one round per level, three of the four
// waves idle at level 2 and two at level 1).  For fp64: 64 KB of slots per tile allow two tiles per CU at
// most, a wave per tile would leave half of the SIMDs without work, and the second register set of the
// look-ahead does not fit (fp64 d = 8 at 2^21 rows: 2.9 ms against 4.2 ms in the kernel above with
// NT = 256, 4.9 ms level by level).
template <typename T>
__global__ __launch_bounds__(256, 1) void inverse_tile_quad_wg_kernel(
    const T* __restrict__ Dp, const T* __restrict__ Fp, const T* __restrict__ Gp, InverseLevels lv,
    const T* __restrict__ Sd_in, const T* __restrict__ So_in, int64_t n, T* __restrict__ Sd_out, T* __restrict__ So_out) {
  using QR = QuadRow<T>;
  using SIO = SlotIO<T, 8>;
  using V = typename Vec16<T>::type;
  constexpr int D = 8, DD = 64, RP = 2, VN = Vec16<T>::N, GR = DD / VN, NT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sd = reinterpret_cast<T*>(smem);
  T* so = sd + 64 * DD;
  T* halo = so + 64 * DD;
  T* xpose = halo + DD;
  const int tid = threadIdx.x, q = tid & 3, Q = tid >> 2;
  const int64_t ntiles = (n + INV_TS - 1) / INV_TS;

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * INV_TS;
    const int n0 = (int)((n - row0) < INV_TS ? (n - row0) : INV_TS);
    const int64_t g3 = row0 >> INV_LP;
    const int n3 = n0 >> INV_LP;
    {                                                   // input level: row m -> slot 4 (m + 1) - 1
      const int first_o = (g3 >= 1) ? 0 : 1;
      const V* gd = reinterpret_cast<const V*>(Sd_in + g3 * DD);
      const V* go = reinterpret_cast<const V*>(So_in + (g3 - 1) * DD);
      V* sdv = reinterpret_cast<V*>(sd);
      V* sov = reinterpret_cast<V*>(so);
      for (int v = tid; v < n3 * GR; v += NT) {
        const int m = v / GR, g = v % GR;
        sdv[SIO::pos(4 * (m + 1) - 1, g)] = gd[v];
        if (m >= first_o) sov[SIO::pos(4 * (m + 1) - 1, g)] = go[v];
      }
      if (g3 >= 1 && tid < GR) reinterpret_cast<V*>(halo)[tid] = (gd - GR)[tid];
    }
    __syncthreads();

    // ---- relative levels 2 and 1: quad Q takes the level's even row 2Q ---------------------------------
#pragma unroll 1
    for (int t = INV_LP - 1; t >= 1; --t) {
      const int st = 1 << (t - 1), M = n0 >> t, m = 2 * Q;
      const bool even = m < M, has_odd = even && (m + 1 < M);
      const int64_t kg = (row0 >> (t + 1)) + Q;
      const bool has_left = even && kg >= 1;
      const int u = (m + 1) * st - 1;                   // the row's slot
      if (even) {
        T Dl[RP][D], F[RP][D], G[RP][D];
#pragma unroll
        for (int a = 0; a < RP; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) { F[a][b] = T(0); G[a][b] = T(0); }
        QR::load_rows_global(Dp + (lv.offD[t] + kg) * DD, q, Dl);
        if (has_odd) QR::load_rows_global(Fp + (lv.offF[t] + kg) * DD, q, F);
        if (has_left) QR::load_rows_global(Gp + (lv.offG[t] + kg - 1) * DD, q, G);
        QR::even_row(
            q, has_odd, has_left, Dl, F, G,
            [&](T (&A)[RP][D]) { QR::load_rows_slot(sd, u + st, q, A); },
            [&](T (&A)[RP][D]) { QR::load_rows_slot(so, u + st, q, A); },
            [&](T (&A)[RP][D]) { QR::load_colpairs_slot(so, u + st, q, A); },
            [&](T (&A)[RP][D]) {
              if (m == 0) QR::load_rows_global(halo, q, A);               // (LDS, plain layout) the previous tile's last row
              else QR::load_rows_slot(sd, u - st, q, A);
            },
            [&](const T (&A)[RP][D]) { QR::store_rows_slot(so, u + st, q, A); },   // the right neighbour's previous row is now this one
            [&](const T (&A)[RP][D]) { QR::store_colpairs_slot(so, u, q, A); },
            [&](const T (&A)[RP][D]) { QR::store_rows_slot(sd, u, q, A); });
      }
      __syncthreads();
    }

    // ---- relative level 0: quad Q takes row 2Q; row 2Q+1 is slot Q ---------------------------------------
    {
      const bool even = 2 * Q < n0, has_odd = 2 * Q + 1 < n0;
      const int64_t kg = (row0 >> 1) + Q;
      const bool has_left = even && kg >= 1;
      const int64_t r = row0 + 2 * Q;
      if (even) {
        T Dl[RP][D], F[RP][D], G[RP][D];
#pragma unroll
        for (int a = 0; a < RP; ++a)
#pragma unroll
          for (int b = 0; b < D; ++b) { F[a][b] = T(0); G[a][b] = T(0); }
        QR::load_rows_global(Dp + (lv.offD[0] + kg) * DD, q, Dl);
        if (has_odd) QR::load_rows_global(Fp + (lv.offF[0] + kg) * DD, q, F);
        if (has_left) QR::load_rows_global(Gp + (lv.offG[0] + kg - 1) * DD, q, G);
        T* xq = xpose + (size_t)Q * DD;
        QR::even_row(
            q, has_odd, has_left, Dl, F, G,
            [&](T (&A)[RP][D]) { QR::load_rows_slot(sd, Q, q, A); },
            [&](T (&A)[RP][D]) { QR::load_rows_slot(so, Q, q, A); },
            [&](T (&A)[RP][D]) { QR::load_colpairs_slot(so, Q, q, A); },
            [&](T (&A)[RP][D]) {
              if (Q == 0) QR::load_rows_global(halo, q, A);
              else QR::load_rows_slot(sd, Q - 1, q, A);
            },
            [&](const T (&A)[RP][D]) { QR::store_rows_global(So_out + r * DD, q, A); },          // Sigma[2k+1, 2k]
            [&](const T (&A)[RP][D]) {                                                          // Sigma[2k, 2k-1]
              // column pairs -> the quad's transposing block -> rows (the four lanes are one wave: in order)
#pragma unroll
              for (int mm = 0; mm < D; ++mm) {
                xq[mm * D + 2 * q] = A[0][mm];
                xq[mm * D + 2 * q + 1] = A[1][mm];
              }
              __builtin_amdgcn_wave_barrier();
              T rows[RP][D];
              QR::load_rows_global(xq, q, rows);
              __builtin_amdgcn_wave_barrier();
              QR::store_rows_global(So_out + (r - 1) * DD, q, rows);
            },
            [&](const T (&A)[RP][D]) { QR::store_rows_global(Sd_out + r * DD, q, A); });
      }
      // Sigma[2u+1, 2u+1] = slot u, unchanged: consecutive lanes copy consecutive 16-byte granules
      const int nodd = n0 >> 1;
      const V* sdv = reinterpret_cast<const V*>(sd);
      V* od = reinterpret_cast<V*>(Sd_out + row0 * DD);
      for (int v = tid; v < nodd * GR; v += NT) {
        const int u = v / GR, g = v % GR;
        od[(size_t)(2 * u + 1) * GR + g] = sdv[SIO::pos(u, g)];
      }
      __syncthreads();
    }
  }
}

}  // namespace cgps
