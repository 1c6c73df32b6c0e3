// One level of tile_cr (cgps_tile.h) for 8 x 8 blocks with FOUR LANES PER ELIMINATION.
// Included from the middle of cgps_tile.h (inside namespace cgps): uses LdsTile, PivotLog, Chol.
//
// The role-split level gives an elimination to four lanes in four different waves, each doing one role's
// share of whole 8 x 8 matrices (~680 registers wanted: spills) with a barrier between reading and writing:
// 4.5 us per narrow level at fp32 (0.85 us for 4 x 4 fp64 on the matrix cores).  Here a DPP quad takes the
// elimination, lane q owning matrix rows 2q and 2q+1 of every block -- the arithmetic of the streaming
// loop of chunk_reduce_ml_kernel: the 8 x 8 Cholesky and x = D^-1 y redundantly on the four lanes, the
// lane's rows of G = Oc[l]^T D^-T and F = Oc[e] D^-T, one quad gather of G (parked G G^T and G x, new
// coupling -F G^T) and one of F (right neighbour R_o -= F F^T, y_o -= F x).  Eliminations of a level
// touch disjoint slots, a quad reads before it writes: ONE barrier per level.
template <typename T, int D>
struct QuadTile {
  static_assert(D == 4 || D == 8, "a quad shares blocks whose rows split evenly over four lanes");
  static constexpr int DD = D * D, RP = D / 4;             // RP: matrix rows per lane
  using LT = LdsTile<T, D>;
  using V = typename Vec16<T>::type;
  static constexpr int VN = LT::VN, GO = RP * D / VN;      // granules of a lane's rows
  static_assert(LT::SWZ && (RP * D) % VN == 0, "blocks stored with rotated granules, a lane's rows in whole granules");

  static __device__ __forceinline__ void load_rows(const T* base, int slot, int q, T (&own)[RP][D]) {
    const V* b = reinterpret_cast<const V*>(base + (size_t)slot * DD);
    const int kk = LT::key(slot);
#pragma unroll
    for (int k = 0; k < GO; ++k) {
      const V v = b[(q * GO + k) ^ kk];
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int i = 0; i < VN; ++i) own[(k * VN + i) / D][(k * VN + i) % D] = e[i];
    }
  }
  static __device__ __forceinline__ void store_rows(T* base, int slot, int q, const T (&own)[RP][D]) {
    V* b = reinterpret_cast<V*>(base + (size_t)slot * DD);
    const int kk = LT::key(slot);
#pragma unroll
    for (int k = 0; k < GO; ++k) {
      V v;
      T* e = reinterpret_cast<T*>(&v);
#pragma unroll
      for (int i = 0; i < VN; ++i) e[i] = own[(k * VN + i) / D][(k * VN + i) % D];
      b[(q * GO + k) ^ kk] = v;
    }
  }
  // the lane's COLUMNS of a block: ownT[t][m] = block[m][RP q + t]
  static __device__ __forceinline__ void load_colpairs(const T* base, int slot, int q, T (&ownT)[RP][D]) {
    const T* b = base + (size_t)slot * DD;
    const int kk = LT::key(slot);
#pragma unroll
    for (int m = 0; m < D; ++m) {
      const int idx = m * D + RP * q;                 // RP consecutive elements, inside one granule (RP divides VN or RP == VN)
      const T* p = b + (((idx / VN) ^ kk) * VN + (idx % VN));
#pragma unroll
      for (int t = 0; t < RP; ++t) ownT[t][m] = p[t];
    }
  }
  // the lane's rows of a contiguous block in global memory (the quad writes the block together)
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
  static __device__ __forceinline__ void gather(const T (&own)[RP][D], T (&full)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) full[i][j] = quad_from<T>(own[i % RP][j], i / RP);
  }
};

// tid: threadIdx.x as an OPAQUE value (see tile_cr): everything addressed through it is computed here, after
// the streaming loop of the calling kernel, instead of being hoisted above that loop and kept alive through it
// (two more 8-byte spills per streamed row in chunk_reduce_ml_kernel<float, 8>: +6 % on config 3).
template <typename T, int D, int NTHR>
__device__ __forceinline__ void tile_cr_level_quad(LdsTile<T, D>& t, int tid, int K, int M, int s, PivotLog& pl,
                                                   double& mah, bool& fail) {
  using QT = QuadTile<T, D>;
  constexpr int RP = QT::RP, NQ = NTHR / 4;
  const int q = tid & 3, Q = tid >> 2, h = s >> 1;
  const int n_elim = (M + 1) / 2;
#pragma unroll 1
  for (int k0 = 0; k0 < n_elim; k0 += NQ) {
    const int k = k0 + Q;
    const int e = (2 * k + 1) * s - 1;
    const bool act = (2 * k < M) && (e != K);
    if (act) {
      const int o = (2 * k + 1 < M) ? e + s : K;
      // the eliminated row (less what is parked for it), factored on all four lanes
      T A[RP][D], ye[RP];
      QT::load_rows(t.R, e, q, A);
#pragma unroll
      for (int a = 0; a < RP; ++a) ye[a] = t.y[e * D + RP * q + a];
      if ((s > 1) && (e + h < K)) {
        T P[RP][D];
        QT::load_rows(t.R, e + h, q, P);
#pragma unroll
        for (int a = 0; a < RP; ++a) {
          ye[a] -= t.y[(e + h) * D + RP * q + a];
#pragma unroll
          for (int b = 0; b < D; ++b) A[a][b] -= P[a][b];
        }
      }
      Chol<T, D> c;
      T x[D];
      {
        T Af[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) Af[i][j] = quad_from<T>(A[i % RP][j], i / RP);
        bool f = false;
        const double piv = chol_lower<T, D>(Af, c, f);
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = quad_from<T>(ye[i % RP], i / RP);
        fwd_subst<T, D>(c, x);
        if (q == 0) {
          pl.mul(piv);
          fail = fail || f;
#pragma unroll
          for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
        }
      }
      // the lane's rows of G = Oc[l]^T D^-T and F = Oc[e] D^-T
      T G[RP][D], F[RP][D];
      QT::load_colpairs(t.Oc, e - s + 1, q, G);
      QT::load_rows(t.Oc, e + 1, q, F);
#pragma unroll
      for (int a = 0; a < RP; ++a) {
        fwd_subst<T, D>(c, G[a]);
        fwd_subst<T, D>(c, F[a]);
      }
      T W[RP][D], wv[RP], Cn[RP][D];                 // parked G G^T, G x; the new coupling -F G^T
      {
        T Gf[D][D];
        QT::gather(G, Gf);
#pragma unroll
        for (int a = 0; a < RP; ++a) {
          T sv = T(0);
#pragma unroll
          for (int m = 0; m < D; ++m) sv = fmaT(G[a][m], x[m], sv);
          wv[a] = sv;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T sw = T(0), sc = T(0);
#pragma unroll
            for (int m = 0; m < D; ++m) {
              sw = fmaT(G[a][m], Gf[j][m], sw);
              sc = fmaT(-F[a][m], Gf[j][m], sc);
            }
            W[a][j] = sw;
            Cn[a][j] = sc;
          }
        }
      }
      // right neighbour: R_o -= F F^T, y_o -= F x
      T Ro[RP][D], yo[RP];
      QT::load_rows(t.R, o, q, Ro);
#pragma unroll
      for (int a = 0; a < RP; ++a) yo[a] = t.y[o * D + RP * q + a];
      if ((s > 1) && (o + h < K)) {
        T P[RP][D];
        QT::load_rows(t.R, o + h, q, P);
#pragma unroll
        for (int a = 0; a < RP; ++a) {
          yo[a] -= t.y[(o + h) * D + RP * q + a];
#pragma unroll
          for (int b = 0; b < D; ++b) Ro[a][b] -= P[a][b];
        }
      }
      {
        T Ff[D][D];
        QT::gather(F, Ff);
#pragma unroll
        for (int a = 0; a < RP; ++a) {
          T sv = yo[a];
#pragma unroll
          for (int m = 0; m < D; ++m) sv = fmaT(-F[a][m], x[m], sv);
          yo[a] = sv;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            T sr = Ro[a][j];
#pragma unroll
            for (int m = 0; m < D; ++m) sr = fmaT(-F[a][m], Ff[j][m], sr);
            Ro[a][j] = sr;
          }
        }
      }
      // everything this elimination reads has been read (the quad runs in lockstep): write
      QT::store_rows(t.R, e, q, W);
      QT::store_rows(t.Oc, e - s + 1, q, Cn);
      QT::store_rows(t.R, o, q, Ro);
#pragma unroll
      for (int a = 0; a < RP; ++a) {
        t.y[e * D + RP * q + a] = wv[a];
        t.y[o * D + RP * q + a] = yo[a];
      }
    }
  }
  __syncthreads();                                  // the level's results are visible
}
