// Streaming stage of the fused solve + log-det for LARGE blocks (d = 8): one block row is
// spread over LPR = 4 adjacent lanes, each owning RP = d / LPR matrix rows of every block.
//
// Included from the middle of cgps_tile.h (inside namespace cgps): it uses LdsTile, StageSmem,
// tile_cr, collect_left_updates, write_partial and RecordLayout defined there.
//
// Why: one lane per 8 x 8 block row needs ~680 registers (spills, one wave per SIMD, which then
// issues at half rate).  Sliced over four lanes the state is ~250 registers, two waves per SIMD
// fit, and a group's four lanes load one contiguous 256-byte block together.
//
// Per eliminated row (all four lanes in lockstep, same maths as eliminate_forward):
//   the 8 x 8 Cholesky and x = D^-1 y are computed redundantly by every lane (about 1/5 of the
//   instructions, no exchange);
//   lane q solves its RP rows of G = Cc^T D^-T and F = On D^-T (it holds the matching RP rows of
//   Cc^T and On);
//   one all-gather of G and one of F inside the group (wave shuffles), then lane q updates its RP
//   rows of dRa -= G G^T, Rn -= F F^T, Cc'^T = -G F^T and its RP entries of dya, yn.
// The group's kept row is written to the LDS tile in the ordinary layout, so the per-workgroup
// cyclic reduction (tile_cr) and the record stage are the single-lane ones (1/16 of the rows).
template <typename T, int D, int LPR>
struct MlGroup {
  static constexpr int RP = D / LPR;
  static_assert(D % LPR == 0 && (LPR & (LPR - 1)) == 0, "block size must split evenly over a power-of-two group");
  // value of `v` held by lane `src_q` of this lane's group
  // (src_q is a compile-time constant at every call site once the loops are unrolled.  A group of
  // four lanes is a DPP quad: quad_perm broadcasts inside it on the VALU, no LDS round trip.)
  template <int Q>
  static __device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xf, 0xf, true));
  }
  static __device__ __forceinline__ T from(T v, int src_q) {
    if constexpr (LPR == 4 && sizeof(T) == 4) {
      switch (src_q) {
        case 0: return quad_bcast<0>(v);
        case 1: return quad_bcast<1>(v);
        case 2: return quad_bcast<2>(v);
        default: return quad_bcast<3>(v);
      }
    } else {
      const int lane = threadIdx.x & 63;
      return __shfl(v, (lane & ~(LPR - 1)) | src_q, 64);
    }
  }
  // all D rows of a row-sliced matrix: full[i][j] = own[i % RP][j] of lane i / RP
  static __device__ __forceinline__ void gather(const T (&own)[RP][D], T (&full)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) full[i][j] = from(own[i % RP][j], i / RP);
  }
  static __device__ __forceinline__ void gather_lower(const T (&own)[RP][D], T (&full)[D][D]) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) full[i][j] = from(own[i % RP][j], i / RP);
  }
  static __device__ __forceinline__ void gather_vec(const T (&own)[RP], T (&full)[D]) {
#pragma unroll
    for (int i = 0; i < D; ++i) full[i] = from(own[i % RP], i / RP);
  }
};

// (fp32: two workgroups per CU; fp64 blocks this large take the whole register file)
template <typename T, int D, int C, int NT, int LPR, bool FOLD = false>
__global__ __launch_bounds__(NT, (sizeof(T) == 4 ? 2 : 1)) void chunk_reduce_ml_kernel(const T* __restrict__ Rg, const T* __restrict__ Og,
                                                                const T* __restrict__ yg, int64_t N,
                                                                const T* __restrict__ Oleft, T* __restrict__ rec,
                                                                double* __restrict__ partial, FoldArgs fold) {
  using MG = MlGroup<T, D, LPR>;
  using LT = LdsTile<T, D>;
  using RL = RecordLayout<T, D>;
  constexpr int DD = D * D, RP = D / LPR, NG = NT / LPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  StageSmem<T, D, NG, NT> sm(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid % LPR, grp = tid / LPR;
  if (tid == 0) *sm.sfail = 0x7fffffff;
  const int64_t grp0 = (int64_t)blockIdx.x * NG;
  const int64_t r0 = (grp0 + grp) * C;
  const int row_lo = q * RP;                        // first matrix row this lane owns
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;

  T Rown[RP][D], yown[RP], Ct[RP][D], dRa[RP][D], dya[RP];
#pragma unroll
  for (int t = 0; t < RP; ++t) {
    yown[t] = T(0);
    dya[t] = T(0);
#pragma unroll
    for (int j = 0; j < D; ++j) { Rown[t][j] = T(0); Ct[t][j] = T(0); dRa[t][j] = T(0); }
  }
  if (r0 < N) {
#pragma unroll
    for (int t = 0; t < RP; ++t) {
      yown[t] = yg[r0 * D + row_lo + t];
#pragma unroll
      for (int j = 0; j < D; ++j) Rown[t][j] = Rg[r0 * DD + (row_lo + t) * D + j];
    }
    const T* Cl = (r0 >= 1) ? Og + (r0 - 1) * DD : Oleft;       // J[row r0, row r0-1]
    if (Cl != nullptr) {
#pragma unroll
      for (int t = 0; t < RP; ++t)
#pragma unroll
        for (int j = 0; j < D; ++j) Ct[t][j] = Cl[j * D + row_lo + t];    // rows of Cc^T = columns of Cc
    }
  }
#pragma unroll 1
  for (int step = 0; step < C - 1; ++step) {
    const int64_t rn = r0 + step + 1;
    if (rn >= N) break;
    T Rn[RP][D], On[RP][D], yn[RP];
#pragma unroll
    for (int t = 0; t < RP; ++t) {
      yn[t] = yg[rn * D + row_lo + t];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        Rn[t][j] = Rg[rn * DD + (row_lo + t) * D + j];
        On[t][j] = Og[(rn - 1) * DD + (row_lo + t) * D + j];
      }
    }
    // redundant Cholesky and x
    T x[D];
    Chol<T, D> c;
    {
      T Rf[D][D];
      MG::gather_lower(Rown, Rf);
      bool f = false;
      const double piv = chol_lower<T, D>(Rf, c, f);
      if (q == 0) { pl.mul(piv); fail = fail || f; }
      MG::gather_vec(yown, x);
      fwd_subst<T, D>(c, x);
      if (q == 0) {                        // one row's sum of squares in the block's precision, then into the fp64 sum
        T sq = T(0);
#pragma unroll
        for (int i = 0; i < D; ++i) sq = fmaT(x[i], x[i], sq);
        mah += (double)sq;
      }
    }
    // own rows of G = Cc^T D^-T, then dRa -= G G^T, dya -= G x
#pragma unroll
    for (int t = 0; t < RP; ++t) fwd_subst<T, D>(c, Ct[t]);          // Ct now holds this lane's rows of G
    {
      T Gall[D][D];
      MG::gather(Ct, Gall);
#pragma unroll
      for (int t = 0; t < RP; ++t) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          T s = dRa[t][j];
#pragma unroll
          for (int m = 0; m < D; ++m) s = fmaT(-Ct[t][m], Gall[j][m], s);
          dRa[t][j] = s;
        }
        T s = dya[t];
#pragma unroll
        for (int m = 0; m < D; ++m) s = fmaT(-Ct[t][m], x[m], s);
        dya[t] = s;
      }
    }
    // own rows of F = On D^-T, then Rn -= F F^T, yn -= F x, Cc'^T = -G F^T
#pragma unroll
    for (int t = 0; t < RP; ++t) fwd_subst<T, D>(c, On[t]);          // On now holds this lane's rows of F
    {
      T Fall[D][D];
      MG::gather(On, Fall);
#pragma unroll
      for (int t = 0; t < RP; ++t) {
        T cn[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
          T s = Rn[t][j], u = T(0);
#pragma unroll
          for (int m = 0; m < D; ++m) {
            s = fmaT(-On[t][m], Fall[j][m], s);
            u = fmaT(-Ct[t][m], Fall[j][m], u);
          }
          Rn[t][j] = s;
          cn[j] = u;
        }
        T s = yn[t];
#pragma unroll
        for (int m = 0; m < D; ++m) s = fmaT(-On[t][m], x[m], s);
        yown[t] = s;
#pragma unroll
        for (int j = 0; j < D; ++j) { Rown[t][j] = Rn[t][j]; Ct[t][j] = cn[j]; }
      }
    }
  }

  // ---- the group's kept row -> LDS tile slot `grp` (ordinary layout) -------------------------------
  int64_t nreal64 = (N + C - 1) / C - grp0;              // groups of this tile that hold real rows
  const int n_real = nreal64 > NG ? NG : (int)nreal64;
  // what group g+1 owes group g's kept row: same lane position, LPR lanes further (LDS across waves)
  {
    T* xw = sm.xch;                                      // [NT/64][DD + D]
    if (lane < LPR && wave > 0) {
      T* p = xw + (wave - 1) * (DD + D);
#pragma unroll
      for (int t = 0; t < RP; ++t) {
#pragma unroll
        for (int j = 0; j < D; ++j) p[(row_lo + t) * D + j] = dRa[t][j];
        p[DD + row_lo + t] = dya[t];
      }
    }
    __syncthreads();
    T nR[RP][D], ny[RP];
#pragma unroll
    for (int t = 0; t < RP; ++t) {
      ny[t] = __shfl_down(dya[t], LPR, 64);
#pragma unroll
      for (int j = 0; j < D; ++j) nR[t][j] = __shfl_down(dRa[t][j], LPR, 64);
    }
    if (lane >= 64 - LPR && wave < NT / 64 - 1) {
      const T* p = xw + wave * (DD + D);
#pragma unroll
      for (int t = 0; t < RP; ++t) {
#pragma unroll
        for (int j = 0; j < D; ++j) nR[t][j] = p[(row_lo + t) * D + j];
        ny[t] = p[DD + row_lo + t];
      }
    }
    if (grp < n_real - 1) {
#pragma unroll
      for (int t = 0; t < RP; ++t) {
        yown[t] += ny[t];
#pragma unroll
        for (int j = 0; j < D; ++j) Rown[t][j] += nR[t][j];
      }
    }
    __syncthreads();                                     // xch is reused below
  }
  if (grp < n_real) {
    T* Rp = sm.t.R + (size_t)grp * DD;
    T* Op = sm.t.Oc + (size_t)grp * DD;
    const int kk = LT::SWZ ? LT::key(grp) : 0;
#pragma unroll
    for (int t = 0; t < RP; ++t) {
      sm.t.y[grp * D + row_lo + t] = yown[t];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int idx = (row_lo + t) * D + j;            // R[i][j], i = row_lo + t
        const int idc = j * D + row_lo + t;              // Cc[j][i] = Ct[t][j]
        if constexpr (LT::SWZ) {
          Rp[((idx / LT::VN) ^ kk) * LT::VN + (idx % LT::VN)] = Rown[t][j];
          Op[((idc / LT::VN) ^ kk) * LT::VN + (idc % LT::VN)] = Ct[t][j];
        } else {
          Rp[idx] = Rown[t][j];
          Op[idc] = Ct[t][j];
        }
      }
    }
  }
  // group 0's share for the row left of the tile waits in LDS (thread 0 needs all of it later)
  if (grp == 0) {
#pragma unroll
    for (int t = 0; t < RP; ++t) {
#pragma unroll
      for (int j = 0; j < D; ++j) sm.xch[(row_lo + t) * D + j] = dRa[t][j];
      sm.xch[DD + row_lo + t] = dya[t];
    }
  }
  const bool fail_stream = fail;                 // (what fails later may be a consequence of this: fail_code)
  // barrier, cyclic reduction of the tile's kept rows, the record (D * D lanes, write-through stores)
  reduce_staged_tile_and_emit<T, D, NT>(sm.t, n_real, sm.xch, rec + (size_t)blockIdx.x * RL::STRIDE, pl, mah, fail);
  int64_t frow = r0 < N ? r0 : N - 1;
  write_partial<NT>(mah, pl.value(), fail_code(fail_stream, fail, frow), partial + PARTIAL_STRIDE * (size_t)blockIdx.x, sm.red, sm.sfail);
  if constexpr (FOLD)
    fold_record_stages<T, D, NT, false>(smem, sm.sfail + 1, rec, partial, fold, (int64_t)C * NG, N);
}
