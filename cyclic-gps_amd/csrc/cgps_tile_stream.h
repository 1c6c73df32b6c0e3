// Fused solve + log-det for systems of MORE than one round of the chip (N > 2^20 rows of 4 x 4 fp64
// blocks; every shard of BASELINE config 4): ONE persistent launch, one workgroup per CU, whose waves
// are SPECIALISED -- waves 0..3 stream, waves 4..7 reduce -- so that the in-LDS reduction of a tile
// runs underneath the streaming of the next one.  Included at the end of cgps_tile.h (namespace cgps).
//
// chunk_reduce_kernel (cgps_tile.h) gives every workgroup ONE tile of 256 lanes x C rows: stream, then
// reduce the 256 kept rows in LDS (HBM idle), then the record stages.  With more tiles than the chip
// holds workgroups, the grid runs in rounds at two workgroups per CU whose phases line up: measured
// 5.7 TB/s of streaming at 2^21..2^24 rows against 6.6-7.7 TB/s for the one-round grid of 2^20 rows, and
// every round pays its own in-LDS reduction with the memory system idle.  Here workgroup w owns the
// CONSECUTIVE rows [w RW, (w+1) RW) and walks them as T tiles:
//
//   streamers (waves 0..3, 256 lanes): tile k exactly as chunk_reduce_kernel streams it (C = 16 rows per
//     lane, 16-byte loads, the right-hand-side line staged in the lane's 128 bytes of the tile buffer it
//     will fill), kept rows -> LDS buffer k & 1, ONE workgroup barrier, on to tile k + 1;
//   reducers (waves 4..7): after that barrier wave 4 + q reduces rows [64 q, 64 q + 64) of buffer k & 1 as a
//     system of its own, four lanes per elimination (cgps_tile_quad.h), WAVE-LOCAL: no workgroup barrier
//     (the streamers are in the middle of their loop), a wave's LDS instructions execute in order.  The
//     four arrive on a counter in LDS; lane 0 of wave 4 then folds the four boundary rows, left to right,
//     into the workgroup's CARRY row: the one row that is left of everything the workgroup has reduced so
//     far, with its coupling to the row left of the workgroup's range and the update it owes that row
//     (eliminate_forward, the streaming stage's own step).  ~10 us per tile, under ~45 us of streaming.
//   last tile of the workgroup: all eight waves reduce it with tile_cr (the barrier-synchronised form of
//     cgps_tile.h: quads for the wide levels, the matrix cores for the narrow ones) -- this is the only
//     reduction on the critical path -- thread 0 folds the carry into its boundary row, and the workgroup
//     leaves ONE record whatever T is: at most 256 records per launch, so the record stages always run
//     inside the launch (fold_record_stages), N = 2^24 included (chunk_reduce_kernel: three launches).
//
// Buffer reuse is ordered by the one barrier per tile (S_k): the streamers write buffer k & 1 (y lines during
// the stream, kept rows at its end) only after S_{k-1}, which the reducers reach only after they are done with
// tile k - 2 = the previous user of that buffer; the wave-boundary exchange slots are double-buffered the
// same way.  Results do not depend on timing: the elimination order is fixed by (N, grid).
#pragma once
// (included inside namespace cgps)

template <typename T, int D, int NT>
struct StreamSmem {
  static constexpr int DD = D * D;
  static constexpr size_t TILE_BYTES = ((((size_t)NT * (2 * DD + D) + DD) * sizeof(T)) + 15) & ~(size_t)15;
  static constexpr int NWV = NT / 64;                       // streaming waves = reducing waves
  static constexpr int CARRY = 3 * DD + 2 * D;              // R, C, dRa, y, dya of the carry row
  LdsTile<T, D> t[2];
  double* red;
  T* xch[2];          // [NWV][DD + D]: slot w < NWV-1 = what lane 0 of streaming wave w+1 owes row 64 w + 63;
                      // slot NWV-1 = what lane 0 of wave 0 owes the row left of the tile
  T* carry;
  T* recbuf;
  int* sfail;         // sfail[0]: first failing row; sfail[1]: last_flag of fold_record_stages
  unsigned* ctr;      // arrivals of the reducing waves (monotonic)
  char* base0;
  char* base1;
  __device__ __forceinline__ StreamSmem(char* smem) {
    t[0].carve(smem, NT);
    t[1].carve(smem + TILE_BYTES, NT);
    base0 = smem;
    base1 = smem + TILE_BYTES;
    char* tail = smem + 2 * TILE_BYTES;
    red = reinterpret_cast<double*>(tail);
    T* p = reinterpret_cast<T*>(red + 2 * (2 * NT / 64));
    xch[0] = p;
    xch[1] = p + NWV * (DD + D);
    carry = p + 2 * NWV * (DD + D);
    recbuf = carry + ((CARRY + 7) & ~7);
    sfail = reinterpret_cast<int*>(recbuf + ((CARRY + 7) & ~7));
    ctr = reinterpret_cast<unsigned*>(sfail + 2);
  }
};
template <typename T, int D, int NT>
constexpr size_t stream_lds_bytes() {
  using S = StreamSmem<T, D, NT>;
  return 2 * S::TILE_BYTES + 2 * (2 * NT / 64) * sizeof(double) +
         (size_t)(2 * S::NWV) * (D * D + D) * sizeof(T) + 2 * (size_t)((S::CARRY + 7) & ~7) * sizeof(T) + 64;
}

// The carry row of a workgroup lives in LDS (`cp`): R | C | dRa | y | dya, blocks row-major, R and dRa with a valid
// LOWER triangle.  fold_into_carry (ONE lane) folds the boundary row of a reduced run of tile rows into it, piece by
// piece through LDS, so that the lane never holds more than a handful of blocks (kept in registers, the carry and
// the step's temporaries are ~300 registers: scratch, and scratch costs this kernel its one-workgroup-per-CU grid).
//   slot_b : LDS slot of the run's boundary row (R, y);  oc_slot: Oc index of its coupling to the row left of the run
//   park0, levels: the run parked what it owes that left row in slots park0 + 2^l - 1, l < levels
//   share (may be null): a further share owed to the left row, lower triangle at share[i*D+j], vector at share[D*D+i]
//   first: there is no carry yet -- the run's left row is the row left of the workgroup's range, the boundary row
//          becomes the carry as it stands
template <typename T, int D>
__device__ __forceinline__ void fold_into_carry(LdsTile<T, D>& t, T* cp, bool first, int slot_b, int oc_slot, int park0,
                                                int levels, const T* share, PivotLog& pl, double& mah, bool& fail) {
  using LT = LdsTile<T, D>;
  constexpr int DD = D * D;
  T* cR = cp;
  T* cC = cp + DD;
  T* cdR = cp + 2 * DD;
  T* cy = cp + 3 * DD;
  T* cdy = cp + 3 * DD + D;
  // what the run owes its left row, as a (negative) sum to ADD to that row
  T uR[D][D], uy[D];
  set_zero<T, D>(uR);
  set_zero<T, D>(uy);
  if (share != nullptr) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      uy[i] = share[DD + i];
#pragma unroll
      for (int j = 0; j <= i; ++j) uR[i][j] = share[i * D + j];
    }
  }
#pragma unroll 1
  for (int l = 0; l < levels; ++l) {
    const int slot = park0 + (1 << l) - 1;
    T P[D][D], p[D];
    LT::load_blk(t.R, slot, P);
    load_vec<T, D>(t.y + slot * D, p);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      uy[i] -= p[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) uR[i][j] -= P[i][j];
    }
  }
  if (first) {
    T B[D][D], v[D];
    LT::load_blk(t.R, slot_b, B);
    load_vec<T, D>(t.y + slot_b * D, v);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      cy[i] = v[i];
      cdy[i] = uy[i];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        cR[i * D + j] = B[i][j];
        cdR[i * D + j] = (j <= i) ? uR[i][j] : T(0);
      }
    }
    LT::load_blk(t.Oc, oc_slot, B);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) cC[i * D + j] = B[i][j];
    return;
  }
  // eliminate the carry row between the row left of the workgroup's range and the run's boundary row
  // (eliminate_forward of cgps_tile.h, its operands fetched from LDS when they are needed):
  Chol<T, D> c;
  T x[D];
  {
    T A[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      x[i] = cy[i] + uy[i];
#pragma unroll
      for (int j = 0; j < D; ++j) A[i][j] = (j <= i) ? cR[i * D + j] + uR[i][j] : T(0);
    }
    pl.mul(chol_lower<T, D>(A, c, fail));
  }
  fwd_subst<T, D>(c, x);                         // x = D^-1 y
#pragma unroll
  for (int i = 0; i < D; ++i) mah += (double)x[i] * (double)x[i];
  T G[D][D];
  {
    T Cc[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) Cc[i][j] = cC[i * D + j];
    rsolve_lt_transposed<T, D>(c, Cc, G);        // G = C^T D^-T: coupling to the row left of the range
  }
  {
    T dR[D][D], dy[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      dy[i] = cdy[i];
#pragma unroll
      for (int j = 0; j < D; ++j) dR[i][j] = cdR[i * D + j];
    }
    syrk_sub_lower<T, D>(dR, G);
    gemv_sub<T, D>(dy, G, x);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      cdy[i] = dy[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) cdR[i * D + j] = dR[i][j];
    }
  }
  T F[D][D];
  LT::load_blk(t.Oc, oc_slot, F);
  rsolve_lt<T, D>(c, F);                         // F = O D^-T: coupling to the run's boundary row
  {
    T B[D][D], v[D];
    LT::load_blk(t.R, slot_b, B);
    load_vec<T, D>(t.y + slot_b * D, v);
    syrk_sub_lower<T, D>(B, F);
    gemv_sub<T, D>(v, F, x);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      cy[i] = v[i];
#pragma unroll
      for (int j = 0; j <= i; ++j) cR[i * D + j] = B[i][j];
    }
  }
  {
    T Cn[D][D];
    neg_abt<T, D>(Cn, F, G);                     // boundary row <-> row left of the range: -F G^T
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) cC[i * D + j] = Cn[i][j];
  }
}

// What lane 0 of a streaming wave owes the kept row of the previous wave's last lane: added to that row in LDS
// (symmetric block) by ONE lane, after the barrier that follows the streamers' stores.
template <typename T, int D>
__device__ __forceinline__ void apply_wave_boundary_share(LdsTile<T, D>& t, int row, const T* p) {
  using LT = LdsTile<T, D>;
  T Rl[D][D], yl[D];
  LT::load_blk(t.R, row, Rl);
  load_vec<T, D>(t.y + row * D, yl);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    yl[i] += p[D * D + i];
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      Rl[i][j] += p[i * D + j];
      Rl[j][i] = Rl[i][j];
    }
  }
  LT::store_blk(t.R, row, Rl);
  store_vec<T, D>(t.y + row * D, yl);
}

template <typename T, int D, int C, int NT, bool FOLD>
__global__ __launch_bounds__(2 * NT, 1) void stream_reduce_kernel(const T* __restrict__ Rg, const T* __restrict__ Og,
                                                                  const T* __restrict__ yg, int64_t N,
                                                                  const T* __restrict__ Oleft, T* __restrict__ rec,
                                                                  double* __restrict__ partial, FoldArgs fold,
                                                                  int64_t rows_per_wg, int64_t tile_rows) {
  static_assert(std::is_same<T, double>::value && D == 4 && NT == 256, "built for 4 x 4 fp64 blocks");
  constexpr int DD = D * D, NW = 2 * NT, NWV = NT / 64;
  using LT = LdsTile<T, D>;
  using RL = RecordLayout<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  StreamSmem<T, D, NT> sm(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool streamer = tid < NT;
  if (tid == 0) {
    sm.sfail[0] = 0x7fffffff;
    *sm.ctr = 0u;
  }
  const int64_t wg_lo = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t wg_hi = (wg_lo + rows_per_wg < N) ? wg_lo + rows_per_wg : N;
  const int n_tiles = (int)((wg_hi - wg_lo + tile_rows - 1) / tile_rows);
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;
  int ftile = -1;                            // first tile in which this lane saw a non-positive pivot
  __syncthreads();                           // the counter and the fail word are initialised

  constexpr int YR = 4;                      // (chunk_reduce_kernel: the right-hand-side line of four rows goes through LDS)
  static_assert(C % YR == 0, "whole right-hand-side lines per chunk");
#pragma unroll 1
  for (int k = 0; k < n_tiles; ++k) {
    const bool last = (k == n_tiles - 1);
#if defined(CGPS_STREAM_EXP) && CGPS_STREAM_EXP == 2
    const int64_t t_lo = ((int64_t)k * gridDim.x + blockIdx.x) * tile_rows;       // timing experiment: results are wrong
#else
    const int64_t t_lo = wg_lo + (int64_t)k * tile_rows;
#endif
#if defined(CGPS_STREAM_EXP) && CGPS_STREAM_EXP == 2
    const int64_t t_hi = (t_lo + tile_rows < N) ? t_lo + tile_rows : N;
#else
    const int64_t t_hi = (t_lo + tile_rows < wg_hi) ? t_lo + tile_rows : wg_hi;
#endif
    const int n_real = (int)((t_hi - t_lo + C - 1) / C);           // lanes of this tile that hold rows
    LT& t = sm.t[k & 1];
    T* xk = sm.xch[k & 1];
    if (streamer) {
      // ---- the streaming stage of chunk_reduce_kernel on rows [r0, r0 + C) ----
      T Rc[D][D], yc[D], Cc[D][D], dRa[D][D], dya[D];
      set_zero<T, D>(dRa);
      set_zero<T, D>(dya);
      set_zero<T, D>(Rc);
      set_zero<T, D>(yc);
      set_zero<T, D>(Cc);
      const int64_t r0 = t_lo + (int64_t)tid * C;
      T* ylds = reinterpret_cast<T*>((k & 1) ? sm.base1 : sm.base0) + (size_t)tid * (YR * D);
      const bool yfull = (r0 + C <= t_hi);
      auto stage_y_line = [&](int64_t row) {
        using V = typename Vec16<T>::type;
        const V* src = reinterpret_cast<const V*>(yg + row * D);
        V* dst = reinterpret_cast<V*>(ylds);
#pragma unroll
        for (int g = 0; g < YR * D / Vec16<T>::N; ++g) dst[g] = src[g];
      };
      if (r0 < t_hi) {
        load_block<T, D>(Rg + r0 * DD, Rc);
        if (yfull) {
          stage_y_line(r0);
          load_vec<T, D>(ylds, yc);
        } else {
          load_vec<T, D>(yg + r0 * D, yc);
        }
        if (r0 >= 1) load_block<T, D>(Og + (r0 - 1) * DD, Cc);
        else if (Oleft != nullptr) load_block<T, D>(Oleft, Cc);
#pragma unroll 1
        for (int j = 0; j < C - 1; ++j) {
          const int64_t rn = r0 + j + 1;
          if (rn >= t_hi) break;
          T Rn[D][D], On[D][D], yn[D];
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
      // the update a lane owes the previous lane's kept row: inside a wave by shuffle; across a wave boundary, and to
      // the row left of the tile, through the exchange slots that ONE lane applies after the barrier
      T nR[D][D], ny[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j2 = 0; j2 <= i; ++j2) nR[i][j2] = __shfl_down(dRa[i][j2], 1, 64);
        ny[i] = __shfl_down(dya[i], 1, 64);
      }
      if (lane != 63 && tid < n_real - 1) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
          for (int j2 = 0; j2 <= i; ++j2) Rc[i][j2] += nR[i][j2];
          yc[i] += ny[i];
        }
      }
      if (lane == 0) {
        T* p = xk + ((wave + NWV - 1) % NWV) * (DD + D);       // wave 0 -> slot NWV-1 (the row left of the tile)
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
          for (int j2 = 0; j2 <= i; ++j2) p[i * D + j2] = dRa[i][j2];
          p[DD + i] = dya[i];
        }
      }
      if (tid < n_real) {
        mirror_lower<T, D>(Rc);
        LT::store_blk(t.R, tid, Rc);
        store_vec<T, D>(t.y + tid * D, yc);
        LT::store_blk(t.Oc, tid, Cc);
      }
    }
    __syncthreads();                                   // S_k: tile k is in LDS; the reducers are done with tile k - 1
    if (!last) {
#if defined(CGPS_STREAM_EXP) && CGPS_STREAM_EXP == 1
      if (false) {
#else
      if (!streamer) {
#endif
        const int q = wave - NWV;                        // this wave's quarter of the tile
        const int base = 64 * q;
        const int nq = (n_real - base) > 64 ? 64 : (n_real - base);      // its rows (<= 0: none)
        if (nq > 0) {
          if (lane == 0 && q + 1 < NWV && base + 64 < n_real) apply_wave_boundary_share<T, D>(t, base + 63, xk + q * (DD + D));
          const int K = nq - 1;
          int olane = lane;
          asm volatile("" : "+v"(olane));                // (as in tile_cr: keeps this addressing out of other loops)
#pragma unroll 1
          for (int s = 1; (s - 1) < K; s <<= 1)
            tile_cr_level_quad<T, D, 64, false>(t, olane, K, (K + 1) / s, s, pl, mah, fail, base);
        }
        // the four quarters are reduced: arrive; wave NWV (the first reducing wave) folds them into the carry
        if (lane == 0) __hip_atomic_fetch_add(sm.ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (q == 0) {
          const unsigned want = (unsigned)NWV * (unsigned)(k + 1);
          while (__hip_atomic_load(sm.ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
          if (lane == 0) {
            bool first = (k == 0);
#pragma unroll 1
            for (int qq = 0; qq < NWV; ++qq) {
              const int b0 = 64 * qq;
              const int nqq = (n_real - b0) > 64 ? 64 : (n_real - b0);
              if (nqq <= 0) break;
              int lv = 0;                                  // levels quarter qq has run (tile_cr's loop condition)
              for (int s = 1; (s - 1) < nqq - 1; s <<= 1) ++lv;
              fold_into_carry<T, D>(t, sm.carry, first, b0 + nqq - 1, b0, b0, lv,
                                    qq == 0 ? xk + (NWV - 1) * (DD + D) : (const T*)nullptr, pl, mah, fail);
              first = false;
            }
          }
        }
      }
    } else {
      // ---- last tile: all eight waves reduce its kept rows to one boundary row (tile_cr of cgps_tile.h) ----
      if (!streamer && lane == 0) {
        const int q = wave - NWV;
        if (q + 1 < NWV && 64 * q + 64 < n_real) apply_wave_boundary_share<T, D>(t, 64 * q + 63, xk + q * (DD + D));
      }
      T* share = xk + (NWV - 1) * (DD + D);              // what the tile's first lane owes the row left of the tile
      if (n_tiles == 1) {
        // no carry: the record is the tile's, written exactly as chunk_reduce_kernel writes it
        reduce_staged_tile_and_emit<T, D, NW>(t, n_real, share, rec + (size_t)blockIdx.x * RL::STRIDE, pl, mah, fail);
      } else {
        reduce_staged_tile_and_emit<T, D, NW>(t, n_real, share, (T*)nullptr, pl, mah, fail);
        // (the reducers' work on tile k - 1, the carry included, precedes these barriers in their program order)
        if (tid == 0) {
          int levels = 0;
          for (int s = 1; (s - 1) < n_real - 1; s <<= 1) ++levels;
          fold_into_carry<T, D>(t, sm.carry, false, n_real - 1, 0, 0, levels, share, pl, mah, fail);
          // the workgroup's record, symmetric blocks, where the D*D lanes below pick it up
          T* rb = sm.recbuf;
          const T* cp = sm.carry;
#pragma unroll
          for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
              rb[RL::RS + i * D + j] = (j <= i) ? cp[i * D + j] : cp[j * D + i];
              rb[RL::CS + i * D + j] = cp[DD + i * D + j];
              rb[RL::DRA + i * D + j] = (j <= i) ? cp[2 * DD + i * D + j] : cp[2 * DD + j * D + i];
            }
            rb[RL::YS + i] = cp[3 * DD + i];
            rb[RL::DYA + i] = cp[3 * DD + D + i];
          }
        }
        // the D*D lanes that write the record are lanes of wave 0, as is thread 0: LDS order within the wave
        if (tid < DD) {
          T* r = rec + (size_t)blockIdx.x * RL::STRIDE;
          const T* rb = sm.recbuf;
          store_wt(r + RL::RS + tid, rb[RL::RS + tid]);
          store_wt(r + RL::CS + tid, rb[RL::CS + tid]);
          store_wt(r + RL::DRA + tid, rb[RL::DRA + tid]);
          if (tid < D) {
            store_wt(r + RL::YS + tid, rb[RL::YS + tid]);
            store_wt(r + RL::DYA + tid, rb[RL::DYA + tid]);
          }
        }
      }
    }
    if (fail && ftile < 0) ftile = k;
  }
  int64_t frow = wg_lo + (ftile < 0 ? 0 : (int64_t)ftile * tile_rows) + (streamer ? (int64_t)tid * C : 0);
  if (frow >= N) frow = N - 1;
  write_partial<NW>(mah, pl.value(), fail ? (int)(frow + 1) : 0, partial + PARTIAL_STRIDE * (size_t)blockIdx.x, sm.red, sm.sfail);
  if constexpr (FOLD) fold_record_stages<T, D, NW, true>(smem, sm.sfail + 1, rec, partial, fold, rows_per_wg, N);
}

// rows per workgroup and per tile for an N-row system on `cus` compute units (multiples of the chunk length C)
struct StreamPlan {
  int64_t rows_per_wg, tile_rows, grid;
  int tiles_per_wg;
};
inline StreamPlan stream_plan(int64_t N, int C, int lanes, int cus) {
  StreamPlan p;
  int64_t rw = (N + cus - 1) / cus;
  rw = ((rw + C - 1) / C) * C;
  const int64_t full = (int64_t)C * lanes;
  p.tiles_per_wg = (int)((rw + full - 1) / full);
  int64_t tr = (rw + p.tiles_per_wg - 1) / p.tiles_per_wg;
  tr = ((tr + C - 1) / C) * C;
  p.rows_per_wg = rw;
  p.tile_rows = tr;
  p.grid = (N + rw - 1) / rw;
  return p;
}

