// Narrow levels of the in-LDS cyclic reduction for 4 x 4 fp64 blocks on the matrix cores:
// SIXTEEN lanes per elimination, one block element per lane, v_mfma_f64_4x4x4 for every 4x4x4
// product.  Included from cgps_tile.h (inside namespace cgps) right before tile_cr.
//
// Why: tile_cr splits an elimination over four waves by role; that costs ~470 instructions per
// wave and ~2 KB of LDS operand reads per elimination whatever the number of eliminations, and a
// level with 16 of them leaves three quarters of the lanes idle.  Five of the nine passes over a
// 256-row tile are that narrow.  Here a wave takes four eliminations (lanes 16 r + 4 b + c hold
// element [r][c] of block b's operands), a 256-thread workgroup sixteen:
//   * every operand is read once, 8 bytes per lane;
//   * the Cholesky factor is never formed: a right-looking elimination over the 16 lanes yields
//     Li = L^-1 directly (row-k broadcasts are MFMAs with a selector matrix, column-k broadcasts
//     DPP quad permutes), and G^T = Li Ol, F^T = Li Or^T, x = Li y are three more MFMAs;
//   * G G^T, F F^T, F G^T, G x, F x: five MFMAs;
//   * the eliminations of a level touch disjoint slots, so ONE barrier per level is enough.
// Same LDS protocol as the role-split passes (slots, parked updates, pending rule), so a reduction
// switches form from one level to the next.
//
// v_mfma_f64_4x4x4 computes, for each of the 4 blocks b of a wave, D_b = A_b B_b + C_b with
// A_b[i][k] in lane 16k+4b+i, B_b[k][j] in lane 16k+4b+j and D_b[i][j] in lane 16i+4b+j.  With every
// matrix kept in the "standard" layout (element [r][c] in lane 16r+4b+c) that reads
//     mfma(X, Y, C) = X^T Y + C.

// double-offset of granule-element (g, tb) -- g = (4 r + c) / 2, tb = (4 r + c) % 2 -- of the block
// in `slot`: the granule swizzle of LdsTile<double, 4> (its key() without the >> 9 fold: the
// tile of this path has at most 257 slots)
__device__ __forceinline__ int mfma_elem_offset(int slot, int g, int tb) {
  const int key = (slot ^ (slot >> 3) ^ (slot >> 6)) & 7;
  return slot * 16 + (((g ^ key) << 1) | tb);
}

template <int Q>
__device__ __forceinline__ double quad_bcast_f64(double v) {          // element c == Q of the lane's quad
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), Q * 0x55, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), Q * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double mfma444(double x, double y, double c) {  // X^T Y + C, standard layout
  return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, c, 0, 0, 0);
}

// One level of the reduction (all its eliminations), then a barrier.  K = n_real - 1, M = rows of
// this level, s = its stride (see tile_cr).  Every lane of the workgroup must call this.
// (Letting wave 0 run the last levels -- at most four eliminations each -- without barriers was
// measured: no gain.)
//
// U = eliminations a 16-lane group carries through the arithmetic AT THE SAME TIME.  One chain is
// a string of dependent fp64 instructions (a lone wave issues one per ~8-10 cycles); the
// eliminations of a level are independent of each other, so U chains written side by side fill
// each other's latency slots: a level with U times as many eliminations costs well under U times
// one pass.  That is what lets the two widest levels of a 256-row tile (128 and 64 eliminations)
// run in this form too instead of the role-split form of tile_cr.
template <int NTHR, int U>
__device__ __forceinline__ void tile_cr_level_mfma(LdsTile<double, 4>& t, int K, int M, int s, PivotLog& pl, double& mah,
                                                   bool& fail) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = (lane >> 2) & 3, r = lane >> 4, c = lane & 3;
  const int h = s >> 1;
  const int n_elim = (M + 1) / 2;
  const double ident = (r == c) ? 1.0 : 0.0;
  const int g = (4 * r + c) >> 1, tb = c & 1;                   // this lane's element, straight and
  const int gT = (4 * c + r) >> 1, tT = r & 1;                  // transposed
  const bool c0 = (c == 0);
  constexpr int PER = NTHR / 16;                                 // eliminations per chain set
#pragma unroll 1
  for (int k0 = 0; k0 < n_elim; k0 += PER * U) {
    if (k0 + 4 * wave >= n_elim) continue;                       // (wave-uniform) nothing for this wave
    bool act[U];
    int e[U], o[U], oRe[U], oRo[U], oOl[U];
    double A[U], Ol[U], OrT[U], Ro[U], Y[U], yo[U];
    // ---- operands: one element per lane, branch-free (lanes without work read valid slots and
    // discard; y: every lane reads its row's entry, column 0 keeps it) --------------------------
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + PER * u + 4 * wave + b;
      e[u] = (2 * k + 1) * s - 1;
      act[u] = (2 * k < M) && (e[u] != K);
      o[u] = (2 * k + 1 < M) ? e[u] + s : K;
      const bool pend_e = act[u] && (s > 1) && (e[u] + h < K);
      const bool pend_o = act[u] && (s > 1) && (o[u] + h < K);
      oRe[u] = mfma_elem_offset(act[u] ? e[u] : K, g, tb);
      oRo[u] = mfma_elem_offset(act[u] ? o[u] : K, g, tb);
      oOl[u] = mfma_elem_offset(act[u] ? e[u] - s + 1 : 0, g, tb);
      const int ye = (act[u] ? e[u] : K) * 4 + r, yoff = (act[u] ? o[u] : K) * 4 + r;
      const int pe = pend_e ? e[u] + h : K, po = pend_o ? o[u] + h : K;
      const double lA = t.R[oRe[u]], lOl = t.Oc[oOl[u]], lOr = t.Oc[mfma_elem_offset(act[u] ? e[u] + 1 : 0, gT, tT)];
      const double lRo = t.R[oRo[u]], lY = t.y[ye], lyo = t.y[yoff];
      const double pA = t.R[mfma_elem_offset(pe, g, tb)], pY = t.y[pe * 4 + r];
      const double pR = t.R[mfma_elem_offset(po, g, tb)], py = t.y[po * 4 + r];
      A[u] = act[u] ? (pend_e ? lA - pA : lA) : ident;
      Ol[u] = act[u] ? lOl : 0.0;
      OrT[u] = act[u] ? lOr : 0.0;
      Ro[u] = pend_o ? lRo - pR : lRo;
      Y[u] = (act[u] && c0) ? (pend_e ? lY - pY : lY) : 0.0;
      yo[u] = pend_o ? lyo - py : lyo;
    }
    // (every writer of a row or of a parked update stores the full symmetric block, so A and Ro
    // are symmetric as loaded)
    // ---- Li = L^-1 of A = L L^T, right-looking over the 16 lanes ------------------------------
    double B[U], piv[U];
    bool f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { B[u] = ident; piv[u] = 1.0; f[u] = false; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double sel = (r == kk) ? 1.0 : 0.0;
      double rowA[U], rowB[U], p[U], col[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        rowA[u] = mfma444(sel, A[u], 0.0);                       // A[kk][c] in every row
        rowB[u] = mfma444(sel, B[u], 0.0);                       // B[kk][c]
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (kk == 0) { p[u] = quad_bcast_f64<0>(rowA[u]); col[u] = quad_bcast_f64<0>(A[u]); }
        else if (kk == 1) { p[u] = quad_bcast_f64<1>(rowA[u]); col[u] = quad_bcast_f64<1>(A[u]); }
        else if (kk == 2) { p[u] = quad_bcast_f64<2>(rowA[u]); col[u] = quad_bcast_f64<2>(A[u]); }
        else { p[u] = quad_bcast_f64<3>(rowA[u]); col[u] = quad_bcast_f64<3>(A[u]); }
        f[u] = f[u] || !(p[u] > 0.0);
        piv[u] *= p[u];
      }
      double rs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) rs[u] = rsqrt_fast(p[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double lik = col[u] * rs[u];                       // L[r][kk]
        const double ra = rowA[u] * rs[u], rb = rowB[u] * rs[u]; // L[c][kk];  row kk of B / L[kk][kk]
        if (r > kk) {
          A[u] = __builtin_fma(-lik, ra, A[u]);
          B[u] = __builtin_fma(-lik, rb, B[u]);
        } else if (r == kk) {
          B[u] = rb;
        }
      }
    }
    // ---- the products ----------------------------------------------------------------------------
    double Um[U], Gt[U], Ft[U], X[U];
#pragma unroll
    for (int u = 0; u < U; ++u) Um[u] = mfma444(B[u], ident, 0.0);        // Li^T
#pragma unroll
    for (int u = 0; u < U; ++u) {
      Gt[u] = mfma444(Um[u], Ol[u], 0.0);                        // Li Ol   = G^T
      Ft[u] = mfma444(Um[u], OrT[u], 0.0);                       // Li Or^T = F^T
      X[u] = mfma444(Um[u], Y[u], 0.0);                          // column 0: x = Li y
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double GGt = mfma444(Gt[u], Gt[u], 0.0);             // G G^T   (owed to the left neighbour: parked)
      const double Gx = mfma444(Gt[u], X[u], 0.0);               // column 0: G x
      const double FFt = mfma444(Ft[u], Ft[u], 0.0);
      const double Fx = mfma444(Ft[u], X[u], 0.0);
      const double FGt = mfma444(Ft[u], Gt[u], 0.0);             // F G^T
      // ---- results (disjoint slots per elimination: no barrier between reads and writes) ---------
      if (act[u]) {
        t.R[oRe[u]] = GGt;
        t.R[oRo[u]] = Ro[u] - FFt;
        t.Oc[oOl[u]] = -FGt;
      }
      if (act[u] && c0) {
        t.y[e[u] * 4 + r] = Gx;
        t.y[o[u] * 4 + r] = yo[u] - Fx;
        mah += X[u] * X[u];
      }
      if (act[u] && c0 && r == 0) {
        pl.mul(piv[u]);
        fail = fail || f[u];
      }
    }
  }
  __syncthreads();
}
