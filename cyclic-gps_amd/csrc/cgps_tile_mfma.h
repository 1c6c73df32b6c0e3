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
template <int NTHR>
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
#pragma unroll 1
  for (int k0 = 0; k0 < n_elim; k0 += NTHR / 16) {
    if (k0 + 4 * wave >= n_elim) continue;                       // (wave-uniform) nothing for this wave
    const int k = k0 + 4 * wave + b;
    const int e = (2 * k + 1) * s - 1;
    const bool act = (2 * k < M) && (e != K);
    const int o = (2 * k + 1 < M) ? e + s : K;
    const bool pend_e = act && (s > 1) && (e + h < K);
    const bool pend_o = act && (s > 1) && (o + h < K);
    // ---- operands: one element per lane, branch-free (lanes without work read valid slots and
    // discard; y: every lane reads its row's entry, column 0 keeps it) --------------------------
    const int oRe = mfma_elem_offset(act ? e : K, g, tb), oRo = mfma_elem_offset(act ? o : K, g, tb);
    const int oOl = mfma_elem_offset(act ? e - s + 1 : 0, g, tb);
    const int ye = (act ? e : K) * 4 + r, yoff = (act ? o : K) * 4 + r;
    const int pe = pend_e ? e + h : K, po = pend_o ? o + h : K;
    const double lA = t.R[oRe], lOl = t.Oc[oOl], lOr = t.Oc[mfma_elem_offset(act ? e + 1 : 0, gT, tT)];
    const double lRo = t.R[oRo], lY = t.y[ye], lyo = t.y[yoff];
    const double pA = t.R[mfma_elem_offset(pe, g, tb)], pY = t.y[pe * 4 + r];
    const double pR = t.R[mfma_elem_offset(po, g, tb)], py = t.y[po * 4 + r];
    double A = act ? (pend_e ? lA - pA : lA) : ident;
    const double Ol = act ? lOl : 0.0, OrT = act ? lOr : 0.0;
    double Ro = pend_o ? lRo - pR : lRo;
    double Y = (act && c0) ? (pend_e ? lY - pY : lY) : 0.0;
    double yo = pend_o ? lyo - py : lyo;
    // (every writer of a row or of a parked update stores the full symmetric block, so A and Ro
    // are symmetric as loaded)
    // ---- Li = L^-1 of A = L L^T, right-looking over the 16 lanes ------------------------------
    double B = ident, piv = 1.0;
    bool f = false;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double sel = (r == kk) ? 1.0 : 0.0;
      const double rowA = mfma444(sel, A, 0.0);                  // A[kk][c] in every row
      const double rowB = mfma444(sel, B, 0.0);                  // B[kk][c]
      double p, col;
      if (kk == 0) { p = quad_bcast_f64<0>(rowA); col = quad_bcast_f64<0>(A); }
      else if (kk == 1) { p = quad_bcast_f64<1>(rowA); col = quad_bcast_f64<1>(A); }
      else if (kk == 2) { p = quad_bcast_f64<2>(rowA); col = quad_bcast_f64<2>(A); }
      else { p = quad_bcast_f64<3>(rowA); col = quad_bcast_f64<3>(A); }
      f = f || !(p > 0.0);
      piv *= p;
      const double rs = rsqrt_fast(p);
      const double lik = col * rs;                               // L[r][kk]
      const double ra = rowA * rs, rb = rowB * rs;               // L[c][kk];  row kk of B / L[kk][kk]
      if (r > kk) {
        A = __builtin_fma(-lik, ra, A);
        B = __builtin_fma(-lik, rb, B);
      } else if (r == kk) {
        B = rb;
      }
    }
    // ---- the products ----------------------------------------------------------------------------
    const double U = mfma444(B, ident, 0.0);                      // Li^T
    const double Gt = mfma444(U, Ol, 0.0);                        // Li Ol   = G^T
    const double Ft = mfma444(U, OrT, 0.0);                       // Li Or^T = F^T
    const double X = mfma444(U, Y, 0.0);                          // column 0: x = Li y
    const double GGt = mfma444(Gt, Gt, 0.0);                      // G G^T   (owed to the left neighbour: parked)
    const double Gx = mfma444(Gt, X, 0.0);                        // column 0: G x
    const double FFt = mfma444(Ft, Ft, 0.0);
    const double Fx = mfma444(Ft, X, 0.0);
    const double FGt = mfma444(Ft, Gt, 0.0);                      // F G^T
    // ---- results (disjoint slots per elimination: no barrier between reads and writes) -----------
    if (act) {
      t.R[oRe] = GGt;
      t.R[oRo] = Ro - FFt;
      t.Oc[oOl] = -FGt;
    }
    if (act && c0) {
      t.y[e * 4 + r] = Gx;
      t.y[o * 4 + r] = yo - Fx;
      mah += X * X;
    }
    if (act && c0 && r == 0) {
      pl.mul(piv);
      fail = fail || f;
    }
  }
  __syncthreads();
}
