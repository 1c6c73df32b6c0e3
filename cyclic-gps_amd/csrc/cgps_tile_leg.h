// Stage 1 of the fused solve + log-det with the operands of a LEG model ASSEMBLED IN REGISTERS (SURVEY.md 8(f) N2,
// second half): the block rows of  J = PEG precision(ts, G) + blockdiag(A)  (reference models.py:181-239, :254-268)
// are computed by the lane that eliminates them, from the time stamps and the d x d generator, instead of being
// written to HBM by cgps_peg_precision and read back -- for LEG workloads the compulsory read of Rs / Os disappears
// (what is left is the right-hand side, d values per row) and so do two launches of a log-likelihood.
// Included from cgps_tile.h (inside namespace cgps); the arithmetic is that of cgps_leg.h:
//     E_g = exp(-1/2 (t_{g+1} - t_g) G),   a_g = (I - E_g^T E_g)^-1 E_g^T   (one symmetric positive definite solve)
//     b_g = (I - E_g E_g^T)^-1 E_g = a_g^T                                   (push-through identity)
//     row g+1 gets  toRight_g = E_g a_g,   row g gets  toLeft_g = E_g^T b_g = (a_g E_g)^T,   J[g+1, g] = -b_g
//     R_i = I + toLeft_i + toRight_{i-1} + A
// A lane walks its chunk left to right: one gap evaluation per row (the gap AFTER the row; toRight and b are carried
// to the next row), plus the gap before its first row.
#pragma once

// the three terms of the gap between rows g and g+1; false when the gap is singular (zero length)
template <typename T, int D>
__device__ __forceinline__ bool leg_gap(const T* __restrict__ ts, const T* __restrict__ Gg, int64_t g, T (&toRight)[D][D],
                                        T (&toLeft)[D][D], T (&b)[D][D]) {
  const T dt = ts[g + 1] - ts[g];
  T E[D][D], a[D][D];
  {
    T A[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) A[i][j] = T(-0.5) * dt * Gg[i * D + j];
    mat_exp<T, D>(E, A);
  }
  bool ok;
  {
    T Et[D][D], S[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) Et[i][j] = E[j][i];
    mat_mul<T, D>(S, Et, E);
    ok = spd_solve_i_minus<T, D>(S, Et, a);
  }
  mat_mul<T, D>(toRight, E, a);
  {
    T aE[D][D];
    mat_mul<T, D>(aE, a, E);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        toLeft[i][j] = aE[j][i];
        b[i][j] = a[j][i];
      }
  }
  return ok;
}

// Row r of the system and its coupling to row r-1:  R = I + cR (toRight of the gap before it, carried) + toLeft of the
// gap after it + A;  O = -cB (b of the gap before it).  Leaves the gap after r in (cR, cB) for row r+1.
template <typename T, int D>
__device__ __forceinline__ void leg_row(const T* __restrict__ ts, const T* __restrict__ Gg, const T* __restrict__ Ag,
                                        const T* __restrict__ vg, int64_t r, int64_t N, T (&cR)[D][D], T (&cB)[D][D],
                                        T (&R)[D][D], T (&O)[D][D], T (&y)[D], bool& fail) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    y[i] = vg ? vg[r * D + i] : T(0);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      R[i][j] = ((i == j) ? T(1) : T(0)) + cR[i][j] + (Ag ? Ag[i * D + j] : T(0));
      O[i][j] = -cB[i][j];
    }
  }
  if (r + 1 < N) {
    T tl[D][D];
    if (!leg_gap<T, D>(ts, Gg, r, cR, tl, cB)) fail = true;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) R[i][j] += tl[i][j];
  }
}
