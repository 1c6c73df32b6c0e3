// Small dense d x d primitives, one block per lane, everything in registers.
// Written for gfx950 (CDNA4): all loops are compile-time unrolled so the blocks
// live in VGPRs; no runtime-indexed arrays (those go to scratch).
//
// Conventions: J block tridiagonal, R_i = J[i,i],
// O_i = J[i+1,i] (lower off-diagonal).  Eliminating block row e with Cholesky
// factor D (lower) gives  F = O_e D^-T  (coupling to row e+1) and
// G = O_{e-1}^T D^-T (coupling to row e-1); cf. cyclic_reduction.py:225-254.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgps {

template <typename T> struct Vec16;  // 16-byte vector of T for wide global/LDS access
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };
template <> struct Vec16<float>  { using type = float4;  static constexpr int N = 4; };

// fused multiply-add in the block's own precision (__builtin_fma alone is the double one: with
// float operands it converts, multiplies in fp64 and converts back)
__device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- exchanges inside a DPP quad (four adjacent lanes sharing one block row) ------------------
template <typename T> struct QuadOps;
template <> struct QuadOps<float> {
  template <int CTRL> static __device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
  }
};
template <> struct QuadOps<double> {
  template <int CTRL> static __device__ __forceinline__ double dpp(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
};
// value of v held by lane `src` (a compile-time constant after unrolling) of this lane's quad
template <typename T>
__device__ __forceinline__ T quad_from(T v, int src) {
  switch (src) {
    case 0: return QuadOps<T>::template dpp<0x00>(v);
    case 1: return QuadOps<T>::template dpp<0x55>(v);
    case 2: return QuadOps<T>::template dpp<0xAA>(v);
    default: return QuadOps<T>::template dpp<0xFF>(v);
  }
}
// sum over the four lanes of the quad, in every lane
template <typename T>
__device__ __forceinline__ T quad_sum(T v) {
  v += QuadOps<T>::template dpp<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v += QuadOps<T>::template dpp<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  return v;
}

// threads per 128-row tile: one wave (16 quads, several rounds per level) at fp32, four waves at fp64
// ---- reciprocal square root -------------------------------------------------
// v_rsq_f64 is good to ~2^-23 relative; two Newton steps bring it to ~1 ulp.
__device__ __forceinline__ double rsqrt_fast(double s) {
  double r = __builtin_amdgcn_rsq(s);
  double h = 0.5 * s;
  r = r * __builtin_fma(-h * r, r, 1.5);
  r = r * __builtin_fma(-h * r, r, 1.5);
  return r;
}
__device__ __forceinline__ float rsqrt_fast(float s) {
  float r = __builtin_amdgcn_rsqf(s);
  float h = 0.5f * s;
  r = r * __builtin_fmaf(-h * r, r, 1.5f);
  return r;
}

// ---- reciprocal -----------------------------------------------------------------
// hardware estimate + Newton steps (a full IEEE division is ~20 instructions in fp64)
__device__ __forceinline__ double rcp_fast(double s) {
  double r = __builtin_amdgcn_rcp(s);
  r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float rcp_fast(float s) {
  float r = __builtin_amdgcn_rcpf(s);
  r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
  return r;
}

// A lower-triangular Cholesky factor kept as its strict lower part, its
// diagonal and the reciprocal diagonal (so substitutions multiply, never divide).
template <typename T, int D>
struct Chol {
  T l[D][D];    // l[i][j], j <= i used
  T inv[D];     // 1 / l[j][j]
};

// Cholesky of the LOWER triangle of A (torch.linalg.cholesky semantics).
// Returns the product of the pivots (= det(A) = prod diag(L)^2) in double and
// sets fail when a pivot is not strictly positive (or NaN).
template <typename T, int D>
__device__ __forceinline__ double chol_lower(const T (&A)[D][D], Chol<T, D>& c, bool& fail) {
  // product of the pivots: in the block's own precision while that cannot overflow (fp64 always;
  // fp32: the float product is kept when it stays well inside the range, otherwise -- very large or
  // very small blocks -- the product is redone in double from the factor's diagonal)
  T piv = T(1);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T s = A[j][j];
#pragma unroll
    for (int m = 0; m < j; ++m) s = fmaT(-c.l[j][m], c.l[j][m], s);
    fail = fail || !(s > T(0));
    piv *= s;
    T r = rsqrt_fast(s);
    c.inv[j] = r;
    c.l[j][j] = s * r;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      T t = A[i][j];
#pragma unroll
      for (int m = 0; m < j; ++m) t = fmaT(-c.l[i][m], c.l[j][m], t);
      c.l[i][j] = t * r;
    }
  }
  if constexpr (sizeof(T) == 4 && D > 1) {
    if (!(piv > T(1e-30) && piv < T(1e30))) {
      double p = 1.0;
#pragma unroll
      for (int j = 0; j < D; ++j) p *= (double)c.l[j][j] * (double)c.l[j][j];
      return p;
    }
  }
  return (double)piv;
}

// v <- L^-1 v  (forward substitution).  The same routine gives a row of
// B L^-T from the corresponding row of B.
template <typename T, int D>
__device__ __forceinline__ void fwd_subst(const Chol<T, D>& c, T (&v)[D]) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T s = v[j];
#pragma unroll
    for (int m = 0; m < j; ++m) s = fmaT(-c.l[j][m], v[m], s);
    v[j] = s * c.inv[j];
  }
}

// v <- L^-T v  (backward substitution).
template <typename T, int D>
__device__ __forceinline__ void bwd_subst(const Chol<T, D>& c, T (&v)[D]) {
#pragma unroll
  for (int j = D - 1; j >= 0; --j) {
    T s = v[j];
#pragma unroll
    for (int m = j + 1; m < D; ++m) s = fmaT(-c.l[m][j], v[m], s);
    v[j] = s * c.inv[j];
  }
}

// X <- B L^-T, row by row.
template <typename T, int D>
__device__ __forceinline__ void rsolve_lt(const Chol<T, D>& c, T (&B)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) fwd_subst<T, D>(c, B[i]);
}

// X <- B^T L^-T : row i of X is column i of B pushed through fwd_subst.
template <typename T, int D>
__device__ __forceinline__ void rsolve_lt_transposed(const Chol<T, D>& c, const T (&B)[D][D], T (&X)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) X[i][j] = B[j][i];
    fwd_subst<T, D>(c, X[i]);
  }
}

// lower(S) -= A A^T
template <typename T, int D>
__device__ __forceinline__ void syrk_sub_lower(T (&S)[D][D], const T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      T s = S[i][j];
#pragma unroll
      for (int m = 0; m < D; ++m) s = fmaT(-A[i][m], A[j][m], s);
      S[i][j] = s;
    }
}

// y -= A x
template <typename T, int D>
__device__ __forceinline__ void gemv_sub(T (&y)[D], const T (&A)[D][D], const T (&x)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    T s = y[i];
#pragma unroll
    for (int m = 0; m < D; ++m) s = fmaT(-A[i][m], x[m], s);
    y[i] = s;
  }
}

// y -= A^T x
template <typename T, int D>
__device__ __forceinline__ void gemvT_sub(T (&y)[D], const T (&A)[D][D], const T (&x)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    T s = y[i];
#pragma unroll
    for (int m = 0; m < D; ++m) s = fmaT(-A[m][i], x[m], s);
    y[i] = s;
  }
}

// C = -A B^T
template <typename T, int D>
__device__ __forceinline__ void neg_abt(T (&C)[D][D], const T (&A)[D][D], const T (&B)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T s = T(0);
#pragma unroll
      for (int m = 0; m < D; ++m) s = fmaT(-A[i][m], B[j][m], s);
      C[i][j] = s;
    }
}

template <typename T, int D>
__device__ __forceinline__ void mirror_lower(T (&S)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i + 1; j < D; ++j) S[i][j] = S[j][i];
}

// dense lower-triangular block (upper part zero), as torch.linalg.cholesky returns it
template <typename T, int D>
__device__ __forceinline__ void chol_to_dense(const Chol<T, D>& c, T (&L)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) L[i][j] = (j <= i) ? c.l[i][j] : T(0);
}

// rebuild the substitution form from a stored dense factor
template <typename T, int D>
__device__ __forceinline__ void chol_from_dense(const T (&L)[D][D], Chol<T, D>& c) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) c.l[i][j] = L[i][j];
    c.inv[i] = rcp_fast(L[i][i]);
  }
}

// ---- block load / store from a contiguous [.., D, D] or [.., D] array --------
template <typename T, int D>
__device__ __forceinline__ void load_block(const T* __restrict__ p, T (&A)[D][D]) {
  constexpr int VN = Vec16<T>::N;
  if constexpr ((D * D) % VN == 0) {
    using V = typename Vec16<T>::type;
    const V* q = reinterpret_cast<const V*>(p);
    T flat[D * D];
#pragma unroll
    for (int i = 0; i < D * D / VN; ++i) {
      V v = q[i];
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int t = 0; t < VN; ++t) flat[i * VN + t] = e[t];
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) A[i][j] = flat[i * D + j];
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) A[i][j] = p[i * D + j];
  }
}

template <typename T, int D>
__device__ __forceinline__ void store_block(T* __restrict__ p, const T (&A)[D][D]) {
  constexpr int VN = Vec16<T>::N;
  if constexpr ((D * D) % VN == 0) {
    using V = typename Vec16<T>::type;
    V* q = reinterpret_cast<V*>(p);
#pragma unroll
    for (int i = 0; i < D * D / VN; ++i) {
      V v;
      T* e = reinterpret_cast<T*>(&v);
#pragma unroll
      for (int t = 0; t < VN; ++t) e[t] = A[(i * VN + t) / D][(i * VN + t) % D];
      q[i] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) p[i * D + j] = A[i][j];
  }
}

// RG consecutive blocks / vectors whose total size is a multiple of 16 bytes, from a 16-byte aligned address
template <typename T, int D, int RG>
__device__ __forceinline__ void load_rows(const T* __restrict__ p, T (&A)[RG][D][D]) {
  constexpr int VN = Vec16<T>::N, DD = D * D;
  static_assert((RG * DD) % VN == 0, "a row group must fill whole 16-byte granules");
  using V = typename Vec16<T>::type;
  const V* q = reinterpret_cast<const V*>(p);
  T flat[RG * DD];
#pragma unroll
  for (int i = 0; i < RG * DD / VN; ++i) {
    V v = q[i];
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int t = 0; t < VN; ++t) flat[i * VN + t] = e[t];
  }
#pragma unroll
  for (int r = 0; r < RG; ++r)
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) A[r][i][j] = flat[r * DD + i * D + j];
}
template <typename T, int D, int RG>
__device__ __forceinline__ void load_row_vecs(const T* __restrict__ p, T (&y)[RG][D]) {
  constexpr int VN = Vec16<T>::N;
  if constexpr ((RG * D) % VN == 0) {
    using V = typename Vec16<T>::type;
    const V* q = reinterpret_cast<const V*>(p);
    T flat[RG * D];
#pragma unroll
    for (int i = 0; i < RG * D / VN; ++i) {
      V v = q[i];
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int t = 0; t < VN; ++t) flat[i * VN + t] = e[t];
    }
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int i = 0; i < D; ++i) y[r][i] = flat[r * D + i];
  } else {
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int i = 0; i < D; ++i) y[r][i] = p[r * D + i];
  }
}

template <typename T, int D>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, T (&v)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = p[i];
}
template <typename T, int D>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const T (&v)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) p[i] = v[i];
}

// ---- reductions ---------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum (a, b) over the workgroup; result valid in thread 0.  scratch: 2*nwaves doubles.
template <int NTHREADS>
__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch) {
  constexpr int NW = NTHREADS / 64;
  a = wave_sum(a);
  b = wave_sum(b);
  if constexpr (NW > 1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { scratch[2 * w] = a; scratch[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      a = 0.0; b = 0.0;
#pragma unroll
      for (int i = 0; i < NW; ++i) { a += scratch[2 * i]; b += scratch[2 * i + 1]; }
    }
  }
}

// Record the first failing block row: info holds 0 (ok) or 1 + the smallest
// ORIGINAL (level-0) row index whose pivot was not positive.
__device__ __forceinline__ void report_fail(int* info, int64_t orig_row) {
  int mine = (int)(orig_row + 1);
  int cur = *reinterpret_cast<volatile int*>(info);
  while (cur == 0 || cur > mine) {
    int prev = atomicCAS(info, cur, mine);
    if (prev == cur) break;
    cur = prev;
  }
}

}  // namespace cgps
