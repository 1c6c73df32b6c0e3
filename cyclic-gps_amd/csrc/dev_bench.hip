// Developer micro-benchmarks for the fused path (not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o cyclic-gps_amd/lib/dev_bench cyclic-gps_amd/csrc/dev_bench.hip
// Run on the GPU box: ./cyclic-gps_amd/lib/dev_bench [log2N]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CGPS_FIN_STAMPS 1
#ifndef DEVB_D
#define DEVB_D 4          // block size of the tile_cr and timeline modes (-DDEVB_D=5 ...)
#endif
#ifndef DEVB_T
#define DEVB_T double     // scalar type of the tile_cr mode (-DDEVB_T=float)
#endif
#include "cgps_tile.h"

using namespace cgps;

template <typename T, int D>
__device__ __forceinline__ void set_identity(T (&A)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) A[i][j] = (i == j) ? T(1) : T(0);
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// cheap deterministic SPD block-tridiagonal generator: R = (2 + u) I + small symmetric, O small
template <typename T, int D>
__global__ void gen_kernel(T* R, T* O, T* y, int64_t N) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  unsigned h = (unsigned)(i * 2654435761u) ^ 0x9e3779b9u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (T)((h >> 8) & 0xffff) / (T)65536 - (T)0.5; };
  for (int a = 0; a < D; ++a)
    for (int b = 0; b <= a; ++b) {
      T v = (a == b) ? (T)2.5 + (T)0.2 * rnd() : (T)0.1 * rnd();
      R[i * D * D + a * D + b] = v;
      R[i * D * D + b * D + a] = v;
    }
  if (i < N - 1)
    for (int a = 0; a < D * D; ++a) O[i * D * D + a] = (T)0.3 * rnd();
  for (int a = 0; a < D; ++a) y[i * D + a] = rnd();
}

// ---- V2: the stage-1 access pattern with no arithmetic to speak of ----------------------
template <typename T, int D, int C, int NT>
__global__ __launch_bounds__(NT) void loads_only_kernel(const T* __restrict__ Rg, const T* __restrict__ Og,
                                                        const T* __restrict__ yg, int64_t N, double* out) {
  constexpr int DD = D * D;
  const int64_t r0 = ((int64_t)blockIdx.x * NT + threadIdx.x) * C;
  T acc = 0;
#pragma unroll 1
  for (int j = 0; j < C; ++j) {
    const int64_t rn = r0 + j;
    if (rn < N - 1) {
      T Rn[D][D], On[D][D], yn[D];
      load_block<T, D>(Rg + rn * DD, Rn);
      load_block<T, D>(Og + rn * DD, On);
      load_vec<T, D>(yg + rn * D, yn);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        acc += yn[a];
#pragma unroll
        for (int b = 0; b < D; ++b) acc += Rn[a][b] + On[a][b];
      }
    }
  }
  if (acc == (T)123456.789) out[0] = acc;
}

// ---- V2b: fully coalesced streaming read of the same bytes ------------------------------
__global__ __launch_bounds__(256) void coalesced_read_kernel(const double2* __restrict__ p, int64_t n16, double* out) {
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    double2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 123456.789) out[0] = acc;
}

// ---- V1: chunk phase only (no LDS reduction) ---------------------------------------------
template <typename T, int D, int C, int NT>
__global__ __launch_bounds__(NT) void chunk_only_kernel(const T* __restrict__ Rg, const T* __restrict__ Og,
                                                        const T* __restrict__ yg, int64_t N, double* out) {
  constexpr int DD = D * D;
  const int64_t r0 = ((int64_t)blockIdx.x * NT + threadIdx.x) * C;
  PivotLog pl;
  double mah = 0.0;
  bool fail = false;
  T Rc[D][D], yc[D], Cc[D][D], dRa[D][D], dya[D];
  set_zero<T, D>(dRa);
  set_zero<T, D>(dya);
  if (r0 < N) {
    load_block<T, D>(Rg + r0 * DD, Rc);
    load_vec<T, D>(yg + r0 * D, yc);
  } else {
    set_identity<T, D>(Rc);
    set_zero<T, D>(yc);
  }
  if (r0 >= 1 && r0 < N) load_block<T, D>(Og + (r0 - 1) * DD, Cc);
  else set_zero<T, D>(Cc);
#pragma unroll 1
  for (int j = 0; j < C - 1; ++j) {
    const int64_t rn = r0 + j + 1;
    T Rn[D][D], On[D][D], yn[D];
    if (rn < N) {
      load_block<T, D>(Rg + rn * DD, Rn);
      load_block<T, D>(Og + (rn - 1) * DD, On);
      load_vec<T, D>(yg + rn * D, yn);
    } else {
      set_identity<T, D>(Rn);
      set_zero<T, D>(On);
      set_zero<T, D>(yn);
    }
    eliminate_forward<T, D>(Rc, yc, Cc, dRa, dya, On, Rn, yn, pl, mah, fail);
  }
  double s = mah + pl.value() + (double)Rc[0][0] + (double)Cc[0][0] + (double)dRa[0][0] + (double)dya[0] + (double)yc[0];
  if (s == 123456.789 || fail) out[0] = s;
}

// ---- probe: lane maps of v_mfma_f64_4x4x4_4b_f64 ------------------------------------------
__global__ void mfma_probe_kernel(int* table) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) table[la * 64 + lb] = lane;
    }
}
static void run_mfma_probe() {
  int* table;
  CK(hipMalloc(&table, 4096 * sizeof(int)));
  CK(hipMemset(table, 0xff, 4096 * sizeof(int)));
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, table);
  CK(hipDeviceSynchronize());
  std::vector<int> h(4096);
  CK(hipMemcpy(h.data(), table, 4096 * sizeof(int), hipMemcpyDeviceToHost));
  printf("mfma_f64_4x4x4 probe: for A-lane la: list of (B-lane -> D-lane)\n");
  for (int la = 0; la < 64; ++la) {
    if (!(la < 20 || la == 32 || la == 48)) continue;
    printf("  la=%2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf(" (%d->%d)", lb, h[la * 64 + lb]);
    printf("\n");
  }
}


// ---- tile_cr alone: one 256-slot tile per workgroup, per-policy wall time of the in-LDS reduction ----
// (wall_clock64: 100 MHz).  n_real rows are regenerated before every repetition; the time of the
// reduction itself is stamped by thread 0 between two barriers.
template <int NTHR, int MW>
__global__ __launch_bounds__(NTHR) void tilecr_bench_kernel(int n_real, int reps, long long* ticks, double* sums) {
  using T = DEVB_T;
  constexpr int D = DEVB_D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  StageSmem<T, D, 256, NTHR> sm(smem);
  const int tid = threadIdx.x;
  long long acc = 0;
  double mah = 0.0;
  PivotLog pl;
  bool fail = false;
  for (int rep = 0; rep < reps; ++rep) {
    if (tid < n_real) {
      unsigned h = (unsigned)((tid + 977 * blockIdx.x) * 2654435761u) ^ 0x9e3779b9u;
      auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (T)((h >> 8) & 0xffff) / (T)65536 - (T)0.5; };
      T R[D][D], O[D][D], y[D];
      for (int a = 0; a < D; ++a) {
        for (int b = 0; b <= a; ++b) { T v = (a == b) ? (T)2.5 + (T)0.2 * rnd() : (T)0.1 * rnd(); R[a][b] = v; R[b][a] = v; }
        for (int b = 0; b < D; ++b) O[a][b] = (T)0.3 * rnd();
        y[a] = rnd();
      }
      LdsTile<T, D>::store_blk(sm.t.R, tid, R);
      LdsTile<T, D>::store_blk(sm.t.Oc, tid, O);
      store_vec<T, D>(sm.t.y + tid * D, y);
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    tile_cr<T, D, NTHR, MW>(sm.t, n_real, pl, mah, fail);
    const long long t1 = wall_clock64();
    __syncthreads();
    acc += t1 - t0;
  }
  double logp = pl.value();
  block_sum2<NTHR>(mah, logp, sm.red);
  if (tid == 0) {
    ticks[blockIdx.x] = acc;
    sums[2 * blockIdx.x] = mah;
    sums[2 * blockIdx.x + 1] = logp;
  }
}

template <int NTHR, int MW>
void run_tilecr(int grid, hipStream_t st) {
  const size_t lds = stage_lds_bytes<DEVB_T, DEVB_D>(256, NTHR);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tilecr_bench_kernel<NTHR, MW>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long long* ticks;
  double* sums;
  CK(hipMalloc(&ticks, grid * sizeof(long long)));
  CK(hipMalloc(&sums, 2 * grid * sizeof(double)));
  const int reps = 50;
  printf("tile_cr NTHR=%d MW=%d grid=%d :", NTHR, MW, grid);
  double prev = 0;
  for (int n_real : {2, 4, 8, 16, 32, 64, 128, 256}) {
    hipLaunchKernelGGL((tilecr_bench_kernel<NTHR, MW>), dim3(grid), dim3(NTHR), lds, st, n_real, reps, ticks, sums);
    CK(hipStreamSynchronize(st));
    std::vector<long long> h(grid);
    std::vector<double> hs(2 * grid);
    CK(hipMemcpy(h.data(), ticks, grid * sizeof(long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), sums, 2 * grid * sizeof(double), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg = avg / grid / reps * 10.0;   // ns
    printf("  n=%d: %.0f ns (+%.0f)", n_real, avg, avg - prev);
    if (n_real == 256) printf("  [check %.10g %.10g]", hs[0], hs[1]);
    prev = avg;
  }
  printf("\n");
  CK(hipFree(ticks));
  CK(hipFree(sums));
}

template <typename F>
float time_ms(F&& launch, int reps, hipStream_t st) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / reps;
}

template <typename T, int D, int C, int NT>
void run_chunk_variants(const T* R, const T* O, const T* y, int64_t N, double* out, hipStream_t st, double bytes) {
  const int64_t tiles = (N + (int64_t)C * NT - 1) / ((int64_t)C * NT);
  float a = time_ms([&] { hipLaunchKernelGGL((loads_only_kernel<T, D, C, NT>), dim3((unsigned)tiles), dim3(NT), 0, st, R, O, y, N, out); }, 20, st);
  float b = time_ms([&] { hipLaunchKernelGGL((chunk_only_kernel<T, D, C, NT>), dim3((unsigned)tiles), dim3(NT), 0, st, R, O, y, N, out); }, 20, st);
  printf("C=%2d NT=%3d tiles=%6lld : loads-only %7.2f us (%5.2f TB/s)   chunk-only %7.2f us (%5.2f TB/s)\n", C, NT,
         (long long)tiles, a * 1e3, bytes / (a * 1e-3) / 1e12, b * 1e3, bytes / (b * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  using T = double;
  constexpr int D = DEVB_D;
  const int lg = argc > 1 ? atoi(argv[1]) : 20;
  const int64_t N = (int64_t)1 << lg;
  if (argc > 2 && atoi(argv[2]) == 2) {           // dev_bench 20 2: stamps inside the final reduction of the real pipeline
    hipStream_t s2;
    CK(hipStreamCreate(&s2));
    T *R2, *O2, *y2;
    CK(hipMalloc(&R2, N * D * D * sizeof(T)));
    CK(hipMalloc(&O2, N * D * D * sizeof(T)));
    CK(hipMalloc(&y2, N * D * sizeof(T)));
    hipLaunchKernelGGL((gen_kernel<T, D>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s2, R2, O2, y2, N);
    const size_t wsb = tile_ws_bytes(N, D, sizeof(T));
    char* ws;
    double* out2;
    int* info;
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc(&out2, 16));
    CK(hipMalloc(&info, 4));
    for (int mode = (argc > 3 ? atoi(argv[3]) : 0); mode < 2; ++mode) {     // dev_bench 20 2 <0: two launches | 1: folded>
      setenv("CGPS_NO_FOLD", mode == 0 ? "1" : "0", 1);
      for (int it = 0; it < 30; ++it) {
        hipEvent_t ea, eb;
        CK(hipEventCreate(&ea));
        CK(hipEventCreate(&eb));
        CK(hipEventRecord(ea, s2));
        int rc = run_tile_mahal_logdet<T, D>(R2, O2, y2, N, ws, wsb, out2, info, s2);
        CK(hipEventRecord(eb, s2));
        CK(hipStreamSynchronize(s2));
        float ms;
        CK(hipEventElapsedTime(&ms, ea, eb));
        long long st[16];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fin_stamps), sizeof(st)));
        double o[2];
        CK(hipMemcpy(o, out2, 16, hipMemcpyDeviceToHost));
        if (it >= 28) {
          static long long ks[512][12];
          CK(hipMemcpyFromSymbol(ks, HIP_SYMBOL(g_k_stamps), sizeof(ks)));
          long long t0 = ks[0][0];
          for (int b = 0; b < 256; ++b) t0 = ks[b][0] < t0 ? ks[b][0] : t0;
          static int xcc[512];
          CK(hipMemcpyFromSymbol(xcc, HIP_SYMBOL(g_k_xcc), sizeof(xcc)));
          double sx[8] = {0}, mx[8] = {0};
          int cx[8] = {0};
          for (int b = 0; b < 256; ++b) {
            const int x = xcc[b] & 7;
            const double d = (ks[b][1] - t0) * 0.01;
            sx[x] += d; cx[x]++; mx[x] = d > mx[x] ? d : mx[x];
          }
          printf("   stream end per XCD (avg/max us):");
          for (int x = 0; x < 8; ++x) printf(" [%d: n=%d %.1f/%.1f]", x, cx[x], cx[x] ? sx[x] / cx[x] : 0.0, mx[x]);
          printf("\n   stream end by block index (us):");
          for (int b = 0; b < 256; b += 17) printf(" b%d:%.1f", b, (ks[b][1] - t0) * 0.01);
          printf("\n");
          const char* names[10] = {"start", "streamed", "tile reduced+emitted", "partial written", "arrived(group)",
                                   "group reduced", "arrived(top)", "final done", "rows staged in LDS", "tile_cr done"};
          for (int k = 0; k < 10; ++k) {
            long long lo = 1LL << 62, hi = 0;
            int cnt = 0;
            for (int b = 0; b < 256; ++b) {
              if (ks[b][k] < t0) continue;            // not reached in this launch
              lo = ks[b][k] < lo ? ks[b][k] : lo;
              hi = ks[b][k] > hi ? ks[b][k] : hi;
              ++cnt;
            }
            if (cnt) printf("   %-22s first %.2f us  last %.2f us  (%d workgroups)\n", names[k], (lo - t0) * 0.01, (hi - t0) * 0.01, cnt);
          }
        }
        if (it >= 27)
          printf("mode %s rc %d: op %.1f us | final: loads %.2f  lds-stores %.2f  reduce %.2f  last-row %.2f  sums+write %.2f  total %.2f us  [%.10g %.10g]\n",
                 mode == 0 ? "two launches" : "folded", rc, ms * 1e3, (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01,
                 (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01, (st[5] - st[4]) * 0.01, (st[5] - st[0]) * 0.01, o[0], o[1]);
      }
      break;   // fold_final_enabled() latches the environment at first use: one mode per process
    }
    return 0;
  }
  if (argc > 2 && atoi(argv[2]) == 1) {           // dev_bench 20 1: only the in-LDS reduction timings
    hipStream_t s2;
    CK(hipStreamCreate(&s2));
    for (int grid : {1, 256}) {
      run_tilecr<256, 1>(grid, s2);
      run_tilecr<256, 2>(grid, s2);
      run_tilecr<256, 4>(grid, s2);
      run_tilecr<512, 1>(grid, s2);
      run_tilecr<512, 2>(grid, s2);
      run_tilecr<512, 4>(grid, s2);
    }
    return 0;
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  T *R, *O, *y;
  double* out;
  CK(hipMalloc(&R, N * D * D * sizeof(T)));
  CK(hipMalloc(&O, N * D * D * sizeof(T)));
  CK(hipMalloc(&y, N * D * sizeof(T)));
  CK(hipMalloc(&out, 4096));
  hipLaunchKernelGGL((gen_kernel<T, D>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, R, O, y, N);
  CK(hipStreamSynchronize(st));
  const double bytes = ((2.0 * N - 1) * D * D + N * D) * sizeof(T);
  printf("N=2^%d d=%d fp64  algorithmic bytes %.1f MB\n", lg, D, bytes / 1e6);
  run_mfma_probe();

  {  // coalesced streaming read of R and O back to back (ceiling for read-only streaming)
    const int64_t n16 = N * D * D * sizeof(T) / 16;
    float ms = time_ms([&] {
      hipLaunchKernelGGL(coalesced_read_kernel, dim3(2048), dim3(256), 0, st, (const double2*)R, n16, out);
      hipLaunchKernelGGL(coalesced_read_kernel, dim3(2048), dim3(256), 0, st, (const double2*)O, n16, out);
    }, 20, st);
    printf("coalesced read of R+O (2 launches): %7.2f us  (%5.2f TB/s)\n", ms * 1e3, 2.0 * n16 * 16 / (ms * 1e-3) / 1e12);
  }
  run_chunk_variants<T, D, 4, 256>(R, O, y, N, out, st, bytes);
  run_chunk_variants<T, D, 8, 256>(R, O, y, N, out, st, bytes);
  run_chunk_variants<T, D, 16, 256>(R, O, y, N, out, st, bytes);
  run_chunk_variants<T, D, 8, 128>(R, O, y, N, out, st, bytes);
  run_chunk_variants<T, D, 8, 64>(R, O, y, N, out, st, bytes);
  run_chunk_variants<T, D, 16, 64>(R, O, y, N, out, st, bytes);
  run_chunk_variants<T, D, 32, 64>(R, O, y, N, out, st, bytes);

  return 0;
}
