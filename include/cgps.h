/* libcgps -- C ABI of the MI355X (gfx950) block-tridiagonal cyclic-reduction library.
 *
 * The reference (cunningham-lab/cyclic-gps) has no FFI: its boundary for this
 * path is the Python module cyclic_gps/cyclic_reduction.py.  Each entry point
 * below is what a ctypes binding for one function of that module calls; the
 * function it replaces is cited as file:line (relative to the reference root).
 * The Python mirror of the module lives in cyclic-gps_amd/cyclic_gps/.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers (hipMalloc / torch CUDA tensors),
 *     C-contiguous, batch-major: Rs[N][d][d], Os[N-1][d][d], vectors [N][d].
 *     O_i is the LOWER off-diagonal block J[i+1][i].
 *   - The library never allocates, frees or synchronises: the caller passes a
 *     scratch buffer of cgps_workspace_bytes() bytes, a HIP stream (hipStream_t
 *     as void*; NULL = default stream), and every call is asynchronous.
 *   - dtype: CGPS_F32 / CGPS_F64.  1 <= d <= 8.
 *   - Return value: CGPS_OK, or an error code; cgps_last_error() gives a
 *     thread-local message.
 *   - Non-positive-definite blocks are reported through the device word
 *     `info` (like torch.linalg.cholesky_ex): 0 = fine, otherwise 1 + the
 *     smallest original block-row index whose pivot was not positive.  The
 *     library zeroes it at the start of the call.
 *   - Factor storage ("packed factor"): the reference's decomp = (ms, Ds, Fs, Gs)
 *     (cyclic_reduction.py:287-309) with the per-level tensors concatenated in
 *     level order: Dp[sum ceil(m_l/2)] = Dp[N], Fp[sum floor(m_l/2)],
 *     Gp[sum floor((m_l-1)/2)], m_0 = N, m_{l+1} = floor(m_l/2).  Allocate N
 *     blocks for each.  cgps_level_layout() returns the sizes and offsets.
 *     A vector in "CRR layout" (halfsolve output, cyclic_reduction.py:312-338)
 *     uses the Dp offsets: [N][d].
 */
#ifndef CGPS_H
#define CGPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGPS_VERSION 320

enum { CGPS_F32 = 0, CGPS_F64 = 1 };

enum {
  CGPS_OK = 0,
  CGPS_ERR_ARG = 1,        /* bad argument (null pointer, N < 1, workspace too small ...) */
  CGPS_ERR_HIP = 2,        /* HIP runtime error at launch */
  CGPS_ERR_UNSUPPORTED = 3 /* d or dtype outside the compiled range */
};

/* which operation a workspace is sized for */
enum {
  CGPS_OP_MAHAL_LOGDET = 0,
  CGPS_OP_DECOMPOSE = 1,
  CGPS_OP_HALFSOLVE = 2,
  CGPS_OP_BACKSOLVE = 3,
  CGPS_OP_SOLVE = 4,
  CGPS_OP_LOGDET_FACTOR = 5,
  CGPS_OP_INVERSE_BLOCKS = 6,
  CGPS_OP_MAHAL_LOGDET_LEVELWISE = 7,
  CGPS_OP_DECOMPOSE_SOLVE = 8
};

#define CGPS_MAX_LEVELS 64

int cgps_version(void);
const char* cgps_last_error(void);

/* Level sizes and packed-factor offsets for N block rows (host-side arithmetic only).
 * ms[l] = rows at level l (ms[nlevels-1] == 1); offD/offF/offG[l] = first block of
 * level l in Dp/Fp/Gp; arrays must hold CGPS_MAX_LEVELS entries (offsets: +1 for the total). */
int cgps_level_layout(int64_t N, int* nlevels, int64_t* ms, int64_t* offD, int64_t* offF, int64_t* offG);

int cgps_workspace_bytes(int64_t N, int d, int dtype, int op, size_t* bytes);

/* mahal_and_det(Rs, Os, x) -> (x^T J^-1 x, log|J|)        cyclic_reduction.py:380-438
 * out2[0] = mahal, out2[1] = logdet (device doubles, accumulated in fp64 for both dtypes).
 * When a block is not positive definite (info != 0) cgps_mahal_logdet and cgps_finish_records
 * write NaN to both, so a caller that never reads info cannot take a wrong number for a result. */
int cgps_mahal_logdet(const void* Rs, const void* Os, const void* x, int64_t N, int d, int dtype,
                      void* ws, size_t ws_bytes, double* out2, int* info, void* stream);
/* same result, one kernel launch per reduction level (simple form, kept as cross-check) */
int cgps_mahal_logdet_levelwise(const void* Rs, const void* Os, const void* x, int64_t N, int d, int dtype,
                                void* ws, size_t ws_bytes, double* out2, int* info, void* stream);

/* decompose_step(Rs, Os) -> (n, D, F, G), (R', O')           cyclic_reduction.py:203-259
 * Dk[ceil(n/2)], Fk[n/2], Gk[(n-1)/2], Rn[n/2], On[n/2-1]; n >= 2. */
int cgps_decompose_step(const void* Rs, const void* Os, int64_t n, int d, int dtype,
                        void* Dk, void* Fk, void* Gk, void* Rn, void* On, int* info, void* stream);

/* decompose(Rs, Os) -> packed factor                          cyclic_reduction.py:287-309 */
int cgps_decompose(const void* Rs, const void* Os, int64_t N, int d, int dtype,
                   void* Dp, void* Fp, void* Gp, void* ws, size_t ws_bytes, int* info, void* stream);

/* Right-hand sides: the reference's einsums carry a trailing "..." (cyclic_reduction.py:52-57,
 * 76-84), so halfsolve / backhalfsolve / solve take y[N][d] or Y[N][d][m].  nrhs = m (1 for a plain
 * vector); all vectors are [N][d][nrhs], C-contiguous.  For nrhs > 1 the factor is read once per
 * sweep and per panel of up to eight columns (csrc/cgps_solve_tile_m.h) instead of once per column;
 * size the workspace with cgps_solve_workspace_bytes(). */
int cgps_solve_workspace_bytes(int64_t N, int d, int dtype, int op, int nrhs, size_t* bytes);

/* halfsolve(decomp, y) -> L^-1 T y in CRR layout              cyclic_reduction.py:312-338
 * mahal_out (optional, device double) receives ||L^-1 T y||^2 = mahal(decomp, y), :461-467
 * (summed over all nrhs columns, like the reference's torch.sum). */
int cgps_halfsolve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, int nrhs,
                   const void* y, void* xcrr, void* ws, size_t ws_bytes, double* mahal_out, void* stream);

/* backhalfsolve(decomp, ycrr) -> T^T L^-T ycrr, natural order  cyclic_reduction.py:341-377 */
int cgps_backsolve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, int nrhs,
                   const void* ycrr, void* x, void* ws, size_t ws_bytes, void* stream);

/* solve(decomp, y) -> J^-1 y                                   cyclic_reduction.py:441-444 */
int cgps_solve(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype, int nrhs,
               const void* y, void* x, void* ws, size_t ws_bytes, void* stream);

/* decompose(Rs, Os) AND solve(decomp, y) in one call, for callers that factor and solve together
 * (compute_insample_posterior, models.py:288-292; the backward of mahal_and_det): the factor comes out as from
 * cgps_decompose, xcrr[N][d] = halfsolve(decomp, y) in CRR layout, x[N][d] = J^-1 y.  The forward substitution of
 * y rides along in the first pass of the factorisation, so the forward sweep proper starts three levels down and
 * never reads the factor blocks of those levels (7/8 of the factor): the factor is read once, not twice.
 * Workspace: cgps_workspace_bytes(N, d, dtype, CGPS_OP_DECOMPOSE_SOLVE). */
int cgps_decompose_solve(const void* Rs, const void* Os, const void* y, int64_t N, int d, int dtype,
                         void* Dp, void* Fp, void* Gp, void* xcrr, void* x, void* ws, size_t ws_bytes, int* info,
                         void* stream);

/* det(decomp) -> log|J| = 2 sum log diag(D)                    cyclic_reduction.py:447-458 */
int cgps_logdet_factor(const void* Dp, int64_t N, int d, int dtype,
                       void* ws, size_t ws_bytes, double* out, void* stream);

/* inverse_blocks(decomp) -> diag and lower off-diag blocks of J^-1   cyclic_reduction.py:470-503
 * Sd[N][d][d], So[N-1][d][d]. */
int cgps_inverse_blocks(const void* Dp, const void* Fp, const void* Gp, int64_t N, int d, int dtype,
                        void* Sd, void* So, void* ws, size_t ws_bytes, void* stream);

/* Adjoint of mahal_and_det in the blocks (what autograd computes through cyclic_reduction.py:380-438
 * for LEGFamily.training_step, models.py:374-381).  Given Sd/So = inverse_blocks(decomp),
 * w[N][d] = solve(decomp, x) and the two upstream gradients gm, gl (DEVICE scalars of the blocks'
 * dtype: d loss / d mahal, d loss / d logdet), overwrites in place
 *   Sd[i] <- gl Sd[i] - gm w_i w_i^T            (= d loss / d Rs[i])
 *   So[i] <- 2 (gl So[i] - gm w_i+1 w_i^T)      (= d loss / d Os[i])
 * d loss / d x = 2 gm w is the caller's.  No workspace. */
int cgps_mahal_logdet_adjoint(void* Sd, void* So, const void* w, int64_t N, int d, int dtype,
                              const void* gm, const void* gl, void* stream);

/* ---- operand assembly for LEG models (the caller's step right before the path) ---------------
 * Blocks of the PEG prior precision from the time stamps and the generator G
 * (models.py:181-239: E_i = exp(-1/2 (t_{i+1}-t_i) G), two d x d solves per gap):
 * ts[N] (same dtype as the blocks), G[d][d]  ->  Rs[N][d][d], Os[N-1][d][d].
 * info: 0, or 1 + the index of a row next to a gap whose systems are not positive definite
 * (zero-length gap, NaN).  No workspace. */
int cgps_peg_precision(const void* ts, const void* G, int64_t N, int d, int dtype,
                       void* Rs, void* Os, int* info, void* stream);

/* cgps_mahal_logdet of a LEG system WITHOUT materialising its blocks (the assembly fused into the first pass of the
 * reduction; replaces cgps_peg_precision + cgps_mahal_logdet of models.py:349-367 when no factor is kept):
 *     J = PEG precision(ts, G) + blockdiag(A),   out2 = {v^T J^-1 v, log|J|}
 * ts[N], G[d][d], A[d][d] (added to EVERY diagonal block; NULL: nothing added -- the prior precision itself),
 * v[N][d] (NULL: zeros, out2[0] = 0).  Every lane assembles the block rows it eliminates in registers from the time
 * stamps: nothing but ts and v is read from HBM.  One launch at any N.  Workspace, out2 and info as for
 * cgps_mahal_logdet (cgps_workspace_bytes(N, d, dtype, CGPS_OP_MAHAL_LOGDET)); info also reports a singular gap
 * (zero length).  CGPS_ERR_UNSUPPORTED for d = 8 and fp64 d = 6 (their first pass shares a block row between lanes):
 * assemble with cgps_peg_precision and call cgps_mahal_logdet there. */
int cgps_leg_mahal_logdet(const void* ts, const void* G, const void* A, const void* v, int64_t N, int d, int dtype,
                          void* ws, size_t ws_bytes, double* out2, int* info, void* stream);

/* Both reductions of a LEG log-likelihood (models.py:349-367) in ONE launch, side by side:
 *   out4[0..1] = {v^T K^-1 v, log|K|},  K = PEG precision(ts, G) + blockdiag(A)      (posterior precision, info2[0])
 *   out4[2..3] = {0, log|PEG precision(ts, G)|}                                       (prior precision,     info2[1])
 * Workspace: twice cgps_workspace_bytes(N, d, dtype, CGPS_OP_MAHAL_LOGDET), each half rounded up to 256 bytes. */
int cgps_leg_mahal_logdet_pair(const void* ts, const void* G, const void* A, const void* v, int64_t N, int d, int dtype,
                               void* ws, size_t ws_bytes, double* out4, int* info2, void* stream);

/* Adjoint of cgps_peg_precision in G and in the time gaps (training through the assembly; what
 * autograd computes through models.py:181-239 for LEGFamily.training_step, models.py:374-381).
 * gRs[N][d][d], gOs[N-1][d][d]: d loss / d Rs, d loss / d Os.  One lane per time gap, 64 gaps per
 * workgroup; gG_partial[ceil((N-1)/64)][d][d] receives each workgroup's share of d loss / d G
 * (the caller adds them up: a few d x d blocks, deterministic order); gtau[N-1] (may be NULL)
 * receives d loss / d (t_{i+1} - t_i).  N >= 2.  No workspace. */
int cgps_peg_precision_adjoint(const void* ts, const void* G, int64_t N, int d, int dtype,
                               const void* gRs, const void* gOs, void* gG_partial, void* gtau, void* stream);

/* Prediction glue of LEG models, the step AFTER the path (reference models.py:455-514 intercast with
 * forecast :394-408, interpolate :410-452 and gaussian_stitch model_utils.py:64-107, which the reference
 * runs as a Python loop over the targets): posterior mean and covariance of the latent at p target
 * times from the in-sample posterior.  ts[n] sorted; target_ts[p]; G[d][d]; ip_mean[n][d];
 * ip_cov_diag[n][d][d]; ip_cov_offdiag[n-1][d][d] = Cov(z_{i+1}, z_i) (what cgps_inverse_blocks leaves
 * in So)  ->  out_mean[p][d], out_cov[p][d][d].  One lane per target; no workspace, no info word (the
 * systems solved are positive definite whenever ts is strictly increasing). */
int cgps_leg_intercast(const void* ts, int64_t n, const void* target_ts, int64_t p, const void* G, int d, int dtype,
                       const void* ip_mean, const void* ip_cov_diag, const void* ip_cov_offdiag,
                       void* out_mean, void* out_cov, void* stream);

/* ---- time-axis sharding (one shard per GPU / rank) -----------------------------------------
 * The reference has no distributed code; this is the multi-GPU form BASELINE.json asks for.
 * A shard is n_loc consecutive block rows: Rs[n_loc], Os[n_loc-1] (couplings INSIDE the shard),
 * x[n_loc], plus O_left = J[first row of the shard, last row of the previous shard] (NULL for the
 * first shard).  cgps_shard_reduce eliminates every row of the shard but its last one and leaves
 *   record_out  : cgps_record_elems() elements = that last row (R, y), its new coupling to the
 *                 previous shard's last row, and the additive update it owes that row;
 *   partial_out : 4 doubles {sum of squares, sum of log pivots, 1 + first failing LOCAL row or 0, 0}.
 * The caller gathers the P records and partials in shard order (ONE all-gather over RCCL; the
 * library itself never communicates) and every rank calls cgps_finish_records, which reduces
 * the P-row boundary system and writes out2 = {mahal, logdet} and info (1 + a failing row --
 * local to the shard that saw it -- or 0).  record_stride_bytes / partial_stride_bytes are the
 * distances between consecutive shards' records / partials (record stride: a multiple of 16 bytes,
 * records 16-byte aligned), so both can be read in place from
 * the receive buffer of an all-gather of [record | partial] messages (record_out and
 * partial_out of cgps_shard_reduce may point straight into the send buffer; partial_out must
 * be 8-byte aligned).  P <= 1024 (256 for fp32 d = 8 and fp64 d = 7; 64 for fp64 d = 6 and d = 8).  Built for every block size
 * 1 <= d <= 8 in both precisions.  A rank may send several records (its rows cut into consecutive sub-shards,
 * each reduced by its own cgps_shard_reduce with O_left = the coupling to the previous sub-shard's last row):
 * cgps_finish_records simply takes all of them in row order. */
int cgps_record_elems(int d, int dtype, int64_t* elems);
int cgps_shard_reduce(const void* Rs, const void* Os, const void* x, const void* O_left, int64_t n_loc, int d,
                      int dtype, void* ws, size_t ws_bytes, void* record_out, double* partial_out, void* stream);
int cgps_finish_records(const void* records, size_t record_stride_bytes, const double* partials,
                        size_t partial_stride_bytes, int64_t P, int64_t rows_per_shard, int64_t N_total, int d,
                        int dtype, double* out2, int* info, void* stream);

/* The small systems a sharded SOLVE needs from the gathered records (cyclic_gps/sharded.py), one launch each:
 * cgps_boundary_solve: x at every shard's last row, xsep[P][d] -- the P-row block-tridiagonal system of the shards' last
 *   rows (R_w = Rs_w + dRa_{w+1}, y_w = ys_w + dya_{w+1}, J[w+1, w] = Cs_{w+1}) solved by block Cholesky; P <= 64;
 *   info: 0 or 1 + the first row whose pivot block is not positive definite.
 * cgps_boundary_recursions: what the rest of the system leaves on rank's two ends, out = [Pa (d*d) | pa (d) | dR (d*d) | dy (d)]:
 *   (Pa, pa) = diagonal block and right-hand side of the PREVIOUS shard's last row once every row left of it has been
 *   eliminated (zeros for rank 0), (dR, dy) = what eliminating every row right of rank's last row adds to that row. */
int cgps_boundary_solve(const void* records, size_t record_stride_bytes, int64_t P, int d, int dtype, void* xsep,
                        int* info, void* stream);
int cgps_boundary_recursions(const void* records, size_t record_stride_bytes, int64_t P, int64_t rank, int d, int dtype,
                             void* out, int* info, void* stream);

/* The one-launch forms of cgps_mahal_logdet / cgps_shard_reduce hand records from workgroup to workgroup inside the
 * launch through arrival counters in the library's own device memory.  Every completed launch leaves them at zero;
 * a launch that did NOT complete (a GPU fault in another kernel of the process, a killed context that was revived)
 * can leave a count half-way.  cgps_reset_counters enqueues a memset of all of them on `stream` (current device).
 * Call it with no library call in flight on that device.  Never needed in normal operation. */
int cgps_reset_counters(void* stream);

/* Measurement hook (bench.py): the next cgps_mahal_logdet call on this host thread
 * records `start` right before and `stop` right after its dominant kernel (the one
 * that streams Rs/Os/x from HBM) on the call's stream, then the hook clears itself.
 * start/stop are hipEvent_t created with timing enabled; nothing synchronises. */
int cgps_profile_next_call(void* start_event, void* stop_event);

#ifdef __cplusplus
}
#endif
#endif /* CGPS_H */
