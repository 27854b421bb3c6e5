/* nngp_hip.h -- C ABI of libnngp_hip.so, the MI355X (gfx950) NNGP/NTK hot path.
 *
 * The reference (Kangfei/NNGP-src) has no FFI: its hot path is three neural-tangents calls made
 * from Python.  This header is therefore the boundary a reference
 * maintainer would bind with ctypes (INTEGRATION.md shows the stub); each entry point cites the
 * reference call it replaces.  Plain C types only -- no torch/HIP types in the signatures.
 *
 * Conventions
 *   - every pointer named x*, y, mean, var*, out* is a DEVICE pointer (HBM) unless it says "host";
 *   - matrices are row-major; inputs are float64 like the reference (train.py:24, estimator.py:12);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value 0 = success, <0 = error (nngp_last_error() holds a thread-local message);
 *   - entry points are re-entrant across models; one model must be driven from one stream at a time.
 */
#ifndef NNGP_HIP_H
#define NNGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNGP_ABI_VERSION 1
#define NNGP_GET_NNGP 1 /* get='nngp' */
#define NNGP_GET_NTK 2  /* get='ntk'  */
#define NNGP_MAX_DENSE 16

#define NNGP_DTYPE_F32 0
#define NNGP_DTYPE_F64 1

#define NNGP_COV_NONE 0 /* compute_cov=False                                  */
#define NNGP_COV_DIAG 1 /* only diag(cov): all the reference consumes (train.py:180, estimator.py:55) */
#define NNGP_COV_FULL 2 /* compute_cov=True: the full M x M covariance        */

/* Architecture of stax.serial(Dense, (Relu, Dense) x (n_dense-1)); the widths (512, 1) do not enter
 * the closed-form kernel.  Reference: train.py:161-164, estimator.py:27-30 (n_dense = 2, W_std = 1,
 * b_std = 0). */
typedef struct nngp_arch {
    int32_t n_dense;
    int32_t reserved;
    double w_std[NNGP_MAX_DENSE];
    double b_std[NNGP_MAX_DENSE];
} nngp_arch;

/* Fit diagnostics (host struct filled by nngp_model_info). */
typedef struct nngp_fit_info {
    double reg;            /* diag_reg * trace(K)/N actually added (train.py:171-172 semantics)      */
    double trace_mean;     /* trace(K)/N                                                            */
    double rel_residual;   /* |y - (K + reg I) alpha|_2 / |y|_2 after refinement (float64 residual)  */
    int32_t refine_iters;  /* preconditioned-CG iterations used (max over output columns)            */
    int32_t clamped_pivots;/* pivots the float32 factorisation had to clamp (preconditioner only)    */
    int64_t n;             /* training rows                                                          */
    int64_t n_padded;      /* leading dimension of the device matrices                               */
} nngp_fit_info;

typedef struct nngp_model nngp_model;

int nngp_version(void);
#ifdef NNGP_TIMING_KNOBS
/* Timing-experiment switches (ablations, some of which produce WRONG results on purpose).  NOT in libnngp_hip.so:
 * only libnngp_hip_knobs.so, built with -DNNGP_TIMING_KNOBS for scripts/ and the A/B tests, exports it. */
int nngp_debug_set(int32_t key, int32_t value);
#endif
const char* nngp_last_error(void);

/* ---- a1: kernel_fn(x1, x2, get) ---------------------------------------------------------------
 * Replaces the closure returned by stax.serial(...) (train.py:161-164; estimator.py:27-30).
 * x1: [n1, d] f64; x2: [n2, d] f64 or NULL (symmetric, x2 = x1, n2 ignored).
 * Writes rows [row_begin, row_end) of the n1 x n2 kernel (row-block shard: the slot
 * nt.batch(device_count>0) would fill, train.py:166-168); pass 0, n1 for everything.
 * out_nngp / out_ntk: f32 or f64 (out_dtype) matrices with leading dimension ld, either may be NULL;
 * element (i, j) is written at out[i * ld + j] (i is the global row).                            */
int nngp_kernel_build(const double* x1, int64_t n1, const double* x2, int64_t n2, int32_t d,
                      const nngp_arch* arch, int32_t out_dtype, void* out_nngp, void* out_ntk,
                      int64_t ld, int64_t row_begin, int64_t row_end, void* stream);

/* K(x, x) and Theta(x, x) per row (closed form from |x|^2/d); either output may be NULL. */
int nngp_kernel_diag(const double* x, int64_t n, int32_t d, const nngp_arch* arch,
                     double* diag_nngp, double* diag_ntk, void* stream);

/* ---- a3: gradient_descent_mse_ensemble(kernel_fn, X, Y, diag_reg) -----------------------------
 * Replaces nt.predict.gradient_descent_mse_ensemble (train.py:171-172; estimator.py:34-35).
 * nngp_model_create allocates every device buffer the model will need for up to n_cap training
 * rows and m_cap test rows (no allocation happens inside fit/predict unless m > m_cap).
 * get: NNGP_GET_NNGP or NNGP_GET_NTK (the posterior for `get`); ny: columns of Y (reference: 1).   */
int nngp_model_create(nngp_model** out, int64_t n_cap, int64_t m_cap, int32_t d, int32_t ny,
                      const nngp_arch* arch, int32_t get, double diag_reg,
                      int32_t diag_reg_absolute_scale);
int nngp_model_destroy(nngp_model* m);

/* Whole fit = set_train + build_rows(0, n) + factor + solve. */
int nngp_model_fit(nngp_model* m, const double* x, const double* y, int64_t n, void* stream);

/* Stage-level entry points (bench.py times them; the multi-GPU path shards build_rows):
 *   set_train : copy X [n, d], Y [n, ny]; row norms, closed-form diagonal, regulariser
 *   build_rows: rows [row_begin, row_end) of the float64 train-train kernel (all columns);
 *               (0, n) takes the symmetric path (lower tiles computed once, mirrored)
 *   factor    : A32 = float32(K) + reg I, blocked lower Cholesky in place (panels on the float32 MFMA, the large
 *               trailing updates as split-float16 MFMA products with float32 accumulation).  The factor is the
 *               preconditioner of the float64 solves, not the solver: when the float32 factorisation breaks down
 *               (pivots at the rounding-noise floor, cond(K + reg I) * eps32 >> 1) it is rebuilt from the float64
 *               kernel with a 16x larger diagonal shift, up to four times -- more CG iterations, same alpha and
 *               refined covariances.  nngp_model_factor_shift returns the shift in use (= reg normally); the call
 *               waits for the factorisation (one 4-byte read-back of the clamped-pivot count).
 *   solve     : alpha = (K + reg I)^-1 Y by CG on the float64 kernel, preconditioned by the
 *               float32 factor (max_iters <= 0 and tol <= 0 select the defaults 60 / 1e-10; 60 is scaled by
 *               sqrt(shift / reg) when the factor carries a raised shift).  With one output column, a predict that
 *               also forms covariance rows Z ~ K_td (K + reg I)^-1 stops this CG at 1e-6 and corrects the mean through
 *               them: mu = K_td a_k + Z r_k, exact up to (K_td A^-1 - Z) r_k, the product of two small errors (means
 *               agree with the converged solve to ~1e-9).  nngp_model_alpha, nngp_model_info and mean-only predicts
 *               take the solve up again where it stopped and run it to tol.
 *               The call records the request; the CG itself runs on the model's own high-priority stream when alpha
 *               is first needed -- nngp_model_alpha / _info (which then block for it), or inside nngp_model_predict
 *               after the covariance work has been enqueued, so that it overlaps it.                          */
int nngp_model_set_train(nngp_model* m, const double* x, const double* y, int64_t n, void* stream);
int nngp_model_build_rows(nngp_model* m, int64_t row_begin, int64_t row_end, void* stream);
int nngp_model_factor(nngp_model* m, void* stream);
double nngp_model_factor_shift(nngp_model* m);
int nngp_model_solve(nngp_model* m, int32_t max_iters, double tol, void* stream);
/* Appends b training rows (device pointers x_new [b, d], y_new [b, ny]) to a fitted model: the kernel rows of the new
 * queries are built and the float32 factor is EXTENDED (L10 by a blocked triangular solve against the existing factor,
 * L11 by a small Cholesky) instead of refactored -- the active-learning loop of the reference
 * (active/ActiveLearner.py:43-77) refits from scratch after every batch of `budget` queries.  n + b <= n_cap.
 * Follow with nngp_model_solve: alpha is the exact float64 solution for all n + b rows (the extended factor is the
 * CG preconditioner; with a relative regulariser its old part belongs to the previous trace, which costs no accuracy). */
int nngp_model_append(nngp_model* m, const double* x_new, const double* y_new, int64_t b, void* stream);


/* Block-column pieces of `factor` for the multi-GPU right-looking Cholesky (host: nngp-src_amd/distributed.py):
 * block columns of width w (multiple of 128) are dealt cyclically to the ranks; the owner factors its column
 * (factor_panel: Cholesky of the diagonal block + triangular solve of the rows below), the column travels by RCCL
 * broadcast into every rank's copy of the factor buffer, and each rank applies it to the block columns it owns
 * (factor_update).  factor == factor_begin + [single-GPU recursion] + factor_end. */
int nngp_model_factor_begin(nngp_model* m, void* stream);
int nngp_model_factor_panel(nngp_model* m, int64_t col0, int64_t width, void* stream);
int nngp_model_factor_update(nngp_model* m, int64_t panel_col0, int64_t panel_width, int64_t col0, int64_t width,
                             void* stream);
/* ... the same update for several target block columns at once (HOST array of their first rows / columns, ascending; all of
 * `width` except possibly the last): the block columns one rank owns go out four to a split-float16 launch. */
int nngp_model_factor_update_cols(nngp_model* m, int64_t panel_col0, int64_t panel_width, const int64_t* cols, int32_t ncols,
                                  int64_t width, void* stream);
int nngp_model_factor_end(nngp_model* m, void* stream);
/* float32 factor buffer [n_padded, ld] (lower triangle) and the inverted 128-blocks [n_padded/128][128*128]. */
int nngp_model_factor_buffers(nngp_model* m, float** a32, int64_t* ld, float** dinv);

/* The float64 train-train kernel buffer (device pointer, leading dimension in elements) so the host
 * can all-gather row blocks over RCCL between build_rows and factor. */
int nngp_model_kernel_buffer(nngp_model* m, double** k64, int64_t* ld);
/* ---- the row-sharded layout (round 4, nngp-src_amd/shard32.py): what travels between GPUs is the FLOAT32 factor input
 * (4 N^2 / G bytes per rank, the exchange 8e sizes), and each rank's float64 kernel rows -- the operand of the CG's matrix-vector product
 * and of the covariance's residual product -- stay where nt.batch's device slot built them (reference: nt.batch(kernel_fn,
 * device_count), train.py:166-168).  After nngp_model_build_rows(row_begin, row_end):
 *   _factor_input_rows   a32 rows [row_begin, row_end) = float32(K rows) + reg * shift_scale on the diagonal (lower tiles; shift_scale
 *                        >= 1: a caller that saw clamped pivots redoes the exchange with a larger shift, as nngp_model_factor does alone);
 *                        marks the model's float64 kernel as partial: nngp_model_factor then never rebuilds the input from it
 *   (all-gather of the float32 buffer -- nngp_model_factor_buffers gives it -- e.g. nngp_allgather_rows with NNGP_DTYPE_F32)
 *   _factor_input_complete  every row is in place: the next nngp_model_factor / _factor_begin only adds the padding
 *   _precond             z = (L L^T)^-1 r with the float32 factor (r, z: n doubles): the preconditioner of a CG driven by the caller,
 *   _matvec_rows         q[0 : row_end - row_begin] = K[row_begin : row_end, :] p (no regulariser): its sharded matrix-vector product
 *   _set_alpha           installs the solution (n x ny doubles, device) with the solve's iteration count and relative residual:
 *                        nngp_model_predict(..., NNGP_COV_NONE) then serves means; covariances of this layout are formed by the
 *                        caller from nngp_model_apply_factor, nngp_kernel_build and nngp_gemm_nt_f64 on its row block (shard32.py). */
int nngp_model_factor_input_rows(nngp_model* m, int64_t row_begin, int64_t row_end, double shift_scale, void* stream);
int nngp_model_factor_input_complete(nngp_model* m);
int nngp_model_precond(nngp_model* m, const double* r, double* z, void* stream);
int nngp_model_matvec_rows(nngp_model* m, const double* p, double* q, int64_t row_begin, int64_t row_end, void* stream);
int nngp_model_set_alpha(nngp_model* m, const double* alpha, int32_t iters, double rel_residual, void* stream);
int nngp_model_info(nngp_model* m, nngp_fit_info* info /* host */);
/* Ownership rule of SURVEY.md 8b ("allocated in nngp_fit, never inside timed launch functions") for the predict side: allocates, now,
 * everything an nngp_model_predict of up to `rows` test rows with this cov_mode would otherwise allocate on first use (cross-kernel
 * and right-hand-side blocks, the workspace of the blocked solves, refinement rows, the digit planes and products of the int8
 * residual path, the float32 path's L^T, the full-covariance blocks).  nngp_model_create(m_cap > 0) covers the mean-only part;
 * the Python GPModel calls this with (m_cap, NNGP_COV_DIAG).  nngp_alloc_count(): device allocations the library has made so far
 * in this process (diagnostic: a predict on a reserved model adds none -- tests/test_gpu_api.py). */
int nngp_model_reserve(nngp_model* m, int64_t rows, int32_t cov_mode);
int64_t nngp_alloc_count(void);
/* Live timing of the path's dominant kernel, the split-float16 trailing update of the Cholesky (k_gemm_nt_h3, lower):
 * with the timer on, nngp_model_factor brackets every such launch with a pair of HIP events on the stream it is launched
 * on (bench.py's `roofline` object; no reference counterpart -- the reference prints wall-clock seconds, train.py:176,195).
 * _read waits for the last factorisation and returns, for it: the launches, their summed duration and their algorithmic
 * work, 2 * (updated entries on or below the diagonal) * (panel width) flops, float32-grade (the kernel executes three
 * float16 MFMA products per term).  Any out pointer may be NULL. */
int nngp_model_update_timer(nngp_model* m, int32_t enable);
int nngp_model_update_timer_read(nngp_model* m, int64_t* launches, double* ms_total, double* flops_total);
/* ... and their algorithmic bytes: C read and written once per launch (8 B per updated float32 entry) plus the operands' split
 * rows once (4 B per row and k) -- what a launch must move at least, for the roofline's `traffic` comparison. */
int nngp_model_update_timer_bytes(nngp_model* m, double* bytes_total);
/* Live timing of the posterior's residual products on the int8 matrix pipe (k_gemm_nt_i8s, gemm_i8s.hip; reference op: the
 * covariance of predict_fn(..., compute_cov=True), train.py:157-158): with the timer on, every plane-product launch of
 * nngp_model_predict is bracketed by a pair of HIP events on the caller's stream.  _read waits for them and returns what ran since
 * the last read: the launches, their summed duration, the float64 flops they stand for (2 m n k) and the int8 operations executed
 * (x the number of plane pairs), then starts over.  Any out pointer may be NULL. */
int nngp_model_residual_timer(nngp_model* m, int32_t enable);
int nngp_model_residual_timer_read(nngp_model* m, int64_t* launches, double* ms_total, double* flops_total, double* int8_ops_total);
/* Live timing of the posterior's blocked triangular solves (solve.hip: trsm_rlt_blocks_h3 / trsm_rut_blocks_h3 and their float32
 * forms; reference op: the K_dd^-1 products inside predict_fn(..., compute_cov=True), train.py:157-158): with the timer on, every
 * forward or backward solve of a block of right-hand sides is bracketed by a pair of HIP events on the caller's stream.  _read waits
 * for them and returns what ran since the last read -- the solves, their summed duration and their algorithmic flops (N^2 M each: a
 * triangular matrix against M right-hand sides) -- then starts over.  At most 32 solves are kept between reads. */
int nngp_model_trsm_timer(nngp_model* m, int32_t enable);
int nngp_model_trsm_timer_read(nngp_model* m, int64_t* solves, double* ms_total, double* flops_total);
/* Guard of that path.  The first predict of a fit that forms a level-1 variance with the int8 residual also estimates what the
 * digit pairs it dropped may have cost the variances (|z|_2 x the dropped pairs' random-sign sum, relative to each variance; maximum
 * over the rows) and reads it back once; above 1e-5 the predict is redone with the float64 product and the fit stays on the float64
 * pipe.  ratio: that estimate for the current fit (-1: not measured yet); distrusted: 1 if the fit was taken off the int8 path. */
int nngp_model_residual_floor(nngp_model* m, double* ratio, int32_t* distrusted);
/* alpha = (K + reg I)^-1 Y, [n, ny] f64, copied to a device buffer. */
int nngp_model_alpha(nngp_model* m, double* alpha_out, void* stream);

/* ---- a4: predict_fn(x_test, get, compute_cov) --------------------------------------------------
 * Replaces predict_fn(x_test=..., get=..., compute_cov=True) (train.py:157-158; estimator.py:66-67).
 * x_test: [mt, d] f64, or NULL for x_test=None (predict on the training rows, estimator.py:37-40).
 * mean: [mt, ny] f64.  cov_mode DIAG: var_or_cov is [mt] f64; FULL: [mt, mt] f64; NONE: ignored.
 * Asynchronous on `stream` with two exceptions, both small read-backs followed by a wait for the call's own covariance work:
 * an NTK covariance whose alpha solve took fewer than 6 CG iterations reads the sweeps' 16-byte error estimate
 * (nngp_model_sweep_estimate), and any covariance whose alpha solve took fewer than 3 iterations reads the 4-byte row flag
 * (see nngp_model_set_refine).  The estimate is calibrated for the default two NTK sweeps; at levels >= 3 it describes the
 * first two corrections only and errs on the side of continuing by CG.   */
int nngp_model_predict(nngp_model* m, const double* x_test, int64_t mt, int32_t cov_mode,
                       double* mean, double* var_or_cov, void* stream);

/* Precision level of the posterior covariance (means are float64-accurate at every level).  Z ~ K_td (K + reg I)^-1
 * comes from the float32 factor; each correction sweep costs one [mt, N] x [N, N] float64 MFMA product and one pair
 * of blocked triangular solves and shrinks the error by rho ~ 2e-2 (N = 32768).
 * (max relative error of diag(cov) measured at N = 32768, d = 128, M = 1024; profiles/r1o_var_study.json):
 *   0      float32 only, cov = K_tt - V^T V                                                        1e-2
 *   1      (default) diag: z0.(k + r0) + |L^-1 r0|^2 with r0 = k - A z0: ONE float64 product, a solve pair for    2e-6
 *          z0 and a forward solve for the remainder e0^T A e0 = r0^T A^-1 r0 ~ r0^T M^-1 r0 (the exact identity
 *          k^T A^-1 k = z0.(k + r0) + e0^T A e0 holds for whatever z0 the float32 solves return);
 *          full covariance and the NTK covariance run at level 2
 *   2      one sweep + the second-order formula sym(z_i . (k_j + r_j)) (the diagonal through the quadratic form    4e-7
 *          z^T A z on the lower triangle of K: 1.5 products)
 *   L > 2  L-1 sweeps + the second-order formula                                     2e-10 at L = 3
 * Levels >= 1 are adaptive: the sweeps contract by the spectral radius of I - M^-1 A (M = the float32 factor), which
 * approaches 1 when cond(K + reg I) * eps32 does (small diag_reg, low-dimensional encodings).  After the fixed sweeps
 * predict checks the signs of that -- the alpha solve took >= 7 CG iterations (level >= 2 diag: >= 8; NTK: >= 6, or
 * the two sweeps' own estimate of what they left exceeds 3e-7, see nngp_model_sweep_estimate), or a row's first-order term
 * z.r is too large for its second-order error to be small (a loose lower bound: the backstop when the alpha solve says
 * nothing, e.g. y = 0) -- and then continues the rows by preconditioned
 * CG (one float64 product + one pair of solves per iteration, each row with its own scalars) until every row's step
 * lowers e^T A e by less than 1e-8 of its variance, and forms the covariance again.  The row check needs a 4-byte
 * read-back (the call then waits for its own covariance work); it is made only when the alpha solve took < 3 iterations;
 * nngp_model_cov_iters returns the iterations the last predict spent there (0: the fixed sweeps were enough). */
int nngp_model_set_refine(nngp_model* m, int32_t sweeps);
int nngp_model_cov_iters(nngp_model* m);
/* NTK covariance only: what the last predict expected its two fixed sweeps to leave, from the error energies
 * e = r . M^-1 r the two corrections saw (e2 ~ e1 * (e1 / e0)), worst row: row_rel = sqrt(e2 / |z . k|), the relative
 * energy-norm error of the row, and var_rel = rho / (1 - rho) * |dv| / |var| with rho = sqrt(e1 / e0) and dv the change of
 * the variance under the second correction: the predicted relative variance error (measured within 0.4x .. 120x of the
 * real one wherever that exceeds 1e-8).  var_rel above 3e-7 sent the rows on by CG; costs a 16-byte read-back.  Both
 * are -1 when the last predict did not measure them (NNGP, or the alpha solve's iteration count had already decided).
 * Either pointer may be NULL. */
int nngp_model_sweep_estimate(nngp_model* m, double* row_rel, double* var_rel);

/* Serving mode (the reference's Estimator keeps its factor and solves per query batch: estimator.py:34-67).  Builds the
 * explicit float64 inverse X = (K + reg I)^-1 once per fit -- rows of the identity, 1024 at a time, through the
 * float32 solves + float64 corrections above (to CG convergence when the factor is a weak preconditioner), then
 * symmetrised; N^2 doubles of HBM.  Afterwards nngp_model_predict at levels >= 1 takes Z = K_td X (one float64
 * product, no triangular solves) in place of the float32 solves + sweeps and applies the same second-order formulas
 * (diag: K_tt - (2 z.k - z^T A z); full: K_tt - sym(Z (K_td + R)^T)) -- Z K_dt alone would not do: k^T X k cancels to
 * the variance from terms 1e3..1e6 larger, and no float64 inverse is that accurate.  NNGP posterior only (no-op for
 * NTK, whose covariance has no such formula).  Any later set_train / factor / append drops the inverse.
 * Cost: ~N/1024 covariance-sized solves. */
int nngp_model_prepare_serving(nngp_model* m, void* stream);

/* ---- e: multi-GPU row-block shard -- the nt.batch(kernel_fn, device_count=G) slot (train.py:166-168) -----------
 * One process per GPU.  Rank g builds rows [g*chunk, (g+1)*chunk), chunk = ceil(n / world), with
 * nngp_model_build_rows (or nngp_kernel_build's row range) and ONE in-place RCCL all-gather over xGMI completes K on
 * every rank.  librccl is bound at run time (dlopen): the library loads and everything else works without it.
 * Rendezvous is the caller's business: rank 0 calls nngp_comm_unique_id, ships the 128 bytes to the other ranks by
 * any channel (torch.distributed / MPI / a file), every rank calls nngp_comm_create on its own device.
 *   nngp_allgather_rows  k: [>= world*chunk, ld] device matrix of dtype NNGP_DTYPE_F32/F64, own row block already written;
 *   nngp_bcast           panel broadcast of the block-cyclic Cholesky (nngp_model_factor_panel output) from `root`. */
typedef struct nngp_comm nngp_comm;
int nngp_comm_unique_id(void* id128 /* host, 128 bytes */);
int nngp_comm_create(nngp_comm** out, const void* id128 /* host */, int32_t world, int32_t rank);
int nngp_comm_destroy(nngp_comm* c);
const char* nngp_comm_library(void); /* which librccl was bound ("" if none) */
int nngp_allgather_rows(void* k, int64_t n, int64_t ld, int32_t dtype, nngp_comm* c, void* stream);
int nngp_bcast(void* buf, int64_t count, int32_t dtype, int32_t root, nngp_comm* c, void* stream);

/* ---- N2: native query-line encoder (host code; replaces the per-line Python of estimator/encoder.py:59-97,187-250
 * and QuerySampler.py:157-221) --------------------------------------------------------------------------------------
 * schema_text, one directive per line:  "table <name>" | "num <column> <min> <max>" | "cat <column> <num_categories>".
 * mode 0: multi-join lines "t1,t2@preds_t1@preds_t2@t1,t2,col#...[@card]"; mode 1: single-table lines
 * "COL,upper,lower#...@card".  nngp_encoder_encode parses '\n'-separated lines into HOST buffers x_out [max_lines, dim]
 * (float64) and, when with_card != 0, card_out [max_lines]; bit-identical to the reference encoder. */
typedef struct nngp_encoder nngp_encoder;
int nngp_encoder_create(nngp_encoder** out, const char* schema_text, int32_t chunk_size, int32_t mode);
int nngp_encoder_destroy(nngp_encoder* e);
int32_t nngp_encoder_dim(const nngp_encoder* e);
int nngp_encoder_encode(const nngp_encoder* e, const char* text, int64_t text_len, int32_t with_card, double* x_out,
                        double* card_out, int64_t max_lines, int64_t* n_lines_out);

/* ---- building blocks exported for parity tests and the integration notes ----------------------
 * Blocked lower Cholesky of a float32 matrix in place (n multiple of 128, ld >= n).  dinv: workspace of
 * (n/128) * 128*128 floats receiving the inverses of the diagonal blocks.  clamped: device int32.   */
int nngp_potrf_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, void* stream);
/* C[M,N] = beta*C + alpha * A[M,K] B[N,K]^T on float32 MFMA (all of M, N, K multiples of 128).
 * lower_only != 0: only tiles on or below the diagonal are touched (SYRK-style, M == N).            */
int nngp_gemm_nt_f32(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb,
                     int64_t m, int64_t n, int64_t k, float alpha, float beta, int32_t lower_only,
                     void* stream);
/* Same product with both operands split into two float16 planes (a*scale = hi + lo) and three float16 MFMA products
 * per term, float32 accumulation: float32-grade results at 3/16 of the float32 matrix-pipe cost.  scale: power of two
 * with max|a*scale|, max|b*scale| <= 2^14.  M, N multiples of 128, K of 32.  Allocates and frees its own split
 * workspace (the Cholesky keeps one per model); a test / integration primitive like the two above.      */
int nngp_gemm_nt_h3(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb,
                    int64_t m, int64_t n, int64_t k, float alpha, float beta, float scale, int32_t lower_only,
                    void* stream);
/* C[M,N] = beta*Cin + alpha * A[M,K] B[N,K]^T on float64 MFMA (M, N multiples of 128, K of 16; Cin may be C). */
int nngp_gemm_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                     const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                     void* stream);
/* The same float64 product in float64 GRADE on the int8 matrix pipe (gemm_i8s.hip): every row of A and of B is scaled by its own
 * power of two and cut into `slices_a` / `slices_b` exact balanced 8-bit digit planes; the plane pairs with ia + ib <= cut are
 * multiplied exactly (int32 accumulation) and combined in float64.  Error: ~256^-(cut+2) and 256^-slices of (row maximum of A) x
 * (row maximum of B) x K per entry -- with 5 x 5 planes and cut = 4, about 2^-40 of that bound.  This is the residual product of the
 * posterior (reference: predict_fn(..., compute_cov=True), train.py:157-158) as a test / integration primitive: it allocates and
 * frees its own planes.  M, N multiples of 128, any K >= 1; lda, ldb even; 2 <= slices <= 7 (7 x 7 planes with cut 6, 28 products, is float64 grade proper).  Cin may be C or NULL (beta = 0). */
int nngp_gemm_nt_i8s(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                     const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                     int32_t slices_a, int32_t slices_b, int32_t cut, void* stream);
/* y = (A + diag_add I) x for a symmetric n x n float64 matrix stored in full (leading dimension lda), read from its LOWER
 * triangle only: the product of the alpha CG (half the bytes of a plain GEMV, fixed summation order).  Any n >= 1; what
 * lies beyond row / column n in a padded buffer is never read. */
int nngp_symv_f64(const double* a, int64_t lda, int64_t n, const double* x, double* y, double diag_add, void* stream);
/* Pool scoring of the active-learning loop on the device (reference: active/ActiveLearner.py:43-55, active_test):
 * score_i = sqrt(max(var_i, 0)) / max_j mean[j * ny] (np.max(pred_mean, 0) of the single output column).
 *   biased = 0: the `count` largest scores in ascending order of score -- np.argsort(score)[-count:];
 *   biased = 1: `count` indices drawn without replacement with probability proportional to the score, AS THE REFERENCE DRAWS THEM --
 *               jax.random.choice(PRNGKey(seed), m, (count,), replace=False, p=score / sum(score)) of jax 0.3.23 (nngp.yaml:78):
 *               Gumbel top-k, argsort(-gumbel - log p)[:count], with u_i from Threefry-2x32-20 on the counter pair (i, m + i) under the
 *               key (seed >> 32, seed & 0xffffffff) -- restated in nngp-src_amd/jaxrand.py, pinned by the Random123 known-answer
 *               vectors; the same draw on every device and in the host build, in draw order.
 * mean [m, ny], var [m], indices [count] are device pointers (host pointers in the host build): only `count` indices have to
 * travel to the host.  count <= m. */
int nngp_pool_select(const double* mean, int64_t m, int32_t ny, const double* var, int64_t count, int32_t biased,
                     uint64_t seed, int64_t* indices, void* stream);
/* The float32 factor of a fitted model applied to a block of right-hand-side ROWS, as the posterior does it (reference:
 * the cho_solve inside predict_fn, train.py:157-158 through neural-tangents' nt.predict): b [rows, n] float32, row stride n
 * (n = the model's training rows; rows a multiple of 128 is not required), overwritten with
 *   mode 0:  b L^-T            (the forward half: what the variance's L^-1 k needs)
 *   mode 1:  b (L L^T)^-1      (both halves: the preconditioner M^-1 of the refinement sweeps)
 * through the blocked solves the model would take for a block of that size (inverted diagonal blocks, the large updates on the
 * float16 pipe from 256 rows and N = 8192 on).  L is the factor of float32(K) + reg I: compare against a triangular solve with
 * the factor itself (nngp_model_factor_buffers), not against the float64 kernel. */
int nngp_model_apply_factor(nngp_model* m, float* b, int64_t rows, int32_t mode, void* stream);
/* Round 5: from 128 right-hand-side rows and N = 2048 on, each of those blocked solves is ONE persistent launch whose workgroups
 * draw work items (128 x 128 tiles of the diagonal products and of the updates, 16-row split passes) from a ticket counter and wait
 * on device counters for the items they depend on (csrc/trsm_tickets.hip).  The order of the tickets is fixed on the host, and the
 * launch cannot hang if every item's dependencies hold LOWER tickets.  This entry point returns that table for a shape -- row_tiles
 * tiles of 128 rows, block_cols block columns of 1024 (the last one tail_tiles <= 8 tiles of 128 wide), forward (0) or backward (1)
 * solve, as scheduled for `workers` resident workgroups, with (merged = 1) or without the chain updates formed straight from the
 * previous block's split rows (type 5) -- as four int32 per item {type | panels << 4, row tile, column tile or
 * 32-row group, block column}; type 0 = split of an updated block, 1 = diagonal-product tile, 2 = split of a solved block,
 * 3 = update of one 128 x 128 tile, 4 = update of the 2 x 2 group of tiles (2 r, 2 r + 1) x (2 c, 2 c + 1) as one 256 x 256 tile
 * (an update's `panels` finished block columns start at `block column` and go back in solve order), 5 = the last update of a
 * tile from the split rows of the block column right before its own.  Host memory, no GPU
 * needed; items == NULL only counts.  tests/test_host.py replays the tables against the kernel's own wait conditions. */
int nngp_trsm_ticket_order(int32_t row_tiles, int32_t block_cols, int32_t tail_tiles, int32_t backward, int32_t workers, int32_t merged,
                           int32_t* items /* host, 4 * cap */, int64_t cap, int64_t* count /* host */);
/* The same table as the product launches it on MI355X: `queues` = 8 tables, one per XCD -- a workgroup draws from the table of the XCD
 * it runs on (and from the others' once its own has run out), so that the items which read one pair of row tiles' split rows meet in one
 * L2.  `items` is the GLOBAL start order of the host's schedule (every item's dependencies precede it there), queue_of[i] the table item
 * i is in; a table holds its items in that order.  Hang freedom: the earliest unfinished item of the global order is at the head of its
 * table, whose workgroups hold nothing later than it.  queues = 1: nngp_trsm_ticket_order.  tests/test_host.py replays both properties
 * and runs the draw protocol with random item durations and worker counts. */
int nngp_trsm_ticket_queues(int32_t row_tiles, int32_t block_cols, int32_t tail_tiles, int32_t backward, int32_t workers, int32_t merged,
                            int32_t queues /* 1 or 8 */, int32_t* items /* host, 4 * cap */, int32_t* queue_of /* host, cap */, int64_t cap,
                            int64_t* count /* host */);
/* B[m, n] <- B L^-T using the factor and dinv from nngp_potrf_f32 (m, n multiples of 128). */
int nngp_trsm_rlt_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv,
                      int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNGP_HIP_H */
