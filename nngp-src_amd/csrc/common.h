// Internal helpers shared by the HIP translation units of libnngp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <string.h>

#include "../../include/nngp_hip.h"

namespace nngp {

constexpr int TB = 128;  // Cholesky / GEMM tile edge; all float32 device matrices are padded to it

void set_error(const char* fmt, ...);
void note_alloc();  // counts a device allocation (nngp_alloc_count)
// Timing-experiment switches.  They exist ONLY in libnngp_hip_knobs.so (built with -DNNGP_TIMING_KNOBS for scripts/ and the
// A/B tests, entry point nngp_debug_set); in the product library NNGP_KNOB(i) is the constant 0 and every branch on it is
// compiled out.  Keys:
//   0  bit mask: 1, 2, 4 leaf-kernel ablations (wrong results); 1, 2, 8 split-float16 GEMM ablations (no loads / no MFMA /
//      no C traffic; wrong results); 64 = start the alpha CG ahead, without waiting on the host (nngp_model_solve);
//      128 = never stop the alpha CG early (no mean correction through the covariance rows)
//   3  (also) 40 + e = stopping tolerance 10^-e of the early-stopped alpha CG (scripts/partial_tol_study.py)
//   1  block-column width of the look-ahead Cholesky (default 1024)
//   2  1 = no look-ahead (recursion on one stream); 2 = trailing updates on the float32 MFMA; 3 = panel-solve product on the
//      float32 MFMA (trailing updates stay on the float16 pipe); 4 = round-1 panel solves (GEMM recursion instead of trsm_panel.hip);
//      5 = fused panel solves entirely in float32 (no float16-pipe products)
//   3  1 = slower leaf variant; 3 = kernel build with the float64-MFMA Gram product; 10 + n = first n block columns of the
//      Cholesky on the float32 MFMA; 20 + c = float32 lead of 128 c columns in the first trailing update (default 256)
//   4  compute units the persistent split-float16 grid leaves free (default 32 in the Cholesky); panel CUs of the CU-mask
//      experiment
//   5  2 = CU-masked streams; 9 = no split-K in the float64 GEMM; 8 = plain blockIdx tile order in the float64 GEMM; 10 + v = tile-block shape of the split-float16 GEMM
//   6  >= 128: block size of the inverted diagonal blocks (default 1024); 1 = fixed covariance sweeps only and no retry
//      of a broken-down factorisation; 2..30 = e: row-flag threshold 10^-e of the adaptive covariance
//   7  1 = 128-wide recursion in the posterior solves; any non-zero value = float32 solve path; 3 = CG solve in stream
//      order instead of deferred
//   5  (also) 1 = old level-1 variance formula (full float64 residual + preconditioned remainder)
//   5  (also) 50..57 int8 residual path (api.hip: use_i8s, ensure_i8s); round 4: 58 = five digit planes of z instead of three rounded ones;
//      59 = row statistics and float32 copy in passes of their own instead of the fused combination; 60 = K chunks of 16384 always;
//      61 = the factor's inverted blocks built in line (not beside the first predict's cross-kernel build); 62 = digit planes of K row by row; 63 = per-layer ReLU recursion in the kernel build (no composite map); 64 = seven z planes in FINE products; 65 = an NTK model's NNGP kernel always in a build of its own
//   0  (also) 32 = alpha CG runs in stream order inside nngp_model_solve, early-stopped (resumed by whoever needs alpha itself)
//   8  round 4, grouped Cholesky (set BEFORE the model is created): bit 1 = the schedule with the panel solves off the update stream
//      (potrf_lookahead_grouped_v4; needs its extra streams); with it: 2 = bulk panel solves on the panel stream itself; 4 = early part
//      of the next group's first diagonal-block update on its own stream; 8 = helper grids behind the bulk solves; 16 = a far chunk's
//      next-column region in a launch of its own; 32 = finished diagonal blocks inverted on a side stream under the last block columns
//   9  round 4, posterior solves: 2 = 2048-wide inverted blocks / two 1024-column panels per step of the blocked solves from np = 8192 on
//      (measured: posterior -0.5 ms, block inverses +1.1 ms at N = 32768 -- off)
//  10  block columns from the end where key 8 = 32 issues the inverses; 11  priority of that side stream; 12  print the schedule's stream end times
//  13  compute units the posterior solves' update grids leave to the alpha CG; 14  2 = CG iterations replayed from a hipGraph
#ifdef NNGP_TIMING_KNOBS
extern std::atomic<int> g_knobs[16];
#define NNGP_KNOB(i) (nngp::g_knobs[i].load(std::memory_order_relaxed))
#else
#define NNGP_KNOB(i) 0
#endif

#define NNGP_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            nngp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

#define NNGP_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            nngp::set_error(__VA_ARGS__);  \
            return -2;                     \
        }                                  \
    } while (0)

#define NNGP_TRY(expr)          \
    do {                        \
        int _rc = (expr);       \
        if (_rc != 0) return _rc; \
    } while (0)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Squared layer parameters in kernel-argument form.
struct ArchDev {
    int n_dense;
    double w2[NNGP_MAX_DENSE];
    double b2[NNGP_MAX_DENSE];
};

int make_arch_dev(const nngp_arch* arch, ArchDev* out);

// ---- kernel_build.hip ----
struct BuildArgs {
    const double* x1;
    const double* x2;   // == x1 when symmetric
    const double* q1;   // |x1_i|^2 / d
    const double* q2;
    int64_t n1, n2;
    int d;
    int64_t row_begin, row_end;
    int sym;            // 1: x2 == x1, only tiles with col-tile <= row-tile are computed, then mirrored
    double* nngp64;     // any of the four outputs may be NULL
    double* ntk64;
    float* nngp32;
    float* ntk32;
    int64_t ld64, ld32;
    double diag_add_nngp32;  // added on the diagonal of the float32 outputs (regulariser), sym only
    double diag_add_ntk32;
    int lower32;        // 1: float32 outputs only get the lower triangle (factorisation input)
    const double* comp; // set by launch_kernel_build itself: the composite ReLU map's table (kernel_build.hip), or NULL
    int no_comp;        // 1: keep the per-layer recursion (an NTK model's NNGP kernel: bit-identical whether it comes from the fit's own build or a build of its own)
};
int launch_row_sqnorm(const double* x, int64_t n, int d, double* q, hipStream_t s);
int launch_diag_from_q(const double* q, int64_t n, const ArchDev& arch, double* dn, double* dt, hipStream_t s);
int launch_kernel_build(const BuildArgs& a, const ArchDev& arch, hipStream_t s);

// ---- gemm_f32.hip ----
// tri: the square B is lower (1) / upper (2) triangular -- every column tile only walks the k range where its rows of B are non-zero
int launch_gemm_nt_f32(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb,
                       int64_t m, int64_t n, int64_t k, float alpha, float beta, bool lower_only, hipStream_t s, int tri = 0);

int launch_gemm_nt_f32_batched(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb,
                               int64_t m, int64_t n, int64_t k, float alpha, float beta, bool lower_only, int batch,
                               int64_t stride_c, int64_t stride_a, int64_t stride_b, hipStream_t s, int tri = 0);

// ---- gemm_h3.hip: float32-grade products on the float16 matrix pipe (two float16 planes per operand) ----
int launch_split_rows(const float* p, int64_t ld, int64_t rows, int64_t k, float scale, char* out, int64_t out_ld,
                      hipStream_t s);
int launch_split_rows_rowscale(const float* src, int64_t ld_src, int64_t rows, int64_t k, float* copy_dst, int64_t ld_copy,
                               char* out, int64_t out_ld, float* row_inv, hipStream_t s, int64_t pan_stride = 0);  // k <= 2048: panels of 1024 columns
int launch_split_lower_t(const float* l, int64_t ld, int64_t n, int64_t bs, float scale, char* out, int64_t col_stride,
                         hipStream_t s);
int launch_gemm_nt_h3(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t m, int64_t n, int64_t k,
                      float alpha, float beta, bool lower_only, int64_t diag_shift, int* counters, int reserve_cus,
                      hipStream_t s, const float* row_alpha = nullptr);  // row_alpha: per-row factor of the product
// up to four regions of C per launch, all with the same panels: C(row0 + i, col0 + j) for i < m, j < n (lower_only: j <= i + shift);
// their A rows start row0 rows, their B rows col0 rows into the split rows at a / b
struct H3RegionSpec {
    int64_t row0, col0, m, n, shift;
};
int launch_gemm_nt_h3r(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t pstride, int npanels, int64_t lead,
                       const H3RegionSpec* spec, int nreg, int64_t k, float alpha, float beta, bool lower_only, int* counters,
                       int reserve_cus, hipStream_t s, const float* row_alpha = nullptr, int helper = 0);
int launch_gemm_nt_h3x(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t pstride, int npanels, int64_t lead,
                       int64_t m, int64_t n, int64_t k, float alpha, float beta, bool lower_only, int64_t diag_shift, int* counters,
                       int reserve_cus, hipStream_t s, const float* row_alpha = nullptr);  // sum over npanels panels, one pass over C

// ---- potrf.hip ----
int launch_potrf_leaf(float* a, int64_t ld, float* dinv_block, int32_t* clamped, float pivot_floor, hipStream_t s);
int potrf_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, hipStream_t s);
int trsm_rlt_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t n,
                 hipStream_t s);

// ---- trsm_panel.hip: b [m, w] <- b L^-T in one launch (w <= 1024), optionally with the rows' float16 split copy ----
// planes_t != NULL: also the rows' transposed split copy (SplitWork::planes_t; tstride = its block-row stride, row0 / col0 = the global
// row of b's first row / global column of the panel's first column)
int launch_trsm_panel_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t w,
                          char* planes, int64_t ldp, float scale, hipStream_t s, char* planes_t = nullptr, int64_t tstride = 0,
                          int64_t row0 = 0, int64_t col0 = 0);
int launch_split_diag_frag(const float* a, int64_t ld, int64_t w, float scale, char* out, const float* dinv, float* dfrag,
                           float* dscale, hipStream_t s);  // dfrag: split halves of the inverted 128-blocks in fragment order; dscale: their rows' 1 / scale
int launch_trsm_panel_h3(float* b, int64_t ldb, int64_t m, const char* lfrag, const float* dfrag, const float* dscale, int64_t w,
                         char* planes, int64_t ldp, float scale, hipStream_t s, char* planes_t = nullptr, int64_t tstride = 0,
                         int64_t row0 = 0, int64_t col0 = 0);


struct SplitWork {   // float16-split copies of the factor (gemm_h3.hip); one per model
    char* planes = nullptr;    // [ncols][rows_cap][k_cap] x 4 bytes: L by block column k, row index = global row
    int64_t rows_cap = 0, k_cap = 0, col_stride = 0;  // col_stride in bytes
    float scale = 1.0f;        // power of two with max |L_ij| * scale < 2^15: |L_ij| <= sqrt(max_i A_ii).  (A scale per
                               // block column from its measured maximum changed nothing: the float16 pipe keeps
                               // subnormal low planes exact; accuracy is limited by accumulator truncation instead.)
    int* counters = nullptr;   // 8 work counters of the persistent GEMM grid (one per XCD)
    bool l_ready = false;      // every block column of the current factor has been written
    int64_t split_panel = -1;  // block-column offset whose split copy was last written by the block-column ABI
    char* ldiag = nullptr;     // 2 x 4 k_cap^2 bytes: split copy of the diagonal block being solved against, in the fragment order of k_trsm_panel_h3
    float* dfrag = nullptr;    // 2 x k_cap x 128 floats: its inverted 128-blocks in the same fragment order (round 5: as split halves)
    float* dscale = nullptr;   // 2 x k_cap floats: 1 / scale of every row of those blocks
                               // (two of each, by block-column parity: the bulk rows of block column k are still being solved on the
                               // solve stream of the grouped Cholesky while the panel stream prepares block column k + 1)
    char* planes_t = nullptr;  // same shape: L^T by block row j, rows r < j*k_cap (built on the first posterior solve)
    bool lt_ready = false;
    char* planes_b = nullptr;  // [mb_cap + 256][k_cap] x 4 bytes: the right-hand-side block of a blocked solve; round 4: one such panel per
                               // 1024 columns of a solve step, col_stride bytes apart (the split-float16 kernel walks the panels of both
                               // operands with ONE stride, and the factor's is col_stride)
    int b_panels = 1;          // panels planes_b has room for
    int solve_reserve = 0;     // compute units the blocked solves' update launches leave free (api.hip sets it while the alpha CG runs beside them)
    float* row_inv = nullptr;  // [mb_cap] per-row 1/scale of planes_b
    int64_t mb_cap = 0;
};

struct LookAhead {  // streams and events of the look-ahead Cholesky (one per model)
    static constexpr int kMaxSteps = 64;
    hipStream_t panel = nullptr, update = nullptr;
    hipStream_t bulk = nullptr;  // round 4: the panel solves of the rows beyond the next diagonal block (above the update stream's priority)
    hipStream_t aux = nullptr;   // round 4: the early part of a group's product onto the next group's first diagonal block
    hipStream_t side = nullptr;  // round 4: lowest priority -- the inverses of finished diagonal blocks under the chain-bound last block columns
    hipEvent_t ev_side_done = nullptr;
    int prio_levels = 0;         // stream priority levels of the device (the side stream needs three)
    bool masked = false;  // streams own disjoint CU sets (hipExtStreamCreateWithCUMask)
    hipEvent_t ev_in = nullptr, ev_panel_done = nullptr, ev_update_done = nullptr;
    hipEvent_t ev_panel[kMaxSteps] = {}, ev_col[kMaxSteps] = {};
    hipEvent_t ev_chunk[kMaxSteps] = {}, ev_helper[kMaxSteps] = {};  // grouped form: a far chunk may start / its helper grid has finished
    // round 4 (panel solves off the update stream): the update stream up to the far chunk of step k; up to the near update of block
    // column k; the first rows / the other rows of block column k are solved; its diagonal block is split; the early part of the next
    // group's first diagonal-block update is in place
    hipEvent_t ev_far[kMaxSteps] = {}, ev_near[kMaxSteps] = {}, ev_tc[kMaxSteps] = {}, ev_tb[kMaxSteps] = {}, ev_split[kMaxSteps] = {},
               ev_gp[kMaxSteps] = {};
    hipEvent_t ev_c1[kMaxSteps] = {};     // the far chunk's launch over the next block column has been issued (and everything before it)
    hipEvent_t ev_chain[kMaxSteps] = {};  // not owned: ev_c1[k] or ev_far[k], whichever the chain's next links wait for
    hipEvent_t ev_bulk_done = nullptr;
    // live timing of the split-float16 trailing updates (nngp_model_update_timer): event pairs around each launch
    static constexpr int kMaxTimed = 320;  // split-float16 update launches of one factorisation (grouped form: ~3 per block column)
    bool time_updates = false;
    hipEvent_t tu0[kMaxTimed] = {}, tu1[kMaxTimed] = {};
    int tu_count = 0;              // launches of the last factorisation
    double tu_flops[kMaxTimed] = {};
    double tu_bytes[kMaxTimed] = {};  // algorithmic bytes: C read + written once (8 B per updated entry) + the split rows of the operands once
};
int lookahead_create(LookAhead** out);
void lookahead_destroy(LookAhead* la);
struct TriInv;
int potrf_lookahead_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor,
                        LookAhead* la, SplitWork* sw, hipStream_t user, TriInv* ti = nullptr);  // ti: invert finished diagonal blocks on the way
constexpr int64_t kLookAheadNb = 1024;  // block-column width of the look-ahead Cholesky
constexpr int kLookAheadGroup = 4;      // block columns per deep-K far update (potrf.hip, grouped form)
int potrf_panel_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, int64_t o,
                    int64_t w, hipStream_t s, SplitWork* sw = nullptr);
int potrf_update_f32(float* a, int64_t n, int64_t ld, int64_t po, int64_t pw, int64_t o, int64_t w, hipStream_t s,
                     SplitWork* sw = nullptr);  // sw: run the update on the float16 pipe (split copy of the panel kept in sw)
int potrf_update_cols_f32(float* a, int64_t n, int64_t ld, int64_t po, int64_t pw, const int64_t* cols, int ncols, int64_t w,
                          hipStream_t s, SplitWork* sw = nullptr);  // several target block columns, four to a split-float16 launch
int trsm_rut_f32(float* b, int64_t ldb, int64_t m, const float* lt, int64_t ldl, const float* dinvt, int64_t n,
                 hipStream_t s);

// ---- gemm_i8s.hip: float64-grade products on the int8 matrix pipe (exactly sliced operands) ----
struct I8Plan {       // which digit-plane pairs are multiplied, grouped by diagonal ia + ib (largest first)
    int ndiag, npairs;
    int dstart[9];    // first pair of diagonal dd; dstart[ndiag] = npairs
    int pa[32], pb[32];
};
struct I8Planes {     // digit planes of one float64 kernel matrix: [6][rows][np_cap] int8, row stride np_cap bytes
    int8_t* planes = nullptr;
    double* scale = nullptr;     // [np_cap + 1] row scales; the last entry: the sum of their squares (error estimate)
    bool ready = false;          // the planes belong to the current matrix
    int ns_done = 0;             // planes per row they were cut into
    int alloc_planes = 0;        // planes there is room for
};
struct I8Work {       // one per model (api.hip): planes of K (and of the NNGP kernel beside an NTK fit), of a block of right-hand-side rows
    I8Planes k, aux;
    int64_t k_rows = 0;          // rows per plane (np_cap up to the next multiple of 256)
    int8_t* zplanes = nullptr;   // [6][z_rows + 256][np_cap]
    double* zscale = nullptr;    // [z_rows]
    int64_t z_rows = 0;
    int z_planes = 0;            // planes zplanes / partial have room for
    int32_t* partial = nullptr;  // [chunks][diagonals][z_rows][np_cap] exact plane products
    double* rowpart = nullptr;   // [rowpart_rows][np_cap / 1024][4] partial row statistics of the fused combination (I8Fuse)
    int64_t rowpart_rows = 0;
    int* counters = nullptr;     // work counters of the persistent grid
    int ns_k = 5, ns_z = 5, cut = 4;
    // live timing of the plane-product launches since the timer was last read (nngp_model_residual_timer_read)
    static constexpr int kMaxTimed = 64;
    bool timed = false;
    hipEvent_t t0[kMaxTimed] = {}, t1[kMaxTimed] = {};
    int t_count = 0;
    double t_ops[kMaxTimed] = {};   // executed int8 multiply-adds x 2
    double t_flops[kMaxTimed] = {}; // algorithmic: the float64 product they stand for, 2 m n k
};
int i8s_plan(int nsa, int nsb, int cut, I8Plan* pl);
int64_t i8s_chunks(int64_t k, const I8Plan* pl = nullptr);  // K chunks a product over k columns is cut into: of <= 16384, or <= 40960 when no diagonal
                                                                // of the plan has more than 3 pairs (int32 accumulators); without a plan: the larger count
int launch_i8s_diag_bound_scale(const double* src, int64_t ld, int64_t n, double* scale, hipStream_t s);
int launch_i8s_scale_sqsum(const double* scale, int64_t n, double* out, hipStream_t s);
int launch_i8s_floor_ratio(const double* z, int64_t ld, int64_t rows, int64_t cols, const double* var, const I8Plan& pl, int nsa, int nsb,
                           const double* sk2, unsigned long long* out, hipStream_t s);
// the same estimate from per-row statistics of z (|z|_2^2, max |z|: launch_rowdot_f64's zstat) instead of a pass over z
int launch_i8s_floor_ratio_rows(const double* zstat, int64_t rows, const double* var, const I8Plan& pl, int nsa, int nsb, const double* sk2,
                                unsigned long long* out, hipStream_t s);
int launch_i8s_slice_rows(const double* src, int64_t ld, int64_t rows, int64_t cols, int ns, const double* scale_in, double* scale_out,
                          int8_t* planes, int64_t ldp, int64_t pstride, hipStream_t s, double* writeback = nullptr);  // scale_in NULL: scale by the row maxima
int launch_i8s_slice_sym(const double* src, int64_t ld, int64_t n, int ns, const double* scale, int8_t* planes, int64_t ldp,
                         int64_t pstride, hipStream_t s);  // all planes of a bitwise symmetric matrix, every entry read once
int launch_gemm_nt_i8s(int32_t* partial, int64_t ldc, int64_t slab, const int8_t* a, int64_t lda, int64_t sa, const int8_t* b,
                       int64_t ldb, int64_t sb, const I8Plan& pl, int64_t m, int64_t n, int64_t k, int* counters, int reserve_cus,
                       hipStream_t s);
struct I8Fuse {           // what the combination pass of a level-1 residual also delivers (k_i8s_combine<true>)
    float* out32 = nullptr;   // float32 copy of the result, [rows, ld32]
    int64_t ld32 = 0;
    double* part = nullptr;   // [rows][i8s_col_blocks(cols)][4] partial row statistics, summed by launch_i8s_rowstat_finish
};
int64_t i8s_col_blocks(int64_t cols);
int launch_i8s_combine(double* out, int64_t ldo, const double* cin, int64_t ldcin, double beta, double alpha, const double* g,
                       int64_t ldg, double gamma, const int32_t* partial, int64_t ldc, int64_t slab, int nchunk, int ndiag,
                       const double* sa, const double* sb, int64_t rows, int64_t cols, hipStream_t s, const I8Fuse* fuse = nullptr);
// var = base - sum z.(k + r), delta = sum z.r, zstat = (|z|^2, max |z|) per row from the partial statistics of the fused combination
int launch_i8s_rowstat_finish(const double* part, int64_t cols, int64_t rows, const double* base, double* var, double* delta,
                              double* zstat, hipStream_t s);

// ---- gemm_f64.hip ----
int launch_gemm_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                       const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                       hipStream_t s, int kmode = 0);

// ---- solve.hip ----
int launch_transpose_blocks_f32(const float* src, float* dst, int64_t bs, int64_t count, hipStream_t s);
// Inverted diagonal blocks (size bs) of the float32 factor for the blocked triangular solves.
struct TriInv {
    int64_t bs = 0;
    float* tinv = nullptr;     // [nblk][bs*bs]  T_J = L_JJ^-T (upper triangular)
    float* xinv = nullptr;     // [nblk][bs*bs]  X_J = L_JJ^-1 (lower triangular)
    float* partial = nullptr;  // [bs/128][np]   column partial sums of the backward sweep
    float* tmp = nullptr;      // [bs]
    int64_t done_blocks = 0;   // blocks [0, done_blocks) of the CURRENT factor are already inverted (set by the look-ahead Cholesky,
                               // consumed -- and reset -- by triinv_build)
};
int64_t triinv_block(int64_t np);
int triinv_build(const float* l, int64_t ld, const float* dinv, int64_t np, TriInv& ti, hipStream_t s);
int triinv_build_range(const float* l, int64_t ld, const float* dinv, int64_t np, TriInv& ti, int64_t j0, int64_t j1, hipStream_t s);
int trsm_rlt_blocks_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ld, const TriInv& ti, int64_t np,
                        float* tmp, hipStream_t s);
int trsm_rut_blocks_f32(float* b, int64_t ldb, int64_t m, const float* lt, int64_t ld, const TriInv& ti, int64_t np,
                        float* tmp, hipStream_t s);
// the same two solves with the large updates on the float16 pipe (split copies of L / L^T in `sw`)
int trsm_rlt_blocks_h3(float* b, int64_t ldb, int64_t m, const float* l, int64_t ld, const TriInv& ti, int64_t np,
                       float* tmp, const SplitWork& sw, hipStream_t s);
int trsm_rut_blocks_h3(float* b, int64_t ldb, int64_t m, const float* lt, int64_t ld, const TriInv& ti, int64_t np,
                       float* tmp, const SplitWork& sw, hipStream_t s);
// in-place solves L x = b / L^T x = b on a float32 vector of length np
int trsv_forward_f32(const float* l, int64_t ld, const TriInv& ti, int64_t np, float* b, float* x, hipStream_t s);
int trsv_backward_f32(const float* l, int64_t ld, const TriInv& ti, int64_t np, float* b, float* x, hipStream_t s);
// ---- trsm_tickets.hip: one persistent, ticket-ordered launch per blocked solve (round 5) ----
struct TrsmTickets;
int tk_create(TrsmTickets** out, int64_t np_cap, int64_t m_cap);  // 1: no room / not applicable (*out stays NULL)
void tk_destroy(TrsmTickets* tk);
int tk_prepare_inverses(TrsmTickets* tk, const TriInv& ti, int64_t np, const float* l, int64_t ld, hipStream_t s);  // split copies of the inverted 1024-blocks + the merged chain operands from the float32 factor l
bool tk_inverses_ready(const TrsmTickets* tk);
void tk_invalidate_inverses(TrsmTickets* tk);
bool tk_usable(const TrsmTickets* tk, int64_t m, int64_t np);
int tk_solve(TrsmTickets* tk, float* b, int64_t ldb, int64_t m, int64_t np, const SplitWork& sw, bool backward, hipStream_t s, int reserve_cus = 0);
int tk_poll_error(TrsmTickets* tk, bool wait);
int tk_order_export(int mt, int nb, int tail_ct, int backward, int workers, int merged, int queues, int32_t* out, int32_t* queue_of, int64_t cap,
                    int64_t* count);  // error word of a launch that gave up waiting (0: none seen)
// y = (A + diag_add I) x for a SYMMETRIC n x n float64 matrix stored in full, reading only its lower triangle (half the bytes
// of launch_gemv_f64); part: [ceil(n/128)][np] workspace, np = ceil(n/128)*128.  Deterministic (fixed summation order).
int launch_symv_f64(const double* a, int64_t lda, int64_t n, const double* x, double* y, double diag_add, double* part,
                    int64_t np, hipStream_t s);
int launch_gemv_f64(const double* a, int64_t lda, int64_t rows, int64_t cols, const double* x, int64_t incx,
                    double* y, int64_t incy, double diag_add, hipStream_t s);
struct PcgWork {
    double* r; double* z; double* p; double* q; double* xcol; double* bcol;
    float* f32a; float* f32b; float* f32c;
    double* dot_part = nullptr;   // [32] partial sums of the dot-product kernel
    unsigned* dot_ctr = nullptr;  // its arrival counter (wraps to 0 by itself)
    double* symv_part = nullptr;  // [np / 128][np]: per-tile partial results of the symmetric matrix-vector product
    int64_t symv_np = 0;          // its np (0: not allocated -> plain GEMV)
    double* scal;       // device scalars [32]: 0..4 CG scalars, 6..7 trace / max of the diagonal, 8.. CG residual history
    double* host_scal;  // pinned host [32]
    // One CG iteration (from the second on: ~170 launches at N = 32768) captured once as a hipGraph and replayed (round 4); the key says
    // what the captured launches were built for -- any change of it (another size, other buffers) captures again.
    hipGraphExec_t iter_graph = nullptr;
    const void* graph_key[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t graph_dims[4] = {0, 0, 0, 0};
    double graph_reg = 0.0;  // (the regulariser is a by-value kernel argument)
};
int pcg_begin(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32, const TriInv& ti,
              int64_t np, const double* bcol, double* xcol, PcgWork& w, int ahead, hipStream_t s);
int pcg_finish(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32, const TriInv& ti,
               int64_t np, double* xcol, PcgWork& w, int ahead, int max_iters, double tol, int* iters_out,
               double* relres_out, hipStream_t s, bool resume = false);
int pcg_solve(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32,
              const TriInv& ti, int64_t np, const double* bcol, double* xcol, PcgWork& w, int max_iters,
              double tol, int* iters_out, double* relres_out, hipStream_t s);
// z = (L L^T)^-1 r with the float32 factor (the CG's preconditioner on its own: the row-sharded layout drives the CG from the host side)
int precond_apply(const float* l32, int64_t ld32, const TriInv& ti, int64_t n, int64_t np, const double* r, double* z, PcgWork& w,
                  hipStream_t s);

// ---- posterior.hip ----
int launch_convert_f64_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols,
                           int64_t rows_p, int64_t cols_p, hipStream_t s);
int launch_factor_input(const double* k64, int64_t ld64, float* a32, int64_t ld32, int64_t n, int64_t np,
                        double reg, double pad_diag, hipStream_t s, int64_t row_begin = 0, int64_t row_end = -1);  // rows [row_begin, row_end) (-1: to np)
int launch_mirror_rows_f64(double* k, int64_t ld, int64_t n0, int64_t n1, hipStream_t s);
int launch_row_sqsum_f32(const float* v, int64_t ld, int64_t rows, int64_t cols, const double* base,
                         double* out, hipStream_t s);
int launch_cov_finish(const double* ktt, int64_t ldk, const float* vvt, int64_t ldv, int64_t m, double* cov,
                      hipStream_t s);
int launch_zero_pad_f64(double* a, int64_t ld, int64_t n, int64_t np, hipStream_t s);
int launch_f32_to_f64_mat(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols,
                          bool accumulate, hipStream_t s);
int launch_axpby_mat(double* r, double a, const double* k, double b, int64_t ld, int64_t rows, int64_t cols,
                     hipStream_t s);
int launch_rowdot_f64(const double* z, const double* k, double kscale, const double* r, int64_t ld, int64_t rows,
                      int64_t cols, const double* base, double sign, double* out, hipStream_t s, double* zstat = nullptr);  // zstat: [2 rows] |z|^2, max |z|
int launch_copy_mat_f64(const double* src, int64_t lds, double* dst, int64_t m, hipStream_t s);
int launch_skinny_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                         const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta, hipStream_t s);
int launch_identity_rows(double* e, int64_t ld, int64_t cols, int64_t r0, int64_t rows, int64_t rows_p, hipStream_t s);
int launch_symmetrize_f64(double* a, int64_t ld, int64_t n, hipStream_t s);
// per-row preconditioned CG over [rows, cols] blocks (api.hip: rows_pcg_continue)
struct RowsPcg {
    int64_t cap = 0;          // padded row capacity
    double* p = nullptr;      // [cap, np_cap] search directions
    double* q = nullptr;      // [cap, np_cap] (K + reg I) p
    double* rho = nullptr;    // [cap] r . M^-1 r
    double* coef = nullptr;   // [cap] beta, then alpha of the current step
    double* tol = nullptr;    // [cap] stopping threshold on the per-step decrease of e^T A e
    double* delta = nullptr;  // [cap] first-order term z . r of the fixed sweeps
    double* var = nullptr;    // [cap] variance estimate of the fixed sweeps (full-covariance mode)
    double* zstat = nullptr;  // [2 cap] |z_row|_2^2 and max |z_row| of the level-1 variance's rows (guard of the int8 residual)
    int32_t* state = nullptr; // [cap] >= 0: consecutive small steps; -1: finished
    int32_t* live = nullptr;  // [6] rows still iterating; rows flagged by k_rows_prepare; [2..5]: two doubles, the sweep estimates
    int32_t* host = nullptr;  // pinned [6], same layout
};
int launch_rows_prepare(const double* delta, const double* ktt, const double* var, const double* q, int mode, double thr,
                        int64_t rows, double* tol, int32_t* flagged, hipStream_t s);
int launch_rows_energy(const double* r, const float* s32, int64_t ld, int64_t rows, int64_t cols, double* out, hipStream_t s);
int launch_rows_prepare_ntk(const double* e1, const double* dv, double* zk_tol, const double* var, int64_t vstride,
                            int64_t rows, double tau, hipStream_t s);
int launch_rows_dvar(const float* d32, const double* w, const double* k, double kscale, int64_t ld, int64_t rows,
                     int64_t cols, double* out, hipStream_t s);
int launch_sweep_estimate(const double* e0, const double* e1, const double* zk, const double* dv, const double* var,
                          int64_t vstride, int64_t rows, double* out, hipStream_t s);
int launch_rows_rho(const double* r, const float* s32, int64_t ld, int64_t rows, int64_t cols, bool first, RowsPcg& w,
                    hipStream_t s);
int launch_rows_update_p(double* p, const float* s32, int64_t ld, int64_t rows, int64_t cols, RowsPcg& w, hipStream_t s);
int launch_rows_alpha(const double* p, const double* q, int64_t ld, int64_t rows, int64_t cols, RowsPcg& w, hipStream_t s);
int launch_rows_axpy2(double* z, double* r, const double* p, const double* q, int64_t ld, int64_t rows, int64_t cols,
                      RowsPcg& w, hipStream_t s);

int launch_pool_select(const double* mean, int64_t m, int ny, const double* var, int64_t count, int biased, uint64_t seed,
                       double* key_ws, int64_t* indices, hipStream_t s);
int launch_transpose_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t n, hipStream_t s);
int launch_strided_copy_f64(const double* src, int64_t incs, double* dst, int64_t incd, int64_t n, hipStream_t s);

}  // namespace nngp
