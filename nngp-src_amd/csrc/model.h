// Internal to the C-ABI layer (api.hip, api_predict.hip, api_ops.hip): the model object that owns the device buffers of one GP fit, and the
// helpers the entry points share.  Not installed; include/nngp_hip.h is the interface.
#pragma once
#include <stdarg.h>
#include <stdlib.h>
#include <cmath>
#include <new>
#include <mutex>
#include "common.h"
#include <atomic>

namespace nngp {
extern std::atomic<long long> g_alloc_count;  // device allocations made by the library so far (nngp_alloc_count)

template <typename T>
inline int dev_alloc(T** p, int64_t count) {
    *p = nullptr;
    if (count <= 0) return 0;
    NNGP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (size_t)count));
    g_alloc_count.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

template <typename T>
inline void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

template <typename T>
inline bool soft_alloc(T** p, int64_t count) {  // false (and no sticky error) when the device has no room
    g_alloc_count.fetch_add(1, std::memory_order_relaxed);
    *p = nullptr;
    if (NNGP_KNOB(5) == 55) return false;  // test: as if the device were full
    if (hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (size_t)count) == hipSuccess) return true;
    (void)hipGetLastError();
    *p = nullptr;
    return false;
}

// Estimated |z . dr| / variance above which a fit is taken off the int8 residual: level 1 is itself 1e-6 .. 5e-6 from the converged
// variance; the bench fits sit at ~1e-7 (nngp_model_residual_floor)
constexpr double kI8FloorThr = 1e-5;
}  // namespace nngp

using namespace nngp;

struct nngp_model {
    int64_t n_cap = 0, np_cap = 0, m_cap = 0;
    int d = 0, ny = 1, get = NNGP_GET_NNGP;
    ArchDev arch{};
    double diag_reg = 1e-3;
    int absolute = 0;

    int64_t n = 0, np = 0;
    int64_t ld = 0;  // leading dimension of k64 / a32 = np_cap, fixed so that rows can be appended in place
    bool have_train = false, built = false, factored = false, solved = false;
    bool a32_built = false;  // the kernel build also wrote float32(K) + reg I on the lower tiles of a32 (fused factor input)
    // Row-sharded layout (round 4, nngp-src_amd/shard32.py): this rank's float64 kernel rows stay local, the float32 factor input is what
    // travels.  a32_complete: every row of a32 is in place (converted here or gathered) -- factor_begin only adds the padding;
    // k64_partial: k64 holds this rank's rows only -- no refactoring from it, no replicated CG, no covariance inside nngp_model_predict.
    bool a32_complete = false, k64_partial = false;
    bool k64_symmetric = false;  // k64 came from ONE symmetric build (mirrored tiles: bitwise symmetric) -- its digit planes can be cut from the lower triangle

    // training-side buffers
    double* x = nullptr;      // [n_cap, d]
    double* y = nullptr;      // [n_cap, ny]
    double* q = nullptr;      // [n_cap] |x|^2/d
    double* kdiag = nullptr;  // [n_cap] K(x,x) of the `get` kernel
    double* k64 = nullptr;    // [np_cap, np_cap] float64 train-train kernel (`get`), ld = np
    float* a32 = nullptr;     // [np_cap, np_cap] float32 A = K + reg I -> L (lower), ld = np
    float* dinv = nullptr;    // [np_cap/128][128*128] inverted diagonal blocks of L
    int32_t* clamped = nullptr;
    double* alpha = nullptr;  // [n_cap, ny]
    PcgWork pcg{};
    TriInv tri{};
    LookAhead* la = nullptr;
    SplitWork split{};
    TrsmTickets* tk = nullptr;   // the blocked solves as one persistent launch each (trsm_tickets.hip); NULL: step-by-step launches
    bool tk_failed = false;      // a launch gave up waiting (error word): the model stays on the step-by-step solves
    double diag_max = 0.0;

    // predict-side buffers (grown on demand when m > m_cap)
    double* xt_q = nullptr;      // [m_cap]
    double* tt_diag = nullptr;   // [m_cap] nngp K(x_t, x_t)
    double* ktd64 = nullptr;     // [m_cap, np_cap] float64 cross kernel
    float* b32 = nullptr;        // [mp_cap, np_cap] float32 RHS of the triangular solve
    float* trsm_tmp = nullptr;   // [mp_cap, 1024] block-column scratch of the blocked triangular solves
    int64_t ktd_cap = 0;         // capacity (rows) of ktd64
    int64_t full_cap = 0;        // capacity (rows) of the full-covariance buffers
    double* ktt64 = nullptr;     // [full_cap, full_cap]
    float* vvt32 = nullptr;      // [fullp, fullp]

    // float64 refinement of the posterior covariance (grown on demand)
    int var_refine = 1;          // covariance precision level, see nngp_model_set_refine
    float* lt32 = nullptr;       // [np_cap, np_cap] L^T, built lazily after a fit
    float* dinvt = nullptr;      // transposed inverted 128-blocks
    bool lt_ready = false;
    int64_t refine_cap = 0;      // padded row capacity of z64 / r64
    double* z64 = nullptr;       // [refine_cap, np_cap]  Z ~ K_td (K + reg I)^-1
    double* r64 = nullptr;       // [refine_cap, np_cap]  residual / product workspace
    double* covp64 = nullptr;    // [fullp, fullp] padded covariance
    // serving mode (nngp_model_prepare_serving): explicit float64 (K + reg I)^-1, refined to float64 accuracy once per fit;
    // predict then needs one float64 product per query batch instead of blocked solves + correction sweeps
    double* ainv64 = nullptr;    // [np, np] compact (ld = np at the time it was built)
    bool serving_ready = false;
    bool serving_weak = false;   // built from a weak preconditioner: predict adds one correction step
    RowsPcg rows{};              // per-row CG that continues the correction sweeps when the float32 factor is a weak preconditioner
    int64_t rows_pq_cap = 0;     // row capacity of rows.p / rows.q (allocated when a continuation first runs)
    int cov_iters = 0;           // iterations of the last continuation (0: the fixed sweeps were enough)
    double sweep_est_var = -1.0; // NTK: the same as a predicted relative variance error
    double sweep_est = -1.0;     // NTK: predicted relative energy-norm error after the fixed sweeps (-1: not measured)
    // NTK covariance needs the NNGP kernels as well
    double* kaux64 = nullptr;    // [np_cap, np_cap] NNGP train-train kernel when get == ntk
    bool aux_ready = false;
    double* ktd_aux = nullptr;   // [ktd rows, np_cap] NNGP cross kernel when get == ntk
    int64_t ktd_aux_cap = 0;

    // Sliced int8 copies of K for the residual products of the covariance (gemm_i8s.hip): allocated and cut by the first predict
    // that takes that path, cut again after every change of K.
    I8Work i8{};
    hipEvent_t ev_gate = nullptr;  // start of the covariance's int8 plane products on the caller's stream: the deferred alpha CG waits for it
    bool gate_recorded = false;
    hipEvent_t ev_i8 = nullptr;  // the planes of K were cut on the solve stream (beside the first blocked solves of a predict)
    bool i8_k_pending = false;   // ... and the consumer has not waited for that yet
    bool i8_unavailable = false; // no room for its workspace on this device: float64 pipe from then on
    // Guard of the int8 residual (once per fit, on the first predict that used it for a level-1 variance): estimate of what the dropped
    // digit pairs may have cost the variances, relative to them (k_i8s_floor_ratio); above kI8FloorThr the predict is redone on the
    // float64 pipe and the fit stays there.
    bool i8_checked = false, i8_distrusted = false, i8_used_now = false;
    double i8_floor_ratio = -1.0;
    // ... and on EVERY later level-1 predict of the fit (round 4: a later batch may sit closer to training points, its variances orders
    // of magnitude smaller): the estimate comes from statistics the variance's own row-dot pass collects, its one word travels to the host
    // without a wait and is looked at when the NEXT predict starts -- a batch that trips it sends the fit to the float64 pipe from then on
    // (the batch itself is not redone: only the first predict of a fit waits for its own estimate).
    unsigned long long* i8_guard = nullptr;       // device word
    unsigned long long* i8_guard_host = nullptr;  // pinned
    hipEvent_t ev_guard = nullptr;
    bool i8_guard_pending = false;
    bool i8_want_fine = false;  // sticky: see i8s_planes_policy
    bool i8_fuse_request = false, i8_fuse_done = false;  // level-1 variance: row statistics + float32 copy from the combination pass (I8Fuse)
    bool i8_suspended = false;   // prepare_serving: the explicit inverse is refined against residuals of the float64 pipe itself  // prepare_serving: the explicit inverse is refined against float64 residuals proper

    double reg = 0.0, trace_mean = 0.0, relres = 0.0;
    // Diagonal shift of the float32 factor's input.  = reg, unless the float32 factorisation of K + reg I broke down
    // (pivots at the rounding-noise floor: cond * eps32 >> 1); nngp_model_factor then factors K + reg_fac I with a
    // larger shift.  The factor is only the preconditioner: alpha and the refined covariances still solve K + reg I.
    double reg_fac = 0.0;
    int iters = 0;
    // The CG solve for alpha is deferred: nngp_model_solve records where the factor is ready, and the solve runs on its
    // own stream when alpha is first needed -- inside predict AFTER the covariance work has been enqueued, so that the
    // HBM-bound CG (float64 GEMV + float32 TRSVs) overlaps the MFMA-bound covariance products.
    hipStream_t solve_stream = nullptr;
    // The inverted 1024-blocks of the factor (tri: CG preconditioner, blocked solves) are built lazily: nngp_model_factor_end only marks
    // them stale; a predict right after the fit builds them on the look-ahead's panel stream beside its cross-kernel build (a chain of
    // ~17 small launches, 0.5 ms at N = 32768, that leaves the chip idle when it runs in line); every other consumer builds them in
    // order.  tri_join() is the one gate: every reader of `tri` passes it on the stream it reads from.
    bool tri_stale = false, tri_pending = false;
    // live timing of the posterior's blocked triangular solves (nngp_model_trsm_timer): one event pair per forward / backward solve
    struct TrsmTimer {
        static constexpr int kMax = 32;
        bool timed = false;
        int count = 0;
        hipEvent_t t0[kMax] = {}, t1[kMax] = {};
        double flops[kMax] = {};
    } trsm_t;
    hipEvent_t ev_tri = nullptr, ev_tri_fork = nullptr;
    hipEvent_t ev_ready = nullptr, ev_solved = nullptr;
    hipEvent_t ev_lt = nullptr;  // orders the split copy of L^T written on solve_stream (apply_inverse_f32)
    hipEvent_t ev_predict = nullptr;  // end of the last predict on its stream: it reads alpha and the CG residual
    bool have_predict_event = false;
    bool solve_pending = false;
    // Early stop (ny == 1): a predict that also forms the covariance rows Z ~ K_td (K + reg I)^-1 stops the CG at 1e-6 and
    // corrects the mean through them: mu = K_td a_k + Z r_k (exact up to (K_td A^-1 - Z) r_k, the product of two small
    // errors).  cg_partial: alpha holds a_k and the CG state in pcg is intact; anything that needs alpha itself resumes.
    bool cg_partial = false;
    bool have_alpha_event = false;  // ev_solved has been recorded at least once
    int cg_iters_done = 0;
    int solve_ahead = 0;  // > 0: the first `solve_ahead` CG iterations are already in flight on solve_stream (ny == 1)
    int pend_max_iters = 60;
    double pend_tol = 1e-10;

    ~nngp_model() {
        dev_free(x); dev_free(y); dev_free(q); dev_free(kdiag); dev_free(k64); dev_free(a32); dev_free(dinv);
        dev_free(clamped); dev_free(alpha);
        dev_free(pcg.r); dev_free(pcg.z); dev_free(pcg.p); dev_free(pcg.q); dev_free(pcg.xcol); dev_free(pcg.bcol);
        dev_free(pcg.f32a); dev_free(pcg.f32b); dev_free(pcg.f32c); dev_free(pcg.scal); dev_free(pcg.symv_part); dev_free(pcg.dot_part); dev_free(pcg.dot_ctr);
        if (pcg.host_scal) (void)hipHostFree(pcg.host_scal);
        if (pcg.iter_graph) (void)hipGraphExecDestroy(pcg.iter_graph);
        dev_free(tri.tinv); dev_free(tri.xinv); dev_free(tri.partial); dev_free(tri.tmp);
        lookahead_destroy(la);
        if (solve_stream) (void)hipStreamDestroy(solve_stream);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_lt) (void)hipEventDestroy(ev_lt);
        if (ev_tri) (void)hipEventDestroy(ev_tri);
        for (int t = 0; t < TrsmTimer::kMax; ++t) { if (trsm_t.t0[t]) (void)hipEventDestroy(trsm_t.t0[t]); if (trsm_t.t1[t]) (void)hipEventDestroy(trsm_t.t1[t]); }
        if (ev_tri_fork) (void)hipEventDestroy(ev_tri_fork);
        if (ev_solved) (void)hipEventDestroy(ev_solved);
        if (ev_predict) (void)hipEventDestroy(ev_predict);
        if (ev_i8) (void)hipEventDestroy(ev_i8);
        if (ev_gate) (void)hipEventDestroy(ev_gate);
        dev_free(i8.k.planes); dev_free(i8.k.scale); dev_free(i8.aux.planes); dev_free(i8.aux.scale); dev_free(i8.zplanes);
        for (int t = 0; t < I8Work::kMaxTimed; ++t) { if (i8.t0[t]) (void)hipEventDestroy(i8.t0[t]); if (i8.t1[t]) (void)hipEventDestroy(i8.t1[t]); } dev_free(i8.zscale); dev_free(i8.partial); dev_free(i8.rowpart); dev_free(i8.counters);
        dev_free(split.planes); dev_free(split.counters); dev_free(split.planes_t); dev_free(split.planes_b); dev_free(split.row_inv);
        dev_free(split.ldiag); dev_free(split.dfrag); dev_free(split.dscale);
        tk_destroy(tk);
        dev_free(xt_q); dev_free(tt_diag); dev_free(ktd64); dev_free(b32); dev_free(trsm_tmp); dev_free(ktt64); dev_free(vvt32);
        dev_free(lt32); dev_free(dinvt); dev_free(z64); dev_free(r64); dev_free(covp64); dev_free(kaux64); dev_free(ktd_aux);
        dev_free(ainv64);
        dev_free(rows.p); dev_free(rows.q); dev_free(rows.rho); dev_free(rows.coef); dev_free(rows.tol); dev_free(rows.delta);
        dev_free(rows.var); dev_free(rows.state); dev_free(rows.live); dev_free(rows.zstat); dev_free(i8_guard);
        if (i8_guard_host) (void)hipHostFree(i8_guard_host);
        if (ev_guard) (void)hipEventDestroy(ev_guard);
        if (rows.host) (void)hipHostFree(rows.host);
    }
};


namespace nngp {
// ---- api_predict.hip: workspaces, the factor's consumers, the posterior ----
int ensure_predict_capacity(nngp_model* m, int64_t mt, bool need_ktd);
int ensure_full_cov_capacity(nngp_model* m, int64_t mt);
int drop_pending_solve(nngp_model* m);
void set_split_scale(nngp_model* m);
int ensure_refine_capacity(nngp_model* m, int64_t mp);
int ensure_lt_alloc(nngp_model* m);
int ensure_lt(nngp_model* m, hipStream_t s);
int ensure_lt_split(nngp_model* m, hipStream_t s);
bool use_i8s(const nngp_model* m, int64_t mp);
bool use_i8s_fine(const nngp_model* m, int64_t mp);
int i8s_planes_policy(const nngp_model* m);
int ensure_i8s(nngp_model* m, int64_t mp, I8Planes& pk, int planes);
int i8s_cut_planes(nngp_model* m, I8Planes& pk, const double* kmat, int64_t kld, hipStream_t s);
int i8s_product_rows(nngp_model* m, I8Planes& pk, const double* kmat, int64_t kld, double* out, const double* cin, double beta,
                     double alpha, double* z, double gamma, int64_t mp, hipStream_t s, int grade);
int residual_rows(nngp_model* m, double* out, const double* rhs, double* z, int64_t mp, hipStream_t s, bool first_residual);
int tri_join(nngp_model* m, hipStream_t s);
int tri_fork(nngp_model* m, hipStream_t s);
bool use_split_solves(const nngp_model* m, int64_t mp);
bool use_tickets(const nngp_model* m, int64_t mp);
bool cg_from_the_start(const nngp_model* m, int64_t mp);
int tickets_reserve(const nngp_model* m);
int tickets_check(nngp_model* m, bool wait);
int apply_forward_f32(nngp_model* m, int64_t mp, hipStream_t s);
int apply_inverse_f32(nngp_model* m, int64_t mp, hipStream_t s);
int refined_solve_rows(nngp_model* m, const double* rhs, int64_t mp, int sweeps, bool final_residual, hipStream_t s,
                       bool measure = false);
int rows_pcg_continue(nngp_model* m, int64_t mp, int max_iters, hipStream_t s);
int build_cross(nngp_model* m, const double* xt, const double* qt, int64_t mt, int64_t mp, bool nngp, double* out,
                hipStream_t s);
// ---- api.hip ----
int run_pending_solve(nngp_model* m, hipStream_t user, bool order_user, bool allow_partial = false);  // the deferred alpha CG
}  // namespace nngp
