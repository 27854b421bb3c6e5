// The posterior half of the C ABI: workspaces of a predict, the consumers of the factor (blocked solves, refinement sweeps, the int8
// residual products), nngp_model_prepare_serving / _predict / _apply_factor.  Reference: predict_fn(x_test, get, compute_cov),
// train.py:157-158, estimator.py:66-67.
#include "model.h"

namespace nngp {

int ensure_predict_capacity(nngp_model* m, int64_t mt, bool need_ktd) {
    if (mt > m->m_cap || m->b32 == nullptr) {
        const int64_t cap = mt > m->m_cap ? mt : m->m_cap;
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(m->xt_q); dev_free(m->tt_diag); dev_free(m->b32); dev_free(m->trsm_tmp);
        NNGP_TRY(dev_alloc(&m->xt_q, cap));
        NNGP_TRY(dev_alloc(&m->tt_diag, cap));
        NNGP_TRY(dev_alloc(&m->b32, round_up(cap, TB) * m->np_cap));
        NNGP_TRY(dev_alloc(&m->trsm_tmp, round_up(cap, TB) * triinv_block(m->np_cap)));
        if (m->split.planes != nullptr) {  // split copy of one right-hand-side block (float16 path of the blocked solves)
            dev_free(m->split.planes_b); dev_free(m->split.row_inv);
            m->split.planes_b = nullptr; m->split.row_inv = nullptr;
            m->split.mb_cap = round_up(cap, TB);
            // one panel per 1024 columns of a solve step, col_stride bytes apart (see SplitWork)
            m->split.b_panels = (int)((triinv_block(m->np_cap) + m->split.k_cap - 1) / m->split.k_cap);
            NNGP_TRY(dev_alloc(&m->split.planes_b, (int64_t)(m->split.b_panels - 1) * m->split.col_stride + (m->split.mb_cap + 256) * m->split.k_cap * 4));
            NNGP_TRY(dev_alloc(&m->split.row_inv, m->split.mb_cap));
            // ... and the persistent form of the blocked solves (no room for its workspace: the step-by-step form stays)
            tk_destroy(m->tk);
            m->tk = nullptr;
            if (m->split.k_cap == 1024 && triinv_block(m->np_cap) == 1024) NNGP_TRY(tk_create(&m->tk, m->np_cap, m->split.mb_cap) < 0 ? -1 : 0);
        }
        m->m_cap = cap;
    }
    if (need_ktd && mt > m->ktd_cap) {
        const int64_t cap = mt > m->m_cap ? mt : m->m_cap;
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(m->ktd64);
        NNGP_TRY(dev_alloc(&m->ktd64, round_up(cap, TB) * m->np_cap));
        m->ktd_cap = cap;
    }
    return 0;
}

int ensure_full_cov_capacity(nngp_model* m, int64_t mt) {
    if (mt > m->full_cap) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(m->ktt64); dev_free(m->vvt32);
        const int64_t mp = round_up(mt, TB);
        dev_free(m->covp64);
        NNGP_TRY(dev_alloc(&m->ktt64, mp * mp));
        NNGP_TRY(dev_alloc(&m->vvt32, mp * mp));
        NNGP_TRY(dev_alloc(&m->covp64, mp * mp));
        m->full_cap = mt;
    }
    return 0;
}

// A CG that was started ahead (nngp_model_solve) reads y, K, the factor and the CG workspace: anything that is about to
// overwrite those waits for it and drops it.
int drop_pending_solve(nngp_model* m) {
    if (m->solve_pending && m->solve_ahead > 0) NNGP_HIP_CHECK(hipStreamSynchronize(m->solve_stream));
    m->solve_pending = false;
    m->solve_ahead = 0;
    m->cg_partial = false;
    return 0;
}

// |L_ij| <= sqrt(max_i A_ii): scale of the float16 split copies of the factor, largest entry below 2^15
void set_split_scale(nngp_model* m) {
    const double lmax = sqrt(fmax(m->diag_max, m->trace_mean) + m->reg_fac);
    int e = 0;
    (void)frexp(lmax, &e);  // lmax = f * 2^e, f in [0.5, 1)
    m->split.scale = (lmax > 0.0 && std::isfinite(lmax)) ? (float)ldexp(1.0, 15 - e) : 1.0f;
}

int ensure_refine_capacity(nngp_model* m, int64_t mp) {
    if (mp > m->refine_cap) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(m->z64); dev_free(m->r64);
        NNGP_TRY(dev_alloc(&m->z64, mp * m->np_cap));
        NNGP_TRY(dev_alloc(&m->r64, mp * m->np_cap));
        m->refine_cap = mp;
    }
    if (mp > m->rows.cap) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        RowsPcg& w = m->rows;
        dev_free(w.rho); dev_free(w.coef); dev_free(w.tol); dev_free(w.delta); dev_free(w.var); dev_free(w.state); dev_free(w.zstat);
        NNGP_TRY(dev_alloc(&w.rho, mp)); NNGP_TRY(dev_alloc(&w.coef, mp)); NNGP_TRY(dev_alloc(&w.tol, mp));
        NNGP_TRY(dev_alloc(&w.delta, mp)); NNGP_TRY(dev_alloc(&w.var, mp)); NNGP_TRY(dev_alloc(&w.state, mp));
        NNGP_TRY(dev_alloc(&w.zstat, 2 * mp));
        if (w.live == nullptr) {
            NNGP_TRY(dev_alloc(&w.live, 6));
            NNGP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&w.host), 6 * sizeof(int32_t), hipHostMallocDefault));
        }
        w.cap = mp;
    }
    return 0;
}

// L^T and the transposed inverted diagonal blocks: operands of the "B L^-1" half of (L L^T)^-1 on the float32 path.
int ensure_lt_alloc(nngp_model* m) {
    if (m->lt32 != nullptr) return 0;
    NNGP_HIP_CHECK(hipDeviceSynchronize());
    NNGP_TRY(dev_alloc(&m->lt32, m->np_cap * m->np_cap));
    NNGP_TRY(dev_alloc(&m->dinvt, (m->np_cap / TB) * TB * TB));
    return 0;
}
int ensure_lt(nngp_model* m, hipStream_t s) {
    if (m->lt_ready) return 0;
    NNGP_TRY(ensure_lt_alloc(m));
    NNGP_TRY(launch_transpose_f32(m->a32, m->ld, m->lt32, m->np, m->np, s));
    NNGP_TRY(launch_transpose_blocks_f32(m->dinv, m->dinvt, TB, m->np / TB, s));
    m->lt_ready = true;
    return 0;
}

// split copy of L^T by block row (operand of the "B L^-1" half on the float16 pipe); built once per fit
int ensure_lt_split(nngp_model* m, hipStream_t s) {
    if (m->split.lt_ready) return 0;
    if (m->split.planes_t == nullptr) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        const int64_t ncols = (m->np_cap + m->split.k_cap - 1) / m->split.k_cap;
        NNGP_TRY(dev_alloc(&m->split.planes_t, ncols * m->split.col_stride));
    }
    NNGP_TRY(launch_split_lower_t(m->a32, m->ld, m->np, m->split.k_cap, m->split.scale, m->split.planes_t,
                                  m->split.col_stride, s));
    m->split.lt_ready = true;
    return 0;
}


// ---- float64-grade residual products on the int8 matrix pipe (gemm_i8s.hip) ----
constexpr int64_t kI8RowBlock = 2048;  // right-hand-side rows per pass (bounds the int32 partial buffer)

// Where it pays (profiles/r3_i8s_crossover.jsonl, predict with the diagonal variance, float64 / int8 residual, ms): a predict that also
// has to cut the planes of K (the first after a fit) wins from two 128-row tiles of right-hand sides on -- N = 2048, M = 256: 0.57 /
// 0.50; N = 8192: M = 128 2.04 / 2.39, M = 256 2.45 / 2.56, M = 512 3.63 / 3.06, M = 1024 5.02 / 3.99; N = 16384, M = 1024: 15.5 / 11.4
// -- later predicts on the same fit from any size on (N = 16384, M = 128: 4.51 / 4.42).  Debug key 5 = 50: float64 matrix pipe instead.
bool use_i8s(const nngp_model* m, int64_t mp) {
    if (m->i8_suspended || m->i8_unavailable || m->i8_distrusted) return false;
    if (NNGP_KNOB(5) == 54) return true;  // timing experiment: at any size (scripts/i8s_crossover.py)
    return m->np >= 2048 && mp >= 256 && NNGP_KNOB(5) != 50;
}

// Two grades of the product.  COARSE: 3 x 5 planes (z ROUNDED to its three and written back, round 4; 5 x 5 before), pairs with
// ia + ib <= 4 (12 exact plane products, 15 before; error ~2^-32 sqrt(N) of the row maxima, all of it from the kernel's side now)
// -- for a FIRST residual.  FINE: 7 x 7 planes, ia + ib <= 6 (28 products; ~2^-48 sqrt(N): what the float64 matrix pipe
// delivers; since late round 4 z is ROUNDED to five planes there as well and written back: 5 x 7 planes, 25 products, the iterate
// moves by 9e-13 of its row maximum) -- for the later residuals and the NTK's W = Z K_dd, where the coarse floor would show (see residual_rows); 28 products
// still cost 3/4 of the float64 product at N = 32768.  A model that will ask for FINE products (NTK fits, covariance levels >= 2)
// has its kernel matrix cut into 7 planes once; coarse products then read the first five of them.
enum { I8_COARSE = 0, I8_FINE = 1 };
constexpr int kI8FinePlanes = 7, kI8FineCut = 6;
constexpr int kI8CoarseZPlanes = 3;  // a first residual's z is rounded to three digits and written back (i8s_product_rows)
constexpr int kI8FineZPlanes = 5;    // a later residual's z (an iterate good to ~1e-8) and the NTK's final Z: rounded to 40 bits below the row maximum (9e-13 of it)

// FINE pays later than COARSE (28 against 15 products): from N = 4096 and four 128-row tiles of right-hand sides on.  Debug key 5 = 57: off.
bool use_i8s_fine(const nngp_model* m, int64_t mp) { return use_i8s(m, mp) && m->np >= 4096 && mp >= 512 && NNGP_KNOB(5) != 57; }
// (i8_want_fine: a predict of this model has needed a FINE product before -- a full covariance at level 1 is promoted to level 2, a weak
// fit's rows continue with later residuals -- so the planes are cut seven deep from the start instead of being thrown away, reallocated
// and cut again in the middle of a predict)
int i8s_planes_policy(const nngp_model* m) {
    return (m->get == NNGP_GET_NTK || m->var_refine >= 2 || m->i8_want_fine) && NNGP_KNOB(5) != 57 ? kI8FinePlanes : 5;
}


// Workspace of the int8 path: `planes` N^2 bytes of digit planes of the kernel matrix `pk` + the planes and exact plane products of
// one block of rows.  Returns 1 -- and the model stays on the float64 pipe from then on -- when the device has no room for them (a
// kernel matrix that fills most of the 288 GB leaves none): the int8 path is an accelerator, not a requirement.
int ensure_i8s(nngp_model* m, int64_t mp, I8Planes& pk, int planes) {
    I8Work& w = m->i8;
    if (NNGP_KNOB(5) == 51) { w.ns_k = w.ns_z = 4; w.cut = 3; }
    else if (NNGP_KNOB(5) == 52) { w.ns_k = w.ns_z = 6; w.cut = 5; }
    else { w.ns_k = 5; w.ns_z = NNGP_KNOB(5) == 58 ? 5 : kI8CoarseZPlanes; w.cut = 4; }
    if (planes < w.ns_k) planes = w.ns_k;
    w.k_rows = round_up(m->np_cap, 256);
    auto give_up = [&]() -> int {
        dev_free(m->i8.k.planes); dev_free(m->i8.k.scale); dev_free(m->i8.aux.planes); dev_free(m->i8.aux.scale);
        dev_free(w.zplanes); dev_free(w.zscale); dev_free(w.partial); dev_free(w.rowpart);
        w.rowpart_rows = 0;
        m->i8.k.ready = m->i8.aux.ready = false;
        m->i8.k.alloc_planes = m->i8.aux.alloc_planes = 0;
        w.z_rows = 0;
        w.z_planes = 0;
        m->i8_unavailable = true;
        return 1;
    };
    if (pk.alloc_planes < planes) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(pk.planes); dev_free(pk.scale);
        pk.alloc_planes = 0;
        if (!soft_alloc(&pk.planes, planes * w.k_rows * m->np_cap) || !soft_alloc(&pk.scale, m->np_cap + 1)) return give_up();
        NNGP_HIP_CHECK(hipMemset(pk.planes, 0, (size_t)(planes * w.k_rows * m->np_cap)));
        pk.alloc_planes = planes;
        pk.ready = false;
    }
    if (mp > w.rowpart_rows) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(w.rowpart);
        w.rowpart_rows = 0;
        NNGP_TRY(dev_alloc(&w.rowpart, mp * i8s_col_blocks(m->np_cap) * 4));
        w.rowpart_rows = mp;
    }
    if (w.counters == nullptr) {
        NNGP_TRY(dev_alloc(&w.counters, 16));
        NNGP_HIP_CHECK(hipMemset(w.counters, 0, 16 * sizeof(int)));
    }
    const int64_t rows = mp < kI8RowBlock ? mp : kI8RowBlock;
    if (rows > w.z_rows || planes > w.z_planes) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(w.zplanes); dev_free(w.zscale); dev_free(w.partial);
        const int64_t nr = rows > w.z_rows ? rows : w.z_rows;
        const int np_ = planes > w.z_planes ? planes : w.z_planes;
        w.z_rows = 0;
        w.z_planes = 0;
        if (!soft_alloc(&w.zplanes, np_ * (nr + 256) * m->np_cap) || !soft_alloc(&w.zscale, nr) ||
            !soft_alloc(&w.partial, i8s_chunks(m->np_cap) * np_ * nr * m->np_cap))  // diagonals = cut + 1 <= planes
            return give_up();
        NNGP_HIP_CHECK(hipMemset(w.zplanes, 0, (size_t)(np_ * (nr + 256) * m->np_cap)));
        w.z_rows = nr;
        w.z_planes = np_;
    }
    return 0;
}

// the digit planes of a kernel matrix (as many as were allocated for it).  It is positive semi-definite: row i is bounded by
// sqrt(K_ii max_j K_jj) -- no pass over the matrix for the scales
int i8s_cut_planes(nngp_model* m, I8Planes& pk, const double* kmat, int64_t kld, hipStream_t s) {
    I8Work& w = m->i8;
    NNGP_TRY(launch_i8s_diag_bound_scale(kmat, kld, m->np, pk.scale, s));
    // a bitwise symmetric matrix (one symmetric kernel build; the NNGP kernel beside an NTK fit always is): every entry read once
    const bool sym = (kmat == m->kaux64 || m->k64_symmetric) && (pk.alloc_planes == 5 || pk.alloc_planes == 7) && NNGP_KNOB(5) != 62;
    if (sym)
        NNGP_TRY(launch_i8s_slice_sym(kmat, kld, m->np, pk.alloc_planes, pk.scale, pk.planes, m->np_cap, w.k_rows * m->np_cap, s));
    else
        NNGP_TRY(launch_i8s_slice_rows(kmat, kld, m->np, m->np, pk.alloc_planes, pk.scale, nullptr, pk.planes, m->np_cap, w.k_rows * m->np_cap, s));
    NNGP_TRY(launch_i8s_scale_sqsum(pk.scale, m->np, pk.scale + m->np_cap, s));
    pk.ns_done = pk.alloc_planes;
    pk.ready = true;
    return 0;
}

// out [mp, np] = beta cin + alpha z Kmat + gamma z on the int8 pipe; Kmat: a symmetric positive semi-definite [np, np] kernel matrix
// (k64, or kaux64 beside an NTK fit) whose digit planes are kept in pk (workspace: ensure_i8s, by the caller).
// The planes (13 N^2 bytes of HBM traffic to cut 5 of them: 2.7 ms at N = 32768) are cut once per change of the matrix: by
// nngp_model_predict on the solve stream beside its first blocked solves (-0.8 ms against stream order at N = 32768), else here in
// stream order.  Measured and dropped (profiles/r3_i8s_slicing_placement.json): slicing beside the FACTORISATION on a second stream
// costs the Cholesky exactly what the slicing takes, at any stream priority and wherever in the factorisation it starts -- its 32768
// small workgroups settle on every compute unit a trailing-update launch has just left and the next launch waits for them -- and on
// a CU-masked stream (1 / 2 / 4 units per XCD) it needs 95 / 64 / 55 ms: one compute unit moves ~20 GB/s of it.
int i8s_product_rows(nngp_model* m, I8Planes& pk, const double* kmat, int64_t kld, double* out, const double* cin, double beta,
                     double alpha, double* z, double gamma, int64_t mp, hipStream_t s, int grade) {
    const int64_t np = m->np;
    I8Work& w = m->i8;
    // COARSE: z comes straight from the float32 solves and every identity downstream holds for WHATEVER z they returned -- so z is
    // rounded to 24-bit fixed point below its row maximum (three planes; the solves' own error is ~1e-4 of it) and written back:
    // the planes ARE z, planes 3 and 4 do not exist, 12 plane products instead of 15 and no truncation on the z side
    // FINE (round 4, late): the same for the later residuals and the NTK's W = Z K_dd with five planes -- 25 products instead of 28 (timing-knob key 5 = 64: seven)
    const int fine_z = NNGP_KNOB(5) == 64 ? kI8FinePlanes : kI8FineZPlanes;
    const bool round_z = grade == I8_COARSE ? w.ns_z < w.ns_k : fine_z < kI8FinePlanes;
    const bool fuse_rows = grade == I8_COARSE && m->i8_fuse_request && cin != nullptr && w.rowpart_rows >= mp && NNGP_KNOB(5) != 59;
    const int ns_z = grade == I8_FINE ? fine_z : w.ns_z, ns_k = grade == I8_FINE ? kI8FinePlanes : w.ns_k;
    const int cut = grade == I8_FINE ? kI8FineCut : w.cut;
    NNGP_REQUIRE(pk.alloc_planes >= ns_k && w.z_planes >= ns_z, "i8s_product_rows: workspace for %d planes missing", ns_k);
    if (!pk.ready || pk.ns_done < ns_k) {
        NNGP_TRY(i8s_cut_planes(m, pk, kmat, kld, s));
    } else if (&pk == &m->i8.k && m->i8_k_pending) {  // cut on the solve stream at the start of this predict
        NNGP_HIP_CHECK(hipStreamWaitEvent(s, m->ev_i8, 0));
    }
    if (&pk == &m->i8.k) m->i8_k_pending = false;
    I8Plan pl;
    NNGP_TRY(i8s_plan(ns_z, ns_k, cut, &pl));
    const int64_t nchunk = i8s_chunks(np, &pl);
    for (int64_t r0 = 0; r0 < mp; r0 += kI8RowBlock) {
        const int64_t mb = mp - r0 < kI8RowBlock ? mp - r0 : kI8RowBlock;
        const int64_t slab = mb * np;
        NNGP_TRY(launch_i8s_slice_rows(z + r0 * np, np, mb, np, ns_z, nullptr, w.zscale, w.zplanes, m->np_cap, (w.z_rows + 256) * m->np_cap, s,
                                       round_z ? z + r0 * np : nullptr));
        if (r0 == 0 && grade == I8_COARSE && !m->gate_recorded && NNGP_KNOB(2) != 8 && NNGP_KNOB(2) != 9 && !cg_from_the_start(m, mp)) {
            // The deferred alpha CG (solve stream) starts HERE, not with the blocked solves before this product: its hundreds of small
            // GEMV launches settle on compute units between the solves' persistent split-float16 launches (which need a whole unit's
            // LDS) and cost them 3.7 ms at N = 32768 (scripts/cov_alone.py); beside this one long launch they fit the wave slots and
            // the 32 KB of LDS it leaves.
            if (m->ev_gate == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_gate, hipEventDisableTiming));
            NNGP_HIP_CHECK(hipEventRecord(m->ev_gate, s));
            m->gate_recorded = true;
        }
        const bool timed = w.timed && w.t_count < I8Work::kMaxTimed;
        if (timed) {
            const int t = w.t_count;
            if (w.t0[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&w.t0[t]));
            if (w.t1[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&w.t1[t]));
            NNGP_HIP_CHECK(hipEventRecord(w.t0[t], s));
        }
        NNGP_TRY(launch_gemm_nt_i8s(w.partial, np, slab, w.zplanes, m->np_cap, (w.z_rows + 256) * m->np_cap, pk.planes, m->np_cap,
                                    w.k_rows * m->np_cap, pl, mb, np, np, w.counters, 0, s));
        if (timed) {
            const int t = w.t_count++;
            NNGP_HIP_CHECK(hipEventRecord(w.t1[t], s));
            w.t_flops[t] = 2.0 * (double)mb * (double)np * (double)np;
            w.t_ops[t] = w.t_flops[t] * pl.npairs;
        }
        I8Fuse fuse;
        fuse.out32 = m->b32 + r0 * np;
        fuse.ld32 = np;
        fuse.part = w.rowpart + r0 * i8s_col_blocks(np) * 4;
        NNGP_TRY(launch_i8s_combine(out + r0 * np, np, cin ? cin + r0 * np : nullptr, np, beta, alpha, z + r0 * np, np, gamma, w.partial, np,
                                    slab, (int)nchunk, pl.ndiag, w.zscale, pk.scale, mb, np, s, fuse_rows ? &fuse : nullptr));
    }
    if (fuse_rows) m->i8_fuse_done = true;
    if (grade == I8_COARSE && !m->gate_recorded && NNGP_KNOB(2) == 9) {  // timing experiment: the CG starts when the product has ended
        if (m->ev_gate == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_gate, hipEventDisableTiming));
        NNGP_HIP_CHECK(hipEventRecord(m->ev_gate, s));
        m->gate_recorded = true;
    }
    return 0;
}

// out [mp, np] = rhs - z (K + reg I) for z = z64 (or another [mp, np] block).
// first_residual: z comes straight from the float32 solves, so the residual is ~1e-4 of rhs and the COARSE product's error floor
// (2^-32 sqrt(N) of the row maxima, ~1e-3 of such a residual) is harmless: the NNGP level-1 variance moves by 2e-7 (N = 32768) ..
// 7e-7 (ill-conditioned sweep case), a first correction sweep loses nothing.  Every LATER residual is ~1e-8 of rhs and needs float64
// grade proper -- measured with the coarse product there (scripts/i8s_hard_case.py): NTK variances off by 3e-5 .. 2e-4 (first order
// in the rows' error) against 1e-8, NNGP level 2 at 1e-7 instead of 1e-8, the explicit inverse of the serving mode stuck four digits
// short of float64 (serving variances 4e-3 off).  Those take the FINE product where it pays, else the float64 pipe.
int residual_rows(nngp_model* m, double* out, const double* rhs, double* z, int64_t mp, hipStream_t s, bool first_residual) {
    const int64_t np = m->np;
    const bool coarse = first_residual;
    if (!coarse && use_i8s_fine(m, mp)) m->i8_want_fine = true;
    if (coarse ? use_i8s(m, mp) : use_i8s_fine(m, mp)) {
        const int rc = ensure_i8s(m, mp, m->i8.k, coarse ? i8s_planes_policy(m) : kI8FinePlanes);
        if (rc == 0) {
            if (coarse) m->i8_used_now = true;
            return i8s_product_rows(m, m->i8.k, m->k64, m->ld, out, rhs, 1.0, -1.0, z, -m->reg, mp, s, coarse ? I8_COARSE : I8_FINE);
        }
        if (rc != 1) return rc;  // 1: no room for the planes -- the float64 pipe below
    }
    NNGP_TRY(launch_gemm_nt_f64(out, np, rhs, np, z, np, m->k64, m->ld, mp, np, np, -1.0, 1.0, s));
    return launch_axpby_mat(out, 1.0, z, -m->reg, np, mp, np, s);
}

// every reader of m->tri calls this on the stream it reads from: builds the inverted blocks there if nobody has yet, else orders the
// stream behind whoever did
int tri_join(nngp_model* m, hipStream_t s) {
    if (m->tri_stale) {
        if (m->ev_tri == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_tri, hipEventDisableTiming));
        NNGP_TRY(triinv_build(m->a32, m->ld, m->dinv, m->np, m->tri, s));
        if (m->tk != nullptr && m->tri.bs == 1024) NNGP_TRY(tk_prepare_inverses(m->tk, m->tri, m->np, m->a32, m->ld, s));
        NNGP_HIP_CHECK(hipEventRecord(m->ev_tri, s));
        m->tri_stale = false;
        m->tri_pending = true;
        return 0;
    }
    if (m->tri_pending) NNGP_HIP_CHECK(hipStreamWaitEvent(s, m->ev_tri, 0));
    if (m->tk != nullptr && !tk_inverses_ready(m->tk) && m->tri.bs == 1024) {  // the workspace was rebuilt after the blocks were inverted
        NNGP_TRY(tk_prepare_inverses(m->tk, m->tri, m->np, m->a32, m->ld, s));
        if (m->ev_tri == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_tri, hipEventDisableTiming));
        NNGP_HIP_CHECK(hipEventRecord(m->ev_tri, s));
        m->tri_pending = true;
    }
    return 0;
}

// predict: the build goes to the panel stream (idle between factorisations) behind what `s` holds now; tri_join orders the readers
int tri_fork(nngp_model* m, hipStream_t s) {
    if (!m->tri_stale || m->la == nullptr || m->la->panel == nullptr || NNGP_KNOB(5) == 61) return 0;  // key 5 = 61: in line
    if (m->ev_tri == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_tri, hipEventDisableTiming));
    if (m->ev_tri_fork == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_tri_fork, hipEventDisableTiming));
    NNGP_HIP_CHECK(hipEventRecord(m->ev_tri_fork, s));
    NNGP_HIP_CHECK(hipStreamWaitEvent(m->la->panel, m->ev_tri_fork, 0));
    NNGP_TRY(triinv_build(m->a32, m->ld, m->dinv, m->np, m->tri, m->la->panel));
    if (m->tk != nullptr && m->tri.bs == 1024) NNGP_TRY(tk_prepare_inverses(m->tk, m->tri, m->np, m->a32, m->ld, m->la->panel));
    NNGP_HIP_CHECK(hipEventRecord(m->ev_tri, m->la->panel));
    m->tri_stale = false;
    m->tri_pending = true;
    return 0;
}

// the factor has float16-split copies (look-ahead factorisation) and the caller did not ask for the float32 path
// and the block of right-hand sides is large enough for the 256-row tiles of the float16 GEMM to pay (measured, ms per
// diag-variance call at level 2, float16 / float32 solves -- N = 10800: M = 128: 6.5 / 5.6, 512: 9.1 / 8.8, 1024: 11.5 / 12.9;
// N = 32768: M = 128: 21.8 / 20.1, 256: 29.3 / 30.6, 512: 40.6 / 51.5)
bool use_split_solves(const nngp_model* m, int64_t mp) {
    return m->split.l_ready && m->split.planes_b != nullptr && mp <= m->split.mb_cap && m->tri.bs % m->split.k_cap == 0 &&
           m->tri.bs / m->split.k_cap <= m->split.b_panels && m->tri.bs <= 2048 && NNGP_KNOB(7) == 0 && mp >= 256 && mp * m->np >= 7000000;
}

// one persistent, ticket-ordered launch per solve (trsm_tickets.hip) instead of np / 1024 steps of three launches (debug key 9 = 16: the steps)
bool use_tickets(const nngp_model* m, int64_t mp) {
    return !m->tk_failed && m->tri.bs == 1024 && tk_usable(m->tk, mp, m->np) && NNGP_KNOB(9) != 16;
}

// With the blocked solves as persistent launches (224 registers, 128 KB of LDS: a 64-register kernel with a few KB of LDS fits beside
// their workgroups) the deferred alpha CG no longer waits for the plane products' start: it runs from the predict's start on.
bool cg_from_the_start(const nngp_model* m, int64_t mp) {
    static const int env = getenv("NNGP_TK_GATE") ? atoi(getenv("NNGP_TK_GATE")) : 1;  // (development aid)
    return env != 0 && use_split_solves(m, mp) && use_tickets(m, mp);
}

// While the digit planes of K are being cut on the solve stream (one 132 KB workgroup per compute unit: it cannot share a unit with a
// persistent solve) the solves leave it some units; the alpha CG's light kernels fit beside the solve's workgroups as they are.
int tickets_reserve(const nngp_model* m) {
    static const int env = getenv("NNGP_TK_RESERVE") ? atoi(getenv("NNGP_TK_RESERVE")) : 0;  // (development aid; measured 0 / 32 / 48 / 64: posterior 29.8-30.1 / 30.1 / 30.3 / 29.8 ms at cfg3)
    return m->i8_k_pending ? env : 0;
}

// a persistent solve that gave up waiting left its error word behind: report it once, and keep the model off that path
int tickets_check(nngp_model* m, bool wait) {
    const int e = tk_poll_error(m->tk, wait);
    if (e == 0) return 0;
    m->tk_failed = true;
    set_error("a persistent blocked solve gave up waiting for a dependency (error word 0x%x): its results are invalid; the model now uses the step-by-step solves", e);
    return -6;
}

// event pair around one blocked solve of mp right-hand sides (np^2 mp flops: a triangular matrix, multiply-add = 2)
static int trsm_timer_begin(nngp_model* m, int64_t mp, hipStream_t s, int* slot) {
    auto& t = m->trsm_t;
    *slot = -1;
    if (!t.timed || t.count >= nngp_model::TrsmTimer::kMax) return 0;
    const int i = t.count;
    if (t.t0[i] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&t.t0[i]));
    if (t.t1[i] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&t.t1[i]));
    NNGP_HIP_CHECK(hipEventRecord(t.t0[i], s));
    t.flops[i] = (double)m->np * (double)m->np * (double)mp;
    *slot = i;
    return 0;
}
static int trsm_timer_end(nngp_model* m, int slot, hipStream_t s) {
    if (slot < 0) return 0;
    NNGP_HIP_CHECK(hipEventRecord(m->trsm_t.t1[slot], s));
    m->trsm_t.count = slot + 1;
    return 0;
}

static int apply_forward_untimed(nngp_model* m, int64_t mp, hipStream_t s);

// b32 [mp, np] <- b32 L^-T   (rows are right-hand sides)
int apply_forward_f32(nngp_model* m, int64_t mp, hipStream_t s) {
    NNGP_TRY(tickets_check(m, false));
    NNGP_TRY(tri_join(m, s));
    int slot = -1;
    NNGP_TRY(trsm_timer_begin(m, mp, s, &slot));
    NNGP_TRY(apply_forward_untimed(m, mp, s));
    return trsm_timer_end(m, slot, s);
}

static int apply_forward_untimed(nngp_model* m, int64_t mp, hipStream_t s) {
    // While the deferred alpha CG runs on its own stream (from the int8 residual's gate on), the solves' persistent update grids
    // leave it some compute units (debug key 13 = n: n units; default 0 = none -- see DESIGN_NOTES R4)
    m->split.solve_reserve = (m->gate_recorded && NNGP_KNOB(13) > 0) ? NNGP_KNOB(13) : 0;
    if (NNGP_KNOB(7) == 1)  // timing experiment: the 128-wide recursion instead of the 1024-block form
        return trsm_rlt_f32(m->b32, m->np, mp, m->a32, m->ld, m->dinv, m->np, s);
    if (use_split_solves(m, mp)) {
        if (use_tickets(m, mp)) return tk_solve(m->tk, m->b32, m->np, mp, m->np, m->split, false, s, tickets_reserve(m));
        return trsm_rlt_blocks_h3(m->b32, m->np, mp, m->a32, m->ld, m->tri, m->np, m->trsm_tmp, m->split, s);
    }
    return trsm_rlt_blocks_f32(m->b32, m->np, mp, m->a32, m->ld, m->tri, m->np, m->trsm_tmp, s);
}

// b32 [mp, np] <- b32 (L L^T)^-1
int apply_inverse_f32(nngp_model* m, int64_t mp, hipStream_t s) {
    // The split copy of L^T (first covariance predict after a fit) is only read by the second half: it is written on the
    // model's own stream (idle until the alpha CG is asked for) beside the forward solve.  (debug key 2 = 7: in stream order)
    const bool lt_aside = NNGP_KNOB(7) != 1 && use_split_solves(m, mp) && !m->split.lt_ready && m->split.planes_t != nullptr &&
                          m->ev_lt != nullptr && NNGP_KNOB(2) != 7;
    if (lt_aside) {
        NNGP_HIP_CHECK(hipEventRecord(m->ev_lt, s));
        NNGP_HIP_CHECK(hipStreamWaitEvent(m->solve_stream, m->ev_lt, 0));
        NNGP_TRY(ensure_lt_split(m, m->solve_stream));
        NNGP_HIP_CHECK(hipEventRecord(m->ev_lt, m->solve_stream));
    }
    NNGP_TRY(apply_forward_f32(m, mp, s));
    if (NNGP_KNOB(2) == 10 && !m->gate_recorded && m->solve_pending) {  // timing experiment: the deferred alpha CG starts with the backward solve
        if (m->ev_gate == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_gate, hipEventDisableTiming));
        NNGP_HIP_CHECK(hipEventRecord(m->ev_gate, s));
        m->gate_recorded = true;
    }
    if (lt_aside) NNGP_HIP_CHECK(hipStreamWaitEvent(s, m->ev_lt, 0));
    // (whoever gets here on the float32 path finds its operand built: round 4's null-pointer abort came from a caller that had not)
    if (NNGP_KNOB(7) == 1) {
        NNGP_TRY(ensure_lt(m, s));
        return trsm_rut_f32(m->b32, m->np, mp, m->lt32, m->np, m->dinvt, m->np, s);
    }
    if (use_split_solves(m, mp)) NNGP_TRY(ensure_lt_split(m, s));
    else NNGP_TRY(ensure_lt(m, s));
    int slot = -1;
    NNGP_TRY(trsm_timer_begin(m, mp, s, &slot));
    if (use_split_solves(m, mp) && use_tickets(m, mp))
        NNGP_TRY(tk_solve(m->tk, m->b32, m->np, mp, m->np, m->split, true, s, tickets_reserve(m)));
    else if (use_split_solves(m, mp))
        NNGP_TRY(trsm_rut_blocks_h3(m->b32, m->np, mp, m->lt_ready ? m->lt32 : nullptr, m->np, m->tri, m->np, m->trsm_tmp, m->split, s));
    else
        NNGP_TRY(trsm_rut_blocks_f32(m->b32, m->np, mp, m->lt32, m->np, m->tri, m->np, m->trsm_tmp, s));
    return trsm_timer_end(m, slot, s);
}

// NTK covariance: the error of Z enters in first order (no cancellation as in the NNGP form), so what the sweeps leave is
// what the variance gets, ~ rho^(sweeps + 1) with rho the contraction of the float32 factor as a solver.  Measured at
// N = 16384, d = 256, join-block encoding, 4 CG iterations in the alpha solve (scripts/ntk_var_study.py, worst relative
// error over 1024 queries against 4 sweeps / ms per predict): 1 sweep 2.7e-5 / 24.8, 2 sweeps 3.6e-8 / 36.8, 3 sweeps
// 2.0e-8 / 49.0.  Levels <= 2: two sweeps; above: that many.
static int ntk_sweeps(const nngp_model* m) { return m->var_refine < 2 ? 2 : m->var_refine; }

// z64 <- rows of `rhs` [mp, np] times (K + reg I)^-1: float32 solves corrected by `sweeps` float64 residual sweeps.
// final_residual: also leave r64 = rhs - z64 (K + reg I) for the returned z64 (one more float64 product).
// measure (sweeps >= 2): the error energies r . M^-1 r that the first two corrections saw are kept in rows.var and
// rows.delta, and z . rhs in rows.tol, for k_sweep_estimate -- three passes over [mp, np].
int refined_solve_rows(nngp_model* m, const double* rhs, int64_t mp, int sweeps, bool final_residual, hipStream_t s,
                       bool measure) {
    const int64_t np = m->np;
    measure = measure && sweeps >= 2;
    NNGP_TRY(launch_convert_f64_f32(rhs, np, m->b32, np, mp, np, mp, np, s));
    NNGP_TRY(apply_inverse_f32(m, mp, s));
    NNGP_TRY(launch_f32_to_f64_mat(m->b32, np, m->z64, np, mp, np, false, s));
    for (int it = 0; it < sweeps + (final_residual ? 1 : 0); ++it) {
        NNGP_TRY(residual_rows(m, m->r64, rhs, m->z64, mp, s, it == 0));
        if (it == sweeps) break;
        NNGP_TRY(launch_convert_f64_f32(m->r64, np, m->b32, np, mp, np, mp, np, s));
        NNGP_TRY(apply_inverse_f32(m, mp, s));
        if (measure && it < 2) NNGP_TRY(launch_rows_energy(m->r64, m->b32, np, mp, np, it == 0 ? m->rows.var : m->rows.delta, s));
        NNGP_TRY(launch_f32_to_f64_mat(m->b32, np, m->z64, np, mp, np, true, s));
    }
    if (measure) NNGP_TRY(launch_rowdot_f64(m->z64, nullptr, 0.0, rhs, np, mp, np, nullptr, 1.0, m->rows.tol, s));
    return 0;
}

// Continues the correction of z64 (rows of rhs (K + reg I)^-1) by preconditioned CG, every row with its own scalars,
// until each row's step no longer lowers e^T A e by more than rows.tol[row] (twice in a row) or max_iters is reached.
// In: z64 and its residual r64 = rhs - z64 (K + reg I); out: both updated.  The stationary sweeps contract by the
// spectral radius of I - M^-1 A, which approaches 1 when cond(K + reg I) * eps32 does (small diag_reg, low-dimensional
// encodings: seen at cond 2.6e7, where four sweeps left 4e-2 in the variance); CG on the same operator needs ~the
// iteration count of the alpha solve.  One host read-back per iteration (the count of rows still iterating).
int rows_pcg_continue(nngp_model* m, int64_t mp, int max_iters, hipStream_t s) {
    const int64_t np = m->np;
    RowsPcg& w = m->rows;
    if (mp > m->rows_pq_cap) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        dev_free(w.p); dev_free(w.q);
        w.p = w.q = nullptr;
        NNGP_TRY(dev_alloc(&w.p, mp * m->np_cap));
        NNGP_TRY(dev_alloc(&w.q, mp * m->np_cap));
        m->rows_pq_cap = mp;
    }
    m->cov_iters = 0;
    for (int it = 0; it < max_iters; ++it) {
        NNGP_TRY(launch_convert_f64_f32(m->r64, np, m->b32, np, mp, np, mp, np, s));
        NNGP_TRY(apply_inverse_f32(m, mp, s));
        NNGP_TRY(launch_rows_rho(m->r64, m->b32, np, mp, np, it == 0, w, s));
        NNGP_TRY(launch_rows_update_p(w.p, m->b32, np, mp, np, w, s));
        NNGP_TRY(launch_gemm_nt_f64(w.q, np, w.p, np, w.p, np, m->k64, m->ld, mp, np, np, 1.0, m->reg, s));
        NNGP_TRY(launch_rows_alpha(w.p, w.q, np, mp, np, w, s));
        NNGP_TRY(launch_rows_axpy2(m->z64, m->r64, w.p, w.q, np, mp, np, w, s));
        NNGP_HIP_CHECK(hipMemcpyAsync(w.host, w.live, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        NNGP_HIP_CHECK(hipStreamSynchronize(s));
        m->cov_iters = it + 1;
        if (w.host[0] == 0) break;
    }
    return 0;
}

int build_cross(nngp_model* m, const double* xt, const double* qt, int64_t mt, int64_t mp, bool nngp, double* out,
                hipStream_t s) {
    NNGP_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double) * mp * m->np, s));
    BuildArgs a{};
    a.x1 = xt; a.x2 = m->x; a.q1 = qt; a.q2 = m->q;
    a.n1 = mt; a.n2 = m->n; a.d = m->d;
    a.row_begin = 0; a.row_end = mt; a.sym = 0;
    a.ld64 = a.ld32 = m->np;
    if (nngp) a.nngp64 = out; else a.ntk64 = out;
    return launch_kernel_build(a, m->arch, s);
}

}  // namespace nngp

extern "C" {

// Serving mode (the reference's Estimator keeps its Cholesky factor and calls cho_solve per query
// batch, estimator.py:34-67).  Builds X = (K + reg I)^-1 explicitly in float64: rows of the identity, 1024 at a time,
// through the same float32-solve + float64-correction machinery as the covariance (two sweeps; by CG to convergence when
// the factor is a weak preconditioner), then X <- (X + X^T) / 2.  Cost ~ N/1024 covariance-sized solves, once per fit;
// afterwards predict forms Z = K_td X with ONE float64 product and no solves.
int nngp_model_prepare_serving(nngp_model* m, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(m != nullptr && m->solved, "prepare_serving: fit the model first");
    NNGP_REQUIRE(!m->k64_partial, "prepare_serving: not with the row-sharded layout (the model holds only its own kernel rows)");
    if (m->get != NNGP_GET_NNGP) return 0;  // the NTK covariance has no second-order formula to absorb the inverse's error
    NNGP_TRY(run_pending_solve(m, s, true));  // its iteration count says how good the preconditioner is
    const int64_t n = m->n, np = m->np;
    const int64_t blk = np < 1024 ? np : 1024;
    NNGP_TRY(ensure_predict_capacity(m, blk, true));
    NNGP_TRY(ensure_refine_capacity(m, blk));
    if (!use_split_solves(m, blk)) NNGP_TRY(ensure_lt(m, s));
    if (m->ainv64 == nullptr) {
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        NNGP_TRY(dev_alloc(&m->ainv64, m->np_cap * m->np_cap));
    }
    const bool weak = m->iters >= 8 || m->reg_fac > m->reg;
    const double shift = (m->reg > 0.0 && m->reg_fac > m->reg) ? sqrt(m->reg_fac / m->reg) : 1.0;
    // The serving predictions SQUARE the inverse's error (second-order formula) on top of a cancellation of 1e3 .. 1e6: the
    // inverse has to converge to float64 accuracy proper, and the COARSE int8 residual's floor (2^-32 of the row maxima) stops the
    // sweeps four digits short of it -- measured: 4e-3 in the serving variances at N = 2500 against 1e-7 (scripts/i8s_hard_case.py).
    // Nor the FINE product: an inverse refined against it serves variances at 1e-9 .. 6e-9 of level 3 where the float64 pipe's
    // reaches 2e-11 .. 5e-11, for 10-20 % of the build time (scripts/i8s_serving_build.py).
    struct Suspend { bool& f; explicit Suspend(bool& b) : f(b) { f = true; } ~Suspend() { f = false; } } suspend(m->i8_suspended);
    for (int64_t r0 = 0; r0 < n; r0 += blk) {
        const int64_t rows = (n - r0 < blk) ? n - r0 : blk, rp = round_up(rows, TB);
        NNGP_TRY(launch_identity_rows(m->ktd64, np, np, r0, rows, rp, s));
        // three sweeps: the rows' errors add up coherently in Z = K_td X (two left 1.4e-3 in the variance at N = 32768)
        NNGP_TRY(refined_solve_rows(m, m->ktd64, rp, 3, weak, s));
        if (weak) {  // tolerance relative to the row's energy z.e = X_ii
            NNGP_TRY(launch_rowdot_f64(m->z64, m->ktd64, 1.0, nullptr, np, rows, np, nullptr, 1.0, m->rows.delta, s));
            NNGP_TRY(launch_rows_prepare(nullptr, nullptr, nullptr, m->rows.delta, 1, 0.0, rows, m->rows.tol, m->rows.live + 1, s));
            NNGP_TRY(rows_pcg_continue(m, rp, (int)fmin(1000.0, 80.0 * shift), s));
        }
        NNGP_HIP_CHECK(hipMemcpyAsync(m->ainv64 + r0 * np, m->z64, sizeof(double) * rows * np, hipMemcpyDeviceToDevice, s));
    }
    if (np > n) NNGP_HIP_CHECK(hipMemsetAsync(m->ainv64 + n * np, 0, sizeof(double) * (np - n) * np, s));
    NNGP_TRY(launch_symmetrize_f64(m->ainv64, np, np, s));
    m->serving_ready = true;
    m->serving_weak = weak;
    return 0;
}

static int predict_impl(nngp_model* m, const double* x_test, int64_t mt, int32_t cov_mode, double* mean, double* var_or_cov, void* stream);

int nngp_model_predict(nngp_model* m, const double* x_test, int64_t mt, int32_t cov_mode, double* mean,
                       double* var_or_cov, void* stream) {
    const int rc = predict_impl(m, x_test, mt, cov_mode, mean, var_or_cov, stream);
    // whatever path the predict took, the caller's stream leaves behind the build of the inverted blocks it may have forked (tri_fork)
    if (m != nullptr && !m->tri_stale && m->tri_pending) (void)hipStreamWaitEvent((hipStream_t)stream, m->ev_tri, 0);
    return rc;
}

static int predict_impl(nngp_model* m, const double* x_test, int64_t mt, int32_t cov_mode, double* mean, double* var_or_cov,
                        void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(m != nullptr && m->solved, "predict: fit the model first");
    NNGP_REQUIRE(cov_mode >= NNGP_COV_NONE && cov_mode <= NNGP_COV_FULL, "predict: bad cov_mode");
    NNGP_REQUIRE(mean != nullptr && (cov_mode == NNGP_COV_NONE || var_or_cov != nullptr), "predict: NULL output");
    const bool on_train = (x_test == nullptr);
    if (on_train) mt = m->n;
    NNGP_REQUIRE(mt >= 0, "predict: negative row count");
    NNGP_REQUIRE(!m->k64_partial || (cov_mode == NNGP_COV_NONE && !on_train),
                 "predict: this model holds only its own kernel rows (row-sharded layout): it serves means of test rows; covariances are "
                 "formed by the caller from nngp_model_apply_factor and its row block (shard32.py)");
    if (mt == 0) return 0;
    const bool is_ntk = (m->get == NNGP_GET_NTK);
    const int64_t n = m->n, np = m->np, mp = round_up(mt, TB);
    const bool compact_train = on_train && m->ld != m->np;  // K_dd rows have stride ld: copy them to the compact cross buffer
    NNGP_TRY(ensure_predict_capacity(m, mt, !on_train || compact_train));
    // A predict whose covariance will take the int8 residual product right after a fit: the digit planes of K (13 N^2 bytes of HBM
    // traffic, 2.7 ms alone at N = 32768) are cut beside the predict's first kernels.  The cut's workgroups hold 132 KB of LDS and cannot
    // share a compute unit with a workgroup of the persistent solves, so what is left of it when the first solve starts waits for the
    // gaps between and behind the solves: it is enqueued FIRST (round 5: before the cross kernel and the inverted blocks, on the
    // look-ahead's idle update stream -- it needs nothing but K) and has the chip until then.  (debug key 5 = 53: in stream order
    // where the planes are first needed)
    if (cov_mode != NNGP_COV_NONE && m->var_refine >= 1 && use_i8s(m, mp) && !(m->serving_ready && !is_ntk) && NNGP_KNOB(5) != 53) {
        if (cov_mode == NNGP_COV_FULL && use_i8s_fine(m, mp)) m->i8_want_fine = true;  // a full covariance runs at level >= 2: later residuals
        const int rc_i8 = ensure_i8s(m, mp, m->i8.k, i8s_planes_policy(m));
        if (rc_i8 != 0 && rc_i8 != 1) return rc_i8;
        if (rc_i8 == 0 && !m->i8.k.ready) {
            if (m->ev_i8 == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_i8, hipEventDisableTiming));
            NNGP_HIP_CHECK(hipEventRecord(m->ev_i8, s));  // K is complete; earlier readers of the planes are behind us
            // With the alpha CG running from the predict's start (cg_from_the_start) the cut must not sit in front of it on the solve
            // stream, nor in front of the inverted blocks on the panel stream
            static const int cut_env = getenv("NNGP_TK_CUTSTREAM") ? atoi(getenv("NNGP_TK_CUTSTREAM")) : 2;  // (development aid)
            hipStream_t cs = m->solve_stream;
            if (cut_env != 0 && cg_from_the_start(m, mp) && m->la != nullptr && m->la->panel != nullptr)
                cs = (cut_env == 2 && m->la->update != nullptr) ? m->la->update : m->la->panel;
            NNGP_HIP_CHECK(hipStreamWaitEvent(cs, m->ev_i8, 0));
            NNGP_TRY(i8s_cut_planes(m, m->i8.k, m->k64, m->ld, cs));
            NNGP_HIP_CHECK(hipEventRecord(m->ev_i8, cs));
            m->i8_k_pending = true;
        }
    }
    NNGP_TRY(tri_fork(m, s));  // first predict of a fit: the factor's inverted blocks are built beside the cross-kernel build

    // ---- cross kernel of `get` and the mean: mu = K_td alpha (float64) ----
    const double* xt = on_train ? m->x : x_test;
    const double* ktd = m->k64;  // x_test=None: K_td = K_dd (estimator.py:37-40); its padding is zero
    const double* qt = m->q;
    if (!on_train) {
        NNGP_TRY(launch_row_sqnorm(xt, mt, m->d, m->xt_q, s));
        qt = m->xt_q;
        NNGP_TRY(build_cross(m, xt, qt, mt, mp, !is_ntk, m->ktd64, s));
        ktd = m->ktd64;
    }
    if (compact_train) {
        NNGP_HIP_CHECK(hipMemsetAsync(m->ktd64, 0, sizeof(double) * mp * np, s));
        NNGP_HIP_CHECK(hipMemcpy2DAsync(m->ktd64, sizeof(double) * np, m->k64, sizeof(double) * m->ld, sizeof(double) * np, mt,
                                        hipMemcpyDeviceToDevice, s));
        ktd = m->ktd64;
    }
    // The covariance does not depend on alpha: it is enqueued first, then the deferred CG solve runs on its own stream
    // (overlapping it), and the mean follows once alpha is there.
    // Levels >= 2 check afterwards whether the fixed number of correction sweeps was enough (cov_adaptive below):
    // 0 nothing to check, 1 NNGP diag, 2 NNGP full, 3 NTK, 4 NNGP diag at level 1.
    if (m->i8_guard_pending && hipEventQuery(m->ev_guard) == hipSuccess) {  // the previous level-1 batch's estimate has arrived
        double ratio = 0.0;
        memcpy(&ratio, m->i8_guard_host, sizeof(double));
        m->i8_guard_pending = false;
        if (ratio > m->i8_floor_ratio) m->i8_floor_ratio = ratio;
        if (!(ratio <= (NNGP_KNOB(5) == 56 ? 0.0 : kI8FloorThr))) m->i8_distrusted = true;  // float64 pipe from this predict on (key 5 = 56: test)
    }
    int check_kind = 0;
    bool i8_check_pending = false;
    m->i8_used_now = false;
    m->gate_recorded = false;
    bool z_valid = false;  // z64 ends up holding the rows K_td (K + reg I)^-1 (to first order): the mean can be corrected through them
    const bool full = (cov_mode == NNGP_COV_FULL);
    const double* ntk_cross = nullptr;  // NNGP cross kernel of the NTK covariance
    auto ntk_finish = [&]() -> int {    // from z64 = Theta_td (Theta_dd + reg I)^-1; K_tt already in ktt64 (full)
        // W = Z K_dd: its error is the variance's -- float64 grade proper (the FINE product where it pays)
        int rc_w = use_i8s_fine(m, mp) ? ensure_i8s(m, mp, m->i8.aux, kI8FinePlanes) : 1;
        if (rc_w == 0) {
            NNGP_TRY(i8s_product_rows(m, m->i8.aux, m->kaux64, np, m->r64, nullptr, 0.0, 1.0, m->z64, 0.0, mp, s, I8_FINE));
        } else {
            if (rc_w != 1) return rc_w;
            NNGP_TRY(launch_gemm_nt_f64(m->r64, np, nullptr, 0, m->z64, np, m->kaux64, np, mp, np, np, 1.0, 0.0, s));
        }
        if (!full)  // var_i = K_tt,ii + z_i . (w_i - 2 k_i)
            return launch_rowdot_f64(m->z64, ntk_cross, -2.0, m->r64, np, mt, np, m->tt_diag, 1.0, var_or_cov, s);
        NNGP_TRY(launch_axpby_mat(m->r64, 1.0, ntk_cross, -1.0, np, mp, np, s));  // G = W - K_td
        NNGP_TRY(launch_gemm_nt_f64(m->covp64, mp, m->ktt64, mp, m->r64, np, m->z64, np, mp, mp, np, 1.0, 1.0, s));
        NNGP_TRY(launch_gemm_nt_f64(m->covp64, mp, m->covp64, mp, m->z64, np, ntk_cross, np, mp, mp, np, -1.0, 1.0, s));
        return launch_copy_mat_f64(m->covp64, mp, var_or_cov, mt, s);
    };
    // Row flag: a LOWER bound of the remaining relative variance error above this value (debug key 6 = e >= 2: 10^-e).
    // The bound is loose -- measured 1e-10 where the error is 4e-7 (N = 32768, scripts/flag_study.py) -- so it is only the
    // backstop for fits whose alpha solve says nothing about the conditioning (e.g. y = 0 converges at once).
    // NTK: predicted relative variance error after the two sweeps (k_sweep_estimate) above which the rows go on by CG.
    // scripts/ntk_est_study.py, 54 fits (profiles/r2_ntk_est_study.jsonl): the estimate is within 0.4x .. 120x of the error
    // wherever that exceeds 1e-8; every fit it passes is within 5e-7, the five it sends on had 7e-6 .. 6.5e-5; the
    // bench-sized fits (N = 4096 .. 16384) sit at 3.5e-9 .. 8.2e-8.
    const double kSweepEstThr = 3e-7;
    const double kFlagThr = (NNGP_KNOB(6) >= 2 && NNGP_KNOB(6) <= 30) ? pow(10.0, -(double)NNGP_KNOB(6)) : 1e-8;
    auto cov_part = [&]() -> int {
    if (full) NNGP_TRY(ensure_full_cov_capacity(m, mt));
    NNGP_TRY(launch_diag_from_q(qt, mt, m->arch, m->tt_diag, nullptr, s));  // NNGP K(x_t, x_t)
    auto build_ktt = [&]() -> int {  // NNGP K_tt [mt, mt] into ktt64 (ld = mp)
        BuildArgs a{};
        a.x1 = xt; a.x2 = xt; a.q1 = qt; a.q2 = qt;
        a.n1 = mt; a.n2 = mt; a.d = m->d;
        a.row_begin = 0; a.row_end = mt; a.sym = 1;
        a.ld64 = a.ld32 = mp;
        a.nngp64 = m->ktt64;
        return launch_kernel_build(a, m->arch, s);
    };

    if (!is_ntk && m->var_refine == 0) {
        // float32 only: V^T = K_td L^-T, cov = K_tt - V^T V  (fast; error ~ cond * eps32 relative to the prior)
        NNGP_TRY(launch_convert_f64_f32(ktd, np, m->b32, np, mp, np, mp, np, s));
        NNGP_TRY(apply_forward_f32(m, mp, s));
        if (!full) return launch_row_sqsum_f32(m->b32, np, mt, np, m->tt_diag, var_or_cov, s);
        NNGP_TRY(build_ktt());
        NNGP_TRY(launch_gemm_nt_f32(m->vvt32, mp, m->b32, np, m->b32, np, mp, mp, np, 1.0f, 0.0f, false, s));
        return launch_cov_finish(m->ktt64, mp, m->vvt32, mp, mt, var_or_cov, s);
    }

    const bool serving = m->serving_ready && !is_ntk;  // Z = K_td X with the explicit float64 inverse: one product, no solves
    if (!serving && !use_split_solves(m, mp)) NNGP_TRY(ensure_lt(m, s));  // float32 L^T: only the float32 solve path reads it
    NNGP_TRY(ensure_refine_capacity(m, mp));
    // z64 ~ rows of K_td (K + reg I)^-1 [and r64 = their residual]: float32 solves + float64 sweeps, or -- serving mode --
    // one product with the explicit inverse.  Either way z64 only has to be GOOD, not exact: the second-order formulas
    // below square its error.  (K_td X alone would not do: k^T X k cancels to the variance from terms 1e3..1e6 larger,
    // and no float64 inverse is accurate to that.)
    auto solve_rows = [&](int sweeps, bool final_residual) -> int {
        if (!serving) return refined_solve_rows(m, ktd, mp, sweeps, final_residual, s);
        NNGP_TRY(launch_gemm_nt_f64(m->z64, np, nullptr, 0, ktd, np, m->ainv64, np, mp, np, np, 1.0, 0.0, s));
        if (m->serving_weak) {  // ill-conditioned fit: X is less accurate; one correction step with X as the solver
            NNGP_TRY(residual_rows(m, m->r64, ktd, m->z64, mp, s, false));
            NNGP_TRY(launch_gemm_nt_f64(m->z64, np, m->z64, np, m->r64, np, m->ainv64, np, mp, np, np, 1.0, 1.0, s));
        }
        if (final_residual) {
            NNGP_TRY(residual_rows(m, m->r64, ktd, m->z64, mp, s, false));
        }
        return 0;
    };
    if (!is_ntk) {
        // NNGP: cov_ij = K_tt,ij - k_i^T A^-1 k_j with Z ~ K_td A^-1 (float32 solve + float64 correction sweeps).
        //   level 1: one sweep, cov = K_tt - sym(Z K_dt)  (error ~ rho * float32 error); diag only: see below
        //   level L >= 2: L-1 sweeps, then with R = K_td - Z A:  k_i^T A^-1 k_j = sym(z_i . (k_j + r_j)) + O(err^2)
        // the explicit inverse needs the second-order formula; the full covariance has no cheap remainder estimate: level >= 2
        const int level = ((serving || (full && m->var_refine == 1)) && m->var_refine < 2) ? 2 : m->var_refine;
        const bool second_order = level >= 2;
        if (!full && level == 1) {
            // diag, ONE float64 product and three triangular solves (the default).  With z0 = M^-1 k from the float32 factor
            // M = L L^T and r0 = k - A z0:   k^T A^-1 k = z0.(k + r0) + e0^T A e0   exactly, for whatever z0 the solves
            // returned, and   e0^T A e0 = r0^T A^-1 r0 ~ r0^T M^-1 r0 = |L^-1 r0|^2   -- a forward solve only, and only the
            // correction term (<~ 1e-2 of the variance) depends on it.  Measured at N = 32768 against level 4: 2.1e-6 worst
            // relative error (level 2: 4.2e-7 for one more solve pair and half a float64 product; the formula without the
            // last term, z0.(k + r0) via the quadratic form alone: 4.7e-3 -- e0^T A e0 is NOT negligible).
            // (an int8 residual's combination pass also leaves z.(k + r0), z.r0, the statistics of z and float32(r0): I8Fuse)
            m->i8_fuse_request = true;
            m->i8_fuse_done = false;
            const int rc_solve = solve_rows(0, true);
            m->i8_fuse_request = false;
            NNGP_TRY(rc_solve);
            z_valid = true;
            if (m->i8_fuse_done) {
                NNGP_TRY(launch_i8s_rowstat_finish(m->i8.rowpart, np, mt, m->tt_diag, var_or_cov, m->rows.delta, m->rows.zstat, s));
            } else {
                NNGP_TRY(launch_rowdot_f64(m->z64, ktd, 1.0, m->r64, np, mt, np, m->tt_diag, -1.0, var_or_cov, s));
                NNGP_TRY(launch_rowdot_f64(m->z64, nullptr, 0.0, m->r64, np, mt, np, nullptr, 1.0, m->rows.delta, s,
                                           m->i8_used_now ? m->rows.zstat : nullptr));
                NNGP_TRY(launch_convert_f64_f32(m->r64, np, m->b32, np, mp, np, mp, np, s));
            }
            NNGP_TRY(apply_forward_f32(m, mp, s));
            NNGP_TRY(launch_row_sqsum_f32(m->b32, np, mt, np, var_or_cov, var_or_cov, s));
            check_kind = 4;  // like 1, and r64 already holds the residual of z64
            if (m->i8_used_now) {  // an int8 residual: what may the dropped digit pairs have cost THIS batch's variances?
                I8Plan pl;
                NNGP_TRY(i8s_plan(m->i8.ns_z, m->i8.ns_k, m->i8.cut, &pl));
                if (m->i8_guard == nullptr) {
                    NNGP_TRY(dev_alloc(&m->i8_guard, 1));
                    NNGP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m->i8_guard_host), sizeof(unsigned long long), hipHostMallocDefault));
                }
                if (m->ev_guard == nullptr) NNGP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_guard, hipEventDisableTiming));
                NNGP_HIP_CHECK(hipMemsetAsync(m->i8_guard, 0, sizeof(unsigned long long), s));
                NNGP_TRY(launch_i8s_floor_ratio_rows(m->rows.zstat, mt, var_or_cov, pl, m->i8.ns_z, m->i8.ns_k, m->i8.k.scale + m->np_cap,
                                                     m->i8_guard, s));
                if (!m->i8_checked) {
                    i8_check_pending = true;   // first predict of the fit: waits for its own estimate (below)
                } else {
                    NNGP_HIP_CHECK(hipMemcpyAsync(m->i8_guard_host, m->i8_guard, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                    NNGP_HIP_CHECK(hipEventRecord(m->ev_guard, s));
                    m->i8_guard_pending = true;  // looked at when the next predict starts
                }
            }
            return launch_rows_prepare(m->rows.delta, m->tt_diag, var_or_cov, nullptr, 0, kFlagThr, mt, m->rows.tol,
                                       m->rows.live + 1, s);
        }
        if (!full && serving && mt <= (np <= 16384 ? 32 : 16) && !m->serving_weak) {
            z_valid = true;
            // a handful of queries against the explicit inverse: two passes over N x N float64 matrices (X, then K) per
            // group of 8 queries, both HBM-bound; var = K_tt - (2 z.k - z^T (K + reg I) z).  (Measured, ms per call,
            // streamed groups / 128-row MFMA GEMM -- N = 10800: 8 queries 0.62 / 3.06; N = 32768: 4.07 / 9.29.)
            for (int64_t g = 0; g < mt; g += 8) {
                const int64_t gm = (mt - g < 8) ? mt - g : 8;
                NNGP_TRY(launch_skinny_nt_f64(m->z64 + g * np, np, nullptr, 0, ktd + g * np, np, m->ainv64, np, gm, np, np, 1.0, 0.0, s));
                NNGP_TRY(launch_skinny_nt_f64(m->r64 + g * np, np, nullptr, 0, m->z64 + g * np, np, m->k64, m->ld, gm, np, np, 1.0, 0.0, s));
            }
            NNGP_TRY(launch_axpby_mat(m->r64, -1.0, m->z64, -m->reg, np, mt, np, s));
            return launch_rowdot_f64(m->z64, ktd, 2.0, m->r64, np, mt, np, m->tt_diag, -1.0, var_or_cov, s);
        }
        if (!full && second_order) {
            // diag only: z.(k + r) = 2 z.k - z^T (K + reg I) z, and the quadratic form needs only the lower triangle of
            // the symmetric K: W = 2 Z strict_lower_blocks(K) + Z diag_blocks(K) -- HALF the float64 product that the
            // full residual costs (kmode 1 / 2 of the float64 GEMM), then var = K_tt - z.(2 k - W - reg z)
            NNGP_TRY(solve_rows(level - 1, false));
            z_valid = true;
            NNGP_TRY(launch_gemm_nt_f64(m->r64, np, nullptr, 0, m->z64, np, m->k64, m->ld, mp, np, np, 2.0, 0.0, s, 1));
            NNGP_TRY(launch_gemm_nt_f64(m->r64, np, m->r64, np, m->z64, np, m->k64, m->ld, mp, np, np, 1.0, 1.0, s, 2));
            NNGP_TRY(launch_axpby_mat(m->r64, -1.0, m->z64, -m->reg, np, mp, np, s));
            NNGP_TRY(launch_rowdot_f64(m->z64, ktd, 2.0, m->r64, np, mt, np, m->tt_diag, -1.0, var_or_cov, s));
            // delta = z.k - z^T A z: the first-order term that the formula above cancels (cov_adaptive reads the flags)
            NNGP_TRY(launch_rowdot_f64(m->z64, ktd, 1.0, m->r64, np, mt, np, nullptr, 1.0, m->rows.delta, s));
            if (serving) return 0;  // the inverse was refined to convergence when it was built
            check_kind = 1;
            return launch_rows_prepare(m->rows.delta, m->tt_diag, var_or_cov, nullptr, 0, kFlagThr, mt, m->rows.tol,
                                       m->rows.live + 1, s);
        }
        NNGP_TRY(solve_rows(second_order ? level - 1 : 1, second_order));
        z_valid = true;
        if (!full)
            return launch_rowdot_f64(m->z64, ktd, 1.0, second_order ? m->r64 : nullptr, np, mt, np, m->tt_diag, -1.0,
                                     var_or_cov, s);
        const double* g = ktd;
        if (second_order) {
            NNGP_TRY(launch_rowdot_f64(m->z64, ktd, 1.0, m->r64, np, mt, np, m->tt_diag, -1.0, m->rows.var, s));
            NNGP_TRY(launch_rowdot_f64(m->z64, nullptr, 0.0, m->r64, np, mt, np, nullptr, 1.0, m->rows.delta, s));
            NNGP_TRY(launch_rows_prepare(m->rows.delta, m->tt_diag, m->rows.var, nullptr, 0, kFlagThr, mt, m->rows.tol,
                                         m->rows.live + 1, s));
            check_kind = serving ? 0 : 2;
            NNGP_TRY(launch_axpby_mat(m->r64, 1.0, ktd, 1.0, np, mp, np, s));  // G = K_td + R
            g = m->r64;
        }
        NNGP_TRY(build_ktt());
        NNGP_TRY(launch_gemm_nt_f64(m->covp64, mp, m->ktt64, mp, m->z64, np, g, np, mp, mp, np, -1.0, 1.0, s));
        return launch_copy_mat_f64(m->covp64, mp, var_or_cov, mt, s);
    }

    // NTK (train.py --kernel_type ntk): with Z = Theta_td (Theta_dd + reg I)^-1,
    //   cov = K_tt + Z K_dd Z^T - (K_td Z^T + h.c.),  K = NNGP kernels.
    if (!m->aux_ready) {
        if (m->kaux64 == nullptr) {
            NNGP_HIP_CHECK(hipDeviceSynchronize());
            NNGP_TRY(dev_alloc(&m->kaux64, m->np_cap * m->np_cap));
        }
        BuildArgs a{};
        a.x1 = m->x; a.x2 = m->x; a.q1 = m->q; a.q2 = m->q;
        a.n1 = n; a.n2 = n; a.d = m->d;
        a.row_begin = 0; a.row_end = n; a.sym = 1;
        a.ld64 = a.ld32 = np;
        a.nngp64 = m->kaux64;
        a.no_comp = 1;  // the same bits as when the fit's own build writes it (nngp_model_build_rows)
        NNGP_TRY(launch_kernel_build(a, m->arch, s));
        NNGP_TRY(launch_zero_pad_f64(m->kaux64, np, n, np, s));
        m->aux_ready = true;
        m->i8.aux.ready = false;
    }
    const double* ktd_n = m->kaux64;  // NNGP cross kernel; x_test=None: K_dd itself
    if (!on_train) {
        if (mp > m->ktd_aux_cap) {
            NNGP_HIP_CHECK(hipDeviceSynchronize());
            dev_free(m->ktd_aux);
            NNGP_TRY(dev_alloc(&m->ktd_aux, mp * m->np_cap));
            m->ktd_aux_cap = mp;
        }
        NNGP_TRY(build_cross(m, xt, qt, mt, mp, true, m->ktd_aux, s));
        ktd_n = m->ktd_aux;
    }
    NNGP_TRY(refined_solve_rows(m, ktd, mp, ntk_sweeps(m), false, s, m->var_refine >= 1));
    ntk_cross = ktd_n;
    z_valid = true;
    if (full) NNGP_TRY(build_ktt());
    if (m->var_refine >= 1) check_kind = 3;
    NNGP_TRY(ntk_finish());
    if (check_kind == 3) {
        // b32 still holds the second correction; r64 = Z K_dd (diag) or Z K_dd - K_td (full): see ntk_finish
        NNGP_TRY(launch_rows_dvar(m->b32, m->r64, full ? nullptr : ntk_cross, -1.0, np, mt, np, m->rows.coef, s));
        NNGP_TRY(launch_sweep_estimate(m->rows.var, m->rows.delta, m->rows.tol, m->rows.coef, var_or_cov, full ? mt + 1 : 1,
                                       mt, reinterpret_cast<double*>(m->rows.live + 2), s));
    }
    return 0;
    };
    if (cov_mode != NNGP_COV_NONE) NNGP_TRY(cov_part());
    NNGP_TRY(run_pending_solve(m, s, true, z_valid));
    // Were the fixed sweeps enough?  Two signs that the float32 factor is a weak preconditioner: the alpha solve needed
    // many CG iterations, or a row's first-order term is too large for its second-order error to be small (k_rows_prepare).
    // Then the rows go on by preconditioned CG until each has converged, and the covariance is formed again.
    // (debug key 6 = 1: fixed sweeps only.)
    if (i8_check_pending) {
        // (the host has just waited for the alpha solve: this read-back waits for the covariance kernels enqueued before it, once
        // per fit -- later predicts on the fit stay asynchronous)
        NNGP_HIP_CHECK(hipMemcpyAsync(m->i8_guard_host, m->i8_guard, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        NNGP_HIP_CHECK(hipStreamSynchronize(s));
        memcpy(&m->i8_floor_ratio, m->i8_guard_host, sizeof(double));
        m->i8_checked = true;
        const double thr = NNGP_KNOB(5) == 56 ? 0.0 : kI8FloorThr;  // key 5 = 56: distrust whatever the estimate says (test)
        if (!(m->i8_floor_ratio <= thr)) {
            m->i8_distrusted = true;  // use_i8s is false from here on: the covariance again, on the float64 pipe
            m->i8_used_now = false;
            NNGP_TRY(cov_part());
        }
    }
    m->cov_iters = 0;
    m->sweep_est = m->sweep_est_var = -1.0;
    if (check_kind != 0 && NNGP_KNOB(6) != 1) {
        // Iterations of the alpha solve against the variance error of the fixed sweeps, 72 random fits of
        // tests/test_gpu_parity.py (N <= 5200): <= 5: <= 1e-7, 6: <= 1e-5, 7: <= 5e-5, >= 9: up to 8e-2; the bench sizes
        // need 5 (N = 32768) and 6 (N = 65536).  NTK covariance has no second-order formula: stricter.
        // (NTK: two sweeps leave ~rho^3 where the NNGP formula leaves ~rho^4: 6 iterations ~ 2e-4, 5 ~ 6e-6, 4 ~ 4e-8 -- the
        // threshold was 4 until round 2, which sent the N = 16384 bench config (4 iterations, 3.6e-8 after the sweeps) through
        // three continuation steps, 78 ms per predict instead of 37)
        // (round 3: the full covariance continues from 7 iterations too -- a 7-iteration fit of the large-N sweep, N = 5007, d = 3, left 2.1e-4 in
        // its diagonal where the diag path, which already continued from 7, was at 8e-9)
        bool weak = m->iters >= (is_ntk ? 6 : ((check_kind == 4 || check_kind == 2) ? 7 : 8)) || m->reg_fac > m->reg;
        if (!weak && check_kind == 3) {
            // NTK below 6 iterations: the count alone does not separate 4e-8 (N = 16384, d = 256: 4 iterations) from 5e-5
            // (N = 907, d = 2, four layers, diag_reg 1e-4: also 4).  What the two sweeps removed does: 16-byte read-back.
            NNGP_HIP_CHECK(hipMemcpyAsync(m->rows.host + 2, m->rows.live + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
            NNGP_HIP_CHECK(hipStreamSynchronize(s));
            memcpy(&m->sweep_est, m->rows.host + 2, sizeof(double));
            memcpy(&m->sweep_est_var, m->rows.host + 4, sizeof(double));
            weak = !(m->sweep_est_var <= kSweepEstThr);
        }
        // the row flag is the backstop for fits whose alpha solve says nothing (it converged in < 3 iterations, e.g.
        // y = 0); otherwise the iteration count decides and the call stays asynchronous
        if (!weak && check_kind != 3 && m->iters < 3) {
            NNGP_HIP_CHECK(hipMemcpyAsync(m->rows.host + 1, m->rows.live + 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            NNGP_HIP_CHECK(hipStreamSynchronize(s));
            weak = m->rows.host[1] > 0;
        }
        if (weak) {
            if (check_kind == 2) {
                NNGP_TRY(launch_axpby_mat(m->r64, 1.0, ktd, -1.0, np, mp, np, s));  // back from G = K_td + R to R
            } else if (check_kind == 4) {
                // level 1: r64 is the residual of z64 already -- but if it came from the int8 product, its error floor (~1e-3 of it)
                // would stay in the recursively updated residual and in the rows the CG converges to (seen in the parity sweep:
                // continued rows at 1e-6 .. 9e-6 instead of 1e-9): the continuation starts from the float64 residual proper
                if (use_i8s(m, mp)) NNGP_TRY(residual_rows(m, m->r64, ktd, m->z64, mp, s, false));
            } else {
                NNGP_TRY(residual_rows(m, m->r64, ktd, m->z64, mp, s, false));
            }
            if (check_kind == 3)  // tolerance from what the second sweep did to the variance (rows.tol holds z . k)
                NNGP_TRY(launch_rows_prepare_ntk(m->rows.delta, m->rows.coef, m->rows.tol, var_or_cov, full ? mt + 1 : 1, mt,
                                                 NNGP_KNOB(6) >= 2 ? pow(10.0, -(double)NNGP_KNOB(6)) : 1e-10, s));
            const double shift = (m->reg > 0.0 && m->reg_fac > m->reg) ? sqrt(m->reg_fac / m->reg) : 1.0;
            NNGP_TRY(rows_pcg_continue(m, mp, (int)fmin(1000.0, 80.0 * shift), s));
            if (check_kind == 3) {
                NNGP_TRY(ntk_finish());
            } else if (!full) {
                NNGP_TRY(launch_rowdot_f64(m->z64, ktd, 1.0, m->r64, np, mt, np, m->tt_diag, -1.0, var_or_cov, s));
            } else {
                NNGP_TRY(launch_axpby_mat(m->r64, 1.0, ktd, 1.0, np, mp, np, s));  // G = K_td + R
                NNGP_TRY(launch_gemm_nt_f64(m->covp64, mp, m->ktt64, mp, m->z64, np, m->r64, np, mp, mp, np, -1.0, 1.0, s));
                NNGP_TRY(launch_copy_mat_f64(m->covp64, mp, var_or_cov, mt, s));
            }
        }
    }
    for (int c = 0; c < m->ny; ++c)
        NNGP_TRY(launch_gemv_f64(ktd, np, mt, n, m->alpha + c, m->ny, mean + c, m->ny, 0.0, s));
    if (m->cg_partial && z_valid) {
        // the CG stopped at a_k with residual r_k = y - A a_k (still in the CG workspace):
        //   K_td A^-1 y = K_td a_k + (K_td A^-1) r_k = K_td a_k + Z r_k + (K_td A^-1 - Z) r_k,
        // and the last term is the product of two small errors (rows: <= 1e-3 relative; r_k: <= 1e-6 |y|)
        NNGP_TRY(launch_gemv_f64(m->z64, np, mt, n, m->pcg.r, 1, m->rows.delta, 1, 0.0, s));
        NNGP_TRY(launch_axpby_mat(mean, 1.0, m->rows.delta, 1.0, mt, 1, mt, s));
    }
    NNGP_HIP_CHECK(hipEventRecord(m->ev_predict, s));
    m->have_predict_event = true;
    return 0;
}

int nngp_model_apply_factor(nngp_model* m, float* b, int64_t rows, int32_t mode, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(m != nullptr && m->factored, "apply_factor: fit the model first");
    NNGP_REQUIRE(b != nullptr && rows > 0 && (mode == 0 || mode == 1), "apply_factor: bad arguments");
    NNGP_TRY(ensure_predict_capacity(m, rows, false));
    const int64_t mp = round_up(rows, TB), np = m->np, n = m->n;
    NNGP_HIP_CHECK(hipMemsetAsync(m->b32, 0, sizeof(float) * mp * np, s));
    NNGP_HIP_CHECK(hipMemcpy2DAsync(m->b32, sizeof(float) * np, b, sizeof(float) * n, sizeof(float) * n, rows, hipMemcpyDeviceToDevice, s));
    if (mode == 0) {
        NNGP_TRY(apply_forward_f32(m, mp, s));
    } else {
        if (!use_split_solves(m, mp)) NNGP_TRY(ensure_lt(m, s));  // float32 L^T: only the float32 solve path reads it
        NNGP_TRY(apply_inverse_f32(m, mp, s));
    }
    NNGP_HIP_CHECK(hipMemcpy2DAsync(b, sizeof(float) * n, m->b32, sizeof(float) * np, sizeof(float) * n, rows, hipMemcpyDeviceToDevice, s));
    return 0;
}
}  // extern "C"
