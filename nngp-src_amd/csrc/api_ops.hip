// Stand-alone operator entry points of the C ABI (no model object): the factorisation, the GEMMs, pool selection, the ticket tables --
// what tests/, scripts/ and the multi-GPU drivers call directly.
#include "model.h"

extern "C" {

int nngp_trsm_ticket_order(int32_t row_tiles, int32_t block_cols, int32_t tail_tiles, int32_t backward, int32_t workers, int32_t merged,
                           int32_t* items, int64_t cap, int64_t* count) {
    return tk_order_export(row_tiles, block_cols, tail_tiles, backward, workers, merged, 1, items, nullptr, cap, count);
}

int nngp_trsm_ticket_queues(int32_t row_tiles, int32_t block_cols, int32_t tail_tiles, int32_t backward, int32_t workers, int32_t merged,
                            int32_t queues, int32_t* items, int32_t* queue_of, int64_t cap, int64_t* count) {
    return tk_order_export(row_tiles, block_cols, tail_tiles, backward, workers, merged, queues, items, queue_of, cap, count);
}

int nngp_potrf_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, void* stream) {
    NNGP_REQUIRE(a != nullptr && dinv != nullptr, "potrf_f32: NULL argument");
    if (clamped) NNGP_HIP_CHECK(hipMemsetAsync(clamped, 0, sizeof(int32_t), (hipStream_t)stream));
    return potrf_f32(a, n, ld, dinv, clamped, 0.0f, (hipStream_t)stream);
}

int nngp_gemm_nt_f32(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m,
                     int64_t n, int64_t k, float alpha, float beta, int32_t lower_only, void* stream) {
    NNGP_REQUIRE(a != nullptr && b != nullptr && c != nullptr, "gemm_nt_f32: NULL argument");
    return launch_gemm_nt_f32(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, lower_only != 0, (hipStream_t)stream);
}

// Split workspace of nngp_gemm_nt_h3, kept between calls (round 4; until then every call allocated, synchronised and freed -- a host
// wait and a hipMalloc per block column inside the distributed factorisation's collective loop): one grow-only buffer per stream, a
// few streams at most; work on one stream is ordered, so the buffer is reused without waiting.
namespace {
struct H3Scratch {
    hipStream_t stream = nullptr;
    char* p = nullptr;
    size_t bytes = 0;
    bool used = false;
};
std::mutex g_h3_scratch_mutex;
H3Scratch g_h3_scratch[4];

int h3_scratch_get(hipStream_t s, size_t bytes, char** out) {
    std::lock_guard<std::mutex> lock(g_h3_scratch_mutex);
    H3Scratch* e = nullptr;
    for (auto& c : g_h3_scratch)
        if (c.used && c.stream == s) e = &c;
    if (e == nullptr)
        for (auto& c : g_h3_scratch)
            if (!c.used && e == nullptr) e = &c;
    if (e == nullptr) {  // every slot belongs to another stream: take the first one over once its work is through
        e = &g_h3_scratch[0];
        // (the stored handle may belong to a stream its owner has destroyed since: then the whole device is waited for instead)
        if (hipStreamSynchronize(e->stream) != hipSuccess) {
            (void)hipGetLastError();
            NNGP_HIP_CHECK(hipDeviceSynchronize());
        }
        (void)hipFree(e->p);
        *e = H3Scratch();
    }
    if (e->bytes < bytes) {
        if (e->p != nullptr) {
            NNGP_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(e->p);
            e->p = nullptr;
            e->bytes = 0;
        }
        NNGP_HIP_CHECK(hipMalloc((void**)&e->p, bytes));
        e->bytes = bytes;
    }
    e->used = true;
    e->stream = s;
    *out = e->p;
    return 0;
}
}  // namespace

int nngp_gemm_nt_h3(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int64_t n,
                    int64_t k, float alpha, float beta, float scale, int32_t lower_only, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(a != nullptr && b != nullptr && c != nullptr && scale > 0.0f, "gemm_nt_h3: bad argument");
    NNGP_REQUIRE(m > 0 && n > 0 && k > 0 && k % 32 == 0, "gemm_nt_h3: k must be a multiple of 32");
    const int64_t mp = round_up(m, 256), np = round_up(n, 256), ldp = 4 * k;
    char* pa = nullptr;
    NNGP_TRY(h3_scratch_get(s, (size_t)((mp + np) * ldp + 64), &pa));
    char* pb = pa + mp * ldp;
    int* counters = reinterpret_cast<int*>(pb + np * ldp);
    // the padding rows' products are never stored, but they are read: zero (finite) operands; the work counters start at zero
    if (mp > m) NNGP_HIP_CHECK(hipMemsetAsync(pa + m * ldp, 0, (size_t)((mp - m) * ldp), s));
    NNGP_HIP_CHECK(hipMemsetAsync(pb + n * ldp, 0, (size_t)((np - n) * ldp + 64), s));
    NNGP_TRY(launch_split_rows(a, lda, m, k, scale, pa, ldp, s));
    NNGP_TRY(launch_split_rows(b, ldb, n, k, scale, pb, ldp, s));
    return launch_gemm_nt_h3(c, ldc, pa, pb, ldp, m, n, k, alpha / (scale * scale), beta, lower_only != 0, 0, counters,
                             NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 0, s);
}

int nngp_gemm_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                     const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                     void* stream) {
    NNGP_REQUIRE(a != nullptr && b != nullptr && c != nullptr, "gemm_nt_f64: NULL argument");
    return launch_gemm_nt_f64(c, ldc, cin, ldcin, a, lda, b, ldb, m, n, k, alpha, beta, (hipStream_t)stream);
}

int nngp_gemm_nt_i8s(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda, const double* b,
                     int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta, int32_t slices_a, int32_t slices_b,
                     int32_t cut, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(a != nullptr && b != nullptr && c != nullptr && m > 0 && n > 0 && k > 0 && m % TB == 0 && n % TB == 0 && ldc >= n &&
                     lda >= k && ldb >= k,
                 "gemm_nt_i8s: m, n must be multiples of %d", TB);
    I8Plan pl;
    NNGP_TRY(i8s_plan(slices_a, slices_b, cut, &pl));
    const int64_t mp = round_up(m, 256), np = round_up(n, 256), kp = round_up(k, 128);
    const int64_t nchunk = i8s_chunks(k, &pl);
    const int64_t slab = m * n;
    const size_t plane_bytes = (size_t)((mp * slices_a + np * slices_b) * kp);
    const size_t part_bytes = (size_t)(nchunk * pl.ndiag * slab) * sizeof(int32_t);
    NNGP_REQUIRE(part_bytes < (size_t)16 << 30, "gemm_nt_i8s: the test entry keeps all partial products (16 GB limit)");
    char* ws = nullptr;
    NNGP_HIP_CHECK(hipMalloc((void**)&ws, plane_bytes + part_bytes + sizeof(double) * (size_t)(m + n) + 64));
    NNGP_HIP_CHECK(hipMemsetAsync(ws, 0, plane_bytes, s));
    int8_t* pa = reinterpret_cast<int8_t*>(ws);
    int8_t* pb = pa + mp * slices_a * kp;
    int32_t* part = reinterpret_cast<int32_t*>(ws + plane_bytes);
    double* sca = reinterpret_cast<double*>(ws + plane_bytes + part_bytes);
    double* scb = sca + m;
    int* counters = reinterpret_cast<int*>(scb + n);
    NNGP_HIP_CHECK(hipMemsetAsync(counters, 0, 64, s));
    int rc = launch_i8s_slice_rows(a, lda, m, k, slices_a, nullptr, sca, pa, kp, mp * kp, s);
    if (rc == 0) rc = launch_i8s_slice_rows(b, ldb, n, k, slices_b, nullptr, scb, pb, kp, np * kp, s);
    if (rc == 0)
        rc = launch_gemm_nt_i8s(part, n, slab, pa, kp, mp * kp, pb, kp, np * kp, pl, m, n, k, counters,
                                NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 0, s);
    if (rc == 0)
        rc = launch_i8s_combine(c, ldc, cin ? cin : c, cin ? ldcin : ldc, beta, alpha, nullptr, 0, 0.0, part, n, slab, (int)nchunk,
                                pl.ndiag, sca, scb, m, n, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(ws);
    return rc;
}

int nngp_pool_select(const double* mean, int64_t m, int32_t ny, const double* var, int64_t count, int32_t biased, uint64_t seed,
                     int64_t* indices, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(mean != nullptr && var != nullptr && indices != nullptr && m > 0 && ny >= 1 && count >= 0 && count <= m,
                 "pool_select: bad arguments (m=%lld, count=%lld)", (long long)m, (long long)count);
    if (count == 0) return 0;
    double* key = nullptr;
    NNGP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&key), sizeof(double) * (size_t)m, s));
    const int rc = launch_pool_select(mean, m, ny, var, count, biased != 0, seed, key, indices, s);
    NNGP_HIP_CHECK(hipFreeAsync(key, s));
    return rc;
}

int nngp_symv_f64(const double* a, int64_t lda, int64_t n, const double* x, double* y, double diag_add, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    NNGP_REQUIRE(a != nullptr && x != nullptr && y != nullptr && n > 0 && lda >= n, "symv_f64: bad arguments");
    const int64_t np = round_up(n, TB);
    double* part = nullptr;
    NNGP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&part), sizeof(double) * (size_t)(np / TB) * np, s));
    const int rc = launch_symv_f64(a, lda, n, x, y, diag_add, part, np, s);
    NNGP_HIP_CHECK(hipFreeAsync(part, s));
    return rc;
}

int nngp_trsm_rlt_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t n,
                      void* stream) {
    NNGP_REQUIRE(b != nullptr && l != nullptr && dinv != nullptr, "trsm_rlt_f32: NULL argument");
    NNGP_REQUIRE(m % TB == 0 && n % TB == 0 && m > 0 && n > 0, "trsm_rlt_f32: m, n must be multiples of %d", TB);
    return trsm_rlt_f32(b, ldb, m, l, ldl, dinv, n, (hipStream_t)stream);
}

}  // extern "C"
