// RCCL entry points of the C ABI (include/nngp_hip.h, section e): the ONE data-path collective of the row-block kernel
// shard -- an in-place all-gather of the row blocks of K over xGMI -- and the panel broadcast of the block-cyclic
// Cholesky.  Fills the slot of nt.batch(kernel_fn, device_count=...) in the reference (train.py:166-168), which would
// pmap row blocks of x1 over devices and gather the result.
//
// librccl is bound at run time (dlopen/dlsym), not at link time: a process that never shards does not need it, and a
// process that already mapped an RCCL (PyTorch's wheel carries one) keeps exactly one copy -- two RCCLs in one
// process each run their own bootstrap and kernels.  Lookup order: symbols already global in the process, a loaded
// librccl.so(.1), then the loader's search path, then /opt/rocm/lib.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <new>

#include "../../include/nngp_hip.h"

namespace nngp {
void set_error(const char* fmt, ...);
}

namespace {

typedef void* rccl_comm_t;
struct rccl_unique_id { char internal[128]; };  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128 (rccl.h)
enum { kRcclFloat32 = 7, kRcclFloat64 = 8 };     // ncclDataType_t values (rccl.h)

struct RcclApi {
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char where[256] = "";
};

RcclApi g_api;
std::once_flag g_once;

bool bind_from(void* h, const char* where) {
    RcclApi a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(h, "ncclBroadcast"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.Broadcast || !a.GetErrorString) return false;
    a.ok = true;
    snprintf(a.where, sizeof(a.where), "%s", where);
    g_api = a;
    return true;
}

void bind_rccl() {
    if (bind_from(RTLD_DEFAULT, "symbols already in the process")) return;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))
            if (bind_from(h, n)) return;
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* p : paths)
        if (void* h = dlopen(p, RTLD_NOW | RTLD_LOCAL))
            if (bind_from(h, p)) return;
}

int require_rccl() {
    std::call_once(g_once, bind_rccl);
    if (!g_api.ok) {
        nngp::set_error("RCCL not found: librccl.so(.1) is neither loaded in this process nor on the loader path");
        return -4;
    }
    return 0;
}

int rccl_check(int rc, const char* what) {
    if (rc == 0) return 0;
    nngp::set_error("%s failed: RCCL error %d (%s)", what, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
    return -5;
}

}  // namespace

struct nngp_comm {
    rccl_comm_t comm = nullptr;
    int world = 1, rank = 0;
};

extern "C" {

int nngp_comm_unique_id(void* id128) {
    if (id128 == nullptr) { nngp::set_error("comm_unique_id: NULL buffer"); return -2; }
    if (int rc = require_rccl()) return rc;
    rccl_unique_id id;
    if (int rc = rccl_check(g_api.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

int nngp_comm_create(nngp_comm** out, const void* id128, int32_t world, int32_t rank) {
    if (out == nullptr || id128 == nullptr) { nngp::set_error("comm_create: NULL argument"); return -2; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { nngp::set_error("comm_create: bad world/rank %d/%d", world, rank); return -2; }
    if (int rc = require_rccl()) return rc;
    nngp_comm* c = new (std::nothrow) nngp_comm();
    if (c == nullptr) { nngp::set_error("comm_create: out of host memory"); return -2; }
    rccl_unique_id id;
    memcpy(id.internal, id128, sizeof(id.internal));
    if (int rc = rccl_check(g_api.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) {
        delete c;
        return rc;
    }
    c->world = world;
    c->rank = rank;
    *out = c;
    return 0;
}

int nngp_comm_destroy(nngp_comm* c) {
    if (c == nullptr) return 0;
    int rc = 0;
    if (c->comm != nullptr && g_api.ok) rc = rccl_check(g_api.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

const char* nngp_comm_library(void) {
    return require_rccl() == 0 ? g_api.where : "";
}

// Rows [g*chunk, (g+1)*chunk) of k ([>= world*chunk, ld], chunk = ceil(n / world)) were written by rank g; on return every
// rank holds all of them.  In place: the send block is the rank's own slice of the receive buffer.
int nngp_allgather_rows(void* k, int64_t n, int64_t ld, int32_t dtype, nngp_comm* c, void* stream) {
    if (k == nullptr || c == nullptr || n <= 0 || ld <= 0) { nngp::set_error("allgather_rows: bad argument"); return -2; }
    if (dtype != NNGP_DTYPE_F32 && dtype != NNGP_DTYPE_F64) { nngp::set_error("allgather_rows: bad dtype"); return -2; }
    if (int rc = require_rccl()) return rc;
    const int64_t chunk = (n + c->world - 1) / c->world;
    const size_t esz = dtype == NNGP_DTYPE_F64 ? 8 : 4;
    const char* mine = static_cast<const char*>(k) + (size_t)c->rank * (size_t)chunk * (size_t)ld * esz;
    return rccl_check(g_api.AllGather(mine, k, (size_t)chunk * (size_t)ld, dtype == NNGP_DTYPE_F64 ? kRcclFloat64 : kRcclFloat32,
                                      c->comm, stream), "ncclAllGather");
}

int nngp_bcast(void* buf, int64_t count, int32_t dtype, int32_t root, nngp_comm* c, void* stream) {
    if (buf == nullptr || c == nullptr || count < 0 || root < 0 || root >= c->world) { nngp::set_error("bcast: bad argument"); return -2; }
    if (dtype != NNGP_DTYPE_F32 && dtype != NNGP_DTYPE_F64) { nngp::set_error("bcast: bad dtype"); return -2; }
    if (count == 0) return 0;
    if (int rc = require_rccl()) return rc;
    return rccl_check(g_api.Broadcast(buf, buf, (size_t)count, dtype == NNGP_DTYPE_F64 ? kRcclFloat64 : kRcclFloat32, root,
                                      c->comm, stream), "ncclBroadcast");
}

}  // extern "C"
