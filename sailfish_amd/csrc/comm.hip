// comm.hip -- RCCL (xGMI) transport for the multi-GPU path, bound at RUN time.
//
// SURVEY.md 8e / BASELINE.json configs[3]: classes sharded over the GPUs of a node, alpha replicated, one SUM all-reduce of
// alphaOut (M doubles) per EM iteration.  sfgpu_em_optimize_sharded (em.hip) runs that loop in C with a caller-supplied
// all-reduce; this file supplies the one a host gets for free on ROCm: ncclAllReduce enqueued on the handle's stream, so
// that NOTHING of the host runs between two iterations (the Python mirror used to make three calls per iteration --
// ctypes sweep, torch all_reduce, ctypes update -- which cost more than the ~30 us of device work they wrapped).
//
// libsfgpu.so does not LINK librccl (a single-GPU host needs none): the library is dlopen'ed on first use -- the copy the
// process already holds (PyTorch's) if there is one, /opt/rocm/lib's otherwise.  The reference has no counterpart (it is a
// single-node, shared-memory program: the per-transcript atomics of src/CollapsedEMOptimizer.cpp:224-281 are what the
// all-reduce replaces).
#include <dlfcn.h>

#include <mutex>

#include <rccl/rccl.h>          // types and enums only; every function is resolved with dlsym

#include "common.h"

namespace sfgpu {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};

static RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // a copy that is already mapped (PyTorch ships its own) wins: two RCCL instances in one process would not share state
        for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (api.lib) break; }
        if (!api.lib) for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.lib) break; }
        if (!api.lib) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
    });
    return api;
}

#define SF_RCCL(expr)                                                                                         \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            set_error("%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "rccl error"); \
            return SFGPU_ERR_HIP;                                                                             \
        }                                                                                                     \
    } while (0)

}  // namespace sfgpu

struct sfgpu_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
};

using namespace sfgpu;

extern "C" {

int sfgpu_comm_available(void) { return rccl().ok ? 1 : 0; }

int sfgpu_comm_unique_id(void* id_out) {
    SF_REQUIRE(id_out, SFGPU_ERR_INVALID, "sfgpu_comm_unique_id: null pointer");
    SF_REQUIRE(rccl().ok, SFGPU_ERR_STATE, "sfgpu_comm: librccl.so could not be loaded");
    ncclUniqueId id;
    SF_RCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == SFGPU_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, sizeof(id));
    return SFGPU_OK;
}

int sfgpu_comm_create(sfgpu_comm** out, const void* id128, int world, int rank) {
    SF_REQUIRE(out && id128, SFGPU_ERR_INVALID, "sfgpu_comm_create: null pointer");
    SF_REQUIRE(world >= 1 && rank >= 0 && rank < world, SFGPU_ERR_INVALID, "sfgpu_comm_create: need 0 <= rank < world");
    SF_REQUIRE(rccl().ok, SFGPU_ERR_STATE, "sfgpu_comm: librccl.so could not be loaded");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    sfgpu_comm* c = new sfgpu_comm();
    c->world = world; c->rank = rank;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);          // (uses the calling thread's current device)
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
        delete c;
        return SFGPU_ERR_HIP;
    }
    *out = c;
    return SFGPU_OK;
}

// how many ranks the communicator REALLY spans (ncclCommCount): what a benchmark line quotes as proof of what it ran on
int sfgpu_comm_count(sfgpu_comm* c, int* ranks) {
    SF_REQUIRE(c && c->comm && ranks, SFGPU_ERR_INVALID, "sfgpu_comm_count: null pointer");
    SF_REQUIRE(rccl().ok && rccl().CommCount, SFGPU_ERR_STATE, "sfgpu_comm: ncclCommCount is not available");
    SF_RCCL(rccl().CommCount(c->comm, ranks));
    return SFGPU_OK;
}

int sfgpu_comm_destroy(sfgpu_comm* c) {
    if (!c) return SFGPU_OK;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
    return SFGPU_OK;
}

int sfgpu_comm_allreduce_sum_f64(sfgpu_comm* c, double* d_buf, uint64_t n, sfgpu_stream stream) {
    SF_REQUIRE(c && c->comm && d_buf, SFGPU_ERR_INVALID, "sfgpu_comm_allreduce_sum_f64: null pointer");
    SF_RCCL(rccl().AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, as_stream(stream)));
    return SFGPU_OK;
}

// the callback form sfgpu_em_optimize_sharded takes: user = the sfgpu_comm
static int comm_allreduce_cb(double* d_buf, uint64_t n, void* user, sfgpu_stream stream) {
    return sfgpu_comm_allreduce_sum_f64(static_cast<sfgpu_comm*>(user), d_buf, n, stream);
}
sfgpu_allreduce_fn sfgpu_comm_allreduce_fn(void) { return &comm_allreduce_cb; }

// average duration of one in-place SUM all-reduce of n doubles on this communicator (events on `stream`; `reps` back to back
// after `reps / 4` warm-up calls): what the EM-mode decision of the host rests on
int sfgpu_comm_time_allreduce(sfgpu_comm* c, double* d_buf, uint64_t n, uint32_t reps, sfgpu_stream stream, double* avg_us) {
    SF_REQUIRE(c && d_buf && avg_us && reps, SFGPU_ERR_INVALID, "sfgpu_comm_time_allreduce: null pointer");
    hipStream_t st = as_stream(stream);
    hipEvent_t e0, e1;
    SF_HIP(hipEventCreate(&e0)); SF_HIP(hipEventCreate(&e1));
    int rc = SFGPU_OK;
    for (uint32_t i = 0; i < reps / 4 + 1 && !rc; ++i) rc = sfgpu_comm_allreduce_sum_f64(c, d_buf, n, stream);
    if (!rc) {
        (void)hipEventRecord(e0, st);
        for (uint32_t i = 0; i < reps && !rc; ++i) rc = sfgpu_comm_allreduce_sum_f64(c, d_buf, n, stream);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *avg_us = (double)ms * 1e3 / reps;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
