// common.h -- shared host-side plumbing of libsfgpu (error slot, HIP checks, logging).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/sfgpu.h"

namespace sfgpu {

void set_error(const char* fmt, ...);
void log_msg(int level, const char* fmt, ...);

inline hipStream_t as_stream(sfgpu_stream s) { return reinterpret_cast<hipStream_t>(s); }

// Caching device allocator: hipMalloc / hipFree cost tens of microseconds each and hipFree
// synchronises the whole device; a quantification step creates and drops ~60 scratch buffers.
// Blocks are rounded up to a power of two (>= 256 B) and kept in per-size free lists for reuse;
// sfgpu_pool_trim() returns them to the driver.  Callers free only after the work that used the
// block has been synchronised (handles synchronise their stream before releasing scratch).
hipError_t pool_malloc(void** p, size_t bytes);
void pool_free(void* p);
void pool_free_on(void* p, hipStream_t s);     // back to the cache once the work enqueued on s so far is done (no host wait)
void pool_free_on_many(void* const* ps, int n, hipStream_t s);      // ... several blocks behind ONE event (null entries are skipped)
void pool_trim();
void pool_set_large_limit(long long bytes);     // cached device blocks >= 1 GiB are kept up to this many bytes per device
template <typename T>
inline hipError_t pool_malloc(T** p, size_t bytes) { return pool_malloc(reinterpret_cast<void**>(p), bytes); }
// same recycling for pinned host blocks and for the handles' private non-blocking streams
hipError_t pinned_malloc(void** p, size_t bytes);
void pinned_free(void* p);
template <typename T>
inline hipError_t pinned_malloc(T** p, size_t bytes) { return pinned_malloc(reinterpret_cast<void**>(p), bytes); }
// device memory with the UNCACHED memory type (hipExtMallocWithFlags(.., hipDeviceMallocUncached)): no XCD's L2 ever holds a line of
// it, so what one workgroup stores is what every other workgroup's next load returns -- the exchange buffer of the persistent EM loop
// (em_persist.h).  Recycled like the other kinds (hipFree synchronises the device).
hipError_t uncached_malloc(void** p, size_t bytes);
void uncached_free(void* p);
template <typename T>
inline hipError_t uncached_malloc(T** p, size_t bytes) { return uncached_malloc(reinterpret_cast<void**>(p), bytes); }
hipError_t stream_acquire(hipStream_t* s);
void stream_release(hipStream_t s);

// HIP call -> SFGPU_ERR_HIP with the failing expression recorded.
#define SF_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::sfgpu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SFGPU_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

#define SF_CHECK_LAUNCH() SF_HIP(hipGetLastError())

#define SF_REQUIRE(cond, code, msg)                 \
    do {                                            \
        if (!(cond)) {                              \
            ::sfgpu::set_error("%s", msg);          \
            return code;                            \
        }                                           \
    } while (0)

// grow-only device buffer (hipMalloc/hipFree are synchronising and slow: callers over-reserve)
template <typename T>
struct DevBuf {
    T* p = nullptr;
    uint64_t cap = 0;  // elements
    ~DevBuf() { if (p) pool_free(p); }
    int reserve(uint64_t n, hipStream_t s, bool keep, uint64_t used = 0) {
        if (n <= cap) return SFGPU_OK;
        uint64_t nc = cap ? cap : 1;
        while (nc < n) nc *= 2;
        T* q = nullptr;
        SF_HIP(pool_malloc(&q, nc * sizeof(T)));
        if (keep && p && used) SF_HIP(hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, s));
        if (p) { SF_HIP(hipStreamSynchronize(s)); pool_free(p); }
        p = q; cap = nc;
        return SFGPU_OK;
    }
};

// Tuning / experiment switches that lost their A/Bs (profiles/r*_notes.md name each) are read only by builds made with
// -DSFGPU_VARIANTS (tools/*_variants.sh -> csrc/variants/libsfgpu_<name>.so): the product library neither reads them nor contains the
// kernels they select.  The switches the product keeps are listed in include/sfgpu.h.
#ifdef SFGPU_VARIANTS
#define SF_DEV_ENV(name) getenv(name)
#else
#define SF_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
// SFGPU_TIMING=1: the library says on stderr what its plans look like (dev)
inline bool env_timing() { static const bool on = getenv("SFGPU_TIMING") != nullptr; return on; }

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace sfgpu
