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
    ~DevBuf() { if (p) (void)hipFree(p); }
    int reserve(uint64_t n, hipStream_t s, bool keep, uint64_t used = 0) {
        if (n <= cap) return SFGPU_OK;
        uint64_t nc = cap ? cap : 1;
        while (nc < n) nc *= 2;
        T* q = nullptr;
        SF_HIP(hipMalloc(&q, nc * sizeof(T)));
        if (keep && p && used) SF_HIP(hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, s));
        if (p) { SF_HIP(hipStreamSynchronize(s)); SF_HIP(hipFree(p)); }
        p = q; cap = nc;
        return SFGPU_OK;
    }
};

constexpr int kWave = 64;  // gfx950 wavefront

}  // namespace sfgpu
