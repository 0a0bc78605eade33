// primitives.hip -- rocPRIM-backed device sort / scan (see primitives.h).
#include "primitives.h"
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace sfgpu {

int sort_pairs_u64_u32(const uint64_t* d_keys_in, uint64_t* d_keys_out, const uint32_t* d_vals_in,
                       uint32_t* d_vals_out, uint64_t n, hipStream_t s, int end_bit, bool sync) {
    if (n == 0) return SFGPU_OK;
    size_t tmp_bytes = 0;
    SF_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out,
                                     (size_t)n, 0, end_bit, s));
    void* tmp = nullptr;
    SF_HIP(pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 8));
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out,
                                             (size_t)n, 0, end_bit, s);
    hipError_t e2 = hipSuccess;
    if (sync) { e2 = hipStreamSynchronize(s); pool_free(tmp); } else pool_free_on(tmp, s);
    SF_HIP(e);
    SF_HIP(e2);
    return SFGPU_OK;
}

struct WidenU32 {
    __device__ __host__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};

int exclusive_scan_u32(const uint32_t* d_in, uint64_t* d_out, uint64_t n, hipStream_t s, bool sync) {
    // scan n+1 positions; the caller guarantees d_in has n+1 readable entries with d_in[n] arbitrary
    auto in = rocprim::make_transform_iterator(d_in, WidenU32());
    size_t tmp_bytes = 0;
    SF_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, d_out, (uint64_t)0, (size_t)(n + 1),
                                   rocprim::plus<uint64_t>(), s));
    void* tmp = nullptr;
    SF_HIP(pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 8));
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, in, d_out, (uint64_t)0, (size_t)(n + 1),
                                           rocprim::plus<uint64_t>(), s);
    hipError_t e2 = hipSuccess;
    if (sync) { e2 = hipStreamSynchronize(s); pool_free(tmp); } else pool_free_on(tmp, s);
    SF_HIP(e);
    SF_HIP(e2);
    return SFGPU_OK;
}

int exclusive_scan_u32_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, hipStream_t s) {
    size_t tmp_bytes = 0;
    SF_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_in, d_out, (uint32_t)0, (size_t)(n + 1), rocprim::plus<uint32_t>(), s));
    void* tmp = nullptr;
    SF_HIP(pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 8));
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, d_in, d_out, (uint32_t)0, (size_t)(n + 1), rocprim::plus<uint32_t>(), s);
    pool_free_on(tmp, s);
    SF_HIP(e);
    return SFGPU_OK;
}

}  // namespace sfgpu
