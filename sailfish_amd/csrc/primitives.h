// primitives.h -- device-wide sort / scan used outside the hot loops (finish(), plan building).
// Implemented in primitives.hip on top of rocPRIM so the heavy templates compile once.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sfgpu {

// Stable LSD radix sort of (key u64, value u32) pairs, ascending by key.  Scratch is allocated and freed inside.
// sync = true: the call returns after the sort has run (the stream is synchronised) -- callers may free or reuse anything.
// sync = false: nothing is waited for (the scratch is freed stream-ordered); the caller keeps the arrays alive until it has
// synchronised the stream itself.
int sort_pairs_u64_u32(const uint64_t* d_keys_in, uint64_t* d_keys_out, const uint32_t* d_vals_in,
                       uint32_t* d_vals_out, uint64_t n, hipStream_t s, int end_bit = 64, bool sync = true);

// out[i] = sum_{j<i} in[j] for i in [0, n]; out has n+1 entries (out[n] = total).  sync as above.
int exclusive_scan_u32(const uint32_t* d_in, uint64_t* d_out, uint64_t n, hipStream_t s, bool sync = true);

// the same with 32-bit sums (the caller knows the total fits), nothing waited for
int exclusive_scan_u32_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, hipStream_t s);

}  // namespace sfgpu
