// sampling.h -- device multinomial resampling used by the bootstrap (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sfgpu {

// Multinomial(n_total, p_c = base_c / sum(base)) over C classes, exact, as a binary tree of
// conditional binomials evaluated level by level (MultinomialSampler.hpp:13-64 draws the same
// distribution with n_total sequential inverse-CDF draws).
//   d_prefix : exclusive prefix sums of the base counts, C+1 entries
//   d_flags  : per class word whose bit 31 is copied into the output (the singleton flag), may be null
//   d_out    : C sampled counts (31 bits) | flag bit
//   scratch  : two buffers of tree_width(C) uint32 each
uint64_t multinomial_tree_width(uint64_t C);
int multinomial_tree(const uint64_t* d_prefix, uint64_t C, uint32_t n_total, uint64_t seed, uint64_t draw,
                     const uint32_t* d_flags, uint32_t* d_out, uint32_t* scratch_a, uint32_t* scratch_b, hipStream_t s);

}  // namespace sfgpu
