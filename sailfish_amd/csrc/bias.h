// bias.h -- what em.hip needs from bias.hip (the recompute hook of optimize()).
#pragma once
#include <cstdint>

struct sfgpu_bias;

namespace sfgpu {
uint64_t bias_num_transcripts(const sfgpu_bias* b);
}  // namespace sfgpu
