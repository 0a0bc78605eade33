// sampling.hip -- exact multinomial resampling of the class counts on the device (rows a15/a17).
//
// Replaces  MultinomialSampler::operator() (include/MultinomialSampler.hpp:13-64) as used by
//           doBootstrap (src/CollapsedEMOptimizer.cpp:468): n = sum(count) draws over the classes
//           with p = count/total.  The reference draws them one at a time through an O(k^2) prefix
//           table and std::mt19937; the same distribution is produced here in O(C) work and
//           O(log C) dependent steps: the root holds n, every node splits its count between its two
//           half-ranges with ONE conditional binomial  n_left ~ Binomial(n, mass_left / mass_node)
//           (Philox4x32-10 stream per node, exact BINV/BTPE sampler from rng.h).
#include "common.h"
#include "rng.h"
#include "sampling.h"

namespace sfgpu {

constexpr int kSampBlock = 256;

__global__ void k_mn_root(uint32_t* buf, uint32_t n_total) { buf[0] = n_total; }

// level -> level+1 : node i covers classes [i*width, (i+1)*width)
__global__ void k_mn_level(uint32_t level, uint64_t width, const uint64_t* __restrict__ prefix, uint64_t C,
                           const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t seed, uint64_t draw) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> level) return;
    uint32_t n = in[i];
    uint64_t lo = i * width, mid = lo + width / 2, hi = lo + width;
    uint64_t pl = prefix[lo < C ? lo : C], pm = prefix[mid < C ? mid : C], ph = prefix[hi < C ? hi : C];
    uint64_t s_left = pm - pl, s_all = ph - pl;
    uint32_t n_left;
    if (n == 0 || s_left == 0) n_left = 0;
    else if (s_left == s_all) n_left = n;
    else {
        Philox g; g.init(seed, ((uint64_t)1 << level) + i, draw);
        n_left = binomial(g, n, (double)s_left / (double)s_all);
    }
    out[2 * i] = n_left; out[2 * i + 1] = n - n_left;
}

__global__ void k_mn_emit(uint64_t C, const uint32_t* __restrict__ leaf, const uint32_t* __restrict__ flags,
                          uint32_t* __restrict__ out) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[c] = leaf[c] | (flags ? (flags[c] & 0x80000000u) : 0u);
}

uint64_t multinomial_tree_width(uint64_t C) { uint64_t w = 1; while (w < C) w <<= 1; return w; }

int multinomial_tree(const uint64_t* d_prefix, uint64_t C, uint32_t n_total, uint64_t seed, uint64_t draw,
                     const uint32_t* d_flags, uint32_t* d_out, uint32_t* scratch_a, uint32_t* scratch_b, hipStream_t s) {
    if (C == 0) return SFGPU_OK;
    const uint64_t W = multinomial_tree_width(C);
    int depth = 0; while (((uint64_t)1 << depth) < W) ++depth;
    uint32_t *cur = scratch_a, *nxt = scratch_b;
    hipLaunchKernelGGL(k_mn_root, dim3(1), dim3(1), 0, s, cur, n_total);
    for (int level = 0; level < depth; ++level) {
        uint64_t nodes = (uint64_t)1 << level;
        hipLaunchKernelGGL(k_mn_level, dim3((unsigned)((nodes + kSampBlock - 1) / kSampBlock)), dim3(kSampBlock), 0, s,
                           (uint32_t)level, W >> level, d_prefix, C, cur, nxt, seed, draw);
        uint32_t* t = cur; cur = nxt; nxt = t;
    }
    hipLaunchKernelGGL(k_mn_emit, dim3((unsigned)((C + kSampBlock - 1) / kSampBlock)), dim3(kSampBlock), 0, s, C, cur, d_flags,
                       d_out);
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

}  // namespace sfgpu
