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

#include <cstdlib>
#include <cstring>

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

// ---- the same tree in TWO launches (round 6).  A level per launch is ~21 dependent launches of ~20 us (0.44 ms per bootstrap replicate
// on cfg3's 1.62 M classes: 8 % of a replicate), most of them a handful of nodes wide.  Same nodes, same Philox streams, same counts:
//   k_mn_top    : ONE block walks the levels 0 .. T - 1 in LDS (<= 512 nodes wide) and leaves the 2^T counts of level T;
//   k_mn_bottom : one block per level-T node walks ITS subtree of kMnSub leaves in LDS (11 levels) and emits the leaves' counts with
//                 the singleton flag (k_mn_emit's job).
constexpr int kMnBlock = 1024;
constexpr int kMnSubBits = 11;
constexpr uint32_t kMnSub = 1u << kMnSubBits;                          // leaves per bottom block
__device__ __forceinline__ uint32_t mn_split(uint32_t level, uint64_t i, uint64_t width, const uint64_t* __restrict__ prefix, uint64_t C,
                                             uint32_t n, uint64_t seed, uint64_t draw) {
    const uint64_t lo = i * width, mid = lo + width / 2, hi = lo + width;
    const uint64_t pl = prefix[lo < C ? lo : C], pm = prefix[mid < C ? mid : C], ph = prefix[hi < C ? hi : C];
    const uint64_t s_left = pm - pl, s_all = ph - pl;
    if (n == 0 || s_left == 0) return 0u;
    if (s_left == s_all) return n;
    Philox g; g.init(seed, ((uint64_t)1 << level) + i, draw);
    return binomial(g, n, (double)s_left / (double)s_all);
}
__global__ void __launch_bounds__(kMnBlock)
k_mn_top(uint32_t T, uint64_t W, const uint64_t* __restrict__ prefix, uint64_t C, uint32_t n_total, uint32_t* __restrict__ out, uint64_t seed, uint64_t draw) {
    __shared__ uint32_t buf[2][kMnBlock];
    if (threadIdx.x == 0) buf[0][0] = n_total;
    __syncthreads();
    for (uint32_t level = 0; level < T; ++level) {
        const uint32_t* in = buf[level & 1u]; uint32_t* o = buf[(level + 1u) & 1u];
        const uint32_t i = threadIdx.x;
        if (i < (1u << level)) {
            const uint32_t n = in[i];
            const uint32_t nl = mn_split(level, i, W >> level, prefix, C, n, seed, draw);
            o[2u * i] = nl; o[2u * i + 1u] = n - nl;
        }
        __syncthreads();
    }
    if (threadIdx.x < (1u << T)) out[threadIdx.x] = buf[T & 1u][threadIdx.x];
}
__global__ void __launch_bounds__(kMnBlock)
k_mn_bottom(uint32_t T, uint32_t depth, uint64_t W, const uint64_t* __restrict__ prefix, uint64_t C, const uint32_t* __restrict__ top,
            const uint32_t* __restrict__ flags, uint32_t* __restrict__ out, uint64_t seed, uint64_t draw) {
    __shared__ uint32_t buf[2][kMnSub];
    const uint32_t b = blockIdx.x, levels = depth - T;                  // this block's node at level T; its subtree has 2^levels leaves
    if (threadIdx.x == 0) buf[0][0] = top[b];
    __syncthreads();
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t* in = buf[l & 1u]; uint32_t* o = buf[(l + 1u) & 1u];
        for (uint32_t j = threadIdx.x; j < (1u << l); j += kMnBlock) {
            const uint32_t n = in[j];
            const uint32_t nl = mn_split(T + l, ((uint64_t)b << l) + j, W >> (T + l), prefix, C, n, seed, draw);
            o[2u * j] = nl; o[2u * j + 1u] = n - nl;
        }
        __syncthreads();
    }
    const uint32_t* leaf = buf[levels & 1u];
    for (uint32_t j = threadIdx.x; j < (1u << levels); j += kMnBlock) {
        const uint64_t c = ((uint64_t)b << levels) + j;
        if (c < C) out[c] = leaf[j] | (flags ? (flags[c] & 0x80000000u) : 0u);
    }
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
    // (SFGPU_MN_TREE=levels keeps a launch per level: the tests hold the two forms to the same counts)
    const bool by_levels = []() { const char* e = getenv("SFGPU_MN_TREE"); return e && strcmp(e, "levels") == 0; }();
    if (!by_levels) {
        const uint32_t T = depth > kMnSubBits ? (uint32_t)(depth - kMnSubBits) : 0u;         // levels of the top block; 2^T bottom blocks
        if (T <= 10u) {                                                                      // (wider tops -- more than 2^21 classes -- keep the levels)
            hipLaunchKernelGGL(k_mn_top, dim3(1), dim3(kMnBlock), 0, s, T, W, d_prefix, C, n_total, scratch_a, seed, draw);
            hipLaunchKernelGGL(k_mn_bottom, dim3(1u << T), dim3(kMnBlock), 0, s, T, (uint32_t)depth, W, d_prefix, C, scratch_a, d_flags, d_out, seed, draw);
            SF_CHECK_LAUNCH();
            return SFGPU_OK;
        }
    }
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
