// test harness: compiles the product's sampler header (sailfish_amd/csrc/rng.h) as plain C++ so the
// CPU suite can check the samplers' distributions without a GPU
#include "../sailfish_amd/csrc/rng.h"
extern "C" {
void draw_binomial(uint64_t seed, uint32_t n, double p, uint32_t count, uint32_t* out) {
    for (uint32_t i = 0; i < count; ++i) {
        sfgpu::Philox g; g.init(seed, i, 0);
        out[i] = sfgpu::binomial(g, n, p);
    }
}
void draw_uniform(uint64_t seed, uint64_t stream, uint32_t count, double* out) {
    sfgpu::Philox g; g.init(seed, stream, 0);
    for (uint32_t i = 0; i < count; ++i) out[i] = g.uniform();
}
void philox_block(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
    sfgpu::Philox g; g.key[0] = k0; g.key[1] = k1; g.ctr[0] = c0; g.ctr[1] = c1; g.ctr[2] = c2; g.ctr[3] = c3; g.have = 0;
    g.block();
    for (int i = 0; i < 4; ++i) out[i] = g.out[i];
}
}
