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
void draw_binomial_by_inversion(uint64_t seed, uint32_t n, double p, uint32_t count, uint32_t* out) {
    for (uint32_t i = 0; i < count; ++i) {
        sfgpu::Philox g; g.init(seed, i, 0);
        out[i] = sfgpu::binomial_by_inversion(g, n, p);
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

// the multinomial tree of csrc/sampling.hip (k_mn_level), node for node and stream for stream, on the host:
// prefix[0..C] = exclusive prefix sums of the class counts, out[0..C) = one resample of n_total reads
extern "C" void tree_multinomial(uint64_t seed, uint64_t draw, uint64_t C, const uint64_t* prefix, uint32_t n_total, uint32_t* out) {
    uint64_t W = 1; while (W < C) W <<= 1;
    int depth = 0; while (((uint64_t)1 << depth) < W) ++depth;
    uint32_t* cur = new uint32_t[W + 1]; uint32_t* nxt = new uint32_t[W + 1];
    cur[0] = n_total;
    for (int level = 0; level < depth; ++level) {
        const uint64_t width = W >> level;
        for (uint64_t i = 0; i < ((uint64_t)1 << level); ++i) {
            const uint32_t n = cur[i];
            const uint64_t lo = i * width, mid = lo + width / 2, hi = lo + width;
            const uint64_t pl = prefix[lo < C ? lo : C], pm = prefix[mid < C ? mid : C], ph = prefix[hi < C ? hi : C];
            const uint64_t s_left = pm - pl, s_all = ph - pl;
            uint32_t n_left;
            if (n == 0 || s_left == 0) n_left = 0;
            else if (s_left == s_all) n_left = n;
            else { sfgpu::Philox g; g.init(seed, ((uint64_t)1 << level) + i, draw); n_left = sfgpu::binomial(g, n, (double)s_left / (double)s_all); }
            nxt[2 * i] = n_left; nxt[2 * i + 1] = n - n_left;
        }
        uint32_t* t = cur; cur = nxt; nxt = t;
    }
    for (uint64_t c = 0; c < C; ++c) out[c] = cur[c];
    delete[] cur; delete[] nxt;
}
extern "C" void draw_uniform_sub(uint64_t seed, uint64_t stream, uint64_t substream, uint32_t count, double* out) {
    sfgpu::Philox g; g.init(seed, stream, substream);
    for (uint32_t i = 0; i < count; ++i) out[i] = g.uniform();
}
