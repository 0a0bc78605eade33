// gibbs.hip -- collapsed Gibbs sampling over the equivalence classes (row a16).
//
// Replaces  src/CollapsedGibbsSampler.cpp:35-94 (initCountMap_), :96-186 (sampleRound_),
//           :198-291 (sample<ReadExperiment>).
//
// The reference runs one chain per TBB chunk of the sample range (:223-246): initCountMap_ assigns
// every class's reads to its transcripts with one multinomial draw, then each sample is the previous
// one plus ONE sampleRound_ (`bool numInternalRounds = 10` is 1, :248).  A round visits every class
// once, in eqVec order -- which is hash-table order, i.e. arbitrary -- and each visit reads the
// transcript counts earlier visits left behind.
//
// Device design.
//   * Chains: one LANE per chain, 64 chains per wavefront in lock step.  The class structure is the
//     same for every chain (wave-uniform loads) and the per-chain state is chain-minor --
//     countMap[nonzero][chain], txpCount[transcript][chain] -- so every state access of a wavefront
//     is one coalesced 256-byte transaction and there is no cross-lane traffic.
//   * Visiting order: two class visits commute exactly when the classes share no transcript, so a
//     round may visit conflict-free classes concurrently and is then identical to SOME sequential
//     scan.  Classes are cut into tiles of 64 consecutive classes (canonical order = sorted by first
//     id, so a tile touches a narrow band of transcripts [lo, hi]); with K = the largest number of
//     following tiles a tile's band reaches into, the tiles {p, p+K, p+2K, ...} of "phase" p have
//     pairwise disjoint bands.  A round is K phase launches; inside a phase every (tile, 64-chain
//     group) is an independent block.  The scan order is phase-major instead of index-major -- a
//     different but equally systematic scan (the reference's own order is an accident of its hash
//     table).  Classes whose members span more than kWideSpan transcripts would inflate K; they are
//     taken out of the tiles and visited one after another in a final "wide" phase.
//     This turns ~0.5 M sequential class visits per round per chain (16 wavefronts on the chip for 1024
//     chains: 5.9 s per round, measured) into K ~ tens of launches of thousands of blocks.
//   * The multinomial over a class's k members is a chain of k-1 conditional binomials (exact
//     BINV/BTPE sampler, rng.h) instead of n inverse-CDF draws (MultinomialSampler.hpp:13-64).
//   * The aux weight of member t is (count/effLen_t)/sum, proportional to 1/effLen_t inside a class;
//     its normaliser cancels in the multinomial probabilities, so 1/effLen_t is used directly.
// Random numbers: Philox4x32-10 keyed by (seed; chain, round, class): results do not depend on how
// blocks are scheduled.  The reference seeds std::mt19937 from std::random_device (:104-105,
// :227-228), so parity is distributional.
#include "colour.h"
#include "common.h"
#include "rng.h"

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

namespace sfgpu {

constexpr int kGibbsBlock = 64;            // one wavefront of chains per block
constexpr int kGibbsTile = 64;             // classes per tile
constexpr uint32_t kWideSpan = 2048;       // classes spanning more transcripts than this are "wide"
constexpr uint32_t kMaxPhases = 1024;      // more phases than this: fall back to one sequential scan
constexpr uint32_t kWideSerial = 64;       // up to this many wide classes are visited one after another by one launch
constexpr uint32_t kMaxColours = 1u << 16; // more colours than this: visit the wide classes one after another after all
constexpr uint32_t kThinWidth = 12;        // a component of wide classes with fewer classes per colour than this is visited serially
#if !defined(SFGPU_GIBBS_LIGHT_MAX)
#define SFGPU_GIBBS_LIGHT_MAX 1500
#endif
constexpr uint64_t kGibbsLightMax = SFGPU_GIBBS_LIGHT_MAX;   // classes of more reads than this are HEAVY: visited by the phase kernel that carries BTPE (see k_gibbs_phase)
constexpr double kGibbsPrior = 1e-8;       // priorAlpha (:215)
constexpr double kGibbsTiny = 4.9406564584124654e-324;

struct GibbsArgs {
    uint32_t n_chains; uint64_t C;
    const uint32_t* rowptr; const uint32_t* ids; const uint64_t* counts;
    const double* inv_len; const double* w_mass;        // 1/effLen_t ; (prior + mass_t)/effLen_t
    uint32_t* count_map; int32_t* txp_count;
    const uint8_t* wide;                                 // per class: visited in the wide phase
    const uint32_t* wide_list; uint32_t n_wide;
    uint64_t seed; uint32_t round;
};

// multinomial(n; p_0..p_{k-1}) as conditional binomials; calls put(i, r_i) for every member.  Member `last` gets what the others
// leave: a BINV walk costs its mean, so the chain costs n (1 - p_last) steps in all -- least when the LARGEST member is not sampled
// at all.  `last` is the same for every chain of the wavefront (the member with the largest EM weight: the chains live around the EM
// solution), so no lane idles; any order gives the same distribution.
template <bool LIGHT, typename ProbFn, typename PutFn>
__device__ __forceinline__ void multinomial_chain(Philox& g, uint32_t n, uint32_t k, double p_total, uint32_t last, ProbFn prob, PutFn put) {
    double p_rem = p_total;
    uint32_t n_rem = n;
    for (uint32_t i = 0; i < k; ++i) {
        if (i == last) continue;
        const double p = prob(i);
        const double ratio = (p_rem > 0.0) ? ratio_of(p, p_rem) : 0.0;
        const double pr = ratio < 1.0 ? ratio : 1.0;
        const uint32_t r = (n_rem == 0) ? 0u : (LIGHT ? binomial_by_inversion(g, n_rem, pr) : binomial(g, n_rem, pr));
        p_rem -= p; if (p_rem < 0.0) p_rem = 0.0;
        put(i, r);
        n_rem -= r;
    }
    put(last, n_rem);
}

// initCountMap_ (:45-92) for one class and one chain
template <bool LIGHT = false>
__device__ __forceinline__ void gibbs_init_class(const GibbsArgs& a, uint64_t c, uint32_t ch) {
    const uint32_t b = a.rowptr[c], k = a.rowptr[c + 1] - b;
    const uint32_t n = (uint32_t)a.counts[c];            // uint32 n in MultinomialSampler (:15)
    const uint32_t nch = a.n_chains;
    if (k == 0) return;
    if (k == 1) {                                        // :81-83
        a.count_map[(uint64_t)b * nch + ch] = n;
        a.txp_count[(uint64_t)a.ids[b] * nch + ch] += (int32_t)n;
        return;
    }
    double denom = 0.0, w_top = -1.0; uint32_t last = 0;
    for (uint32_t i = 0; i < k; ++i) { const double w = a.w_mass[a.ids[b + i]]; denom += w; if (w > w_top) { w_top = w; last = i; } }      // :58-63
    if (!(denom > kGibbsTiny)) {                                                      // :65 -- nothing assigned
        for (uint32_t i = 0; i < k; ++i) a.count_map[(uint64_t)(b + i) * nch + ch] = 0;
        return;
    }
    Philox g; g.init(a.seed, ch, c);
    multinomial_chain<LIGHT>(g, n, k, denom, last,
        [&](uint32_t i) { return a.w_mass[a.ids[b + i]]; },
        [&](uint32_t i, uint32_t r) {
            a.count_map[(uint64_t)(b + i) * nch + ch] = r;                            // :76-79
            a.txp_count[(uint64_t)a.ids[b + i] * nch + ch] += (int32_t)r;             // :86-89
        });
}

// sampleRound_ (:113-184) for one class and one chain
template <bool LIGHT = false>
__device__ __forceinline__ void gibbs_round_class(const GibbsArgs& a, uint64_t c, uint32_t ch) {
    const uint32_t b = a.rowptr[c], k = a.rowptr[c + 1] - b;
    if (k <= 1) return;                                     // singletons keep their full count (:128)
    const uint32_t nch = a.n_chains;
    Philox g; g.init(a.seed, ((uint64_t)(a.round + 1) << 32) | ch, c);
    const double frac = 0.25 + 0.5 * g.uniform();           // U(0.25, 0.75) per class (:106, :115)
    // pass 1: take round(frac * current) reads away from every member (:138-148)
    uint32_t n_res = 0, last = 0; double denom = 0.0, w_top = -1.0;
    for (uint32_t i = 0; i < k; ++i) {
        const uint64_t at = (uint64_t)(b + i) * nch + ch;
        const uint32_t t = a.ids[b + i];
        { const double w = a.w_mass[t]; if (w > w_top) { w_top = w; last = i; } }        // (wavefront-uniform: the member the chain does not sample)
        const uint32_t cur = a.count_map[at];
        const uint32_t r = (uint32_t)(frac * (double)cur + 0.5);                       // std::round, values >= 0 (:142)
        n_res += r;
        a.count_map[at] = cur - r;
        const int32_t tc = a.txp_count[(uint64_t)t * nch + ch] - (int32_t)r;
        a.txp_count[(uint64_t)t * nch + ch] = tc;
        denom += (kGibbsPrior + (double)tc) * a.inv_len[t];                            // :147
    }
    // pass 2: re-draw them from p_i ~ (prior + txpCount_i) * aux_i (:150-170).  denom >= k*1e-8/len > 0,
    // so the reference's "did not sample" branch (:172-179) cannot trigger for finite inputs.
    multinomial_chain<LIGHT>(g, n_res, k, denom, last,
        [&](uint32_t i) { const uint32_t t = a.ids[b + i]; return (kGibbsPrior + (double)a.txp_count[(uint64_t)t * nch + ch]) * a.inv_len[t]; },
        [&](uint32_t i, uint32_t r) {
            if (r) { a.count_map[(uint64_t)(b + i) * nch + ch] += r; a.txp_count[(uint64_t)a.ids[b + i] * nch + ch] += (int32_t)r; }
        });
}

// one phase: blockIdx.x -> tile (phase + K * x), blockIdx.y -> group of 64 chains
// Two forms (round 5).  The sampler is long chains of dependent f64 operations: what a SIMD needs is wavefronts to switch between, and
// what limits them is BTPE's registers -- the kernel with it needs 128 VGPRs for 4 wavefronts per SIMD (the compiler's own choice is
// 142 and 3; at 5 it spills), without it 76: 6 wavefronts.  A class of <= kGibbsLightMax reads does without BTPE: a binomial of a
// larger mean than one BINV walk takes is drawn as a sum of equal parts (binomial_by_inversion: the sum of binomials over a split
// of n IS the binomial) -- rarely more than two at that size.  So the LIGHT form visits those classes at 6 wavefronts per SIMD, the
// other form the HEAVY classes of the same tiles in a launch of its own behind it (when the phase has any): still one systematic
// scan -- the light classes of a phase's tiles, then their heavy ones.  cfg3 (classes of 250 reads on average, 21 tiles with a heavy
// one): init 136 -> 109 ms, a round 173 -> 141; with the bound at 440 (every tile has heavy classes: two launches per phase) 117 / 150.
template <bool INIT, bool LIGHT>
__global__ void __launch_bounds__(kGibbsBlock) __attribute__((amdgpu_waves_per_eu(LIGHT ? 6 : 4, LIGHT ? 6 : 4)))
k_gibbs_phase(GibbsArgs a, uint32_t phase, uint32_t K, uint32_t n_tiles) {
    const uint32_t tile = phase + K * blockIdx.x;
    const uint32_t ch = blockIdx.y * kGibbsBlock + threadIdx.x;
    if (tile >= n_tiles || ch >= a.n_chains) return;
    const uint64_t c0 = (uint64_t)tile * kGibbsTile;
    const uint64_t c1 = (c0 + kGibbsTile < a.C) ? c0 + kGibbsTile : a.C;
    for (uint64_t c = c0; c < c1; ++c) {
        if (a.wide[c] != (LIGHT ? 0 : 2)) continue;                  // 0 light, 1 wide (not here), 2 heavy
        if (INIT) gibbs_init_class<LIGHT>(a, c, ch); else gibbs_round_class<LIGHT>(a, c, ch);
    }
}

// the wide classes, one after another (they may share transcripts with each other)
template <bool INIT>
__global__ void __launch_bounds__(kGibbsBlock)
k_gibbs_wide(GibbsArgs a) {
    const uint32_t ch = blockIdx.x * kGibbsBlock + threadIdx.x;
    if (ch >= a.n_chains) return;
    for (uint32_t i = 0; i < a.n_wide; ++i) {
        const uint64_t c = a.wide_list[i];
        if (INIT) gibbs_init_class(a, c, ch); else gibbs_round_class(a, c, ch);
    }
}

// one COLOUR of the wide classes (no two classes of a colour share a transcript): blockIdx.x -> kListChunk classes of the
// list, blockIdx.y -> group of 64 chains
// `chunk` classes per block: 8 when the colour is large (fewer, longer blocks), down to 1 when it is small -- a colour of 49
// classes at 8 per block was 7 blocks of 8 visits one after another (~100 us per launch); at 1 per block it is ~20.
constexpr uint32_t kListChunk = 8;
template <bool INIT>
__global__ void __launch_bounds__(kGibbsBlock)
k_gibbs_list(GibbsArgs a, const uint32_t* __restrict__ list, uint32_t n, uint32_t chunk) {
    const uint32_t ch = blockIdx.y * kGibbsBlock + threadIdx.x;
    if (ch >= a.n_chains) return;
    const uint32_t i0 = blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint64_t c = list[i];
        if (INIT) gibbs_init_class(a, c, ch); else gibbs_round_class(a, c, ch);
    }
}

// the THIN components of the wide classes (see the plan): blockIdx.x -> component, whose classes are visited one after another,
// in class order; blockIdx.y -> group of 64 chains
template <bool INIT>
__global__ void __launch_bounds__(kGibbsBlock)
k_gibbs_components(GibbsArgs a, const uint32_t* __restrict__ list, const uint32_t* __restrict__ comp_off) {
    const uint32_t ch = blockIdx.y * kGibbsBlock + threadIdx.x;
    if (ch >= a.n_chains) return;
    for (uint32_t i = comp_off[blockIdx.x]; i < comp_off[blockIdx.x + 1]; ++i) {
        const uint64_t c = list[i];
        if (INIT) gibbs_init_class(a, c, ch); else gibbs_round_class(a, c, ch);
    }
}

// plan: per class wide flag; per tile the band [lo, hi] of its non-wide classes
__global__ void k_gibbs_plan(uint64_t C, uint32_t n_tiles, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                             const uint64_t* __restrict__ counts, uint8_t* wide, uint32_t* tile_lo, uint32_t* tile_hi, uint8_t* tile_kind,
                             uint32_t* wide_list, unsigned int* n_wide) {
    uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles) return;
    const uint64_t c0 = (uint64_t)tile * kGibbsTile;
    const uint64_t c1 = (c0 + kGibbsTile < C) ? c0 + kGibbsTile : C;
    uint32_t lo = 0xFFFFFFFFu, hi = 0, kind = 0;                 // kind: bit 0 the tile has light classes, bit 1 heavy ones
    for (uint64_t c = c0; c < c1; ++c) {
        uint32_t b = rowptr[c], e = rowptr[c + 1];
        uint32_t mn = 0xFFFFFFFFu, mx = 0;
        for (uint32_t j = b; j < e; ++j) { uint32_t t = ids[j]; mn = t < mn ? t : mn; mx = t > mx ? t : mx; }
        bool w = (e > b) && (mx - mn > kWideSpan);
        const bool heavy = !w && e - b > 1u && counts[c] > kGibbsLightMax;       // (a singleton is never sampled)
        wide[c] = w ? 1 : (heavy ? 2 : 0);
        if (w) wide_list[atomicAdd(n_wide, 1u)] = (uint32_t)c;
        else if (e > b) { lo = mn < lo ? mn : lo; hi = mx > hi ? mx : hi; kind |= heavy ? 2u : 1u; }
    }
    tile_lo[tile] = lo; tile_hi[tile] = hi;                  // lo > hi: the tile has no banded class
    tile_kind[tile] = (uint8_t)kind;
}

// the labels of the listed classes as a compact CSR (the host colours the wide classes: it needs their labels, not all 9 M nonzeros)
__global__ void k_gibbs_list_lens(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ rowptr, uint32_t* lens) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens[i] = rowptr[list[i] + 1] - rowptr[list[i]];
}
__global__ void k_gibbs_list_rows(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                                  const uint32_t* __restrict__ out_off, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = rowptr[list[i]], k = rowptr[list[i] + 1] - b;
    for (uint32_t j = 0; j < k; ++j) out[out_off[i] + j] = ids[b + j];
}

__global__ void k_gibbs_weights(uint64_t M, const double* __restrict__ len, const double* __restrict__ mass, double num_mapped,
                                double* __restrict__ inv_len, double* __restrict__ w_mass) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    double l = len[t]; if (l <= 1.0) l = 1.0;                    // effLens clamp of optimize() (:738 in the optimizer)
    double il = 1.0 / l;
    inv_len[t] = il;
    double m = kGibbsPrior + mass[t] * num_mapped;               // txp.setMass(priorAlpha + mass * numMapped) (:219-221)
    w_mass[t] = (kGibbsPrior + m) * il;                          // (priorAlpha + mass(false)) * aux (:61)
}

// txp_count[t][chain] -> out[sample = base + chain][t]
__global__ void k_gibbs_emit(uint32_t n_chains, uint32_t n_emit, uint64_t M, const int32_t* __restrict__ txp_count,
                             int32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_emit * M) return;
    uint64_t ch = i / M, t = i % M;
    out[i] = txp_count[t * n_chains + ch];
}

}  // namespace sfgpu

using namespace sfgpu;

// number of phases: the smallest K such that every tile's band ends before the band of the K-th
// following tile begins (suffix minima make the test monotone even if the bands are not)
static uint32_t gibbs_phase_count(const std::vector<uint32_t>& lo, const std::vector<uint32_t>& hi) {
    const size_t n = lo.size();
    std::vector<uint32_t> slo(n + 1, 0xFFFFFFFFu);
    for (size_t i = n; i-- > 0;) slo[i] = std::min(slo[i + 1], lo[i]);
    uint32_t K = 1;
    for (size_t i = 0; i < n; ++i) {
        if (lo[i] > hi[i]) continue;                       // empty band
        // first d >= 1 with slo[i + d] > hi[i]  (slo is non-decreasing in its index)
        size_t a = i + 1, b = n;                           // search in [i+1, n]; slo[n] = +inf
        while (a < b) { size_t m = (a + b) / 2; if (slo[m] > hi[i]) b = m; else a = m + 1; }
        uint32_t d = (uint32_t)(a - i);
        if (d > K) K = d;
    }
    return K;
}

extern "C" {

int sfgpu_gibbs_sample(const sfgpu_problem* prob, const double* d_mass, uint32_t n_samples, uint32_t n_chains,
                       uint64_t seed, int32_t* d_out, sfgpu_gibbs_cb cb, void* user, sfgpu_stream stream) {
    SF_REQUIRE(prob && d_mass, SFGPU_ERR_INVALID, "sfgpu_gibbs_sample: null pointer");
    SF_REQUIRE(prob->M > 0 && prob->d_len, SFGPU_ERR_INVALID, "sfgpu_gibbs_sample: need M > 0 and d_len");
    SF_REQUIRE(n_samples > 0, SFGPU_ERR_INVALID, "sfgpu_gibbs_sample: n_samples == 0");
    const uint64_t M = prob->M, C = prob->C;
    hipStream_t st = as_stream(stream);
    if (n_chains == 0) {                   // default: one wavefront-multiple of chains, at most 1024
        n_chains = n_samples < 1024 ? n_samples : 1024;
        n_chains = (n_chains + kGibbsBlock - 1) / kGibbsBlock * kGibbsBlock;
    }
    uint32_t L = 0;
    if (C) { SF_HIP(hipMemcpyAsync(&L, prob->d_rowptr + C, 4, hipMemcpyDeviceToHost, st)); SF_HIP(hipStreamSynchronize(st)); }
    const uint32_t n_tiles = (uint32_t)((C + kGibbsTile - 1) / kGibbsTile);
    uint32_t* count_map = nullptr; int32_t* txp_count = nullptr; double *inv_len = nullptr, *w_mass = nullptr;
    uint8_t *wide = nullptr, *tile_kind = nullptr; uint32_t *tile_lo = nullptr, *tile_hi = nullptr, *wide_list = nullptr; unsigned int* d_nwide = nullptr;
    int32_t* d_tmp = nullptr; int32_t* h_tmp = nullptr; uint32_t* d_thin_off = nullptr;
    uint32_t *d_lens = nullptr, *d_off = nullptr, *d_rows = nullptr;          // plan scratch: the wide classes' labels
    int rc = SFGPU_OK;
    const bool timing = env_timing();                 // where a call's time goes (stderr)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    auto t_mark = now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "gibbs timing: %-12s %9.2f ms\n", what, ms_since(t_mark));
        t_mark = now();
    };
#define G_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); rc = SFGPU_ERR_HIP; goto done; } } while (0)
    G_TRY(pool_malloc(&count_map, ((uint64_t)L * n_chains + 1) * 4));
    G_TRY(pool_malloc(&txp_count, (uint64_t)M * n_chains * 4));
    G_TRY(pool_malloc(&inv_len, M * 8)); G_TRY(pool_malloc(&w_mass, M * 8));
    G_TRY(pool_malloc(&wide, C ? C : 1)); G_TRY(pool_malloc(&wide_list, (C ? C : 1) * 4)); G_TRY(pool_malloc(&d_nwide, 4));
    G_TRY(pool_malloc(&tile_kind, (size_t)(n_tiles ? n_tiles : 1)));
    G_TRY(pool_malloc(&tile_lo, (size_t)(n_tiles ? n_tiles : 1) * 4)); G_TRY(pool_malloc(&tile_hi, (size_t)(n_tiles ? n_tiles : 1) * 4));
    if (!d_out) G_TRY(pool_malloc(&d_tmp, (uint64_t)n_chains * M * 4));
    if (cb) G_TRY(pinned_malloc(&h_tmp, (uint64_t)n_chains * M * 4));
    G_TRY(hipMemsetAsync(txp_count, 0, (uint64_t)M * n_chains * 4, st));
    G_TRY(hipMemsetAsync(d_nwide, 0, 4, st));
    lap("allocate");
    hipLaunchKernelGGL(k_gibbs_weights, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, M, prob->d_len, d_mass,
                       (double)prob->num_mapped, inv_len, w_mass);
    {
        // ---- plan: bands, wide classes, number of phases
        uint32_t K = 1; unsigned int n_wide = 0;
        std::vector<uint32_t> colour_off;                  // wide classes by colour (empty: visited one after another)
        std::vector<uint32_t> thin_off, thin_list;         // ... and those of thin components, by component
        unsigned int n_rest = 0;                           // wide classes outside the thin components: wide_list[0, n_rest)
        std::vector<uint8_t> phase_kind;                   // per phase: bit 0 some tile has light classes, bit 1 heavy ones
        uint64_t n_heavy_tiles = 0;
        if (n_tiles) {
            hipLaunchKernelGGL(k_gibbs_plan, dim3((n_tiles + 255) / 256), dim3(256), 0, st, C, n_tiles, prob->d_rowptr, prob->d_ids,
                               prob->d_counts, wide, tile_lo, tile_hi, tile_kind, wide_list, d_nwide);
            G_TRY(hipGetLastError());
            std::vector<uint32_t> lo(n_tiles), hi(n_tiles);
            std::vector<uint8_t> kind(n_tiles);
            G_TRY(hipMemcpyAsync(kind.data(), tile_kind, (size_t)n_tiles, hipMemcpyDeviceToHost, st));
            G_TRY(hipMemcpyAsync(lo.data(), tile_lo, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, st));
            G_TRY(hipMemcpyAsync(hi.data(), tile_hi, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, st));
            G_TRY(hipMemcpyAsync(&n_wide, d_nwide, 4, hipMemcpyDeviceToHost, st));
            G_TRY(hipStreamSynchronize(st));
            K = gibbs_phase_count(lo, hi);
            if (K > kMaxPhases) K = n_tiles;             // no locality to exploit: one tile per launch == a sequential scan
            phase_kind.assign(K, 0);
            for (uint32_t t = 0; t < n_tiles; ++t) { phase_kind[t % K] |= kind[t]; n_heavy_tiles += (kind[t] >> 1) & 1u; }
            n_rest = n_wide;
            if (n_wide > 1) {
                // the plan appends wide classes with an atomic cursor: put them into class order, so that the wide
                // phase visits them in the same order on every run (they may share transcripts -> order matters
                // for the draws of a fixed seed)
                std::vector<uint32_t> wl(n_wide);
                G_TRY(hipMemcpyAsync(wl.data(), wide_list, (size_t)n_wide * 4, hipMemcpyDeviceToHost, st));
                G_TRY(hipStreamSynchronize(st));
                std::sort(wl.begin(), wl.end());
                if (n_wide > kWideSerial) {
                    // many wide classes (every class of a gene also names a far pseudogene / paralog): one after another they
                    // took 108 ms per round for 722 k of them (measured).  Colour them and visit a colour per launch.
                    // the wide classes' labels as a compact CSR on the host (class wl[i] = row i): gathered on the device, two small
                    // copies instead of the whole class table (43 MB through pageable memory cost 10 ms per call on cfg3)
                    std::vector<uint32_t> h_rowptr(n_wide + 1, 0), h_ids, colour_of, rows(n_wide);
                    {
                        G_TRY(hipMemcpyAsync(wide_list, wl.data(), (size_t)n_wide * 4, hipMemcpyHostToDevice, st));     // (sorted)
                        G_TRY(pool_malloc(&d_lens, (size_t)n_wide * 4)); G_TRY(pool_malloc(&d_off, (size_t)n_wide * 4));
                        hipLaunchKernelGGL(k_gibbs_list_lens, dim3((n_wide + 255) / 256), dim3(256), 0, st, n_wide, wide_list, prob->d_rowptr, d_lens);
                        G_TRY(hipMemcpyAsync(h_rowptr.data() + 1, d_lens, (size_t)n_wide * 4, hipMemcpyDeviceToHost, st));
                        G_TRY(hipStreamSynchronize(st));
                        for (uint32_t i = 0; i < n_wide; ++i) h_rowptr[i + 1] += h_rowptr[i];
                        const uint32_t total = h_rowptr[n_wide];
                        h_ids.resize(total ? total : 1);
                        G_TRY(pool_malloc(&d_rows, (size_t)(total ? total : 1) * 4));
                        G_TRY(hipMemcpyAsync(d_off, h_rowptr.data(), (size_t)n_wide * 4, hipMemcpyHostToDevice, st));
                        hipLaunchKernelGGL(k_gibbs_list_rows, dim3((n_wide + 255) / 256), dim3(256), 0, st, n_wide, wide_list, prob->d_rowptr, prob->d_ids, d_off, d_rows);
                        G_TRY(hipMemcpyAsync(h_ids.data(), d_rows, (size_t)total * 4, hipMemcpyDeviceToHost, st));
                        G_TRY(hipStreamSynchronize(st));
                    }
                    for (uint32_t i = 0; i < n_wide; ++i) rows[i] = i;                  // row numbers of the classes still to be coloured
                    uint32_t n_colours = colour_wide_classes(rows, h_rowptr, h_ids, M, colour_of);
                    // THIN components first.  A colour is a launch (~100 us of latency however few classes it holds), and classes that
                    // all share one transcript need a colour each: 14.7 k launches per round when one far transcript is shared by
                    // the classes of 4096 ids.  But such a chain is only sequential INSIDE its connected component; a component whose
                    // colours hold < kThinWidth classes on average is cheaper as one wavefront (per 64 chains) walking its classes in
                    // order (~8 us per class), all thin components side by side in one launch.  What is left is coloured again.
                    std::vector<uint32_t> comp_of;
                    const uint32_t n_comp = components_of_wide_classes(rows, h_rowptr, h_ids, M, comp_of);
                    std::vector<uint32_t> comp_n(n_comp, 0), comp_colours(n_comp, 0);
                    for (size_t i = 0; i < wl.size(); ++i) { ++comp_n[comp_of[i]]; comp_colours[comp_of[i]] = std::max(comp_colours[comp_of[i]], colour_of[i] + 1); }
                    std::vector<uint32_t> thin_id(n_comp, 0xFFFFFFFFu);
                    uint32_t n_thin = 0;
                    for (uint32_t c = 0; c < n_comp; ++c)
                        if ((uint64_t)comp_n[c] < (uint64_t)kThinWidth * comp_colours[c]) thin_id[c] = n_thin++;
                    std::vector<uint32_t> rest, rest_rows;
                    if (n_thin) {
                        thin_off.assign(n_thin + 1, 0);
                        for (size_t i = 0; i < wl.size(); ++i) if (thin_id[comp_of[i]] != 0xFFFFFFFFu) ++thin_off[thin_id[comp_of[i]] + 1];
                        for (uint32_t c = 0; c < n_thin; ++c) thin_off[c + 1] += thin_off[c];
                        thin_list.resize(thin_off[n_thin]);
                        std::vector<uint32_t> cur(thin_off.begin(), thin_off.end() - 1);
                        for (size_t i = 0; i < wl.size(); ++i) {
                            const uint32_t t = thin_id[comp_of[i]];
                            if (t != 0xFFFFFFFFu) thin_list[cur[t]++] = wl[i]; else { rest.push_back(wl[i]); rest_rows.push_back((uint32_t)i); }   // class order inside a component
                        }
                        wl.swap(rest);
                        colour_of.clear();
                        n_colours = wl.empty() ? 0 : colour_wide_classes(rest_rows, h_rowptr, h_ids, M, colour_of);
                    }
                    if (n_colours <= kMaxColours) {
                        colour_off.assign(n_colours + 1, 0);
                        for (uint32_t c : colour_of) ++colour_off[c + 1];
                        for (uint32_t c = 0; c < n_colours; ++c) colour_off[c + 1] += colour_off[c];
                        std::vector<uint32_t> by_colour(wl.size()), cur(colour_off.begin(), colour_off.end() - 1);
                        for (size_t i = 0; i < wl.size(); ++i) by_colour[cur[colour_of[i]]++] = wl[i];       // class order inside a colour
                        wl.swap(by_colour);
                    }
                    n_rest = (unsigned int)wl.size();
                    wl.insert(wl.end(), thin_list.begin(), thin_list.end());       // device list: [coloured (or serial) rest | thin components]
                }
                G_TRY(hipMemcpyAsync(wide_list, wl.data(), (size_t)n_wide * 4, hipMemcpyHostToDevice, st));
                if (!thin_off.empty()) {
                    G_TRY(pool_malloc(&d_thin_off, thin_off.size() * 4));
                    G_TRY(hipMemcpyAsync(d_thin_off, thin_off.data(), thin_off.size() * 4, hipMemcpyHostToDevice, st));
                }
                G_TRY(hipStreamSynchronize(st));
            }
        }
        log_msg(0, "gibbs: %u chains, %llu heavy tiles, %u tiles in %u phases, %u wide classes%s%s", n_chains, (unsigned long long)n_heavy_tiles, n_tiles, K, n_wide,
                colour_off.empty() ? "" : (": " + std::to_string(n_rest) + " in " + std::to_string(colour_off.size() - 1) + " colours").c_str(),
                thin_off.empty() ? "" : (", " + std::to_string(thin_list.size()) + " in " + std::to_string(thin_off.size() - 1) + " thin components").c_str());
        GibbsArgs a{n_chains, C, prob->d_rowptr, prob->d_ids, prob->d_counts, inv_len, w_mass, count_map, txp_count,
                    wide, wide_list, n_rest, seed, 0};
        const unsigned groups = (n_chains + kGibbsBlock - 1) / kGibbsBlock;
        auto sweep = [&](bool init) -> hipError_t {
            for (uint32_t p = 0; p < K && p < n_tiles; ++p) {
                dim3 g((n_tiles - p + K - 1) / K, groups);
                if (phase_kind[p] & 1u) {
                    if (init) hipLaunchKernelGGL((k_gibbs_phase<true, true>), g, dim3(kGibbsBlock), 0, st, a, p, K, n_tiles);
                    else hipLaunchKernelGGL((k_gibbs_phase<false, true>), g, dim3(kGibbsBlock), 0, st, a, p, K, n_tiles);
                }
                if (phase_kind[p] & 2u) {
                    if (init) hipLaunchKernelGGL((k_gibbs_phase<true, false>), g, dim3(kGibbsBlock), 0, st, a, p, K, n_tiles);
                    else hipLaunchKernelGGL((k_gibbs_phase<false, false>), g, dim3(kGibbsBlock), 0, st, a, p, K, n_tiles);
                }
            }
            if (!thin_off.empty()) {
                dim3 g((unsigned)(thin_off.size() - 1), groups);
                if (init) hipLaunchKernelGGL(k_gibbs_components<true>, g, dim3(kGibbsBlock), 0, st, a, wide_list + n_rest, d_thin_off);
                else hipLaunchKernelGGL(k_gibbs_components<false>, g, dim3(kGibbsBlock), 0, st, a, wide_list + n_rest, d_thin_off);
            }
            if (n_rest && colour_off.empty()) {
                if (init) hipLaunchKernelGGL(k_gibbs_wide<true>, dim3(groups), dim3(kGibbsBlock), 0, st, a);
                else hipLaunchKernelGGL(k_gibbs_wide<false>, dim3(groups), dim3(kGibbsBlock), 0, st, a);
            } else if (n_rest) {
                for (size_t c = 0; c + 1 < colour_off.size(); ++c) {
                    const uint32_t n = colour_off[c + 1] - colour_off[c];
                    uint32_t chunk = (uint32_t)(((uint64_t)n * groups + 4095) / 4096);        // ~4 k wavefronts fill the chip
                    chunk = chunk < 1u ? 1u : (chunk > kListChunk ? kListChunk : chunk);
                    dim3 g((n + chunk - 1) / chunk, groups);
                    if (init) hipLaunchKernelGGL(k_gibbs_list<true>, g, dim3(kGibbsBlock), 0, st, a, wide_list + colour_off[c], n, chunk);
                    else hipLaunchKernelGGL(k_gibbs_list<false>, g, dim3(kGibbsBlock), 0, st, a, wide_list + colour_off[c], n, chunk);
                }
            }
            return hipGetLastError();
        };
        lap("plan");
        G_TRY(sweep(true));                                                           // initCountMap_
        lap("init");
        uint32_t done = 0;
        while (done < n_samples) {
            G_TRY(sweep(false));                                                      // one sampleRound_ per sample
            uint32_t n_emit = (n_samples - done < n_chains) ? (n_samples - done) : n_chains;
            int32_t* dst = d_out ? d_out + (uint64_t)done * M : d_tmp;
            uint64_t tot = (uint64_t)n_emit * M;
            hipLaunchKernelGGL(k_gibbs_emit, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n_chains, n_emit, M,
                               txp_count, dst);
            G_TRY(hipGetLastError());
            if (cb) {
                G_TRY(hipMemcpyAsync(h_tmp, dst, tot * 4, hipMemcpyDeviceToHost, st));
                G_TRY(hipStreamSynchronize(st));
                for (uint32_t s = 0; s < n_emit; ++s)
                    if (!cb(h_tmp + (uint64_t)s * M, M, user)) { set_error("gibbs writer callback failed"); rc = SFGPU_ERR_INVALID; goto done; }
            }
            done += n_emit; ++a.round;
        }
        G_TRY(hipStreamSynchronize(st));
        lap("rounds");
    }
#undef G_TRY
done:
    (void)hipStreamSynchronize(st);
    lap("drain");
    for (void* p : {(void*)count_map, (void*)txp_count, (void*)inv_len, (void*)w_mass, (void*)wide, (void*)wide_list,
                    (void*)d_nwide, (void*)tile_lo, (void*)tile_hi, (void*)tile_kind, (void*)d_tmp, (void*)d_thin_off, (void*)d_lens, (void*)d_off, (void*)d_rows})
        if (p) pool_free(p);
    if (h_tmp) pinned_free(h_tmp);
    // The chain state (4 * nnz * n_chains bytes, 38 GB for cfg3's classes and 1024 chains) stays in the allocator's cache: giving
    // it back to the driver and mapping it again cost the next call 1 - 4 s (measured), ten times what the sampling itself takes.
    // It stays only within the allocator's budget for large blocks (core.hip: a quarter of the device's memory, at most 64 GiB;
    // sfgpu_pool_set_large_limit): beyond it the block goes back to the driver here, so that the process's other allocators
    // (torch's) are not starved by memory they cannot see.  sfgpu_pool_trim() releases everything.
    lap("release");
    return rc;
}

}  // extern "C"
