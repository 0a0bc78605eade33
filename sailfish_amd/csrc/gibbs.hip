// gibbs.hip -- collapsed Gibbs sampling over the equivalence classes (row a16).
//
// Replaces  src/CollapsedGibbsSampler.cpp:35-94 (initCountMap_), :96-186 (sampleRound_),
//           :198-291 (sample<ReadExperiment>).
//
// The reference runs one chain per TBB chunk of the sample range (:223-246): initCountMap_ assigns
// every class's reads to its transcripts with one multinomial draw, then each sample is the previous
// one plus ONE sampleRound_ (`bool numInternalRounds = 10` is 1, :248).  Inside a round the classes are
// visited in order and each visit reads the transcript counts the previous visits left behind, so a
// chain is inherently sequential; the parallelism on the device is across chains:
//   * one LANE per chain, 64 chains per wavefront, walking the classes in lock step.  The class
//     structure (rowptr / ids / counts / effective lengths) is the same for every chain, so those
//     loads are wave-uniform, and the per-chain state is laid out chain-minor --
//     countMap[nonzero][chain], txpCount[transcript][chain] -- so every state access of a
//     wavefront is one coalesced 256-byte transaction.  No cross-lane traffic at all;
//   * the multinomial over a class's k members is a chain of k-1 conditional binomials
//     (exact BINV/BTPE sampler, rng.h) instead of n inverse-CDF draws (MultinomialSampler.hpp:13-64):
//     cost O(k), not O(reads);
//   * the aux weight of member t is (count/effLen_t)/sum, i.e. proportional to 1/effLen_t inside a
//     class; its normaliser cancels in the multinomial probabilities, so 1/effLen_t is used directly.
// Random numbers: Philox4x32-10 keyed by (seed; chain, round, class) -- reproducible; the reference seeds
// std::mt19937 from std::random_device (:104-105, :227-228), so parity is distributional.
#include "common.h"
#include "rng.h"

#include <vector>

namespace sfgpu {

constexpr int kGibbsBlock = 64;            // one wavefront of chains per block
constexpr double kGibbsPrior = 1e-8;       // priorAlpha (:215)

// multinomial(n; p_0..p_{k-1}) as conditional binomials; calls put(i, r_i) for every member
template <typename ProbFn, typename PutFn>
__device__ __forceinline__ void multinomial_chain(Philox& g, uint32_t n, uint32_t k, double p_total, ProbFn prob, PutFn put) {
    double p_rem = p_total;
    uint32_t n_rem = n;
    for (uint32_t i = 0; i < k; ++i) {
        uint32_t r;
        if (i + 1 == k) r = n_rem;
        else {
            double p = prob(i);
            double ratio = (p_rem > 0.0) ? p / p_rem : 0.0;
            r = (n_rem == 0) ? 0u : binomial(g, n_rem, ratio < 1.0 ? ratio : 1.0);
            p_rem -= p; if (p_rem < 0.0) p_rem = 0.0;
        }
        put(i, r);
        n_rem -= r;
    }
}

// initCountMap_ (:35-94): initial assignment from (prior + mass_t) * aux_t
__global__ void __launch_bounds__(kGibbsBlock)
k_gibbs_init(uint32_t n_chains, uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
             const uint64_t* __restrict__ counts, const double* __restrict__ w_mass /* (prior+mass_t)/effLen_t */,
             uint32_t* __restrict__ count_map, int32_t* __restrict__ txp_count, uint64_t seed) {
    const uint32_t ch = blockIdx.x * kGibbsBlock + threadIdx.x;
    if (ch >= n_chains) return;
    for (uint64_t c = 0; c < C; ++c) {
        const uint32_t b = rowptr[c], k = rowptr[c + 1] - b;
        const uint32_t n = (uint32_t)counts[c];          // uint32 n in MultinomialSampler (:15)
        if (k == 1) {                                    // :81-83
            count_map[(uint64_t)b * n_chains + ch] = n;
            txp_count[(uint64_t)ids[b] * n_chains + ch] += (int32_t)n;
            continue;
        }
        if (k == 0) continue;
        double denom = 0.0;
        for (uint32_t i = 0; i < k; ++i) denom += w_mass[ids[b + i]];            // :58-63
        if (!(denom > 4.9406564584124654e-324)) {                                // :65 -- nothing assigned
            for (uint32_t i = 0; i < k; ++i) count_map[(uint64_t)(b + i) * n_chains + ch] = 0;
            continue;
        }
        Philox g; g.init(seed, ch, c);
        multinomial_chain(g, n, k, denom,
            [&](uint32_t i) { return w_mass[ids[b + i]]; },
            [&](uint32_t i, uint32_t r) {
                count_map[(uint64_t)(b + i) * n_chains + ch] = r;                // :76-79
                txp_count[(uint64_t)ids[b + i] * n_chains + ch] += (int32_t)r;  // :86-89
            });
    }
}

// sampleRound_ (:96-186)
__global__ void __launch_bounds__(kGibbsBlock)
k_gibbs_round(uint32_t n_chains, uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
              const double* __restrict__ inv_len, uint32_t* __restrict__ count_map, int32_t* __restrict__ txp_count,
              uint64_t seed, uint32_t round) {
    const uint32_t ch = blockIdx.x * kGibbsBlock + threadIdx.x;
    if (ch >= n_chains) return;
    for (uint64_t c = 0; c < C; ++c) {
        const uint32_t b = rowptr[c], k = rowptr[c + 1] - b;
        if (k <= 1) continue;                                   // singletons keep their full count (:128)
        Philox g; g.init(seed, ((uint64_t)(round + 1) << 32) | ch, c);
        const double frac = 0.25 + 0.5 * g.uniform();           // U(0.25, 0.75) per class (:106, :115)
        // pass 1: take round(frac * current) reads away from every member (:138-148)
        uint32_t n_res = 0; double denom = 0.0;
        for (uint32_t i = 0; i < k; ++i) {
            const uint64_t at = (uint64_t)(b + i) * n_chains + ch;
            const uint32_t t = ids[b + i];
            const uint32_t cur = count_map[at];
            const uint32_t r = (uint32_t)(frac * (double)cur + 0.5);             // std::round, values >= 0 (:142)
            n_res += r;
            count_map[at] = cur - r;
            const int32_t tc = txp_count[(uint64_t)t * n_chains + ch] - (int32_t)r;
            txp_count[(uint64_t)t * n_chains + ch] = tc;
            denom += (kGibbsPrior + (double)tc) * inv_len[t];                    // :147
        }
        // pass 2: re-draw them from p_i ~ (prior + txpCount_i) * aux_i (:150-170).  denom >= k*1e-8/len > 0,
        // so the reference's "did not sample" branch (:172-179) cannot trigger for finite inputs.
        multinomial_chain(g, n_res, k, denom,
            [&](uint32_t i) { const uint32_t t = ids[b + i]; return (kGibbsPrior + (double)txp_count[(uint64_t)t * n_chains + ch]) * inv_len[t]; },
            [&](uint32_t i, uint32_t r) {
                if (r) { count_map[(uint64_t)(b + i) * n_chains + ch] += r; txp_count[(uint64_t)ids[b + i] * n_chains + ch] += (int32_t)r; }
            });
    }
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
    uint32_t* count_map = nullptr; int32_t* txp_count = nullptr; double *inv_len = nullptr, *w_mass = nullptr;
    int32_t* d_tmp = nullptr; int32_t* h_tmp = nullptr;
    int rc = SFGPU_OK;
#define G_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); rc = SFGPU_ERR_HIP; goto done; } } while (0)
    G_TRY(pool_malloc(&count_map, ((uint64_t)L * n_chains + 1) * 4));
    G_TRY(pool_malloc(&txp_count, (uint64_t)M * n_chains * 4));
    G_TRY(pool_malloc(&inv_len, M * 8)); G_TRY(pool_malloc(&w_mass, M * 8));
    if (!d_out) G_TRY(pool_malloc(&d_tmp, (uint64_t)n_chains * M * 4));
    if (cb) G_TRY(hipHostMalloc(&h_tmp, (uint64_t)n_chains * M * 4, hipHostMallocDefault));
    G_TRY(hipMemsetAsync(txp_count, 0, (uint64_t)M * n_chains * 4, st));
    hipLaunchKernelGGL(k_gibbs_weights, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, M, prob->d_len, d_mass,
                       (double)prob->num_mapped, inv_len, w_mass);
    {
        const unsigned g = (n_chains + kGibbsBlock - 1) / kGibbsBlock;
        hipLaunchKernelGGL(k_gibbs_init, dim3(g), dim3(kGibbsBlock), 0, st, n_chains, C, prob->d_rowptr, prob->d_ids,
                           prob->d_counts, w_mass, count_map, txp_count, seed);
        G_TRY(hipGetLastError());
        uint32_t done = 0, round = 0;
        while (done < n_samples) {
            hipLaunchKernelGGL(k_gibbs_round, dim3(g), dim3(kGibbsBlock), 0, st, n_chains, C, prob->d_rowptr, prob->d_ids,
                               inv_len, count_map, txp_count, seed, round);
            G_TRY(hipGetLastError());
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
            done += n_emit; ++round;
        }
        G_TRY(hipStreamSynchronize(st));
    }
#undef G_TRY
done:
    if (count_map) pool_free(count_map);
    if (txp_count) pool_free(txp_count);
    if (inv_len) pool_free(inv_len);
    if (w_mass) pool_free(w_mass);
    if (d_tmp) pool_free(d_tmp);
    if (h_tmp) (void)hipHostFree(h_tmp);
    pool_trim();          // the chain state is large (4 * nnz * n_chains bytes): do not keep it cached
    return rc;
}

}  // extern "C"
