// misc.hip -- effective lengths (row a14) and the quant.sf columns (row a13).
//
// Replaces  src/SailfishQuantify.cpp:648-673 (getNormalFragLengthDist), :769-807
//           (correctionFactorsFromCounts), :809-838 (computeSmoothedEffectiveLengths),
//           :706-715 (setEffectiveLengthsDirect) and src/GZipWriter.cpp:216-245 (TPM).
// The <=1000-entry correction tables are serial prefix sums: they are built on the host in the
// reference's evaluation order (bit-identical), the O(M) transforms run on the device.
#include "common.h"

#include <cmath>
#include <vector>

namespace sfgpu {

constexpr int kMiscBlock = 256;
constexpr int kMiscMaxBlocks = 1024;

__global__ void k_efflen(uint64_t M, const uint32_t* __restrict__ ref_len, const double* __restrict__ cf,
                         uint32_t max_len, double* __restrict__ eff) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    uint32_t L = ref_len[t];
    if (!cf) { eff[t] = (double)L; return; }                       // setEffectiveLengthsDirect
    double c = (L >= max_len) ? cf[max_len - 1] : cf[L];            // :823-825
    double e = (double)L - c + 1.0;                                 // :827-828
    if (e < 1.0) e = (double)L;                                     // :829-831
    eff[t] = e;
}

__device__ __forceinline__ double blk_sum(double v, double* lds) {
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < kMiscBlock / kWave; ++i) t += lds[i];
    __syncthreads();
    return t;
}

// tfracDenom = sum_t (estCount_t / numMapped) / len_t   (GZipWriter.cpp:228-234), two-stage
__global__ void k_tpm_partial(uint64_t M, const double* __restrict__ est, const double* __restrict__ len,
                              double num_mapped, double* partials) {
    __shared__ double lds[kMiscBlock / kWave];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kMiscBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kMiscBlock)
        v += (est[t] / num_mapped) / len[t];
    double s = blk_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ void k_tpm(uint64_t M, const double* __restrict__ est, const double* __restrict__ len, double num_mapped,
                      const double* partials, int nb, double* __restrict__ tpm) {
    __shared__ double lds[kMiscBlock / kWave];
    __shared__ double denom_s;
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += kMiscBlock) v += partials[i];
    double d = blk_sum(v, lds);
    if (threadIdx.x == 0) denom_s = d;
    __syncthreads();
    double denom = denom_s;
    for (uint64_t t = (uint64_t)blockIdx.x * kMiscBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kMiscBlock) {
        double npm = est[t] / num_mapped;                           // :241
        double tfrac = (npm / len[t]) / denom;                      // :242
        tpm[t] = tfrac * 1000000.0;                                 // :243
    }
}

// --unsmoothedFLD (computeEmpiricalEffectiveLengths, src/SailfishQuantify.cpp:745-762): one lane per
// transcript walks the (<= maxFragLen-entry, cache-resident) float pdf table in the reference's order, so the
// double sum is bit-identical to the serial loop.
__global__ void k_efflen_empirical(uint64_t M, const uint32_t* __restrict__ ref_len, const float* __restrict__ pdf,
                                   uint32_t pdf_len, uint32_t max_val, float median, int valid_support,
                                   double* __restrict__ eff) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    const uint32_t L = ref_len[t];
    const double ref = (double)L;
    if (ref <= median || !valid_support) { eff[t] = ref; return; }   // a NaN median (empty table) compares false, as in the reference
    const uint32_t hi = L < max_val ? L : max_val;
    double e = 0.0;
    for (uint32_t l = 0; l <= hi; ++l) {
        const float p = (l < pdf_len) ? pdf[l] : 0.0f;
        e += p * ((double)(L - l) + 1.0);
    }
    eff[t] = e;
}

}  // namespace sfgpu

using namespace sfgpu;

extern "C" {

int sfgpu_cf_gaussian(uint32_t max_frag_len, uint64_t mean, uint64_t sd, double* h_cf) {
    SF_REQUIRE(h_cf && max_frag_len > 0, SFGPU_ERR_INVALID, "sfgpu_cf_gaussian: bad argument");
    double cum_mass = 0.0, cum_dens = 0.0;
    for (uint32_t i = 0; i < max_frag_len; ++i) {
        double inv_std = 1.0 / (double)sd;                          // :657-661 kernel lambda
        double x = inv_std * ((double)i - (double)mean);
        double d = std::exp(-0.5 * x * x) * inv_std;
        cum_mass += (double)i * d;                                  // :666
        cum_dens += d;
        h_cf[i] = (cum_dens > 0) ? cum_mass / cum_dens : 0.0;       // :668-670
    }
    return SFGPU_OK;
}

int sfgpu_cf_counts(const uint32_t* h_fl_counts, uint32_t max_frag_len, double* h_cf) {
    SF_REQUIRE(h_fl_counts && h_cf && max_frag_len > 0, SFGPU_ERR_INVALID, "sfgpu_cf_counts: bad argument");
    double vals_prev = 0.0;
    uint32_t mult_prev = h_fl_counts[0];                            // :779-784
    h_cf[0] = 0.0;
    for (uint32_t i = 1; i < max_frag_len; ++i) {                   // :791-803
        uint32_t v = h_fl_counts[i];
        double vals_i = (double)((uint64_t)v * (uint64_t)i) + vals_prev;
        uint32_t mult_i = v + mult_prev;
        h_cf[i] = (mult_i > 0) ? vals_i / (double)mult_i : 0.0;
        vals_prev = vals_i; mult_prev = mult_i;
    }
    return SFGPU_OK;
}

int sfgpu_efflen_smoothed(const uint32_t* d_ref_len, uint64_t M, const double* h_cf, uint32_t max_frag_len,
                          double* d_eff_len, sfgpu_stream stream) {
    SF_REQUIRE(d_ref_len && d_eff_len, SFGPU_ERR_INVALID, "sfgpu_efflen_smoothed: null pointer");
    if (M == 0) return SFGPU_OK;
    hipStream_t st = as_stream(stream);
    double* d_cf = nullptr;
    if (h_cf) {
        SF_REQUIRE(max_frag_len > 0, SFGPU_ERR_INVALID, "sfgpu_efflen_smoothed: max_frag_len == 0");
        SF_HIP(pool_malloc(&d_cf, (size_t)max_frag_len * 8));
        hipError_t e = hipMemcpyAsync(d_cf, h_cf, (size_t)max_frag_len * 8, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { pool_free(d_cf); SF_HIP(e); }
    }
    hipLaunchKernelGGL(k_efflen, dim3((unsigned)((M + kMiscBlock - 1) / kMiscBlock)), dim3(kMiscBlock), 0, st, M,
                       d_ref_len, d_cf, max_frag_len, d_eff_len);
    hipError_t le = hipGetLastError();
    if (d_cf) pool_free_on(d_cf, st);                   // (stream-ordered: no host wait between the class build and the EM's plan; h_cf was consumed by the copy above -- pageable memory is staged before hipMemcpyAsync returns)
    SF_HIP(le);
    return SFGPU_OK;
}

int sfgpu_efflen_empirical(const uint32_t* h_fl_counts, uint32_t max_frag_len, const uint32_t* d_ref_len, uint64_t M,
                           double* d_eff_len, sfgpu_stream stream) {
    SF_REQUIRE(h_fl_counts && d_ref_len && d_eff_len && max_frag_len > 0, SFGPU_ERR_INVALID, "sfgpu_efflen_empirical: bad argument");
    if (M == 0) return SFGPU_OK;
    // EmpiricalDistribution::buildDistribution (src/EmpiricalDistribution.cpp:29-96) for vals = 0..n-1 (the
    // caller's jointMap holds every fragment length, zero counts included: src/SailfishQuantify.cpp:944-946)
    const uint32_t n = max_frag_len;
    double total = 0.0;
    for (uint32_t i = 0; i < n; ++i) total += h_fl_counts[i];
    uint32_t cut = 0, table_len = 1;                    // table_len = vals[lastval] where the cumulative mass passes 1 - 1e-6
    {
        double cum = 0.0;
        for (; cut < n; ++cut) {
            cum += h_fl_counts[cut] / total;
            table_len = cut;
            if (cum > 1.0 - 1e-6) break;
        }
    }
    double kept = 0.0;
    for (uint32_t i = 0; i < cut; ++i) kept += h_fl_counts[i];
    std::vector<float> pdf(table_len ? table_len : 1, 0.0f);
    for (uint32_t v = 0; v < table_len; ++v) pdf[v] = (float)(h_fl_counts[v] / kept);
    // median: the two-ended walk of :78-92 on the unsigned counts
    size_t lo = 0, hi = (size_t)n - 1;
    unsigned int a = h_fl_counts[lo], b = h_fl_counts[hi];
    while (lo < hi) {
        if (a <= b) { b -= a; a = h_fl_counts[++lo]; }
        else { a -= b; b = h_fl_counts[--hi]; }
    }
    const float median = table_len ? (float)lo : std::nanf("");
    const int valid_support = (n - 1) > 0;              // maxVal > minVal with minVal = 0, maxVal = n - 1
    hipStream_t st = as_stream(stream);
    float* d_pdf = nullptr;
    SF_HIP(pool_malloc(&d_pdf, pdf.size() * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d_pdf, pdf.data(), pdf.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_efflen_empirical, dim3((unsigned)((M + kMiscBlock - 1) / kMiscBlock)), dim3(kMiscBlock), 0, st, M,
                           d_ref_len, d_pdf, table_len, n - 1, median, valid_support, d_eff_len);
        e = hipGetLastError();
    }
    (void)hipStreamSynchronize(st);                     // pdf is a pageable host vector and d_pdf goes back to the pool
    pool_free(d_pdf);
    SF_HIP(e);
    return SFGPU_OK;
}

int sfgpu_tpm(const double* d_est_count, const double* d_len, uint64_t M, double num_mapped, double* d_tpm,
              sfgpu_stream stream) {
    SF_REQUIRE(d_est_count && d_len && d_tpm, SFGPU_ERR_INVALID, "sfgpu_tpm: null pointer");
    if (M == 0) return SFGPU_OK;
    hipStream_t st = as_stream(stream);
    int nb = (int)((M + kMiscBlock - 1) / kMiscBlock);
    if (nb > kMiscMaxBlocks) nb = kMiscMaxBlocks;
    double* partials = nullptr;
    SF_HIP(pool_malloc(&partials, (size_t)kMiscMaxBlocks * 8));
    hipLaunchKernelGGL(k_tpm_partial, dim3(nb), dim3(kMiscBlock), 0, st, M, d_est_count, d_len, num_mapped, partials);
    hipLaunchKernelGGL(k_tpm, dim3(nb), dim3(kMiscBlock), 0, st, M, d_est_count, d_len, num_mapped, partials, nb, d_tpm);
    hipError_t le = hipGetLastError();
    (void)hipStreamSynchronize(st);
    pool_free(partials);
    SF_HIP(le);
    return SFGPU_OK;
}

}  // extern "C"
