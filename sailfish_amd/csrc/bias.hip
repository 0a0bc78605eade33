// bias.hip -- bias-aware effective lengths: sailfish::utils::updateEffectiveLengths,
// /root/reference src/SailfishUtils.cpp:611-926, for gfx950.
//
// What the reference does, per call: (1) walk every expressed transcript and accumulate the EXPECTED
// distribution of the bias feature under the current abundances -- the 6-mer context at a fragment start
// (4096 bins, both strands) or the GC percentage of every (start, fragment length) pair (101 bins);
// (2) turn observed / expected into a per-bin weight; (3) walk every transcript again and sum the weights of
// all positions (x fragment lengths) into its new effective length.  The loops are serial over transcripts in
// the reference.
//
// Device layout.
//   Sequence model: O(sum of lengths) work.  k_seq_expected keeps the 4096 bins in LDS (f64 LDS atomics),
//   one block striding over transcripts, and flushes each block's bins with global f64 atomics;
//   k_seq_efflen is one block per transcript.  A transcript is staged through LDS one code byte per base and a
//   thread builds the 6-mer indices of four consecutive positions (first packed, then the reference's rolling
//   update; a byte that is not ACGTU adds 0 in both).
//   GC model: O(sum of lengths x fragment lengths) -- 6e10 pairs for a human transcriptome.  The GC bin of
//   a pair depends only on the sequence and the fragment-length grid, NOT on the abundances, and both passes
//   of the reference weigh a pair by a factor that depends only on (bin, fragment length) x a per-transcript
//   scalar.  So the pairs are counted ONCE per handle into a per-transcript profile
//       S[t][g] = sum_k w_k * #{ i : gcFrac(i, i + fl_k - 1) == g },   w_k = cdf(fl_k) - cdf(fl_{k-1})
//   (k_gc_profile: lane = fragment length, the block walks the positions; integer counters in LDS, one row of
//   101 per lane so that lanes never share an address; GC prefix counts staged through LDS chunk by chunk),
//   and every later update is two passes over S (808 bytes per transcript):
//       expected[g] = 1 + sum_t (alpha_t / effLen_t) * S[t][g]            (k_gc_expected_*, fixed order)
//       effLen'_t   = (sum_g observed[g] / (prior + expected[g]) * S[t][g]) * (probFwd + probRC) * norm
//   gcFrac's lrint((100.0 * d) / fl) is evaluated without a division (gc_bin): an f32 product, and an exact
//   f32 residual that recognises the ties lrint sends to the even neighbour.
#include <algorithm>
#include <cmath>
#include <vector>

#include "bias.h"
#include "common.h"

namespace sfgpu {
namespace {

constexpr int kK = 6;                      // ReadKmerDist<6>, include/ReadExperiment.hpp:249
constexpr int kNKmer = 4096;
constexpr int kNGC = 101;
constexpr int kBlock = 256;
constexpr double kMinAlphaBias = 1e-8;     // :618
constexpr int kLanesFl = 64;               // fragment lengths per chunk of k_gc_profile: one per lane
constexpr int kGcBlock = 512;              // k_gc_profile: eight wavefronts share one chunk's counters and staged counts
constexpr uint32_t kMaxFldHigh = 16000;
constexpr int kSeqBlocks = 1024;
constexpr int kGcRows = 256;               // transcripts per block of the GC expectation partial sums
constexpr uint32_t kStageMin = 4096;       // GC prefix counts staged per chunk (words), at least

struct BiasDev {
    uint64_t M;
    const char* seq; const uint64_t* seq_off; const uint32_t* ref_len; const double* txp_eff;
    const float* cdf; uint32_t cdf_size;
    const uint32_t* read_bias; const uint32_t* observed_gc;
    double prob_fwd, prob_rc, read_norm, read_gc_norm;
    const double* w; uint32_t nfl; int32_t fld_low; uint32_t gs; uint32_t stage_cap;
    double* S;
    double *exp_seq, *ratio_seq, *exp_gc, *ratio_gc, *gc_partial, *scal;   // scal[0] seq norm ratio, scal[1] gc norm ratio
    uint8_t* corrected;                    // [M] 1 = the last update replaced the length (numCorrected :806)
    unsigned long long* n_corrected;
};

__device__ __forceinline__ float cdf_at(const BiasDev& d, uint32_t x) { return x < d.cdf_size ? d.cdf[x] : 1.0f; }   // cdf() :121-124

// :703-712 / :821-832  max(0, RefLength - (int32)EffectiveLength)
__device__ __forceinline__ int unprocessed_len(uint32_t L, double txp_eff) {
    int u = (int)L - (int)txp_eff;
    return u > 0 ? u : 0;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}
// deterministic block sum, result in every thread
__device__ __forceinline__ double block_sum_all(double v, double* lds /* kBlock / kWave + 1 */) {
    v = wave_sum(v);
    if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < kBlock / kWave; ++i) t += lds[i]; lds[kBlock / kWave] = t; }
    __syncthreads();
    double r = lds[kBlock / kWave];
    __syncthreads();
    return r;
}

// The sequence kernels stage a transcript through LDS one byte per base -- bits 0-1 the base code (A0 C1 G2 T/U3),
// bits 4-5 its complement's code, 0 for any other byte (nextKmerIndex adds nothing for it, include/
// UtilityFunctions.hpp:40-90) -- and every thread builds the 6-mer indices of four consecutive positions from
// three LDS words: the first by packing six codes (indexForKmer :93-148: forward = first base most significant,
// reverse complement = complement of the LAST base most significant), the next three by the rolling update.
constexpr uint32_t kSeqStage = 4096;       // positions per staging round

__device__ __forceinline__ void stage_codes(const char* s, uint32_t n_bytes, uint8_t* codes) {
    for (uint32_t j = threadIdx.x; j < n_bytes; j += kBlock) {
        const unsigned c = (unsigned char)s[j] & 0xDFu;
        uint32_t v = 0;
        if (c == 'A') v = 0x30; else if (c == 'C') v = 0x21; else if (c == 'G') v = 0x12; else if (c == 'T' || c == 'U') v = 0x03;
        codes[j] = (uint8_t)v;
    }
    __syncthreads();
}

__device__ __forceinline__ void kmer4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t f[4], uint32_t r[4]) {
    uint32_t c[9], m[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = (w0 >> (8 * k)) & 3u; m[k] = (w0 >> (8 * k + 4)) & 3u;
        c[4 + k] = (w1 >> (8 * k)) & 3u; m[4 + k] = (w1 >> (8 * k + 4)) & 3u;
    }
    c[8] = w2 & 3u; m[8] = (w2 >> 4) & 3u;
    f[0] = (c[0] << 10) | (c[1] << 8) | (c[2] << 6) | (c[3] << 4) | (c[4] << 2) | c[5];
    r[0] = m[0] | (m[1] << 2) | (m[2] << 4) | (m[3] << 6) | (m[4] << 8) | (m[5] << 10);
#pragma unroll
    for (int u = 1; u < 4; ++u) {
        f[u] = ((f[u - 1] << 2) | c[5 + u]) & 0xFFFu;
        r[u] = (r[u - 1] >> 2) | (m[5 + u] << 10);
    }
}

__global__ void k_fill(double* p, uint32_t n, double v) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- sequence-specific model ---------------------------------------------------------------------------
// pass 1 (:697-785): transcriptKmerDist[idx] += probFwd * contribution * cdf(refLen - i - 1)   (context of a
// fragment starting on the forward strand) and += probRC * contribution * cdf(i + 5) (reverse strand)
__global__ void __launch_bounds__(kBlock) k_seq_expected(BiasDev d, const double* __restrict__ eff_in,
                                                         const double* __restrict__ alpha) {
    __shared__ double hist[kNKmer];
    __shared__ __align__(16) uint8_t codes[kSeqStage + 16];
    for (int j = threadIdx.x; j < kNKmer; j += kBlock) hist[j] = 0.0;
    __syncthreads();
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(codes);
    for (uint64_t t = blockIdx.x; t < d.M; t += gridDim.x) {
        const uint32_t L = d.ref_len[t];
        const double a = alpha[t];
        if (a < kMinAlphaBias || unprocessed_len(L, d.txp_eff[t]) <= 0 || L <= (uint32_t)kK) continue;   // uniform
        const double contribution = a / eff_in[t];
        const double cf = d.prob_fwd * contribution, cr = d.prob_rc * contribution;
        const char* s = d.seq + d.seq_off[t];
        const uint32_t n_pos = L - kK;                                      // i = refLen - K - 1 .. 0
        for (uint32_t c0 = 0; c0 < n_pos; c0 += kSeqStage) {
            const uint32_t np = min(kSeqStage, n_pos - c0);
            __syncthreads();
            stage_codes(s + c0, np + kK - 1, codes);
            for (uint32_t q = threadIdx.x; 4 * q < np; q += kBlock) {
                uint32_t f[4], r[4];
                kmer4(cw[q], cw[q + 1], cw[q + 2], f, r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = c0 + 4 * q + u;
                    if (4 * q + u < np) {
                        atomicAdd(&hist[r[u]], cf * (double)cdf_at(d, L - i - 1));
                        if (i + 5 < L) atomicAdd(&hist[f[u]], cr * (double)cdf_at(d, i + 5));
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < kNKmer; j += kBlock) { double v = hist[j]; if (v != 0.0) atomicAdd(&d.exp_seq[j], v); }
}

// :795-803  txomeNormFactor, seqPrior; ratio = readBias.counts / (transcriptKmerDist + seqPrior)
__global__ void __launch_bounds__(kBlock) k_seq_norm(BiasDev d) {
    __shared__ double lds[kBlock / kWave + 1];
    double v = 0.0;
    for (int j = threadIdx.x; j < kNKmer; j += kBlock) v += d.exp_seq[j];
    const double txome = block_sum_all(v, lds);
    const double pmass = (double)kNKmer;
    const double prior = ((pmass / (d.read_norm - pmass)) * txome) / pmass;
    for (int j = threadIdx.x; j < kNKmer; j += kBlock) d.ratio_seq[j] = (double)d.read_bias[j] / (d.exp_seq[j] + prior);
    if (threadIdx.x == 0) d.scal[0] = txome / d.read_norm;
}

// pass 2 (:810-923): effLength = sum over positions of both strands' weights, x txomeNormFactor / readNormFactor
__global__ void __launch_bounds__(kBlock) k_seq_efflen(BiasDev d, const double* eff_in, const double* __restrict__ alpha,
                                                       double* eff_out) {
    __shared__ double lds[kBlock / kWave + 1];
    const uint64_t t = blockIdx.x;
    const uint32_t L = d.ref_len[t];
    const int unproc = unprocessed_len(L, d.txp_eff[t]);
    const double e_in = eff_in[t];
    double eff_length = 0.0;
    if (alpha[t] >= kMinAlphaBias && unproc > 0 && L > (uint32_t)kK) {      // uniform over the block
        __shared__ __align__(16) uint8_t codes[kSeqStage + 16];
        const uint32_t* cw = reinterpret_cast<const uint32_t*>(codes);
        const char* s = d.seq + d.seq_off[t];
        const uint32_t n_pos = L - kK;
        double acc = 0.0;
        for (uint32_t c0 = 0; c0 < n_pos; c0 += kSeqStage) {
            const uint32_t np = min(kSeqStage, n_pos - c0);
            __syncthreads();
            stage_codes(s + c0, np + kK - 1, codes);
            for (uint32_t q = threadIdx.x; 4 * q < np; q += kBlock) {
                uint32_t f[4], r[4];
                kmer4(cw[q], cw[q + 1], cw[q + 2], f, r);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = c0 + 4 * q + u;
                    if (4 * q + u < np) {
                        acc += (d.prob_fwd * d.ratio_seq[r[u]]) * (double)cdf_at(d, L - i - 1);
                        acc += (d.prob_rc * d.ratio_seq[f[u]]) * (double)cdf_at(d, i + 5);
                    }
                }
            }
        }
        eff_length = block_sum_all(acc, lds) * d.scal[0];
    }
    if (threadIdx.x == 0) {
        const bool corrected = unproc > 0 && eff_length > (double)unproc;   // :915-921
        eff_out[t] = corrected ? eff_length : e_in;
        d.corrected[t] = corrected ? 1 : 0;
    }
}

// numCorrected (:806): one atomic per block, only when the caller asks for the statistics
__global__ void __launch_bounds__(kBlock) k_count_corrected(BiasDev d) {
    __shared__ double lds[kBlock / kWave + 1];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < d.M; t += (uint64_t)gridDim.x * kBlock) v += d.corrected[t];
    const double s = block_sum_all(v, lds);
    if (threadIdx.x == 0 && s > 0.0) atomicAdd(d.n_corrected, (unsigned long long)s);
}

// ---- fragment-GC model ---------------------------------------------------------------------------------
// gcFrac's lrint((100.0 * d) / fl) (default rounding: half to even) without a division.  x = 200 d * (0.5 / fl)
// in f32 is within 1e-5 of the quotient and rint(x) is the right integer unless the exact quotient is a tie: a
// non-tie lies at least 1 / (2 fl) > 3e-5 from the nearest tie (fl < 16000).  The residual 200 d - 2 rint(x) fl is
// an integer below 2^24, so the fma computes it exactly; it is +-fl exactly at a tie m + 1/2, and there the even
// neighbour is 2 rint(x / 2) (x / 2 sits a quarter away from it, far beyond the error).
// (tests/test_bias.py checks the formula against lrint for every d < fl < 16000.)
struct GcLane { float half_rfl, quarter_rfl, flf, neg2fl; };
__device__ __forceinline__ GcLane gc_lane(uint32_t fl) {
    const float h = 0.5f / (float)fl;
    return GcLane{h, 0.5f * h, (float)fl, -2.0f * (float)fl};
}
__device__ __forceinline__ uint32_t gc_bin(uint32_t d, const GcLane& l) {
    const float n200 = (float)__umul24(d, 200u);
    const float r = rintf(n200 * l.half_rfl);
    const float rem2 = fmaf(r, l.neg2fl, n200);
    const float even = rintf(n200 * l.quarter_rfl);
    return (uint32_t)(int)((fabsf(rem2) == l.flf) ? even + even : r);
}

// Gs[j] = number of G/C bases in s[0..j]  (Transcript::computeGCContent_, include/Transcript.hpp:183-196;
// only differences are used, so a chunk-local origin is enough)
template <int BLOCK>
__device__ __forceinline__ void stage_gc_prefix(const char* s, uint32_t n, uint32_t* Gs, uint32_t* runs) {
    for (uint32_t j = threadIdx.x; j < n; j += BLOCK) {
        const unsigned c = (unsigned char)s[j] & 0xDFu;
        Gs[j] = (c == 'G' || c == 'C') ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t R = (n + BLOCK - 1) / BLOCK;
    const uint32_t b = min(n, threadIdx.x * R), e = min(n, b + R);
    uint32_t sum = 0;
    for (uint32_t j = b; j < e; ++j) sum += Gs[j];
    uint32_t incl = sum;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    for (int o = 1; o < kWave; o <<= 1) { uint32_t v = __shfl_up(incl, o, kWave); if (lane >= o) incl += v; }
    if (lane == kWave - 1) runs[wv] = incl;
    __syncthreads();
    uint32_t off = incl - sum;
    for (int w = 0; w < wv; ++w) off += runs[w];
    for (uint32_t j = b; j < e; ++j) { off += Gs[j]; Gs[j] = off; }
    __syncthreads();
}

// Transcript::GCCount_ for gcSampFactor 1 (include/Transcript.hpp:183-196), laid out like the sequence: one block per
// transcript, chunks through LDS with the running count carried over (sfgpu_gc_prefix; read by the GC sampling of
// sfgpu_sample_bias)
constexpr uint32_t kPrefixChunk = 8192;
__global__ void __launch_bounds__(kBlock) k_gc_prefix(const char* __restrict__ seq, const uint64_t* __restrict__ seq_off,
                                                      const uint32_t* __restrict__ ref_len, uint32_t* __restrict__ out) {
    __shared__ uint32_t Gs[kPrefixChunk];
    __shared__ uint32_t runs[kBlock];
    const uint64_t t = blockIdx.x;
    const uint32_t L = ref_len[t];
    const char* s = seq + seq_off[t];
    uint32_t* o = out + seq_off[t];
    uint32_t base = 0;
    for (uint32_t p0 = 0; p0 < L; p0 += kPrefixChunk) {
        const uint32_t n = min(kPrefixChunk, L - p0);
        stage_gc_prefix<kBlock>(s + p0, n, Gs, runs);
        for (uint32_t j = threadIdx.x; j < n; j += kBlock) o[p0 + j] = base + Gs[j];
        base += Gs[n - 1];
        __syncthreads();
    }
}

// S[t][g] for one transcript per block (see the header).  Dynamic LDS: 101 f64 | 64 x 101 u32 | kGcBlock u32 |
// stage_cap u32.
__global__ void __launch_bounds__(kGcBlock) k_gc_profile(BiasDev d) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* Sacc = reinterpret_cast<double*>(smem);
    uint32_t* H = reinterpret_cast<uint32_t*>(smem + 816);
    uint32_t* runs = H + kLanesFl * kNGC;
    uint32_t* Gs = runs + kGcBlock;
    const uint64_t t = blockIdx.x;
    const uint32_t L = d.ref_len[t];
    double* Srow = d.S + t * kNGC;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    if (L <= (uint32_t)kK || unprocessed_len(L, d.txp_eff[t]) <= 0) {       // never walked by the reference
        if (tid < kNGC) Srow[tid] = 0.0;
        return;
    }
    if (tid < kNGC) Sacc[tid] = 0.0;
    const char* s = d.seq + d.seq_off[t];
    const uint32_t n_pos = L - kK;                                          // i = refLen - K - 1 .. 0
    for (uint32_t k0 = 0; k0 < d.nfl; k0 += kLanesFl) {
        const uint32_t fl_min = (uint32_t)d.fld_low + k0 * d.gs;
        if (fl_min > L) break;                                              // fragEnd < refLen fails for every i (:742)
        const uint32_t k_last = min(d.nfl - 1, k0 + kLanesFl - 1);
        const uint32_t fl_max = (uint32_t)d.fld_low + k_last * d.gs;
        const uint32_t n_i = min(n_pos, L - fl_min + 1);                    // positions with at least one valid length
        const uint32_t k = k0 + lane;
        const bool k_on = k < d.nfl;
        const uint32_t one = k_on ? 1u : 0u;                                // lanes past the last length count nothing
        const uint32_t fl = k_on ? (uint32_t)d.fld_low + k * d.gs : fl_max;
        const GcLane gl = gc_lane(fl);
        uint32_t* Hrow = H + lane * kNGC;
        for (int j = tid; j < kLanesFl * kNGC; j += kGcBlock) H[j] = 0;
        const uint32_t T = (d.stage_cap - fl_max) & ~3u;
        for (uint32_t p0 = 0; p0 < n_i; p0 += T) {
            __syncthreads();
            stage_gc_prefix<kGcBlock>(s + p0, min(L - p0, T + fl_max), Gs, runs);
            const uint32_t i_end = min(n_i, p0 + T);
            // groups of four positions for which every lane's fragment ends inside the transcript: no tests
            const uint32_t lim = (fl_max <= L) ? min(i_end, L - fl_max + 1) : p0;   // i < lim: the chunk's longest fragment fits
            const uint32_t n_fast = lim > p0 ? (lim - p0) / 4 : 0;
            const uint32_t* Ge = Gs + (fl - 1);
            for (uint32_t j = wv; j < n_fast; j += kGcBlock / kWave) {
                const uint32_t o = 4 * j;                                   // i - p0
                const uint4 cs = *reinterpret_cast<const uint4*>(Gs + o);
                const uint32_t e0 = Ge[o], e1 = Ge[o + 1], e2 = Ge[o + 2], e3 = Ge[o + 3];
                const uint32_t g0 = gc_bin(e0 - cs.x, gl), g1 = gc_bin(e1 - cs.y, gl);
                const uint32_t g2 = gc_bin(e2 - cs.z, gl), g3 = gc_bin(e3 - cs.w, gl);
                atomicAdd(&Hrow[g0], one); atomicAdd(&Hrow[g1], one);
                atomicAdd(&Hrow[g2], one); atomicAdd(&Hrow[g3], one);
            }
            for (uint32_t i = p0 + 4 * n_fast + wv; i < i_end; i += kGcBlock / kWave) {
                const uint32_t e = i + fl - 1;
                if (k_on && e < L) atomicAdd(&Hrow[gc_bin(Gs[e - p0] - Gs[i - p0], gl)], 1u);   // gcFrac(i, e)
            }
        }
        __syncthreads();
        if (tid < kNGC) {
            double acc = Sacc[tid];
            const uint32_t nk = k_last - k0 + 1;
            for (uint32_t l = 0; l < nk; ++l) acc += d.w[k0 + l] * (double)H[l * kNGC + tid];
            Sacc[tid] = acc;
        }
        __syncthreads();
    }
    if (tid < kNGC) Srow[tid] = Sacc[tid];
}

// gcSampFactor > 1 (--gcSizeSamp): Transcript::gcFrac through gcCountInterp_ (include/Transcript.hpp:91-94, 133-162).
// The reference keeps the G/C count only at every step-th base (plus the last) and interpolates between the two
// samples around a position -- with lambda weighing the LEFT sample, as written: count(p) = lambda * cnt[a] +
// (1 - lambda) * cnt[a + 1], lambda = p / step - a (its "last bin" branch is unreachable: a < ceil((L-1)/step) for
// every p < L - 1).  The sampled counts are the per-base counts at multiples of step, so they are read from the
// per-base table (G, as k_gc_prefix lays it out, offset to this transcript).  The bin is clamped to [0,100] -- the
// interpolated difference can leave that range, where the reference indexes outside its 101 bins.
__device__ __forceinline__ double gc_count_interp(const uint32_t* __restrict__ G, uint32_t L, uint32_t step, uint32_t p) {
    if (p == L - 1) return (double)G[L - 1];
    const double frac_p = (double)p / (double)step;
    const uint32_t samp = (uint32_t)floor(frac_p);
    const double lambda = (frac_p - (double)samp) / ((double)(samp + 1) - (double)samp);
    const uint32_t nxt = min((samp + 1) * step, L - 1);
    return lambda * (double)G[samp * step] + (1.0 - lambda) * (double)G[nxt];
}
__device__ __forceinline__ uint32_t gc_frac_sampled(const uint32_t* __restrict__ G, uint32_t L, uint32_t step, uint32_t s, uint32_t e) {
    const double cs = gc_count_interp(G, L, step, s), ce = gc_count_interp(G, L, step, e);
    long r = lrint((100.0 * (ce - cs)) / (double)(e - s + 1));
    r = r < 0 ? 0 : (r > 100 ? 100 : r);
    return (uint32_t)r;
}

// the slow path of k_gc_profile for gcSampFactor > 1: same counting, bins through the interpolation, counts from the table
__global__ void __launch_bounds__(kBlock) k_gc_profile_sampled(BiasDev d, const uint32_t* __restrict__ gc_table, uint32_t step) {
    __shared__ double Sacc[kNGC];
    __shared__ uint32_t H[kLanesFl * kNGC];
    const uint64_t t = blockIdx.x;
    const uint32_t L = d.ref_len[t];
    double* Srow = d.S + t * kNGC;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    if (L <= (uint32_t)kK || unprocessed_len(L, d.txp_eff[t]) <= 0) { if (tid < kNGC) Srow[tid] = 0.0; return; }
    if (tid < kNGC) Sacc[tid] = 0.0;
    const uint32_t* G = gc_table + d.seq_off[t];
    const uint32_t n_pos = L - kK;
    for (uint32_t k0 = 0; k0 < d.nfl; k0 += kLanesFl) {
        const uint32_t fl_min = (uint32_t)d.fld_low + k0 * d.gs;
        if (fl_min > L) break;
        const uint32_t k_last = min(d.nfl - 1, k0 + kLanesFl - 1);
        const uint32_t n_i = min(n_pos, L - fl_min + 1);
        const uint32_t k = k0 + lane;
        const uint32_t fl = (uint32_t)d.fld_low + k * d.gs;
        for (int j = tid; j < kLanesFl * kNGC; j += kBlock) H[j] = 0;
        __syncthreads();
        for (uint32_t i = wv; i < n_i; i += kBlock / kWave) {
            const uint32_t e = i + fl - 1;
            if (k < d.nfl && e < L) atomicAdd(&H[lane * kNGC + gc_frac_sampled(G, L, step, i, e)], 1u);
        }
        __syncthreads();
        if (tid < kNGC) {
            double acc = Sacc[tid];
            for (uint32_t l = 0; l <= k_last - k0; ++l) acc += d.w[k0 + l] * (double)H[l * kNGC + tid];
            Sacc[tid] = acc;
        }
        __syncthreads();
    }
    if (tid < kNGC) Srow[tid] = Sacc[tid];
}

// expected[g] partial sums over kGcRows transcripts per block, fixed order (the loads of eight rows are in
// flight together; the adds stay in transcript order)
__global__ void __launch_bounds__(128) k_gc_expected_partial(BiasDev d, const double* __restrict__ eff_in,
                                                             const double* __restrict__ alpha) {
    __shared__ double contrib[kGcRows];
    const uint64_t t0 = (uint64_t)blockIdx.x * kGcRows;
    const uint32_t n = (uint32_t)min((uint64_t)kGcRows, d.M - t0);
    for (uint32_t r = threadIdx.x; r < n; r += 128) {
        const uint64_t t = t0 + r;
        const double a = alpha[t];
        const bool live = a >= kMinAlphaBias && unprocessed_len(d.ref_len[t], d.txp_eff[t]) > 0;
        contrib[r] = live ? a / eff_in[t] : 0.0;                            // :716
    }
    __syncthreads();
    const int g = threadIdx.x;
    if (g >= kNGC) return;
    const double* S = d.S + t0 * kNGC + g;
    double acc = 0.0;
    uint32_t r = 0;
    for (; r + 8 <= n; r += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = S[(uint64_t)(r + u) * kNGC];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += contrib[r + u] * v[u];
    }
    for (; r < n; ++r) acc += contrib[r] * S[(uint64_t)r * kNGC];
    d.gc_partial[(uint64_t)blockIdx.x * kNGC + g] = acc;
}

// :788-794  txomeGCNormFactor, gcPrior; ratio = gcCounts / (gcPrior + transcriptGCDist)
__global__ void __launch_bounds__(128) k_gc_norm(BiasDev d, uint32_t n_partials) {
    __shared__ double bins[kNGC];
    __shared__ double bc[2];
    const int g = threadIdx.x;
    if (g < kNGC) {
        double v = 1.0;                                                     // resize(101, 1.0) :667
        uint32_t b = 0;
        for (; b + 8 <= n_partials; b += 8) {
            double p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = d.gc_partial[(uint64_t)(b + u) * kNGC + g];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += p[u];
        }
        for (; b < n_partials; ++b) v += d.gc_partial[(uint64_t)b * kNGC + g];
        d.exp_gc[g] = v; bins[g] = v;
    }
    __syncthreads();
    if (g == 0) {
        double txome = 0.0;
        for (int i = 0; i < kNGC; ++i) txome += bins[i];
        const double pmass = 101.0;
        bc[0] = ((pmass / (d.read_gc_norm - pmass)) * txome) / 101.0;
        bc[1] = txome / d.read_gc_norm;
        d.scal[1] = bc[1];
    }
    __syncthreads();
    if (g < kNGC) d.ratio_gc[g] = (double)d.observed_gc[g] / (bc[0] + bins[g]);
}

// effLength = gcFactors.sum() * txomeGCNormFactor / readGCNormFactor (:907-909); one wave per transcript
__global__ void __launch_bounds__(kBlock) k_gc_efflen(BiasDev d, const double* eff_in, const double* __restrict__ alpha,
                                                      double* eff_out) {
    const uint64_t t = (uint64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (t >= d.M) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int unproc = unprocessed_len(d.ref_len[t], d.txp_eff[t]);
    const double e_in = eff_in[t];
    double eff_length = 0.0;
    if (alpha[t] >= kMinAlphaBias && unproc > 0) {
        const double* Srow = d.S + t * kNGC;
        double acc = d.ratio_gc[lane] * Srow[lane];
        if (lane + kWave < kNGC) acc += d.ratio_gc[lane + kWave] * Srow[lane + kWave];
        const double dot = wave_sum(acc);
        eff_length = (dot * d.prob_fwd + dot * d.prob_rc) * d.scal[1];      // gcFactors[start] += p*probFwd; [end] += p*probRC
    }
    if (lane == 0) {
        const bool corrected = unproc > 0 && eff_length > (double)unproc;
        eff_out[t] = corrected ? eff_length : e_in;
        d.corrected[t] = corrected ? 1 : 0;
    }
}

}  // namespace
}  // namespace sfgpu

using namespace sfgpu;

struct sfgpu_bias {
    BiasDev dev{};
    int status = 0;                        // 0 compute, 1 no mappings, 2 both models
    bool seq_on = false, gc_on = false;
    int32_t fld_low = 0, fld_high = 1;
    uint32_t n_partials = 0;
    hipStream_t last_stream = nullptr;
    unsigned long long* h_count = nullptr; // pinned
    std::vector<void*> bufs;
};

namespace sfgpu {
uint64_t bias_num_transcripts(const sfgpu_bias* b) { return b ? b->dev.M : 0; }
}

static void bias_free(sfgpu_bias* b) {
    if (!b) return;
    (void)hipDeviceSynchronize();
    for (void* p : b->bufs) if (p) pool_free(p);
    if (b->h_count) pinned_free(b->h_count);
    delete b;
}

template <typename T>
static int bias_alloc(sfgpu_bias* b, T** p, size_t bytes) {
    SF_HIP(pool_malloc(p, bytes ? bytes : 8));
    b->bufs.push_back(*p);
    return SFGPU_OK;
}

extern "C" {

int sfgpu_bias_create(sfgpu_bias** out, const sfgpu_bias_inputs* in, sfgpu_stream stream) {
    SF_REQUIRE(out && in, SFGPU_ERR_INVALID, "sfgpu_bias_create: null pointer");
    *out = nullptr;
    const bool seq_on = in->seq_bias != 0, gc_on = in->gc_bias != 0;
    SF_REQUIRE(seq_on || gc_on, SFGPU_ERR_INVALID, "sfgpu_bias_create: neither bias model is enabled");
    const int64_t n_map = in->num_fwd + in->num_rc;
    sfgpu_bias* b = new sfgpu_bias();
    b->seq_on = seq_on; b->gc_on = gc_on;
    b->dev.M = in->M;
    b->status = (n_map == 0) ? 1 : ((seq_on && gc_on) ? 2 : 0);          // :625-638
    hipStream_t st = as_stream(stream);
    b->last_stream = st;
    int rc = SFGPU_OK;
#define B_TRY(expr) do { rc = (expr); if (rc) { bias_free(b); return rc; } } while (0)
#define B_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); bias_free(b); return SFGPU_ERR_HIP; } } while (0)
#define B_REQ(cond, code, msg) do { if (!(cond)) { set_error("%s", msg); bias_free(b); return code; } } while (0)
    B_HIP(pinned_malloc(&b->h_count, 8));
    *b->h_count = 0;
    B_TRY(bias_alloc(b, &b->dev.exp_seq, kNKmer * 8));
    B_TRY(bias_alloc(b, &b->dev.exp_gc, 128 * 8));
    hipLaunchKernelGGL(k_fill, dim3(kNKmer / kBlock), dim3(kBlock), 0, st, b->dev.exp_seq, (uint32_t)kNKmer, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(kBlock), 0, st, b->dev.exp_gc, (uint32_t)kNGC, 1.0);
    if (b->status != 0 || in->M == 0) { B_HIP(hipStreamSynchronize(st)); *out = b; return SFGPU_OK; }

    B_REQ(in->d_seq && in->d_seq_off && in->d_ref_len && in->d_txp_eff_len && in->h_fl_counts && in->max_frag_len > 0,
          SFGPU_ERR_INVALID, "sfgpu_bias_create: null input");
    B_REQ(!seq_on || in->h_read_bias, SFGPU_ERR_INVALID, "sfgpu_bias_create: seq_bias needs h_read_bias");
    B_REQ(!gc_on || in->h_observed_gc, SFGPU_ERR_INVALID, "sfgpu_bias_create: gc_bias needs h_observed_gc");
    B_REQ(!gc_on || in->gc_size_samp >= 1, SFGPU_ERR_INVALID, "sfgpu_bias_create: gc_size_samp must be >= 1");
    B_REQ(!gc_on || in->gc_speed_samp >= 1, SFGPU_ERR_INVALID, "sfgpu_bias_create: gc_speed_samp must be >= 1");

    // EmpiricalDistribution::buildDistribution (src/EmpiricalDistribution.cpp:29-77) for vals = 0..n-1
    const uint32_t n = in->max_frag_len;
    const uint32_t* lens = in->h_fl_counts;
    double total = 0.0;
    for (uint32_t i = 0; i < n; ++i) total += lens[i];
    B_REQ(total > 0.0, SFGPU_ERR_INVALID, "sfgpu_bias_create: empty fragment length distribution");
    uint32_t cut = 0, table_len = 1;
    {
        double cum = 0.0;
        for (; cut < n; ++cut) {
            cum += lens[cut] / total;
            table_len = cut;
            if (cum > 1.0 - 1e-6) break;
        }
    }
    B_REQ(table_len > 0, SFGPU_ERR_INVALID, "sfgpu_bias_create: degenerate fragment length distribution");
    double kept = 0.0;
    for (uint32_t i = 0; i < cut; ++i) kept += lens[i];
    std::vector<float> cdf(table_len);
    {
        float run = 0.0f;
        for (uint32_t v = 0; v < table_len; ++v) {
            const float p = (float)(lens[v] / kept);
            run = (v == 0) ? p : run + p;                                  // float adds, :72-76
            cdf[v] = run;
        }
    }
    auto cdf_of = [&](uint32_t x) -> float { return x < table_len ? cdf[x] : 1.0f; };
    B_TRY(bias_alloc(b, const_cast<float**>(&b->dev.cdf), (size_t)table_len * 4));
    B_HIP(hipMemcpyAsync(const_cast<float*>(b->dev.cdf), cdf.data(), (size_t)table_len * 4, hipMemcpyHostToDevice, st));
    b->dev.cdf_size = table_len;
    b->dev.seq = in->d_seq; b->dev.seq_off = in->d_seq_off; b->dev.ref_len = in->d_ref_len; b->dev.txp_eff = in->d_txp_eff_len;
    b->dev.prob_fwd = (double)in->num_fwd / (double)n_map;                 // :640-641
    b->dev.prob_rc = (double)in->num_rc / (double)n_map;
    B_TRY(bias_alloc(b, &b->dev.scal, 16));
    B_TRY(bias_alloc(b, &b->dev.n_corrected, 8));
    B_TRY(bias_alloc(b, &b->dev.corrected, in->M));

    std::vector<double> w;
    if (seq_on) {
        uint32_t tot32 = 0;                                                // totalCount() sums in CountT = uint32 (ReadKmerDist.hpp:27-31)
        for (int i = 0; i < kNKmer; ++i) tot32 += in->h_read_bias[i];
        b->dev.read_norm = (double)tot32;
        B_TRY(bias_alloc(b, const_cast<uint32_t**>(&b->dev.read_bias), kNKmer * 4));
        B_HIP(hipMemcpyAsync(const_cast<uint32_t*>(b->dev.read_bias), in->h_read_bias, kNKmer * 4, hipMemcpyHostToDevice, st));
        B_TRY(bias_alloc(b, &b->dev.ratio_seq, kNKmer * 8));
    } else {
        // :669-684 the 0.005 / 0.995 quantiles of the FLD bound the fragment lengths considered
        bool first = false, second = false;
        for (uint32_t i = 0; i <= n - 1; ++i) {
            const float density = cdf_of(i);
            if (!first && density >= 0.005) { first = true; b->fld_low = (int32_t)i; }
            if (!second && density >= 0.995) { second = true; b->fld_high = (int32_t)i; }
        }
        B_REQ(b->fld_low >= 1, SFGPU_ERR_INVALID,
              "sfgpu_bias_create: the FLD's 0.005 quantile is 0 (the reference divides by the fragment length)");
        B_REQ((uint32_t)b->fld_high < kMaxFldHigh, SFGPU_ERR_RANGE, "sfgpu_bias_create: FLD 0.995 quantile >= 16000");
        const uint32_t gs = in->gc_speed_samp;
        double prev = (double)cdf_of(0);                                   // prevFLMass :739
        for (int64_t fl = b->fld_low; fl <= b->fld_high; fl += gs) {
            w.push_back((double)cdf_of((uint32_t)fl) - prev);
            prev = (double)cdf_of((uint32_t)fl);
        }
        b->dev.nfl = (uint32_t)w.size(); b->dev.fld_low = b->fld_low; b->dev.gs = gs;
        double gc_norm = 0.0;
        for (int i = 0; i < kNGC; ++i) gc_norm += in->h_observed_gc[i];   // :686
        b->dev.read_gc_norm = gc_norm;
        B_TRY(bias_alloc(b, const_cast<uint32_t**>(&b->dev.observed_gc), 128 * 4));
        B_HIP(hipMemcpyAsync(const_cast<uint32_t*>(b->dev.observed_gc), in->h_observed_gc, kNGC * 4, hipMemcpyHostToDevice, st));
        B_TRY(bias_alloc(b, &b->dev.ratio_gc, 128 * 8));
        b->n_partials = (uint32_t)((in->M + kGcRows - 1) / kGcRows);
        B_TRY(bias_alloc(b, &b->dev.gc_partial, (size_t)b->n_partials * kNGC * 8));
        B_TRY(bias_alloc(b, &b->dev.S, (size_t)in->M * kNGC * 8));
        if (b->dev.nfl) {
            B_TRY(bias_alloc(b, const_cast<double**>(&b->dev.w), w.size() * 8));
            B_HIP(hipMemcpyAsync(const_cast<double*>(b->dev.w), w.data(), w.size() * 8, hipMemcpyHostToDevice, st));
        }
        if (b->dev.nfl && in->gc_size_samp > 1) {
            // --gcSizeSamp: the interpolated counts of the reference, read from a per-base table that lives only here
            std::vector<uint64_t> h_off(in->M); std::vector<uint32_t> h_len(in->M);
            B_HIP(hipMemcpyAsync(h_off.data(), in->d_seq_off, in->M * 8, hipMemcpyDeviceToHost, st));
            B_HIP(hipMemcpyAsync(h_len.data(), in->d_ref_len, in->M * 4, hipMemcpyDeviceToHost, st));
            B_HIP(hipStreamSynchronize(st));
            uint64_t extent = 0;
            for (uint64_t t = 0; t < in->M; ++t) extent = std::max(extent, h_off[t] + h_len[t]);
            uint32_t* table = nullptr;
            B_HIP(pool_malloc(&table, (extent ? extent : 1) * 4));
            hipLaunchKernelGGL(k_gc_prefix, dim3((unsigned)in->M), dim3(kBlock), 0, st, in->d_seq, in->d_seq_off, in->d_ref_len, table);
            hipLaunchKernelGGL(k_gc_profile_sampled, dim3((unsigned)in->M), dim3(kBlock), 0, st, b->dev, table, in->gc_size_samp);
            hipError_t le = hipGetLastError();
            (void)hipStreamSynchronize(st);
            pool_free(table);
            B_HIP(le);
        } else if (b->dev.nfl) {
            uint32_t cap = (uint32_t)b->fld_high + 2048;
            if (cap < kStageMin) cap = kStageMin;
            b->dev.stage_cap = cap;
            const size_t lds = 816 + (size_t)kLanesFl * kNGC * 4 + kGcBlock * 4 + (size_t)cap * 4;
            B_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gc_profile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_gc_profile, dim3((unsigned)in->M), dim3(kGcBlock), lds, st, b->dev);
            B_HIP(hipGetLastError());
        } else {
            B_HIP(hipMemsetAsync(b->dev.S, 0, (size_t)in->M * kNGC * 8, st));
        }
    }
    B_HIP(hipStreamSynchronize(st));                                       // the host tables above are pageable
#undef B_TRY
#undef B_HIP
#undef B_REQ
    *out = b;
    return SFGPU_OK;
}

int sfgpu_bias_destroy(sfgpu_bias* b) { bias_free(b); return SFGPU_OK; }

int sfgpu_gc_prefix(const char* d_seq, const uint64_t* d_seq_off, const uint32_t* d_ref_len, uint64_t M,
                    uint32_t* d_gc_prefix, sfgpu_stream stream) {
    SF_REQUIRE(M == 0 || (d_seq && d_seq_off && d_ref_len && d_gc_prefix), SFGPU_ERR_INVALID, "sfgpu_gc_prefix: null pointer");
    SF_REQUIRE(M < (1ull << 31), SFGPU_ERR_RANGE, "sfgpu_gc_prefix: more than 2^31 transcripts");
    if (M == 0) return SFGPU_OK;
    hipLaunchKernelGGL(k_gc_prefix, dim3((unsigned)M), dim3(kBlock), 0, as_stream(stream), d_seq, d_seq_off, d_ref_len, d_gc_prefix);
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

int sfgpu_bias_update(sfgpu_bias* b, const double* d_eff_in, const double* d_alpha, double* d_eff_out,
                      sfgpu_bias_stats* stats, sfgpu_stream stream) {
    SF_REQUIRE(b && d_eff_in && d_alpha && d_eff_out, SFGPU_ERR_INVALID, "sfgpu_bias_update: null pointer");
    hipStream_t st = as_stream(stream);
    b->last_stream = st;
    const uint64_t M = b->dev.M;
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->status = b->status; stats->fld_low = b->fld_low; stats->fld_high = b->fld_high; }
    if (b->status != 0 || M == 0) {
        if (d_eff_out != d_eff_in && M) SF_HIP(hipMemcpyAsync(d_eff_out, d_eff_in, M * 8, hipMemcpyDeviceToDevice, st));
        if (stats) { SF_HIP(hipStreamSynchronize(st)); stats->n_uncorrected = 0; }
        return SFGPU_OK;
    }
    SF_REQUIRE(M < (1ull << 31), SFGPU_ERR_RANGE, "sfgpu_bias_update: more than 2^31 transcripts");
    const bool timing = env_timing();                 // per-kernel times of this update
    hipEvent_t ev[5] = {};
    int n_ev = 0;
    auto mark = [&]() { if (timing && n_ev < 5) { (void)hipEventCreate(&ev[n_ev]); (void)hipEventRecord(ev[n_ev], st); ++n_ev; } };
    mark();
    if (b->seq_on) {
        hipLaunchKernelGGL(k_fill, dim3(kNKmer / kBlock), dim3(kBlock), 0, st, b->dev.exp_seq, (uint32_t)kNKmer, 1.0);   // :652-653
        const unsigned nb = (unsigned)(M < (uint64_t)kSeqBlocks ? M : (uint64_t)kSeqBlocks);
        hipLaunchKernelGGL(k_seq_expected, dim3(nb), dim3(kBlock), 0, st, b->dev, d_eff_in, d_alpha);
        mark();
        hipLaunchKernelGGL(k_seq_norm, dim3(1), dim3(kBlock), 0, st, b->dev);
        mark();
        hipLaunchKernelGGL(k_seq_efflen, dim3((unsigned)M), dim3(kBlock), 0, st, b->dev, d_eff_in, d_alpha, d_eff_out);
    } else {
        hipLaunchKernelGGL(k_gc_expected_partial, dim3(b->n_partials), dim3(128), 0, st, b->dev, d_eff_in, d_alpha);
        mark();
        hipLaunchKernelGGL(k_gc_norm, dim3(1), dim3(128), 0, st, b->dev, b->n_partials);
        mark();
        hipLaunchKernelGGL(k_gc_efflen, dim3((unsigned)((M + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, st,
                           b->dev, d_eff_in, d_alpha, d_eff_out);
    }
    mark();
    SF_CHECK_LAUNCH();
    if (timing) {
        (void)hipEventSynchronize(ev[n_ev - 1]);
        float t[4] = {};
        for (int i = 0; i + 1 < n_ev; ++i) (void)hipEventElapsedTime(&t[i], ev[i], ev[i + 1]);
        fprintf(stderr, "[sfgpu bias] %s update: expected %.3f ms, norm %.3f ms, lengths %.3f ms\n", b->seq_on ? "seq" : "gc", t[0], t[1], t[2]);
        for (int i = 0; i < n_ev; ++i) (void)hipEventDestroy(ev[i]);
    }
    if (stats) {
        SF_HIP(hipMemsetAsync(b->dev.n_corrected, 0, 8, st));
        hipLaunchKernelGGL(k_count_corrected, dim3((unsigned)((M + kBlock - 1) / kBlock < 512 ? (M + kBlock - 1) / kBlock : 512)), dim3(kBlock), 0, st, b->dev);
        SF_HIP(hipMemcpyAsync(b->h_count, b->dev.n_corrected, 8, hipMemcpyDeviceToHost, st));
        SF_HIP(hipStreamSynchronize(st));
        stats->n_corrected = *b->h_count;
        stats->n_uncorrected = M - stats->n_corrected;
    }
    return SFGPU_OK;
}

int sfgpu_bias_expected(sfgpu_bias* b, double* h_expected_seq, double* h_expected_gc) {
    SF_REQUIRE(b, SFGPU_ERR_INVALID, "sfgpu_bias_expected: null handle");
    SF_HIP(hipStreamSynchronize(b->last_stream));
    if (h_expected_seq) SF_HIP(hipMemcpy(h_expected_seq, b->dev.exp_seq, kNKmer * 8, hipMemcpyDeviceToHost));
    if (h_expected_gc) SF_HIP(hipMemcpy(h_expected_gc, b->dev.exp_gc, kNGC * 8, hipMemcpyDeviceToHost));
    return SFGPU_OK;
}

}  // extern "C"
