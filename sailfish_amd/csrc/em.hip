// em.hip -- CollapsedEMOptimizer on the device (rows a6-a12 of SURVEY.md section 8).
//
// Replaces  src/CollapsedEMOptimizer.cpp:711-893 (optimize), :224-281 (EMUpdate_),
//           :288-369 (VBEMUpdate_), :36-44 (truncateCountVector), :849-861 (convergence).
//
// Design (the reference is TBB parallel_for over an AoS vector of heap vectors with a CAS
// loop per contribution):
//   * classes are a flat CSR (rowptr u32, ids u32, counts u32) that stays resident in HBM;
//   * the per-class aux weights of the reference are NOT materialised: w_i is proportional to
//     1/effLen_i and its per-class normaliser cancels in v_i/denom (:762-768, :256-270), so the
//     sweep gathers x_t = alpha_t/effLen_t (EM) or expTheta_t/effLen_t (VBEM) -- one M-vector
//     rebuilt each iteration -- and streams only labels and counts: 4 B per nonzero instead of
//     12 (algorithmic bytes B_iter' = 4L + 8C + 48M);
//   * an iteration is [sweep over classes] -> [per-transcript update: convergence test, swap,
//     next x]; the loop condition of :820 is evaluated ON THE DEVICE from a small state block,
//     every kernel turns into a no-op once it holds, so the host can enqueue iterations in
//     hipGraph chunks and still stop at exactly the reference's iteration;
//   * all reductions that feed decisions (n_active, sum alpha, alphaSum) are two-stage and
//     order-deterministic; the only nondeterministic order is the f64 atomic accumulation into
//     alphaOut, as in the reference's CAS loop (:70-79).
#include <algorithm>
#include "bias.h"
#include "common.h"
#include "primitives.h"
#include "sampling.h"

#include <atomic>
#include <map>
#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace sfgpu {

constexpr int kEmBlock = 256;
constexpr int kMaxPartials = 1024;          // blocks of the per-transcript kernels
constexpr uint32_t kDoneMark = 0xFFFFFFFFu;
constexpr double kTiny = 4.9406564584124654e-324;   // numeric_limits<double>::denorm_min() (:33-34)
constexpr double kPriorAlpha = 0.01;        // :786
constexpr double kMinAlpha = 1e-8;          // :810
constexpr double kCheckCutoff = 1e-2;       // :811

struct EmState {
    uint32_t it_a;                 // iterations completed; written by update, read by sweep
    uint32_t it_b;                 // iteration in flight (or kDoneMark); written by sweep, read by update
    uint32_t notconv[2];           // per iteration parity: some gated transcript moved by > tol
    uint32_t gated[2];             // per iteration parity: some transcript passed the gate
    unsigned long long n_active;
    double alpha_sum;
    // the FUSED iteration (k_sweep_lds<.., .., true>: the update of iteration it - 1 runs at the head of sweep it, see there).  A launch
    // reads itv[par] and writes itv[par ^ 1] (par: the launch's parity, a kernel argument), the update of iteration u sets
    // notconv3[u % 3], the stop test of iteration it reads notconv3[(it - 1) % 3], the launch clears the slot of the NEXT update:
    // no word is both read and written by the blocks of one launch.
    uint32_t itv[2];
    uint32_t notconv3[3];
};

// loop condition of :820, negated:  stop  <=>  it >= minIter && (it >= maxIter || converged)
__device__ __forceinline__ bool em_stop(uint32_t it, const EmState* s, uint32_t min_iter, uint32_t max_iter) {
    if (it < min_iter) return false;
    if (it >= max_iter) return true;
    return it > 0 && s->notconv[(it - 1) & 1] == 0;
}

__device__ __forceinline__ bool em_stop3(uint32_t it, const EmState* s, uint32_t min_iter, uint32_t max_iter) {
    if (it < min_iter) return false;
    if (it >= max_iter) return true;
    return it > 0 && s->notconv3[(it - 1) % 3] == 0;
}

// the stop test of the host loop, posted where the host can read it without a copy command: `mirror` is pinned host memory
// (see em_poll_start).  low word: iterations completed, high word: 1 = the loop has ended.
__global__ void k_post_state(const EmState* st, uint32_t min_iter, uint32_t max_iter, unsigned long long* mirror, int fused) {
    const uint32_t it = st->it_a;
    const bool stop = fused ? em_stop3(it, st, min_iter, max_iter) : em_stop(it, st, min_iter, max_iter);
    *mirror = (unsigned long long)it | ((unsigned long long)(stop ? 1u : 0u) << 32);
}

// psi(x), x > 0: the recurrence psi(x) = psi(x + 10) - sum_{k<10} 1/(x + k) for x < 10, then the asymptotic
// series through B_14 (boost::math::digamma at :303, :314 in the reference; |err| ~ 1e-15).
// The ten reciprocals are added as ONE fraction (pairwise n/d merges: every term is positive, so nothing
// cancels) -- a single f64 division instead of up to ten dependent ones (a wavefront always holds some
// low-abundance transcript, so the old loop ran all ten rounds for everybody).
__host__ __device__ __forceinline__ double digamma_pos(double x) {
    double r = 0.0;
    if (x < 10.0) {
        const double a0 = x, a1 = x + 1.0, a2 = x + 2.0, a3 = x + 3.0, a4 = x + 4.0,
                     a5 = x + 5.0, a6 = x + 6.0, a7 = x + 7.0, a8 = x + 8.0, a9 = x + 9.0;
        double n01 = a0 + a1, d01 = a0 * a1, n23 = a2 + a3, d23 = a2 * a3, n45 = a4 + a5, d45 = a4 * a5,
               n67 = a6 + a7, d67 = a6 * a7, n89 = a8 + a9, d89 = a8 * a9;
        const double n03 = n01 * d23 + n23 * d01, d03 = d01 * d23;
        const double n47 = n45 * d67 + n67 * d45, d47 = d45 * d67;
        const double n07 = n03 * d47 + n47 * d03, d07 = d03 * d47;
        const double n09 = n07 * d89 + n89 * d07, d09 = d07 * d89;
        r = -(n09 / d09);
        x += 10.0;
    }
    double inv = 1.0 / x, inv2 = inv * inv;
    double s = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 * (1.0 / 240.0
             - inv2 * (1.0 / 132.0 - inv2 * (691.0 / 32760.0 - inv2 * (1.0 / 12.0)))))));
    return r + log(x) - 0.5 * inv - s;
}

// VBEM's x_t = exp(psi(a) - c) / effLen (:300-320) for the FUSED sweep, whose 64-register budget has no room for digamma_pos's
// pairwise tree, a log and an exp.  Same recurrence and series as digamma_pos, but exp(log(y)) is y itself:
//   exp(psi(a) - c) = y exp(-(1 / (2y) + s(y) + r + c)),   y = a (+ 10 below 10),   r = sum_{k<10} 1 / (a + k),
// with r gathered as ONE fraction term by term (all terms positive: nothing cancels; ten numbers below 20 multiply to < 1e13).
// One exp, no log, ~10 live doubles; agrees with exp(digamma_pos(a) - c) to a few ulp.
__device__ __forceinline__ double vb_x_lean(double a, double c, double len) {
    double y = a, q = c;
    if (a < 10.0) {
        double n = 1.0, d = a;
#pragma unroll
        for (int k = 1; k < 10; ++k) { const double t = a + (double)k; n = n * t + d; d = d * t; }
        q += n / d;
        y = a + 10.0;
    }
    const double inv = 1.0 / y, inv2 = inv * inv;
    const double s = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 * (1.0 / 240.0
                   - inv2 * (1.0 / 132.0 - inv2 * (691.0 / 32760.0 - inv2 * (1.0 / 12.0)))))));
    q += 0.5 * inv + s;
    return y * exp(-q) / len;
}

// The same for the kernels that are short of registers AND of issue slots (the fused sweep, the persistent loop: every thread of a
// window slot evaluates this once per iteration, ~9 wavefronts per tile on a dependent chain).  One division instead of three
// (n / d, 1 / y and 1 / effLen share the reciprocal of d y effLen) and an exp of its own: k = rint(t log2 e), r = t - k ln 2 in two
// pieces, the Taylor polynomial to r^13 (|r| <= ln 2 / 2: the remainder is below 4e-18), ldexp -- no special cases: the argument lies
// in (-200, 0] (alpha >= the prior 0.01, c = psi(M prior + numMapped) < 50).  Its 15 constants sit in constant memory: as literals
// the compiler keeps them in VGPR pairs across the persistent loop and spills them.  ~70 instructions against ~150; agrees with
// exp(digamma_pos(a) - c) / len to ~1e-13 at alpha near the prior (the fraction of ten terms carries the rounding; the tests hold every loop to 1e-9).
struct VbConsts { double l2e, ln2_hi, ln2_lo, c[12]; double s[7]; };
__device__ __constant__ VbConsts kVb = {
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
     1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5},
    {1.0 / 12.0, 691.0 / 32760.0, 1.0 / 132.0, 1.0 / 240.0, 1.0 / 252.0, 1.0 / 120.0, 1.0 / 12.0}};
__device__ __forceinline__ double vb_x_fast(double a, double c, double len) {
    double y = a, n = 0.0, d = 1.0;
    if (a < 10.0) {
        n = 1.0; d = a;
#pragma unroll
        for (int k = 1; k < 10; ++k) { const double t = a + (double)k; n = fma(n, t, d); d = d * t; }
        y = a + 10.0;
    }
    const double dy = d * y;
    const double R = 1.0 / (dy * len);                                 // the one division
    const double inv = (d * len) * R;                                  // 1 / y
    const double rlen = dy * R;                                        // 1 / effLen
    const double inv2 = inv * inv;
    const VbConsts& K = kVb;
    double s = fma(-inv2, K.s[0], K.s[1]);                              // inv2 (1/12 - inv2 (1/120 - inv2 (1/252 - inv2 (1/240 - inv2 (1/132 - inv2 (691/32760 - inv2 / 12))))))
    s = fma(-inv2, s, K.s[2]); s = fma(-inv2, s, K.s[3]); s = fma(-inv2, s, K.s[4]); s = fma(-inv2, s, K.s[5]); s = fma(-inv2, s, K.s[6]);
    s = s * inv2;
    // t = -(c + n / d + 1 / (2 y) + s)   (n / d = n y effLen R)
    const double t = -(c + fma(n * y, len * R, fma(0.5, inv, s)));
    const double kf = rint(t * K.l2e);
    double r = fma(-kf, K.ln2_hi, t);
    r = fma(-kf, K.ln2_lo, r);
    double p = K.c[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = fma(p, r, K.c[i]);
    p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p * y, (int)kf) * rlen;
}

// The same arithmetic for the PERSISTENT loop's head (round 6).  There every tile is in the same phase at the same time (a tile needs its
// neighbours' sums of the step before: lockstep), so the head's instructions are not hidden under another block's LDS phases and every
// one of them is on the step's critical path -- and the constants of vb_x_fast arrived through 58 v_readlane per evaluation (22 doubles
// loaded once, far more than the kernel's SGPRs hold: spilled into VGPR lanes).  Here a constant is two s_mov_b32 with literals right where
// it is used (scalar ALU, no memory, no spill, never hoisted: the asm is volatile), and the two divisions are v_rcp_f64 + two Newton steps
// (the operands are far from the denormals: a >= the prior, effLen >= 1) instead of the IEEE sequence.
template <uint64_t B> __device__ __forceinline__ double kd_bits() {
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((uint32_t)B));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((uint32_t)(B >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
#define SF_KD(x) kd_bits<__builtin_bit_cast(uint64_t, (double)(x))>()
__device__ __forceinline__ double fast_rcp(double x) {                 // 1 / x to ~1 ulp for normal x
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double vb_x_head(double a, double c, double len) {
    double y = a, n = 0.0, d = 1.0;
    if (a < 10.0) {
        n = 1.0; d = a;
#pragma unroll
        for (int k = 1; k < 10; ++k) { const double t = a + (double)k; n = fma(n, t, d); d = d * t; }
        y = a + 10.0;
    }
    const double dy = d * y;
    const double R = fast_rcp(dy * len);                               // the one reciprocal
    const double inv = (d * len) * R;                                  // 1 / y
    const double rlen = dy * R;                                        // 1 / effLen
    const double inv2 = inv * inv;
    double s = fma(-inv2, SF_KD(1.0 / 12.0), SF_KD(691.0 / 32760.0));   // inv2 (1/12 - inv2 (1/120 - inv2 (1/252 - inv2 (1/240 - inv2 (1/132 - inv2 (691/32760 - inv2 / 12))))))
    s = fma(-inv2, s, SF_KD(1.0 / 132.0)); s = fma(-inv2, s, SF_KD(1.0 / 240.0)); s = fma(-inv2, s, SF_KD(1.0 / 252.0));
    s = fma(-inv2, s, SF_KD(1.0 / 120.0)); s = fma(-inv2, s, SF_KD(1.0 / 12.0));
    s = s * inv2;
    const double t = -(c + fma(n * y, len * R, fma(0.5, inv, s)));      // -(c + n / d + 1 / (2 y) + s)   (n / d = n y effLen R)
    const double kf = rint(t * SF_KD(1.4426950408889634074));
    double r = fma(-kf, SF_KD(6.93147180369123816490e-01), t);
    r = fma(-kf, SF_KD(1.90821492927058770002e-10), r);
    double p = SF_KD(1.0 / 6227020800.0);
    p = fma(p, r, SF_KD(1.0 / 479001600.0)); p = fma(p, r, SF_KD(1.0 / 39916800.0)); p = fma(p, r, SF_KD(1.0 / 3628800.0));
    p = fma(p, r, SF_KD(1.0 / 362880.0)); p = fma(p, r, SF_KD(1.0 / 40320.0)); p = fma(p, r, SF_KD(1.0 / 5040.0)); p = fma(p, r, SF_KD(1.0 / 720.0));
    p = fma(p, r, SF_KD(1.0 / 120.0)); p = fma(p, r, SF_KD(1.0 / 24.0)); p = fma(p, r, SF_KD(1.0 / 6.0)); p = fma(p, r, 0.5);
    p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p * y, (int)kf) * rlen;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// deterministic block sum; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* lds /* kEmBlock/kWave */) {
    v = wave_sum(v);
    int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) lds[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < kEmBlock / kWave; ++i) t += lds[i];
    __syncthreads();
    return t;
}

// sum of the nb per-block partials, same order in every block -> identical everywhere
__device__ __forceinline__ double sum_partials(const double* partials, int nb, double* lds) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += kEmBlock) v += partials[i];
    double t = block_sum(v, lds);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = t;
    __syncthreads();
    return bc;
}

__global__ void k_clamp_len(uint64_t M, const double* __restrict__ len, double* __restrict__ lenc) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) { double l = len[t]; lenc[t] = (l <= 1.0) ? 1.0 : l; }   // :738
}

// The sweep skips NaN terms (EM, :269) and expTheta <= 0 terms (VBEM, :344, :356).  Both tests depend on the transcript only,
// so they are applied ONCE, where x_t is produced, and the sweep's inner loops carry no guard: x_t = 0 is the skipped term.
template <bool VB> __device__ __forceinline__ double sweep_x(double v) {
    if (VB) return (v > 0.0) ? v : 0.0;
    return (v == v) ? v : 0.0;
}

// after a bias recompute (:824-840): x for the next sweep from the current alpha and the new lengths
__global__ void k_x_from_alpha(uint64_t M, const double* __restrict__ alpha, const double* __restrict__ lenc,
                               double* __restrict__ x) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) x[t] = sweep_x<false>(alpha[t] / lenc[t]);
}

__global__ void k_alpha_partials(uint64_t M, const double* __restrict__ alpha, double* partials) {
    __shared__ double lds[kEmBlock / kWave];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) v += alpha[t];
    double s = block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// device-private copy of the counts: 31 bits of count, bit 31 = the class is a singleton
// (cperm: the plan's own class order, see em_renumber -- class c of the plan is the caller's class cperm[c]; rowptr is the plan's)
__global__ void k_narrow_counts(uint64_t C, const uint64_t* __restrict__ c64, const uint32_t* __restrict__ rowptr,
                                uint32_t* __restrict__ c32, unsigned int* overflow, const uint32_t* __restrict__ cperm) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    uint64_t v = c64[cperm ? cperm[c] : c];
    if (v >= 0x80000000ull) atomicOr(overflow, 1u);
    // a class without members never comes out of the builder (addGroup is only called with hits); the tile plan
    // bounds classes per tile through nonzeros, so an empty class in a caller-made CSR is refused, not planned
    if (rowptr[c + 1] <= rowptr[c]) atomicOr(overflow, 2u);
    c32[c] = (uint32_t)v | ((rowptr[c + 1] - rowptr[c] == 1) ? 0x80000000u : 0u);
}

// :774-782  every transcript that appears in a class is active
__global__ void k_mark_active(uint64_t L, const uint32_t* __restrict__ ids, double* alpha_out) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < L) alpha_out[ids[j]] = 1.0;
}

__global__ void k_count_active(uint64_t M, const double* __restrict__ flags, double* partials) {
    __shared__ double lds[kEmBlock / kWave];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock)
        v += (flags[t] > 0.0) ? 1.0 : 0.0;
    double s = block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// :800-803 alpha_t = active ? (1/n_active) * numMapped : 0 ; also x for iteration 0 (EM) or
// the partial sums of alpha (VBEM)
template <bool VB>
__global__ void k_init_alpha(uint64_t M, double* alpha, double* alpha_out, double* x, const double* __restrict__ lenc,
                             double total_frags, const double* act_partials, double* sum_partials_out, int nb,
                             EmState* st) {
    __shared__ double lds[kEmBlock / kWave];
    double n_act = sum_partials(act_partials, nb, lds);
    double scale = 1.0 / n_act;
    double local = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = (alpha_out[t] > 0.0) ? scale * total_frags : 0.0;
        alpha[t] = a; alpha_out[t] = 0.0;
        if (VB) local += a; else x[t] = sweep_x<false>(a / lenc[t]);
    }
    if (VB) { double s = block_sum(local, lds); if (threadIdx.x == 0) sum_partials_out[blockIdx.x] = s; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->it_a = 0; st->it_b = 0; st->notconv[0] = st->notconv[1] = 0; st->gated[0] = st->gated[1] = 0;
        st->itv[0] = st->itv[1] = 0; st->notconv3[0] = st->notconv3[1] = st->notconv3[2] = 0;
        st->n_active = (unsigned long long)n_act; st->alpha_sum = 0.0;
    }
}

// VBEMUpdate_ prologue (:300-320): expTheta_t = alpha_t > denorm_min ? exp(psi(alpha_t) - psi(sum alpha)) : 0,
// folded with the 1/effLen factor of the aux weight.
__global__ void k_vb_prepare(uint64_t M, const double* __restrict__ alpha, double* __restrict__ x,
                             const double* __restrict__ lenc, const double* sum_partials_in, int nb,
                             const EmState* st, int force) {
    if (!force && st->it_b == kDoneMark) return;
    __shared__ double lds[kEmBlock / kWave];
    double asum = sum_partials(sum_partials_in, nb, lds);
    double log_norm = digamma_pos(asum);
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = alpha[t];
        x[t] = (a > kTiny) ? sweep_x<true>(exp(digamma_pos(a) - log_norm) / lenc[t]) : 0.0;
    }
}

// ---- tiled E-step sweep with LDS accumulators --------------------------------------------------
// Classes are stored in the canonical order (first id ascending), so a run of consecutive classes
// touches a narrow band of transcripts.  Once per problem the CSR is re-packed into the stream the
// sweep reads:
//   * a tile is the run of classes whose first nonzero falls into one tile_nnz-sized bucket of the
//     CSR (nnz-balanced); its window is the band [lo, lo+span) of transcripts it touches (<= kWin);
//   * one 32-bit word per nonzero:  [31 null][29 single][28..16 class index in the tile]
//     [15..0 window offset], tiles padded to 8 words so every lane fetches its 8 consecutive words
//     with two 16-byte loads;
//   * members outside the window ("escapes": labels spanning far-apart transcripts) are null in the
//     stream and sit in a per-tile side list (transcript id, class index) that takes global gathers
//     and global atomics -- any input is handled, locality only buys speed.
// Per tile, one 1024-thread block (EMUpdate_ :236-277 for ~8000 nonzeros at a time; measured on
// MI355X: per-block fixed costs -- window staging, barriers, publishing -- dominate below ~4000):
//   0  stage x[lo, lo+span) in LDS, zero the LDS accumulators
//   A  nonzero-parallel: v = x[member] (LDS gather); denominators den[class] += v with ds_add_f64.
//      A lane owns 8 CONSECUTIVE nonzeros, so lanes of one instruction hit different classes (no
//      same-address serialisation) and a lane first sums its run of equal classes in registers
//   B  class-parallel  : den[class] <- count/denom   (:260-264; singletons carry the full count :275)
//   C  nonzero-parallel: window accumulator acc[member] += v * den[class]   (ds_add_f64)
//   D  publish the window with plain coalesced stores; the update folds the windows per transcript
// HBM sees 4 bytes per nonzero + 8 per class per iteration and the common path has no global atomic.
#ifndef SFGPU_TILE_NNZ
#define SFGPU_TILE_NNZ 7800
#endif
constexpr int kTileNnzMax = 1 << 17;         // nonzeros a tile may hold when its class count allows (see the plan in sfgpu_em_create)
constexpr int kTileNnz = SFGPU_TILE_NNZ;     // largest CSR bucket that defines a tile (<= 8191 classes per tile);
                                             // the bucket actually used is sized per problem so that the tiles fill
                                             // the chip in whole rounds (tile_nnz_for)
constexpr uint64_t kRenumberMinClasses = 4096;   // smaller problems are not worth a second plan
constexpr int kRenumberHops = 5;           // class hops of the label propagation behind the plan's own transcript order (em_renumber)
constexpr int kEscSlots = 128;              // LDS accumulator for escaped members (per tile)
constexpr int kWin = 1023;                   // LDS window (transcripts): 2 x 8 KB.  1023, not 1024 (round 6): a window slot is a 10-bit field in the
                                             // persistent loop's class records (em_persist.h) and 1023 is the null slot, whose x is 0
#ifndef SFGPU_SWEEP_BLOCK
#define SFGPU_SWEEP_BLOCK 1024
#endif
constexpr int kSweepBlock = SFGPU_SWEEP_BLOCK;
#ifndef SFGPU_PER_LANE
#define SFGPU_PER_LANE 8
#endif
constexpr int kPerLane = SFGPU_PER_LANE;     // consecutive stream words per lane (8: two 16-byte loads)
#ifndef SFGPU_SWEEP_REGCHUNKS
#define SFGPU_SWEEP_REGCHUNKS 1
#endif
constexpr int kRegChunks = SFGPU_SWEEP_REGCHUNKS;
#ifndef SFGPU_SWEEP_CNTAHEAD
#define SFGPU_SWEEP_CNTAHEAD 4
#endif
constexpr int kCntAhead = SFGPU_SWEEP_CNTAHEAD;    // class counts per thread requested ahead of phase B
// A NULL word (padding, or a member that escaped the window) names a slot and a class of its own -- window slot kWin, which
// holds x = 0, and class kTileNnz, whose count/denom is 0 -- so the sweep treats it like any other word: no test per word.
constexpr uint32_t kNullBit = 0x80000000u, kSingle = 0x20000000u;
constexpr uint32_t kNull = kNullBit | ((uint32_t)kTileNnz << 16) | (uint32_t)kWin;
static_assert(kTileNnz < (1 << 13) && kWin < (1 << 16), "stream word: 13-bit class, 16-bit window slot");
static_assert(kTileNnz <= 8191, "class index field is 13 bits");
static_assert(kEscSlots == 128, "the escape accumulator's hash takes 7 bits");

// what the host wants to know about a plan before it goes on: the largest class count of a tile (checks[0]) and whether some tile's
// first position lies before that of the tile in front of it (checks[1] = 1: the cover lists then need a sort, see k_cov_fill); the
// three scans over the tiles (window slots, stream words, escapes) by ONE block; and all of it posted into pinned host memory by
// the same kernel (four copy commands and three device-wide scans of 513 numbers before: ~45 us of a plan)
__global__ void __launch_bounds__(1024)
k_tile_scans(uint32_t n_tiles, const uint32_t* __restrict__ tile_c0, const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ span,
             const uint32_t* __restrict__ len8, const uint32_t* __restrict__ nesc, uint64_t* off, uint64_t* s0, uint64_t* esc0,
             unsigned long long* host) {
    __shared__ unsigned long long wsum[3][16];
    __shared__ unsigned long long carry[3];
    __shared__ unsigned int most_s, mono_s;
    if (threadIdx.x < 3) carry[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) { most_s = 0u; mono_s = 0u; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned int most = 0u, mono = 0u;
    for (uint32_t base = 0; base <= n_tiles; base += 1024u) {           // (entry n_tiles holds the totals: the inputs' sentinels there are 0)
        const uint32_t i = base + threadIdx.x;
        const bool in = i < n_tiles;
        unsigned long long v[3] = {in ? span[i] : 0u, in ? len8[i] : 0u, in ? nesc[i] : 0u};
        if (in) {
            const uint32_t nc = tile_c0[i + 1] - tile_c0[i];
            most = nc > most ? nc : most;
            if (i > 0 && span[i]) for (uint32_t U = i; U-- > 0;) { if (!span[U]) continue; if (tile_lo[U] > tile_lo[i]) mono = 1u; break; }
        }
        unsigned long long inc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned long long x = v[k];
            for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(x, o, 64); if ((int)lane >= o) x += y; }
            inc[k] = x;
            if (lane == 63u) wsum[k][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned long long pre = carry[k];
            for (uint32_t w = 0; w < wave; ++w) pre += wsum[k][w];
            const unsigned long long ex = pre + inc[k] - v[k];
            uint64_t* out = k == 0 ? off : (k == 1 ? s0 : esc0);
            if (i <= n_tiles) out[i] = ex;
        }
        __syncthreads();
        if (threadIdx.x == 1023u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { unsigned long long t = carry[k]; for (int w = 0; w < 16; ++w) t += wsum[k][w]; carry[k] = t; }
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned int m = __shfl_down(most, o, 64); most = m > most ? m : most; mono |= __shfl_down(mono, o, 64); }
    if (lane == 0) { atomicMax(&most_s, most); if (mono) mono_s = 1u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        host[1] = carry[0]; host[2] = carry[1]; host[3] = carry[2];   // P, S, E (window slots, stream words, escapes)
        host[5] = (unsigned long long)most_s | ((unsigned long long)mono_s << 32);
    }
}

// tile i = classes [tile_c0[i], tile_c0[i+1]) : those with rowptr[c] in [i*tile_nnz, (i+1)*tile_nnz)
__global__ void k_tile_plan(uint64_t C, uint32_t n_tiles, uint32_t tile_nnz, const uint32_t* __restrict__ rowptr,
                            uint32_t* tile_c0) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_tiles) return;
    uint64_t target = (uint64_t)i * tile_nnz;
    uint64_t lo = 0, hi = C;                         // first class with rowptr[c] >= target
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (rowptr[mid] >= target) hi = mid; else lo = mid + 1; }
    tile_c0[i] = (uint32_t)lo;
}

// window of tile i: [lo, lo + span) with lo = smallest member and span <= kWin; also the padded
// stream length of the tile and its number of escapes (members beyond the window)
__global__ void __launch_bounds__(kEmBlock)
k_tile_window(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ tile_c0,
              uint32_t* tile_lo, uint32_t* tile_span, uint32_t* tile_len8, uint32_t* tile_nesc) {
    __shared__ uint32_t rmin[kEmBlock / kWave], rmax[kEmBlock / kWave];
    __shared__ uint32_t lo_s;
    __shared__ unsigned int esc_s;
    uint32_t b = rowptr[tile_c0[blockIdx.x]], e = rowptr[tile_c0[blockIdx.x + 1]];
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (uint32_t j = b + threadIdx.x; j < e; j += kEmBlock) { uint32_t v = ids[j]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    for (int o = kWave / 2; o > 0; o >>= 1) {
        uint32_t v = __shfl_down(mn, o, kWave); mn = v < mn ? v : mn;
        uint32_t w = __shfl_down(mx, o, kWave); mx = w > mx ? w : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { rmin[threadIdx.x / kWave] = mn; rmax[threadIdx.x / kWave] = mx; }
    if (threadIdx.x == 0) esc_s = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kEmBlock / kWave; ++i) mn = rmin[i] < mn ? rmin[i] : mn;
        if (e == b) mn = 0;
        lo_s = mn;
    }
    __syncthreads();
    // the window ends at the largest member that lies within kWin of the smallest (round 5: a far member -- an escape -- used to
    // stretch every window it touched to the full kWin slots: more slots to publish and fold, more overlapping tiles)
    const uint32_t lo = lo_s;
    unsigned int mine = 0;
    uint32_t top = 0;
    for (uint32_t j = b + threadIdx.x; j < e; j += kEmBlock) {
        const uint32_t d = ids[j] - lo;
        if (d >= (uint32_t)kWin) ++mine; else top = d > top ? d : top;
    }
    for (int o = kWave / 2; o > 0; o >>= 1) { const uint32_t w = __shfl_down(top, o, kWave); top = w > top ? w : top; }
    if ((threadIdx.x & (kWave - 1)) == 0) rmax[threadIdx.x / kWave] = top;
    if (mine) atomicAdd(&esc_s, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kEmBlock / kWave; ++i) top = rmax[i] > top ? rmax[i] : top;
        tile_lo[blockIdx.x] = lo; tile_span[blockIdx.x] = (e == b) ? 0u : top + 1u;
        tile_len8[blockIdx.x] = (e - b + kPerLane - 1) / kPerLane * kPerLane;
        tile_nesc[blockIdx.x] = esc_s;
        if (blockIdx.x == gridDim.x - 1) { tile_span[gridDim.x] = 0; tile_len8[gridDim.x] = 0; tile_nesc[gridDim.x] = 0; }   // scan sentinels
    }
}

// write the tile's stream words (one lane per class walks its label) and its escape list
__global__ void __launch_bounds__(kEmBlock)
k_fill_stream(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ tile_c0,
              const uint32_t* __restrict__ tile_lo, const uint64_t* __restrict__ tile_s0,
              const uint64_t* __restrict__ tile_esc0, uint32_t* stream, uint32_t* esc_id, uint32_t* esc_cls,
              const uint32_t* __restrict__ inv) {          // inv: `ids` are positions of the plan's transcript order; escapes name transcripts
    __shared__ unsigned int esc_cursor;
    if (threadIdx.x == 0) esc_cursor = 0;
    __syncthreads();
    const uint32_t c0 = tile_c0[blockIdx.x], c1 = tile_c0[blockIdx.x + 1];
    const uint32_t lo = tile_lo[blockIdx.x];
    const uint64_t s0 = tile_s0[blockIdx.x], s1 = tile_s0[blockIdx.x + 1], e0 = tile_esc0[blockIdx.x];
    const uint32_t j0 = rowptr[c0], n = rowptr[c1] - j0;
    for (uint32_t c = c0 + threadIdx.x; c < c1; c += kEmBlock) {
        uint32_t b = rowptr[c], k = rowptr[c + 1] - b;
        uint32_t tag = ((c - c0) << 16) | (k == 1 ? kSingle : 0u);
        for (uint32_t m = 0; m < k; ++m) {
            uint32_t t = ids[b + m], d = t - lo, w;
            if (d < (uint32_t)kWin) w = tag | d;
            else {
                unsigned int idx = atomicAdd(&esc_cursor, 1u);
                esc_id[e0 + idx] = inv ? inv[t] : t; esc_cls[e0 + idx] = tag;      // tag = class index << 16 | single flag
                w = kNull;
            }
            stream[s0 + (b - j0) + m] = w;
        }
    }
    for (uint64_t p = s0 + n + threadIdx.x; p < s1; p += kEmBlock) stream[p] = kNull;
}

// ---- GATHER form: one kernel per tile makes the COMPACT class-major stream and sorts the tile's nonzeros by window slot ----------
// slot16[s0 + p] = window slot of nonzero p of the tile (kWin: a member outside the window, or padding -- its x reads as 0),
// chdr[(s0 + p) / 8] = class of the chunk's first nonzero | bit 16 + k for every nonzero k >= 1 of the chunk that starts the next
// class; escapes go to the tile's side list as in k_fill_stream.  The same pass classifies every nonzero for the
// transcript-major copy: key = singleton class << 11 | window slot (kEscBin: an escape), and the tile's nonzeros are sorted by
// that key with a STABLE counting sort -- wavefront w owns the w-th contiguous piece of the tile and a histogram row of its own,
// the column scan of the rows gives every wavefront its first position per key, and lanes of one step that share a key are
// ranked by lane (ballots over the 12 key bits) -- so entries of one slot stay in class order whatever the schedule, and the
// sums of phase C are formed in the same order on every rank and in every plan.  A nonzero finds its class without a walk
// per class: a step's 64 consecutive positions meet at most 64 class starts, which the lanes load side by side and search with
// shuffles.  One pass of ~30 us on cfg3 in place of a key pass, a 3-pass radix sort of all nonzeros and a search per tile (0.4 ms).
constexpr int kBuildBlock = 256, kBuildWaves = kBuildBlock / kWave, kBuildPerThread = 13;      // (round 5: 4 rows of bins = 49 KB, three blocks per CU -- 512 threads and 8 rows were one block per CU, two rounds of tiles at two wavefronts per SIMD)
constexpr uint32_t kEscBin = 0xC00u, kBuildBins = kEscBin + 1u;          // keys 0..0x3FF, 0x800..0xBFF and the escape bin
static_assert(kBuildBlock * kBuildPerThread >= (int)kBuildBins, "bins per thread in the column scan");
static_assert(kWin <= 0x400, "key: 10 bits of window slot under the singleton bit");
__global__ void __launch_bounds__(kBuildBlock)
k_tile_build(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ tile_c0,
             const uint32_t* __restrict__ tile_lo, const uint64_t* __restrict__ tile_s0, const uint64_t* __restrict__ tile_esc0,
             uint16_t* slot16, uint32_t* chdr, uint32_t* esc_id, uint32_t* esc_cls, const uint32_t* __restrict__ inv,
             uint32_t* tmp, uint32_t* kv, uint32_t* idx, uint32_t* tile_in, uint32_t* chunks) {
    extern __shared__ uint32_t hist[];                          // [kBuildWaves][kBuildBins]
    __shared__ unsigned int esc_cursor;
    __shared__ uint32_t wsum[kBuildWaves];
    const uint32_t t = blockIdx.x, c0 = tile_c0[t], c1 = tile_c0[t + 1], lo = tile_lo[t];
    const uint64_t s0 = tile_s0[t], e0 = tile_esc0[t];
    const uint32_t len = (uint32_t)(tile_s0[t + 1] - s0);       // padded to whole chunks
    const uint32_t j0 = rowptr[c0], n = rowptr[c1] - j0;
    for (uint32_t i = threadIdx.x; i < kBuildWaves * kBuildBins; i += kBuildBlock) hist[i] = 0u;
    if (threadIdx.x == 0) esc_cursor = 0u;
    __syncthreads();
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const uint32_t seg = ((len + kBuildWaves - 1) / kBuildWaves + (kWave - 1)) & ~(uint32_t)(kWave - 1);
    const uint32_t p_begin = wave * seg < len ? wave * seg : len, p_end = p_begin + seg < len ? p_begin + seg : len;
    uint32_t* h = hist + wave * kBuildBins;

    // ---- a: classify, write the compact stream, count
    uint32_t cfirst = c0;                                       // class of position p0 (wave-uniform)
    if (p_begin < n) {
        uint32_t a = c0, b = c1;                                // last class with rowptr[c] <= j0 + p_begin
        while (b - a > 1u) { const uint32_t mid = (a + b) >> 1; if (rowptr[mid] <= j0 + p_begin) a = mid; else b = mid; }
        cfirst = a;
    }
    for (uint32_t p0 = p_begin; p0 < p_end; p0 += kWave) {
        if (p0 < n && cfirst + 1u < c1 && rowptr[cfirst + 1u] <= j0 + p0) ++cfirst;
        const uint32_t p = p0 + lane, pos = j0 + p;
        const uint32_t ia = cfirst + lane;
        const uint32_t A = ia <= c1 ? rowptr[ia] : 0xFFFFFFFFu;             // start of class cfirst + lane ...
        const uint32_t B = ia + 1u <= c1 ? rowptr[ia + 1u] : 0xFFFFFFFFu;   // ... and of the one behind it
        uint32_t r = 0;                                         // how many of B_0 <= B_1 <= ... are <= pos (at most 63: see above)
#pragma unroll
        for (uint32_t step = kWave / 2; step; step >>= 1) { const uint32_t probe = __shfl(B, (int)(r + step - 1u), kWave); if (probe <= pos) r += step; }
        const uint32_t a_r = __shfl(A, (int)r, kWave), b_r = __shfl(B, (int)r, kWave);
        const bool valid = p < n;
        const uint32_t crel = cfirst + r - c0;
        const bool single = b_r - a_r == 1u;
        uint32_t slot = (uint32_t)kWin;
        if (valid) {
            const uint32_t tr = ids[pos], d = tr - lo;
            uint32_t key = kEscBin;
            if (d < (uint32_t)kWin) { slot = d; key = (single ? 0x800u : 0u) | d; }
            else {
                const unsigned int q = atomicAdd(&esc_cursor, 1u);
                esc_id[e0 + q] = inv ? inv[tr] : tr; esc_cls[e0 + q] = (crel << 16) | (single ? kSingle : 0u);
            }
            tmp[pos] = (key << 16) | crel;
            atomicAdd(&h[key], 1u);
        }
        const uint64_t starts = __ballot(valid && a_r == pos);
        if (p < len) {
            slot16[s0 + p] = (uint16_t)slot;
            if ((lane & 7u) == 0u) chdr[(s0 + p) >> 3] = (valid ? crel : 0u) | ((uint32_t)((starts >> (lane + 1u)) & 0x7Full) << 17);
        }
        cfirst = __shfl(cfirst + r, kWave - 1, kWave);
        if (cfirst > c1) cfirst = c1;
    }
    __syncthreads();

    // ---- b: first position of every (wavefront, key): column scan of the rows, then a scan over the keys
    uint32_t tot[kBuildPerThread], sum = 0u;
#pragma unroll
    for (int k = 0; k < kBuildPerThread; ++k) {
        const uint32_t b = threadIdx.x * kBuildPerThread + k;
        uint32_t run = 0u;
        if (b < kBuildBins) for (int w = 0; w < kBuildWaves; ++w) { const uint32_t c = hist[w * kBuildBins + b]; hist[w * kBuildBins + b] = run; run += c; }
        tot[k] = run; sum += run;
    }
    uint32_t inc = sum;
    for (int o = 1; o < kWave; o <<= 1) { const uint32_t v = __shfl_up(inc, o, kWave); if ((int)lane >= o) inc += v; }
    if (lane == kWave - 1) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - sum;
    for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
    for (int k = 0; k < kBuildPerThread; ++k) {
        const uint32_t b = threadIdx.x * kBuildPerThread + k;
        if (b < kBuildBins) {
            for (int w = 0; w < kBuildWaves; ++w) hist[w * kBuildBins + b] += base;
            if (b == kEscBin) {                                 // everything in front of the escape bin lies in the window
                tile_in[t] = base; idx[t] = j0; chunks[t] = (base + 7u) >> 3;
                if (t == gridDim.x - 1) chunks[gridDim.x] = 0u;                    // the scan's sentinel
            }
        }
        base += tot[k];
    }
    __syncthreads();

    // ---- c: place (stable: wavefronts in order of their pieces, steps in order, lanes in order)
    for (uint32_t p0 = p_begin; p0 < p_end; p0 += kWave) {
        const uint32_t p = p0 + lane;
        const uint32_t v = p < n ? tmp[j0 + p] : (kEscBin << 16);
        const uint32_t key = v >> 16;
        const bool in = key != kEscBin;
        uint64_t peers = __ballot(in);
#pragma unroll
        for (uint32_t b = 0; b < 12u; ++b) { const bool bit = (key >> b) & 1u; const uint64_t m = __ballot(bit); peers &= bit ? m : ~m; }
        if (in) {
            const uint32_t first = h[key];
            kv[j0 + first + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull))] = v;
            if ((peers >> lane) == 1ull) h[key] = first + (uint32_t)__popcll(peers);    // the last lane of the group moves the cursor
        }
    }
}

// one (transcript, slot) pair per window entry; sorted by transcript this is the cover list that the
// per-transcript update walks to fold the tiles' partial sums in a fixed order
__global__ void __launch_bounds__(kEmBlock)
k_cover_pairs(const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_span,
              const uint64_t* __restrict__ tile_off, uint64_t* keys, uint32_t* vals, const uint32_t* __restrict__ inv) {
    uint32_t lo = tile_lo[blockIdx.x], span = tile_span[blockIdx.x];
    uint64_t off = tile_off[blockIdx.x];
    for (uint32_t d = threadIdx.x; d < span; d += kEmBlock) { keys[off + d] = inv ? (uint64_t)inv[lo + d] : (uint64_t)lo + d; vals[off + d] = (uint32_t)(off + d); }
}

// FUSED iteration: the tiles whose windows overlap a tile's own, in tile order.  A window is an interval of transcripts and the
// tiles follow the classes' canonical order (first id ascending), so a tile's `lo` never decreases: the windows that overlap tile
// T's belong to a few tiles around T.  With them in registers, the thread of window slot i finds every sum the previous sweep
// published for its transcript -- partial[off' + (lo + i - lo')] for each overlapping tile -- without a cover list: ONE round trip
// for everything the update needs (the cover-list form needed three).  Entry: {lo', span', off', tile'}; flags: 1 = more than kNbMax
// overlapping tiles somewhere, 2 = `lo` decreases somewhere (a caller-made class table out of canonical order): such plans keep the
// two-kernel iteration.
constexpr int kNbMax = 6;
constexpr uint32_t kNbByList = 0xFFFFFFFFu;               // TileDesc::nb_n of a tile that more than kNbMax tiles overlap
// everything a sweep block needs to know about its tile, in ONE 192-byte record: three scalar loads issued together instead of a
// dozen words from nine arrays (each array a pointer from the kernel arguments first, then the word: the compiler serialises them
// into a chain of dependent scalar round trips at the head of every launch)
struct alignas(64) TileDesc {
    uint32_t c0, nc, lo, span;
    uint64_t s0; uint32_t n8, n_esc;
    uint64_t e0, off;
    uint64_t qb; uint32_t np, nm;                          // GATHER: the transcript-major copy (pure chunks, mixed chunks)
    uint32_t pr, nb_n, nb_before, f0;                      // nb_*: FUSED, the overlapping tiles below; f0: first far slot (em_persist.h)
    uint4 e[kNbMax];                                       // {lo', span', off', tile'}
    uint32_t nf, ov0, n_ov, pad1;                          // nf: distinct far transcripts of the tile = its far slots; ov0 / n_ov: overflow chunks of its long classes (em_persist.h)
};
static_assert(sizeof(TileDesc) == 192, "three 64-byte scalar loads");
__global__ void k_tile_desc(uint32_t n_tiles, const uint32_t* __restrict__ tile_c0, const uint32_t* __restrict__ tile_lo,
                            const uint32_t* __restrict__ tile_span, const uint64_t* __restrict__ tile_s0, const uint64_t* __restrict__ tile_esc0,
                            const uint64_t* __restrict__ tile_off, const uint64_t* __restrict__ tile_qb, const uint32_t* __restrict__ tile_np,
                            const uint32_t* __restrict__ tile_pr, TileDesc* td) {
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    if (T >= n_tiles) return;
    TileDesc r{};
    r.c0 = tile_c0[T]; r.nc = tile_c0[T + 1] - r.c0; r.lo = tile_lo[T]; r.span = tile_span[T];
    r.s0 = tile_s0[T]; r.n8 = (uint32_t)(tile_s0[T + 1] - r.s0);
    r.e0 = tile_esc0[T]; r.n_esc = (uint32_t)(tile_esc0[T + 1] - r.e0);
    r.off = tile_off[T];
    if (tile_qb) { r.qb = tile_qb[T]; r.np = tile_np[T]; r.nm = (uint32_t)((tile_qb[T + 1] - r.qb - 16ull * r.np) >> 5); r.pr = tile_pr[T]; }
    td[T] = r;
}
// the transcript-major copy's fields of the tile records, written when that copy's offsets are known (it is laid out on the plan's side
// stream, next to the cover lists: sfgpu_em_create)
__global__ void k_tile_desc_csc(uint32_t n_tiles, const uint64_t* __restrict__ tile_qb, const uint32_t* __restrict__ tile_np, const uint32_t* __restrict__ tile_pr, TileDesc* td) {
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    if (T >= n_tiles) return;
    const uint64_t qb = tile_qb[T]; const uint32_t np = tile_np[T];
    td[T].qb = qb; td[T].np = np; td[T].nm = (uint32_t)((tile_qb[T + 1] - qb - 16ull * np) >> 5); td[T].pr = tile_pr[T];
}
__global__ void k_nb_table(uint32_t n_tiles, const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_span,
                           const uint64_t* __restrict__ tile_off, TileDesc* td, unsigned int* flags) {
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    if (T >= n_tiles) return;
    uint4 ent[kNbMax];
    for (int j = 0; j < kNbMax; ++j) ent[j] = make_uint4(0u, 0u, 0u, 0u);
    uint32_t n = 0, n_before = 0;
    const uint32_t lo = tile_lo[T], span = tile_span[T];
    if (span) {
        // below: walk down while a window that starts there could still reach lo (empty tiles -- no classes, span 0 -- are skipped)
        uint32_t first = T;
        for (uint32_t U = T; U-- > 0;) {
            const uint32_t sl = tile_span[U], ll = tile_lo[U];
            if (!sl) continue;
            if (ll > lo) { atomicOr(flags, 2u); break; }
            if ((uint64_t)ll + kWin <= lo) break;
            if ((uint64_t)ll + sl > lo) first = U;
        }
        for (uint32_t U = first; U < n_tiles; ++U) {
            if (U == T) { n_before = n; continue; }
            const uint32_t sl = tile_span[U], ll = tile_lo[U];
            if (!sl) continue;
            if (U > T && ll < lo) { atomicOr(flags, 2u); break; }
            if (U > T && (uint64_t)ll >= (uint64_t)lo + span) break;
            if ((uint64_t)ll + sl > lo && (uint64_t)ll < (uint64_t)lo + span) {
                if (n == (uint32_t)kNbMax) { n = kNbByList; atomicOr(flags, 1u); break; }       // too many for the record: this tile goes by the cover list
                ent[n++] = make_uint4(ll, sl, (uint32_t)tile_off[U], U);
            }
        }
        if (n != kNbByList && n_before > n) n_before = n;
    }
    td[T].nb_n = n; td[T].nb_before = n_before;
    for (int j = 0; j < kNbMax; ++j) td[T].e[j] = ent[j];
}
// FUSED: the first two entries of the cover list of every far member's transcript, inline (0xFFFFFFFF: none; more than two entries:
// the rest comes through cov_ptr / cov_pos) -- a far member's alpha' then costs its thread one look-up less than two dependent ones
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
__global__ void k_esc_slots(uint64_t E, const uint32_t* __restrict__ esc_id, const uint32_t* __restrict__ cov_ptr,
                            const uint32_t* __restrict__ cov_pos, uint2* esc_slots) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const uint32_t t = esc_id[i], k0 = cov_ptr[t], k1 = cov_ptr[t + 1];
    esc_slots[i] = make_uint2(k1 > k0 ? cov_pos[k0] : kNoSlot, k1 > k0 + 1u ? cov_pos[k0 + 1u] : kNoSlot);
}
// FUSED: the cover list of the transcript at POSITION pos of the plan's order ([k0, k1) of cov_pos), and the positions no window holds
// (inactive, or only ever a far member): the update reaches those through the list
__global__ void k_cov_by_pos(uint64_t M, const uint32_t* __restrict__ inv, const uint32_t* __restrict__ cov_ptr, uint2* cov2, uint32_t* list, uint32_t* n) {
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= M) return;
    const uint32_t t = inv ? inv[pos] : (uint32_t)pos;
    const uint2 r = make_uint2(cov_ptr[t], cov_ptr[t + 1]);
    cov2[pos] = r;
    if (r.x == r.y) list[atomicAdd(n, 1u)] = (uint32_t)pos;
}
// out[i] = map[in[i]] (the far members' transcripts as positions) ; out[pos] = in[inv[pos]] (a per-transcript vector in the plan's order)
__global__ void k_map_u32(uint64_t n, const uint32_t* __restrict__ in, const uint32_t* __restrict__ map, uint32_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = map[in[i]];
}
__global__ void k_gather_f64(uint64_t M, const double* __restrict__ in, const uint32_t* __restrict__ inv, double* out) {
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < M) out[pos] = in[inv[pos]];
}
__global__ void k_scatter_f64(uint64_t M, const double* __restrict__ in, const uint32_t* __restrict__ inv, double* out) {
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < M) out[inv[pos]] = in[pos];
}

// ---- the cover lists WITHOUT a sort (round 4).  A window is an interval of positions and, when the tiles' `lo` never decreases (the
// canonical class order: checked by k_tile_mono, read back with the plan's sizes), the tiles that hold a position are found by walking
// back from a tile until a window that starts kWin or more positions earlier: the rank of tile T in the list of position lo_T + d is
// the number of earlier tiles whose window reaches it.  Counts per transcript by atomics, a scan, and every slot writes itself into
// its place: the lists come out in tile order, exactly as the stable sort by transcript leaves them (5 small kernels against the
// ~20 of a merge sort of 283 k pairs: ~0.1 ms of cfg3's plan).
__global__ void k_tile_mono(uint32_t n_tiles, const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_span, unsigned int* flag) {
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    if (T >= n_tiles || T == 0 || !tile_span[T]) return;
    for (uint32_t U = T; U-- > 0;) {
        if (!tile_span[U]) continue;                          // (empty tiles carry no window)
        if (tile_lo[U] > tile_lo[T]) atomicOr(flag, 1u);
        break;
    }
}
__global__ void __launch_bounds__(kEmBlock)
k_cov_count(const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_span, const uint32_t* __restrict__ inv, uint32_t* cnt) {
    const uint32_t lo = tile_lo[blockIdx.x], span = tile_span[blockIdx.x];
    for (uint32_t d = threadIdx.x; d < span; d += kEmBlock) atomicAdd(&cnt[inv ? inv[lo + d] : lo + d], 1u);
}
__global__ void __launch_bounds__(kEmBlock)
k_cov_fill(const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_span, const uint64_t* __restrict__ tile_off,
           const uint32_t* __restrict__ inv, const uint32_t* __restrict__ cov_ptr, uint32_t* cov_pos, uint32_t* pub_pos) {
    const uint32_t T = blockIdx.x, lo = tile_lo[T], span = tile_span[T];
    const uint64_t off = tile_off[T];
    for (uint32_t d = threadIdx.x; d < span; d += kEmBlock) {
        const uint32_t pos = lo + d;
        uint32_t rank = 0;
        for (uint32_t U = T; U-- > 0;) {                      // (the same walk for every thread of the tile: uniform loads)
            const uint32_t sl = tile_span[U], ll = tile_lo[U];
            if (!sl) continue;
            if ((uint64_t)ll + kWin <= lo) break;             // no earlier window reaches this tile's
            if ((uint64_t)ll + sl > pos) ++rank;
        }
        const uint32_t k = cov_ptr[inv ? inv[pos] : pos] + rank;
        cov_pos[k] = (uint32_t)(off + d);
        pub_pos[off + d] = k;
    }
}

// cov_pos[k] = window slot that sorts to position k  ->  pub_pos[slot] = k
__global__ void k_invert_perm(uint64_t P, const uint32_t* __restrict__ cov_pos, uint32_t* pub_pos) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < P) pub_pos[cov_pos[k]] = (uint32_t)k;
}

__global__ void k_cover_ptr(uint64_t M, uint64_t P, const uint64_t* __restrict__ sorted_keys, uint32_t* cov_ptr) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > M) return;
    uint64_t lo = 0, hi = P;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (sorted_keys[mid] >= t) hi = mid; else lo = mid + 1; }
    cov_ptr[t] = (uint32_t)lo;
}

// ---- transcripts renumbered by co-occurrence (em_renumber) -------------------------------------------------------------------
// The window form wants a class's members within kWin transcript ids of each other.  A transcriptome whose isoforms are not
// adjacent in the index (accession order, a shuffled FASTA) makes every member but the first an escape -- correct, ~10x
// slower.  When a first plan finds many escapes, the PLAN (not the caller's vectors) gets a transcript order of its own:
// key_t = the smallest transcript id reachable from t over a few class hops (label propagation), transcripts sorted by
// (key, id) -- a gene family becomes one contiguous run -- and classes sorted by their smallest new position.  The sweep
// stages its window through `inv` (position -> transcript); x, alpha and everything the caller sees keep the caller's order.
__global__ void k_renum_iota(uint64_t n, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}
// one hop of the label propagation, Jacobi style: the keys of the previous hop are only READ (key_in), the minima go to
// key_out (a copy of key_in before the launch), so a hop's result is a pure function of its input -- not of which lanes ran
// first (reading and atomicMin-ing one array made the plan's transcript order, and with it the summation order and the
// bootstrap's class order, depend on the schedule)
__global__ void k_lp_min(uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                         const uint32_t* __restrict__ key_in, uint32_t* key_out) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const uint32_t b = rowptr[c], e = rowptr[c + 1];
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t j = b; j < e; ++j) { const uint32_t k = key_in[ids[j]]; m = k < m ? k : m; }
    for (uint32_t j = b; j < e; ++j) if (m < key_in[ids[j]]) atomicMin(&key_out[ids[j]], m);
}
__global__ void k_renum_pair_keys(uint64_t n, const uint32_t* __restrict__ hi, uint64_t* keys, uint32_t* vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ((uint64_t)hi[i] << 32) | (uint64_t)i; vals[i] = (uint32_t)i; }
}
__global__ void k_class_min_pos(uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                                const uint32_t* __restrict__ perm, uint32_t* ckey) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t j = rowptr[c]; j < rowptr[c + 1]; ++j) { const uint32_t v = perm[ids[j]]; m = v < m ? v : m; }
    ckey[c] = m;
}
__global__ void k_renum_lens(uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ cperm, uint32_t* lens) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) lens[c] = rowptr[cperm[c] + 1] - rowptr[cperm[c]]; else if (c == C) lens[c] = 0;
}
__global__ void k_renum_csr(uint64_t C, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ ids,
                           const uint32_t* __restrict__ cperm, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ off64,
                           uint32_t* rowptr2, uint32_t* vids) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > C) return;
    rowptr2[c] = (uint32_t)off64[c];
    if (c == C) return;
    const uint32_t src = rowptr[cperm[c]], k = rowptr[cperm[c] + 1] - src;
    uint32_t* dst = vids + off64[c];
    for (uint32_t m = 0; m < k; ++m) dst[m] = perm[ids[src + m]];
}
__global__ void k_renum_scatter(uint64_t n, const uint32_t* __restrict__ src, const uint32_t* __restrict__ where, uint32_t* dst) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[where[i]] = src[i];
}

// ---- the TRANSCRIPT-major copy of a tile's nonzeros (round 3): phase C as a gather ------------------------------------------
// Phase C of the sweep adds x_t * (count / denom)_class into the window slot of every nonzero: 9.3 M random f64 LDS atomics
// per sweep on cfg3, and the LDS atomic unit is what bounds the phase (0.29 - 0.37 cycles per lane and CU at random addresses,
// 10.8 of the sweep's 21 us).  A random f64 LDS READ costs 0.085 cycles (tools/probes/lds_atomic_probe.hip).  So the tile's
// nonzeros are kept a second time, sorted by (singleton class or not, window slot), 16 bits per nonzero -- the class index in
// the tile -- in CHUNKS of 8 (one 16-byte load).  A run of one slot is ~47 entries long, so 85 % of the chunks are PURE (one
// slot): such a chunk is 8 class indices plus, in a separate array, its slot (| bit 15: its classes are singletons, which add
// their count, not x_t times it); the thread reads count / denom of the 8 classes, adds them up in a tree and hands acc[slot]
// ONE sum.  The chunks that straddle slots are MIXED: 8 class indices + 8 slots, combined run by run.  Pure chunks come first
// in a tile, mixed ones behind them, so that a wavefront runs one of the two loops, not both.  ~2 700 atomics per tile instead
// of 18 000; within a slot the entries are in class order (stable sort), so the sums are formed in the same order on every rank
// and in every plan.  Costs ~2.3 more bytes per nonzero and iteration, and a sort of the nonzeros when the plan is made.
constexpr uint32_t kCscSingleBit = 0x8000u;
// the sorted nonzeros of tile t: kv[idx[t] .. idx[t] + tile_in[t]), key << 16 | class in the tile (k_tile_build)
__global__ void __launch_bounds__(kEmBlock)
k_csc_pure(const uint32_t* __restrict__ kv, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ tile_in, const uint64_t* __restrict__ cb, uint32_t* pure) {
    const uint32_t t = blockIdx.x, a = idx[t], e = a + tile_in[t];
    const uint64_t g0 = cb[t];
    const uint32_t n = (uint32_t)(cb[t + 1] - g0);
    for (uint32_t j = threadIdx.x; j < n; j += kEmBlock) {
        const uint32_t first = a + 8u * j, last = (first + 7u < e) ? first + 7u : e - 1u;
        pure[g0 + j] = (kv[first] >> 16) == (kv[last] >> 16) ? 1u : 0u;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) pure[cb[t + 1]] = 0u;       // the scan's sentinel
}
// byte offset of tile t's chunks in the copy (pure chunks 16 B, mixed 32 B), its pure chunks and the rank of its first pure chunk
__global__ void k_csc_offsets(uint32_t n_tiles, const uint64_t* __restrict__ cb, const uint64_t* __restrict__ ps,
                              uint64_t* tile_qb, uint32_t* tile_np, uint32_t* tile_pr) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    const uint64_t g = cb[t], p = ps[g];
    tile_qb[t] = 16ull * p + 32ull * (g - p);
    tile_pr[t] = (uint32_t)p;
    tile_np[t] = (t == n_tiles) ? 0u : (uint32_t)(ps[cb[t + 1]] - p);
}
__global__ void __launch_bounds__(kEmBlock)
k_csc_write(const uint32_t* __restrict__ kv, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ tile_in, const uint64_t* __restrict__ cb,
            const uint64_t* __restrict__ ps, const uint64_t* __restrict__ tile_qb, const uint32_t* __restrict__ tile_np,
            unsigned char* csc, uint16_t* slot0, uint32_t null_cls) {
    const uint32_t t = blockIdx.x, a = idx[t], e = a + tile_in[t];
    const uint64_t g0 = cb[t], p0 = ps[g0];
    const uint32_t n = (uint32_t)(cb[t + 1] - g0), np = tile_np[t];
    unsigned char* base = csc + tile_qb[t];
    for (uint32_t j = threadIdx.x; j < n; j += kEmBlock) {
        const uint32_t first = a + 8u * j, last = (first + 7u < e) ? first + 7u : e - 1u;
        uint32_t cl[8], sl[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t i = first + k;
            const uint32_t v = kv[i <= last ? i : last], key = v >> 16;
            cl[k] = i <= last ? (v & 0xFFFFu) : null_cls;                     // padding: the plan's null class (count / denom = 0)
            sl[k] = (key & 0x7FFu) | ((key & 0x800u) ? kCscSingleBit : 0u);
        }
        const uint64_t g = g0 + j, pr = ps[g];
        const uint4 cls4 = make_uint4(cl[0] | (cl[1] << 16), cl[2] | (cl[3] << 16), cl[4] | (cl[5] << 16), cl[6] | (cl[7] << 16));
        if ((kv[first] >> 16) == (kv[last] >> 16)) {
            reinterpret_cast<uint4*>(base)[pr - p0] = cls4;
            slot0[pr] = (uint16_t)sl[0];
        } else {
            uint4* m = reinterpret_cast<uint4*>(base + 16ull * np) + 2ull * ((g - pr) - (g0 - p0));
            m[0] = cls4;
            m[1] = make_uint4(sl[0] | (sl[1] << 16), sl[2] | (sl[3] << 16), sl[4] | (sl[5] << 16), sl[6] | (sl[7] << 16));
        }
    }
}

struct SweepArgs {
    // (what the head of the kernel needs comes first: the kernel arguments are fetched 64 bytes at a time)
    const TileDesc* td; EmState* st; uint32_t min_iter, max_iter, par, first;
    uint32_t sharded;                                                    // FUSED inside the sharded loop: alpha' is the all-reduced vector alone (the fold ran before the all-reduce)
    uint32_t null_cls;                                                   // GATHER: the class index of the transcript-major copy's padding (the plan's largest class count of a tile)
    const uint32_t* stream; const uint32_t* chdr;                        // GATHER: 16-bit window slots, 8 per chunk, + one header word per chunk
    const double* x; const uint32_t* counts;
    double* part_a; double* part_b;                                      // FUSED: the sweeps' window sums, by sweep parity
    double* aout_a; double* aout_b; double* aout_c;                      // FUSED: what the escapes add, by sweep mod 3
    const double* lenc; double* alpha;
    const uint32_t* inv;                                                 // window position -> transcript (null: the caller's order)
    const uint32_t* esc_id; const uint32_t* esc_cls; const uint2* esc_slots;
    const unsigned char* csc; const uint16_t* csc_slot0;                  // transcript-major copy (null: phase C scatters with atomics)
    const uint32_t* pub_pos;                                             // window slot -> index in `partial`
    double* alpha_out; double* partial;
    double* tsum;                                                        // VBEM inside optimize(): what each tile added (else null)
    const uint32_t* cov_ptr; const uint2* cov2; const uint32_t* cov_pos; const uint32_t* unc; const uint32_t* unc_n; int check_mode;
    double* tmax;                                                        // [2][n_tiles][waves]: largest relative change a wavefront saw
    double tol; double log_norm; uint64_t M;
    unsigned long long* dbg;                                             // SFGPU_X_STAMP builds: [tile][16] phase time stamps (dev)
    unsigned long long* post;                                            // FUSED, streamed loop: pinned host word, "updates done | ended << 32" at the head of every launch
    unsigned long long post_tag;                                         // ... | 1 << 33 | the run's number << 34 (the host ignores words of another run)
};
#ifdef SFGPU_X_STAMP
#define SF_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define SF_STAMP(k) do { } while (0)
#endif

// GATHER (the default since round 3): phase C reads the transcript-major copy (above) and phase A a COMPACT class-major stream --
// the classes of a tile follow each other, so a nonzero only has to name its window slot (16 bits); a header word per chunk of 8
// holds the class of the chunk's first nonzero and a bit for every nonzero that starts the next class.  Half the bytes of the
// 32-bit words, and half the instructions per nonzero in phase A (no class field to extract and compare: 6.4 -> ~4 us on cfg3).
// !GATHER: the 32-bit stream words [null | single | class | slot] and one LDS atomic per nonzero in phase C (rounds 1 - 2).
// FUSED (round 4; inside optimize(), GATHER plans, EM or VBEM with the constant normaliser): ONE kernel per iteration.  The launch
// of sweep `it` first runs the per-transcript update of iteration it - 1 -- which the unfused loop runs as k_update between
// the sweeps -- in two pieces:
//   * every tile derives the x of ITS window straight from the previous sweep's window sums (the fold of k_update, the same
//     additions in the same order: alphaOut[t] + the sums of the tiles whose windows hold t, in tile order, + the prior; then
//     psi / exp / 1 / effLen), and the x of its far members the same way: nothing has to travel through a global x vector, so
//     nothing needs a grid-wide hand-over.  The sums are published SLOT-major (tile's offset + slot: coalesced, no index), and
//     the tiles that overlap a tile's window sit in its NbTable (k_nb_table), so the thread of a slot addresses all of them
//     without a look-up: descriptors, then ONE round trip of operands;
//   * the update itself (:849-861: the gate, the relative change, alpha <- alpha') and the zeroing of the NEXT sweep's escape
//     accumulator are done for transcript t by the thread of t's HOME slot (the lowest tile whose window holds t); transcripts no
//     window holds (inactive, or only ever far members) come from a list, at the END of the launch, off the critical path.
// The window sums ping-pong between two arrays (sweep `it` reads it - 1's while it writes its own); the escapes' accumulator
// rotates through three (read it - 1's, add into it's, zero it + 1's).  The convergence flag of iteration it - 1 is complete
// when this launch ends, so the loop ends one launch later than the unfused loop would notice: the stop test at the head of launch
// it + 1 sees it, returns, and alpha holds exactly what the reference's loop leaves (the sweep `it` ran for nothing: ~17 us, once).
// What it removes from every iteration: k_update's launch and its latency chain (6.2 us on cfg3) and one kernel boundary.
template <bool VB, bool GATHER, bool FUSED = false>
__global__ void __launch_bounds__(kSweepBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))      // two 1024-thread blocks per CU: 64 VGPRs
k_sweep_lds(SweepArgs a) {
    // the tile's descriptor does not depend on the loop state: it is requested together with the state (pinned below)
    const TileDesc td = a.td[blockIdx.x];
    const uint32_t c0 = td.c0, nc = td.nc, lo = td.lo, span = td.span, n8 = td.n8, n_esc = td.n_esc;
    const uint64_t s0 = td.s0, e0 = td.e0, off = td.off;
#ifdef SFGPU_X_STAMP
    const unsigned long long t_entry = wall_clock64();
#endif
    // The loop state, read as ONE block of independent words (the flag the stop test needs is picked in registers: indexing memory with
    // `it` would be a second, dependent round trip), and every descriptor above pinned in front of the early exit: left alone, the
    // compiler sinks those loads below the branch and the kernel starts with seven dependent scalar round trips instead of two.
    EmState* st = a.st;
    const uint32_t s_it_a = st->it_a, s_itv0 = st->itv[0], s_itv1 = st->itv[1];
    const uint32_t s_nc0 = st->notconv[0], s_nc1 = st->notconv[1];
    const uint32_t s_n30 = st->notconv3[0], s_n31 = st->notconv3[1], s_n32 = st->notconv3[2];
    asm volatile("" :: "s"(c0), "s"(nc), "s"(lo), "s"(span), "s"(s0), "s"(n8), "s"(e0), "s"(n_esc), "s"(off));
    if constexpr (GATHER) asm volatile("" :: "s"(td.qb), "s"(td.np), "s"(td.nm), "s"(td.pr));
    if constexpr (FUSED) {
        asm volatile("" :: "s"(td.nb_n), "s"(td.nb_before), "s"(td.e[0].x), "s"(td.e[0].y), "s"(td.e[0].z), "s"(td.e[1].x), "s"(td.e[1].y), "s"(td.e[1].z),
                     "s"(td.e[2].x), "s"(td.e[2].y), "s"(td.e[2].z), "s"(td.e[3].x), "s"(td.e[3].y), "s"(td.e[3].z),
                     "s"(td.e[4].x), "s"(td.e[4].y), "s"(td.e[4].z), "s"(td.e[5].x), "s"(td.e[5].y), "s"(td.e[5].z));
        static_assert(kNbMax == 6, "the pin above names six entries");
    }
    uint32_t n_unc_v = 0;                                  // FUSED: length of the list of positions no window holds (needed at the END: requested here)
    if constexpr (FUSED) { n_unc_v = *a.unc_n; asm volatile("" :: "s"(n_unc_v)); }
    asm volatile("" :: "s"(s_it_a), "s"(s_itv0), "s"(s_itv1), "s"(s_nc0), "s"(s_nc1), "s"(s_n30), "s"(s_n31), "s"(s_n32));
    const uint32_t it = FUSED ? (a.par ? s_itv1 : s_itv0) : s_it_a;
    bool stop;
    {
        const uint32_t k3 = (it + 2u) % 3u;                                   // (it - 1) mod 3
        const uint32_t prev_notconv = FUSED ? (k3 == 0u ? s_n30 : (k3 == 1u ? s_n31 : s_n32)) : (((it - 1u) & 1u) ? s_nc1 : s_nc0);
        stop = it >= a.min_iter && (it >= a.max_iter || (it > 0u && prev_notconv == 0u));      // em_stop / em_stop3
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if constexpr (FUSED) {
            const uint32_t it_next = (stop || a.first) ? it : it + 1;       // updates done once this launch has ended
            st->itv[a.par ^ 1u] = it_next; st->it_a = it_next;
            st->it_b = stop ? kDoneMark : (a.first ? it : it + 1u);        // the sweep this launch runs (k_fold_slots, sharded loop)
            if (!stop) st->notconv3[it_next % 3u] = 0;                        // (the slot of the update the NEXT launch runs)
            // the streamed loop of em_run: the host follows the device through this word of pinned memory (no copy, no post kernel)
            if (a.post) __hip_atomic_store(a.post, (unsigned long long)it | ((unsigned long long)(stop ? 1u : 0u) << 32) | a.post_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            st->it_b = stop ? kDoneMark : it;
            if (!stop) { st->notconv[it & 1] = 0; st->gated[it & 1] = 0; }
        }
    }
    if (stop) return;
#ifdef SFGPU_X_STAMP
    if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 16] = t_entry;
#endif
    SF_STAMP(1);
    __shared__ double xs[kWin + 1];                        // (+ the slot of the null words: x = 0)
    __shared__ double acc[kWin + 1];
    __shared__ double den[kTileNnz + 1];                   // denominators, then count/denom, per class of the tile (+ the null class)
    // escaped members: a small accumulator keyed by transcript -- a tile's escapes go to FEW far transcripts again and again (the
    // pseudogene / paralog every read of this gene also hits), and thousands of f64 atomics on one alphaOut address serialise in
    // the L2 (measured: 700 k escapes onto ~50 transcripts 162 us per sweep, 9 us without them).  kTileNnz is 7800, not 8000,
    // to make room for it next to two resident blocks.
    __shared__ unsigned int esc_key[kEscSlots];            // transcript + 1 (0 = free)
    __shared__ double esc_val[kEscSlots];
    const double* __restrict__ x = a.x;
    // FUSED: the arrays of this launch.  u = it: the update this launch runs (none in a `first` launch), fed by sweep u's sums.
    const bool upd = FUSED && a.first == 0u;
    const uint32_t sw = upd ? it + 1u : it;                                   // the sweep this launch runs
    const double* __restrict__ rd_part = (it & 1u) ? a.part_b : a.part_a;
    const uint32_t m3 = it % 3u;
    const double* __restrict__ rd_aout = m3 == 0u ? a.aout_a : (m3 == 1u ? a.aout_b : a.aout_c);
    double* __restrict__ wr_part = FUSED ? ((sw & 1u) ? a.part_b : a.part_a) : a.partial;
    const uint32_t w3 = sw % 3u;
    double* wr_aout = FUSED ? (w3 == 0u ? a.aout_a : (w3 == 1u ? a.aout_b : a.aout_c)) : a.alpha_out;
    double* zr_aout = w3 == 0u ? a.aout_b : (w3 == 1u ? a.aout_c : a.aout_a);
    // alpha' of a FAR transcript as k_update<.., FOLD> forms it (its sums through the cover list: cov_pos names the slots)
    auto new_alpha = [&](uint32_t t) -> double {
        double ap = rd_aout[t];
        const uint2 cr = a.cov2[t];
        if (!a.sharded) for (uint32_t k = cr.x; k < cr.y; ++k) ap += rd_part[a.cov_pos[k]];
        if (VB) ap += kPriorAlpha;
        return ap;
    };
    // (FUSED kernels index per-transcript arrays by POSITION in the plan's order; the x vector init made is in the caller's)
    auto x_first = [&](uint32_t pos_) -> double { return x[a.inv ? a.inv[pos_] : pos_]; };
    auto x_of = [&](double ap, double len) -> double {
#ifdef SFGPU_X_CHEAPX
        return sweep_x<false>(ap / len);
#endif
        if (VB) return (ap > kTiny) ? sweep_x<true>(vb_x_lean(ap, a.log_norm, len)) : 0.0;       // :300-320
        return sweep_x<false>(ap / len);
    };
    auto x_now = [&](uint32_t t) -> double { return upd ? x_of(new_alpha(t), a.lenc[t]) : (FUSED ? x_first(t) : x[t]); };      // (far members)
    double local_max = -1.0; unsigned notconv = 0;
    auto judge = [&](double av_, double ap_) {
        const double gate = a.check_mode ? av_ : ap_;               // :852 vs :499
        if (gate > kCheckCutoff) {
            const double rel = fabs(av_ - ap_) / ap_;
            if (rel > local_max) local_max = rel;                  // NaN never wins, as in the reference (:854)
            if (rel > a.tol) notconv = 1;
            if (local_max < 0.0) local_max = 0.0;                  // gated at least once
        }
    };
    // the end of a FUSED launch: the transcripts no window holds, then what the wavefront saw of the convergence test
    auto fused_tail = [&]() {
#ifndef SFGPU_X_NOUPD
        const uint32_t n_unc = n_unc_v;                         // (known on the device only; requested with the descriptors)
        for (uint32_t j = threadIdx.x * gridDim.x + blockIdx.x; j < n_unc; j += kSweepBlock * gridDim.x) {
            const uint32_t t = a.unc[j];
            if (upd) { const double p = rd_aout[t] + (VB ? kPriorAlpha : 0.0); judge(a.alpha[t], p); a.alpha[t] = p; }
            zr_aout[t] = 0.0;
        }
#endif
        if (upd) {
            for (int o = kWave / 2; o > 0; o >>= 1) {
                const double m = __shfl_down(local_max, o, kWave); if (m > local_max) local_max = m;
                notconv |= __shfl_down(notconv, o, kWave);
            }
            if ((threadIdx.x & (kWave - 1)) == 0) {
                if (notconv) st->notconv3[it % 3u] = 1;
                a.tmax[((uint64_t)(it & 1u) * gridDim.x + blockIdx.x) * (kSweepBlock / kWave) + threadIdx.x / kWave] = local_max;
            }
        }
    };
    if (nc == 0) { if constexpr (FUSED) fused_tail(); return; }
    const uint4* __restrict__ words = reinterpret_cast<const uint4*>(a.stream + s0);

    // The inner loops carry no test per word: x is clean (sweep_x), null words have a slot and a class of their own, a
    // singleton's denominator is never used (phase B overwrites it with the count), and adding a zero changes nothing.  The only
    // branch is the run boundary.  The register chunks keep their gathered x values (xv) for phase C and count/denom is read
    // once per run of a class.  N words at a time; (cur, run) / (cur, f) carry over between calls.
    // What bounds the sweep (round 2, cfg3, 21 us): the LDS f64 atomics -- phase C with plain stores instead of atomics is 5.3 us
    // shorter, 9.3 M + 3.7 M atomics at ~0.3 cycles per lane and CU (tools/probes/lds_atomic_probe.hip: 0.14 conflict free,
    // 0.29 random in 1024 slots, 0.37 random in 300) -- then the fixed costs of a one-round tile (launch + state 2.3 us, staging
    // 0.8, phase B 1.8, D 1.0) and the third, quarter-full chunk pass.  Halving the instructions per nonzero (28 + 24 -> 12 + 12)
    // alone changed nothing (22.0 -> 22.5 us).
    auto den_words = [&](const uint32_t* w, auto n_tag, double* v, uint32_t& cur, double& run) {
        constexpr int N = decltype(n_tag)::value;
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = xs[w[i] & 0xFFFFu];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t cls = (w[i] >> 16) & 0x1FFFu;
            if (cls != cur) { atomicAdd(&den[cur], run); cur = cls; run = v[i]; }
            else run += v[i];
        }
    };
    auto acc_words = [&](const uint32_t* w, auto n_tag, const double* v, uint32_t& cur, double& f) {
        constexpr int N = decltype(n_tag)::value;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t cls = (w[i] >> 16) & 0x1FFFu;
            if (cls != cur) { f = den[cls]; cur = cls; }
            const double m = (w[i] & kSingle) ? 1.0 : v[i];                 // a singleton adds its count (:275 / :364)
            atomicAdd(&acc[w[i] & 0xFFFFu], m * f);
        }
    };
    using N4 = std::integral_constant<int, 4>; using N8 = std::integral_constant<int, kPerLane>;
    static_assert(kPerLane == 8, "two 16-byte loads per lane and chunk");
    // a chunk that is not held in registers: 4 words at a time (x values in registers only while they are used)
    auto den_chunk = [&](const uint4& w0, const uint4& w1) {
        const uint32_t a4[4] = {w0.x, w0.y, w0.z, w0.w}, b4[4] = {w1.x, w1.y, w1.z, w1.w};
        double v[4];
        uint32_t cur = (w0.x >> 16) & 0x1FFFu; double run = 0.0;             // (starts on its first word's class with an empty run: no add to den[null])
        den_words(a4, N4{}, v, cur, run);
        den_words(b4, N4{}, v, cur, run);
        atomicAdd(&den[cur], run);
    };
    auto acc_chunk = [&](const uint4& w0, const uint4& w1) {
        const uint32_t a4[4] = {w0.x, w0.y, w0.z, w0.w}, b4[4] = {w1.x, w1.y, w1.z, w1.w};
        double v[4];
        uint32_t cur = kTileNnz; double f = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = xs[a4[i] & 0xFFFFu];
        acc_words(a4, N4{}, v, cur, f);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = xs[b4[i] & 0xFFFFu];
        acc_words(b4, N4{}, v, cur, f);
    };

    // Everything the block needs from HBM is requested up front, before the first barrier: the
    // lane's first 8 stream words (kept in registers through phases A..C), the x window and the
    // class counts of phase B.  A typical tile (<= 8192 nonzeros) needs nothing else.
    // kRegChunks x 8 stream words per lane stay in registers through phases A..C.  (Measured in round 2 on cfg3, whose
    // one-round tiles hold ~18 000 nonzeros = 2.2 chunks of 8192: 1 chunk in registers 21.9 us per sweep, 2 chunks 23.8,
    // 3 chunks 25.1 -- the re-fetch of the other chunks hits the L2 and costs less than the registers do.)
    const uint32_t g0 = threadIdx.x * kPerLane;
    uint32_t w[GATHER ? 1 : kRegChunks][kPerLane];
    double xv[GATHER ? 1 : kRegChunks][kPerLane];         // x of the register chunks' words, phase A -> phase C (scatter form)
    // GATHER: chunk q of the tile = 8 sixteen-bit slots (one 16-byte load) + its header
    const uint4* __restrict__ slots8 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.stream) + s0);
    const uint32_t* __restrict__ hdrs = GATHER ? a.chdr + (s0 >> 3) : nullptr;
    uint4 sl_first = make_uint4(0u, 0u, 0u, 0u); uint32_t hdr_first = 0;
    if constexpr (GATHER) {
        if (g0 < n8) { sl_first = slots8[threadIdx.x]; hdr_first = hdrs[threadIdx.x]; }
    } else {
#pragma unroll
    for (int c = 0; c < kRegChunks; ++c) {
        const uint32_t g = g0 + (uint32_t)c * kSweepBlock * kPerLane;
        uint4 w0 = make_uint4(kNull, kNull, kNull, kNull), w1 = w0;
        if (g < n8) { w0 = words[g / 4]; w1 = words[g / 4 + 1]; }
        w[c][0] = w0.x; w[c][1] = w0.y; w[c][2] = w0.z; w[c][3] = w0.w; w[c][4] = w1.x; w[c][5] = w1.y; w[c][6] = w1.z; w[c][7] = w1.w;
    }
    }
    // The thread's first TWO far members (most tiles hold at most one per thread): their words are requested here, with everything else,
    // and their x stays in registers through phases A and C.  A far member is a chain of dependent gathers, and inside the phases the
    // whole block paid for that chain twice (cfg3's tile 0, which holds the benchmark's wrapped labels: + 3.5 us in A, + 4 us in C,
    // and the launch lasts as long as its slowest tile).
    // (two per thread: cfg3's tile 0 holds 1266 of them; a tile with more than 2048 walks the rest inside the phases)
    const bool has_esc0 = threadIdx.x < n_esc, has_esc1 = threadIdx.x + kSweepBlock < n_esc;
    uint32_t esc_tag0 = kSingle, esc_t0 = 0u, esc_tag1 = kSingle, esc_t1 = 0u;
    uint2 esc_sl0 = make_uint2(kNoSlot, kNoSlot), esc_sl1 = esc_sl0;
    double esc_x0 = 0.0, esc_x1 = 0.0;
    if (has_esc0) {
        esc_tag0 = a.esc_cls[e0 + threadIdx.x]; esc_t0 = a.esc_id[e0 + threadIdx.x];
        if (FUSED && upd) esc_sl0 = a.esc_slots[e0 + threadIdx.x];
    }
    if (has_esc1) {
        esc_tag1 = a.esc_cls[e0 + threadIdx.x + kSweepBlock]; esc_t1 = a.esc_id[e0 + threadIdx.x + kSweepBlock];
        if (FUSED && upd) esc_sl1 = a.esc_slots[e0 + threadIdx.x + kSweepBlock];
    }
    auto esc_value = [&](bool has, uint32_t tag, uint32_t t, uint2 sl) -> double {      // (called after the window's own requests have been issued)
        if (!has || (tag & kSingle)) return 0.0;
        if (FUSED && upd) {
            double ap = rd_aout[t];
            const double len = a.lenc[t];
            if (a.sharded) sl = make_uint2(kNoSlot, kNoSlot);             // (the reduced vector is complete)
            const double q0 = sl.x != kNoSlot ? rd_part[sl.x] : 0.0, q1 = sl.y != kNoSlot ? rd_part[sl.y] : 0.0;
            if (sl.x != kNoSlot) ap += q0;
            if (sl.y != kNoSlot) {
                ap += q1;
                const uint2 cr = a.cov2[t];
                for (uint32_t k = cr.x + 2u; k < cr.y; ++k) ap += rd_part[a.cov_pos[k]];
            }
            if (VB) ap += kPriorAlpha;
            return x_of(ap, len);
        }
        return FUSED ? x_first(t) : x[t];
    };
    auto esc_values = [&]() { esc_x0 = esc_value(has_esc0, esc_tag0, esc_t0, esc_sl0); esc_x1 = esc_value(has_esc1, esc_tag1, esc_t1, esc_sl1); };
    if constexpr (FUSED) {
        // ---- U + staging: descriptors (the tile's own and its NbTable), then ONE round trip of operands, the math, the stores.
        //      A thread holds at most one window slot (kWin <= kSweepBlock); slot i is position lo + i of the plan's transcript order, and
        //      the fused kernel's per-transcript arrays (alpha, effLen, the escape accumulators, cov2) are IN that order.
        static_assert(kWin <= kSweepBlock, "one window slot per thread");
        const uint32_t nb_n = td.nb_n, nb_before = td.nb_before;
        const bool has = threadIdx.x < span;
        const uint32_t pos = lo + threadIdx.x;
        bool home = has;
        double ap = 0.0, len = 1.0, av = 0.0, xv = 0.0;
        if (nb_n != kNbByList) {
            bool in[kNbMax]; double pj[kNbMax];
#pragma unroll
            for (int j = 0; j < kNbMax; ++j) {
                const uint4 e = td.e[j];                                     // {lo', span', off', tile'}
                in[j] = has && (uint32_t)j < nb_n && (pos - e.x) < e.y;
                pj[j] = (upd && in[j] && !a.sharded) ? rd_part[(uint64_t)e.z + (pos - e.x)] : 0.0;
                if (in[j] && (uint32_t)j < nb_before) home = false;
            }
            if (has) {
                if (upd) {
                    ap = rd_aout[pos]; len = a.lenc[pos];
                    const double own = a.sharded ? 0.0 : rd_part[off + threadIdx.x];
                    if (home) av = a.alpha[pos];
                    // (the LDS accumulators are cleared below, while these words travel)
#pragma unroll
                    for (int j = 0; j < kNbMax; ++j) if ((uint32_t)j < nb_before && in[j]) ap += pj[j];     // tile order, as the cover list
                    ap += own;
#pragma unroll
                    for (int j = 0; j < kNbMax; ++j) if ((uint32_t)j >= nb_before && in[j]) ap += pj[j];
                    if (VB) ap += kPriorAlpha;
                } else xv = x_first(pos);
            }
        } else if (has) {
            // a tile in a crowd: the sums of transcript pos through its cover list (slots in tile order); home = the list's first slot
            const uint2 cr = a.cov2[pos];
            const uint32_t k0 = cr.x, k1 = cr.y;
            home = a.cov_pos[k0] == (uint32_t)off + threadIdx.x;
            if (upd) {
                ap = rd_aout[pos]; len = a.lenc[pos];
                if (home) av = a.alpha[pos];
                if (!a.sharded) for (uint32_t k = k0; k < k1; ++k) ap += rd_part[a.cov_pos[k]];
                if (VB) ap += kPriorAlpha;
            } else xv = x_first(pos);
        }
        esc_values();
        for (uint32_t i = threadIdx.x; i < nc; i += kSweepBlock) den[i] = 0.0;
        if (threadIdx.x == 0) { xs[kWin] = 0.0; acc[kWin] = 0.0; den[kTileNnz] = 0.0; den[a.null_cls] = 0.0; }
        if (threadIdx.x < kEscSlots) { esc_key[threadIdx.x] = 0u; esc_val[threadIdx.x] = 0.0; }
        SF_STAMP(3);
        if (has && upd) xv = x_of(ap, len);
        SF_STAMP(4);
#ifndef SFGPU_X_NOUPD
        if (home) {
            if (upd) { judge(av, ap); a.alpha[pos] = ap; }
            zr_aout[pos] = 0.0;
        }
#endif
        SF_STAMP(5);
        if (has) { xs[threadIdx.x] = xv; acc[threadIdx.x] = 0.0; }
    } else {
    if (a.inv) for (uint32_t i = threadIdx.x; i < span; i += kSweepBlock) { xs[i] = x[a.inv[(uint64_t)lo + i]]; acc[i] = 0.0; }
    else for (uint32_t i = threadIdx.x; i < span; i += kSweepBlock) { xs[i] = x[(uint64_t)lo + i]; acc[i] = 0.0; }
    }
    if constexpr (!FUSED) {
    esc_values();
    for (uint32_t i = threadIdx.x; i < nc; i += kSweepBlock) den[i] = 0.0;
    if (threadIdx.x == 0) { xs[kWin] = 0.0; acc[kWin] = 0.0; den[kTileNnz] = 0.0; den[a.null_cls] = 0.0; }
    if (threadIdx.x < kEscSlots) { esc_key[threadIdx.x] = 0u; esc_val[threadIdx.x] = 0.0; }
    }
    __syncthreads();
    SF_STAMP(6);

    // ---- A: denominators
    {
        if constexpr (GATHER) {
            // a chunk: the class of its first nonzero, a bit per nonzero that starts the next class; classes are consecutive
            auto den_slots = [&](const uint4& s4, uint32_t hdr) {
                uint32_t cur = hdr & 0x1FFFu;
                const uint32_t mask = hdr >> 16;
                double run = xs[s4.x & 0xFFFFu];
                auto step = [&](uint32_t k, uint32_t slot) {
                    const double v = xs[slot];
                    if (mask & (1u << k)) { atomicAdd(&den[cur], run); ++cur; run = v; } else run += v;
                };
                step(1, s4.x >> 16); step(2, s4.y & 0xFFFFu); step(3, s4.y >> 16); step(4, s4.z & 0xFFFFu);
                step(5, s4.z >> 16); step(6, s4.w & 0xFFFFu); step(7, s4.w >> 16);
                atomicAdd(&den[cur], run);
            };
            // (the thread's second and third chunk are requested before the first is worked on: a chunk per round trip otherwise --
            //  cfg3's tiles hold 2.2 chunks per thread)
            const uint32_t q1 = threadIdx.x + kSweepBlock, q2 = threadIdx.x + 2u * kSweepBlock;
            const bool in1 = q1 * kPerLane < n8, in2 = q2 * kPerLane < n8;
            uint4 sl1 = make_uint4(0u, 0u, 0u, 0u), sl2 = sl1; uint32_t hd1 = 0u, hd2 = 0u;
            if (in1) { sl1 = slots8[q1]; hd1 = hdrs[q1]; }
            if (in2) { sl2 = slots8[q2]; hd2 = hdrs[q2]; }
            if (g0 < n8) den_slots(sl_first, hdr_first);
            if (in1) den_slots(sl1, hd1);
            if (in2) den_slots(sl2, hd2);
            for (uint32_t q = threadIdx.x + 3u * kSweepBlock; q * kPerLane < n8; q += kSweepBlock) den_slots(slots8[q], hdrs[q]);
        } else {
#pragma unroll
        for (int c = 0; c < kRegChunks; ++c) {
            // (a lane past the end of the tile holds eight null words: it stays out -- hundreds of lanes adding zeros to the ONE
            //  null class serialise in the LDS atomic unit: cfg2's 5 300-nonzero tiles went from 8.5 to 16.5 us before this test)
            if (g0 + (uint32_t)c * kSweepBlock * kPerLane >= n8) break;
            uint32_t cur = (w[c][0] >> 16) & 0x1FFFu; double run = 0.0;
            den_words(w[c], N8{}, xv[c], cur, run);
            atomicAdd(&den[cur], run);
        }
        for (uint32_t g = g0 + kRegChunks * kSweepBlock * kPerLane; g < n8; g += kSweepBlock * kPerLane) den_chunk(words[g / 4], words[g / 4 + 1]);
        }
        if (esc_x0 != 0.0) atomicAdd(&den[(esc_tag0 >> 16) & 0x1FFFu], esc_x0);       // far members: the first two from their registers,
        if (esc_x1 != 0.0) atomicAdd(&den[(esc_tag1 >> 16) & 0x1FFFu], esc_x1);
        for (uint32_t i = threadIdx.x + 2u * kSweepBlock; i < n_esc; i += kSweepBlock) {      // the others by global gathers
            uint32_t tag = a.esc_cls[e0 + i];
            if (tag & kSingle) continue;
            double v = FUSED ? x_now(a.esc_id[e0 + i]) : x[a.esc_id[e0 + i]];
            if (v != 0.0) atomicAdd(&den[(tag >> 16) & 0x1FFFu], v);
        }
    }
    // the class counts of phase B are requested here, behind phase A's last atomics: their round trip overlaps the barrier
    // (requested at the top of the kernel they held registers through phase A; requested in phase B they were 2.3 us of it)
    uint32_t cw[kCntAhead];                                                         // bit 31: singleton class
#pragma unroll
    for (int i = 0; i < kCntAhead; ++i) { const uint32_t c = threadIdx.x + i * kSweepBlock; cw[i] = (c < nc) ? a.counts[c0 + c] : 0u; }
    __syncthreads();
    SF_STAMP(7);
    // ---- B: count / denom per class (in place); singletons carry the full count (:275 / :364)
    auto invert = [&](uint32_t c, uint32_t cwc) {
        double cnt = (double)(cwc & 0x7FFFFFFFu);
        double d = den[c];
        den[c] = (cwc >> 31) ? cnt : ((d > kTiny) ? cnt / d : 0.0);     // :260-264
    };
#pragma unroll
    for (int i = 0; i < kCntAhead; ++i) { const uint32_t c = threadIdx.x + i * kSweepBlock; if (c < nc) invert(c, cw[i]); }
    for (uint32_t c = threadIdx.x + kCntAhead * kSweepBlock; c < nc; c += kSweepBlock) invert(c, a.counts[c0 + c]);
    // GATHER: the thread's first two pure chunks of phase C are requested here, ahead of the barrier
    uint4 pc_e0 = make_uint4(0u, 0u, 0u, 0u), pc_e1 = pc_e0; uint32_t pc_s0 = 0u, pc_s1 = 0u; bool pc_in0 = false, pc_in1 = false;
    if constexpr (GATHER) {
        const uint4* __restrict__ pure_p = reinterpret_cast<const uint4*>(a.csc + td.qb);
        const uint16_t* __restrict__ slot0_p = a.csc_slot0 + td.pr;
        pc_in0 = threadIdx.x < td.np; pc_in1 = threadIdx.x + kSweepBlock < td.np;
        if (pc_in0) { pc_e0 = pure_p[threadIdx.x]; pc_s0 = slot0_p[threadIdx.x]; }
        if (pc_in1) { pc_e1 = pure_p[threadIdx.x + kSweepBlock]; pc_s1 = slot0_p[threadIdx.x + kSweepBlock]; }
    }
    __syncthreads();
    SF_STAMP(8);
    // ---- C: the window.  With the transcript-major copy: a GATHER -- a thread takes chunks of 8 entries sorted by window slot,
    //      reads count / denom of their classes (random LDS reads: 0.085 cycles per lane and CU against 0.29 - 0.37 for a random
    //      f64 atomic), sums per slot in registers (singletons add their count, the others x_t times the sum) and hands the window
    //      one sum per run.  Without it: one atomic per nonzero.
    double esc_sum = 0.0;
    {
        if constexpr (GATHER) {
            const uint64_t qb = td.qb;
            const uint32_t np = td.np, nm = td.nm;
            const uint4* __restrict__ pure = reinterpret_cast<const uint4*>(a.csc + qb);
            const uint16_t* __restrict__ s0 = a.csc_slot0 + td.pr;
            auto pure_chunk = [&](const uint4& e4, uint32_t sf) {                    // one slot per chunk: 8 reads, a tree of adds, one hand-over
                const double f0 = den[e4.x & 0x1FFFu], f1 = den[(e4.x >> 16) & 0x1FFFu], f2 = den[e4.y & 0x1FFFu], f3 = den[(e4.y >> 16) & 0x1FFFu];
                const double f4 = den[e4.z & 0x1FFFu], f5 = den[(e4.z >> 16) & 0x1FFFu], f6 = den[e4.w & 0x1FFFu], f7 = den[(e4.w >> 16) & 0x1FFFu];
                const double sum = ((f0 + f1) + (f2 + f3)) + ((f4 + f5) + (f6 + f7));
                const uint32_t slot = sf & 0x7FFFu;
                const double v = (sf & kCscSingleBit) ? sum : xs[slot] * sum;           // singletons add their count (:275 / :364)
                if (v != 0.0) atomicAdd(&acc[slot], v);
            };
            if (pc_in0) pure_chunk(pc_e0, pc_s0);                                    // (requested ahead of the barrier, see phase B)
            if (pc_in1) pure_chunk(pc_e1, pc_s1);
            for (uint32_t ch = threadIdx.x + 2u * kSweepBlock; ch < np; ch += kSweepBlock) pure_chunk(pure[ch], s0[ch]);
            const uint4* __restrict__ mixed = pure + np;
            for (uint32_t ch = threadIdx.x; ch < nm; ch += kSweepBlock) {            // chunks that straddle slots: run by run
                const uint4 e4 = mixed[2u * ch], s4 = mixed[2u * ch + 1u];
                uint32_t cur = s4.x & 0xFFFFu;
                double sum = 0.0;
                auto flush_run = [&]() {
                    const uint32_t slot = cur & 0x7FFFu;
                    const double v = (cur & kCscSingleBit) ? sum : xs[slot] * sum;
                    if (v != 0.0) atomicAdd(&acc[slot], v);
                };
                auto entry = [&](uint32_t cls, uint32_t sfk) {
                    if (sfk != cur) { flush_run(); cur = sfk; sum = 0.0; }
                    sum += den[cls & 0x1FFFu];
                };
                entry(e4.x & 0xFFFFu, s4.x & 0xFFFFu); entry(e4.x >> 16, s4.x >> 16); entry(e4.y & 0xFFFFu, s4.y & 0xFFFFu); entry(e4.y >> 16, s4.y >> 16);
                entry(e4.z & 0xFFFFu, s4.z & 0xFFFFu); entry(e4.z >> 16, s4.z >> 16); entry(e4.w & 0xFFFFu, s4.w & 0xFFFFu); entry(e4.w >> 16, s4.w >> 16);
                flush_run();
            }
        } else {
#pragma unroll
        for (int c = 0; c < kRegChunks; ++c) {
            if (g0 + (uint32_t)c * kSweepBlock * kPerLane >= n8) break;
            uint32_t cur = kTileNnz; double f = 0.0;
            acc_words(w[c], N8{}, xv[c], cur, f);
        }
        for (uint32_t g = g0 + kRegChunks * kSweepBlock * kPerLane; g < n8; g += kSweepBlock * kPerLane) acc_chunk(words[g / 4], words[g / 4 + 1]);
        }
        auto esc_add = [&](uint32_t tag, uint32_t t, double xval) {     // escapes: the tile's accumulator, global atomics beyond it
            double f = den[(tag >> 16) & 0x1FFFu];
            double contrib = (tag & kSingle) ? f : xval * f;
            if (contrib != 0.0) {
                esc_sum += contrib;
                uint32_t q = (t * 2654435761u) >> (32 - 7);                      // kEscSlots = 2^7
                bool placed = false;
                for (int pr = 0; pr < 4 && !placed; ++pr, q = (q + 1) & (kEscSlots - 1)) {
                    unsigned int k = esc_key[q];
                    if (k == 0u) k = atomicCAS(&esc_key[q], 0u, t + 1u);
                    if (k == 0u || k == t + 1u) { atomicAdd(&esc_val[q], contrib); placed = true; }
                }
                if (!placed) atomicAdd(&wr_aout[t], contrib);                   // accumulator full around q: straight to memory
            }
        };
        if (has_esc0) esc_add(esc_tag0, esc_t0, esc_x0);
        if (has_esc1) esc_add(esc_tag1, esc_t1, esc_x1);
        for (uint32_t i = threadIdx.x + 2u * kSweepBlock; i < n_esc; i += kSweepBlock) {
            const uint32_t tag = a.esc_cls[e0 + i], t = a.esc_id[e0 + i];
            esc_add(tag, t, (tag & kSingle) ? 0.0 : (FUSED ? x_now(t) : x[t]));
        }
    }
    __syncthreads();
    SF_STAMP(9);
    // ---- D: publish the window into the transcript-major partial array (plain stores): the update
    //         then folds each transcript's entries with contiguous, coalesced loads
    double mine = esc_sum;
    if (threadIdx.x < kEscSlots && esc_key[threadIdx.x]) atomicAdd(&wr_aout[esc_key[threadIdx.x] - 1u], esc_val[threadIdx.x]);
    if constexpr (FUSED) {
        for (uint32_t i = threadIdx.x; i < span; i += kSweepBlock) wr_part[off + i] = acc[i];        // slot-major
        fused_tail();
    } else {
        for (uint32_t i = threadIdx.x; i < span; i += kSweepBlock) { const double v = acc[i]; wr_part[a.pub_pos[off + i]] = v; mine += v; }
    }
    SF_STAMP(10);
    if (VB && !FUSED && a.tsum) {
        // everything this tile added to alphaOut, in a fixed order: the update derives sum(alpha) -- the
        // argument of psi(sum alpha) -- from these n_tiles numbers instead of a second pass over M
        __shared__ double red[kSweepBlock / kWave];
        mine = wave_sum(mine);
        if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < kSweepBlock / kWave; ++i) t += red[i];
            a.tsum[blockIdx.x] = t;
        }
    }
}

#include "em_persist.h"

// alphaOut[t] += sum of the tiles' window entries for t, in cover-list order (deterministic)
__device__ __forceinline__ double fold_partials(uint64_t t, const uint32_t* __restrict__ cov_ptr,
                                                const uint32_t* __restrict__ cov_pos, const double* __restrict__ partial) {
    (void)cov_pos;                                  // partial is transcript-major: t owns [cov_ptr[t], cov_ptr[t+1])
    double s = 0.0;
    for (uint32_t k = cov_ptr[t], e = cov_ptr[t + 1]; k < e; ++k) s += partial[k];
    return s;
}

// the sharded loop with ONE sweep kernel per iteration: the fused sweep published its window sums slot-major; this adds a transcript's
// sums (cover list, tile order) to what the sweep's far members left in the accumulator of sweep `sw` -- the vector the all-reduce
// then sums over the ranks.  sw comes from the state block (the sweep's head wrote it; kDoneMark: the loop has ended)
__global__ void __launch_bounds__(kEmBlock)
k_fold_slots(uint64_t M, const uint2* __restrict__ cov2, const uint32_t* __restrict__ cov_pos, const double* __restrict__ part_a,
             const double* __restrict__ part_b, double* aout_a, double* aout_b, double* aout_c, const EmState* st) {
    const uint32_t sw = st->it_b;
    if (sw == kDoneMark) return;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    const double* __restrict__ part = (sw & 1u) ? part_b : part_a;
    double* aout = (sw % 3u) == 0u ? aout_a : ((sw % 3u) == 1u ? aout_b : aout_c);
    const uint2 r = cov2[t];
    if (r.x == r.y) return;
    double s2 = 0.0;
    for (uint32_t k = r.x; k < r.y; ++k) s2 += part[cov_pos[k]];
    aout[t] += s2;
}

// piecewise API: make alphaOut complete before the caller's all-reduce
__global__ void k_fold(uint64_t M, double* alpha_out, const uint32_t* __restrict__ cov_ptr,
                       const uint32_t* __restrict__ cov_pos, const double* __restrict__ partial, const EmState* st) {
    if (st->it_b == kDoneMark) return;
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) alpha_out[t] += fold_partials(t, cov_ptr, cov_pos, partial);
}

// per-transcript update (:849-861): gate, relative change, alpha <- alphaOut, alphaOut <- 0, ++it.
// No atomics: the convergence flags are idempotent plain stores (every writer stores 1) and the
// per-block maximum goes to blkmax[parity][block], reduced by the host when it polls.
template <bool VB, bool FOLD>
__global__ void __launch_bounds__(kEmBlock)
k_update(uint64_t M, double* alpha, double* alpha_out, double* x, const double* __restrict__ lenc,
         double tol, int check_mode, double* sum_partials_out, double* blkmax, EmState* st,
         const uint32_t* __restrict__ cov_ptr, const uint32_t* __restrict__ cov_pos,
         const double* __restrict__ partial, const double* __restrict__ tsum, uint32_t n_tiles, double const_log_norm, int use_const_norm) {
    // request this thread's first operands before looking at the loop state (they do not depend on it):
    // the state test then costs no extra memory round trip
    const uint64_t t_first = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x;
    double a_first = 0.0, ao_first = 0.0; uint32_t k0_first = 0, k1_first = 0;
    if (t_first < M) {
        a_first = alpha[t_first]; ao_first = alpha_out[t_first];
        if (FOLD) { k0_first = cov_ptr[t_first]; k1_first = cov_ptr[t_first + 1]; }
    }
    double tsum_part = 0.0;
    // x for the next sweep is produced here (no k_vb_prepare pass) when the normaliser psi(sum alpha) is at hand: from the sweep's
    // per-tile sums (tsum), or -- round 4, the default inside optimize() -- as a CONSTANT of the run, psi(M prior + numMapped).
    // expTheta's normaliser scales every x_t alike and cancels in x_t count / denom (:340-366), and in VBEM every class hands out
    // its whole count (alpha >= prior > 0 keeps every denominator far above denorm_min), so sum(alpha') IS M prior + numMapped up
    // to rounding: the constant differs from the reference's psi(sum of the floats) by ~1e-13 in a factor that cancels.  What it
    // buys: no n_tiles-element reduction, no digamma of the sum and no block barrier in front of every update (SFGPU_EM_EXACT_NORM=1
    // keeps the summed form).
    const bool const_norm = VB && use_const_norm != 0;
    const bool fused_vb = VB && (tsum != nullptr || const_norm);
    if (fused_vb && !const_norm) for (uint32_t i = threadIdx.x; i < n_tiles; i += kEmBlock) tsum_part += tsum[i];
    uint32_t it = st->it_b;
    if (it == kDoneMark) return;
    __shared__ double lds[kEmBlock / kWave];
    __shared__ double lmax[kEmBlock / kWave];
    double local_sum = 0.0, local_max = -1.0;
    unsigned notconv = 0;
    double log_norm = 0.0;
    if (const_norm) log_norm = const_log_norm;
    else if (fused_vb) {
        // sum(alpha) of this update = M * prior + what the tiles added; same order in every block
        double sacc = block_sum(tsum_part, lds);
        __shared__ double bc;
        if (threadIdx.x == 0) bc = digamma_pos((double)M * kPriorAlpha + sacc);
        __syncthreads();
        log_norm = bc;
    }
    for (uint64_t t = t_first; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        const bool first = (t == t_first);
        double a = first ? a_first : alpha[t];
        double ap = first ? ao_first : alpha_out[t];
        if (FOLD) {
            const uint32_t k0 = first ? k0_first : cov_ptr[t], k1 = first ? k1_first : cov_ptr[t + 1];
            // transcript-major: contiguous, fixed order.  (The first four entries are requested together: a loop over them is one
            // round trip per entry, and some lane of a wavefront nearly always has three or four.)
            const uint32_t nk = k1 - k0;
            const double p0 = nk > 0u ? partial[k0] : 0.0, p1 = nk > 1u ? partial[k0 + 1u] : 0.0;
            const double p2 = nk > 2u ? partial[k0 + 2u] : 0.0, p3 = nk > 3u ? partial[k0 + 3u] : 0.0;
            if (nk > 0u) ap += p0;
            if (nk > 1u) ap += p1;
            if (nk > 2u) ap += p2;
            if (nk > 3u) ap += p3;
            for (uint32_t k = k0 + 4u; k < k1; ++k) ap += partial[k];
        }
        if (VB) ap += kPriorAlpha;                     // alphaOut starts at the prior (:318)
        double gate = check_mode ? a : ap;             // :852 vs :499
        if (gate > kCheckCutoff) {
            double rel = fabs(a - ap) / ap;
            if (rel > local_max) local_max = rel;      // NaN never wins, as in the reference (:854)
            if (rel > tol) notconv = 1;
            if (local_max < 0.0) local_max = 0.0;      // gated at least once
        }
        alpha[t] = ap; alpha_out[t] = 0.0;
        if (fused_vb) x[t] = (ap > kTiny) ? sweep_x<true>(exp(digamma_pos(ap) - log_norm) / lenc[t]) : 0.0;   // :300-320
        else if (VB) local_sum += ap;
        else x[t] = sweep_x<false>(ap / lenc[t]);
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
        double m = __shfl_down(local_max, o, kWave); if (m > local_max) local_max = m;
        notconv |= __shfl_down(notconv, o, kWave);
    }
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (notconv) st->notconv[it & 1] = 1;
        lmax[w] = local_max;
    }
    if (VB && !fused_vb) { double s = block_sum(local_sum, lds); if (threadIdx.x == 0) sum_partials_out[blockIdx.x] = s; }
    else __syncthreads();
    if (threadIdx.x == 0) {
        double m = lmax[0];
        for (int i = 1; i < kEmBlock / kWave; ++i) if (lmax[i] > m) m = lmax[i];
        blkmax[(it & 1) * kMaxPartials + blockIdx.x] = m;          // -1 = nothing gated in this block
        if (blockIdx.x == 0) st->it_a = it + 1;
    }
}

// truncateCountVector (:36-44) + alphaSum
__global__ void k_truncate(uint64_t M, const double* __restrict__ alpha, double cutoff, double* out, double* partials) {
    __shared__ double lds[kEmBlock / kWave];
    double v = 0.0;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock) {
        double a = alpha[t];
        if (a <= cutoff) a = 0.0;
        out[t] = a; v += a;
    }
    double s = block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// :885-891 mass = alpha / alphaSum
__global__ void k_mass(uint64_t M, const double* __restrict__ est, double* mass, const double* partials, int nb,
                       EmState* st) {
    __shared__ double lds[kEmBlock / kWave];
    double asum = sum_partials(partials, nb, lds);
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha_sum = asum;
    if (!mass) return;
    for (uint64_t t = (uint64_t)blockIdx.x * kEmBlock + threadIdx.x; t < M; t += (uint64_t)gridDim.x * kEmBlock)
        mass[t] = est[t] / asum;
}

static inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kEmBlock - 1) / kEmBlock); }

}  // namespace sfgpu

using namespace sfgpu;

static std::atomic<bool> g_allow_persist{true};      // sfgpu_em_allow_persistent: the process-wide switch (ranks that share a device turn the loop off)

struct sfgpu_em {
    hipStream_t user_stream = nullptr;
    hipStream_t stream = nullptr;          // own stream: graph capture is illegal on the null stream
    hipStream_t cur = nullptr;             // where work goes: `stream` inside optimize(), the caller's stream for the piecewise API
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_join = nullptr, ev_plan = nullptr;
    hipEvent_t ev_poll[2] = {nullptr, nullptr};   // pipelined stop test of the loop (em_poll_start / em_poll_wait)
    unsigned long long* h_mirror = nullptr;       // pinned: what k_post_state wrote last
    sfgpu_problem prob{};
    uint64_t L = 0;
    int nb = 1;                            // blocks of the per-transcript kernels
    double *alpha = nullptr, *alpha_out = nullptr, *x = nullptr, *lenc = nullptr;
    double *partials = nullptr, *sum_partials = nullptr, *scratch = nullptr;
    uint32_t* counts32 = nullptr;
    uint32_t* inv = nullptr; uint32_t* cperm = nullptr;      // the plan's own transcript / class order (em_renumber), or null
    uint32_t* tile_lo = nullptr; uint32_t* tile_c0 = nullptr; uint32_t* tile_span = nullptr; uint32_t n_tiles = 0;
    uint32_t tile_nnz = 0;                                  // nonzeros per tile of the plan
    TileDesc* td = nullptr;                                 // the same, one record per tile (what the sweep reads)
    uint64_t* tile_off = nullptr; uint64_t P = 0;          // window slots over all tiles
    double* partial = nullptr;                              // [P] per-tile window sums of one sweep
    uint32_t* cov_ptr = nullptr; uint32_t* cov_pos = nullptr;   // transcript -> its entries of `partial`
    uint32_t* pub_pos = nullptr;                                // window slot -> its entry of `partial`
    uint32_t* lstream = nullptr; uint32_t* esc_id = nullptr; uint32_t* esc_cls = nullptr;   // re-packed labels (k_sweep_lds)
    uint64_t* tile_s0 = nullptr; uint64_t* tile_esc0 = nullptr;
    bool gather = false;                        // the sweep runs in its GATHER form (compact class-major stream + transcript-major copy)
    uint32_t* chdr = nullptr;                   // ... the compact stream's chunk headers (lstream then holds 16-bit slots)
    unsigned char* csc = nullptr; uint16_t* csc_slot0 = nullptr;                              // transcript-major copy of the tiles (phase C as a gather)
    uint64_t* tile_qb = nullptr; uint32_t* tile_np = nullptr; uint32_t* tile_pr = nullptr;
    double* blkmax = nullptr; double* h_blkmax = nullptr;   // [2][kMaxPartials]
    double* tsum = nullptr;                                 // [n_tiles] what each tile added in the last sweep (VBEM, optimize(), SFGPU_EM_EXACT_NORM=1)
    unsigned long long* h_plan = nullptr;                   // pinned: what the plan reads back (a copy into pageable memory is a round trip of its own)
    bool const_norm = true; double vb_log_norm = 0.0;       // VBEM inside optimize(): psi(M prior + numMapped) as the run's normaliser (k_update)
    bool in_optimize = false;                               // the on-device loop (vs the piecewise API) is driving the kernels
    // the FUSED iteration (k_sweep_lds<.., true, true>): second array of window sums, two more escape accumulators, per-wavefront maxima
    double *partial_b = nullptr, *aout_b = nullptr, *aout_c = nullptr, *tmax = nullptr;
    double* partial_a = nullptr;                            // (slot-major, like partial_b; `partial` stays the two-kernel loop's)
    uint2* esc_slots = nullptr; uint64_t E = 0;             // the far members' inline cover slots (k_esc_slots)
    uint2* cov2 = nullptr;                                  // cover list by position of the plan's order
    uint32_t* esc_pos = nullptr; double *alphaP = nullptr, *lencP = nullptr;      // plans with an order of their own: far members as positions, alpha / effLen in that order
    uint32_t* unc = nullptr;                                // transcripts no window holds ([M] + count + the overlap tables' flags behind them)
    int fused_ok = -1;                                      // -1: not looked at yet; 0: this plan keeps the two-kernel iteration
    unsigned long long* dbg = nullptr;
    bool fused = false;                                     // this optimize() runs fused launches
    bool streamed = false;                                  // ... one by one, the host a few launches ahead of the device (no graph)
    uint32_t run_no = 0;                                    // tags the progress words of this optimize()
    uint32_t par = 0;                                       // parity of the next fused launch
    bool graph_fused = false; bool graph_const_norm = true; double graph_log_norm = 0.0;      // (what the cached graph was built for)
    uint32_t null_cls = kTileNnz;                           // GATHER: class index of the transcript-major copy's padding = the largest class count of a tile
    // the PERSISTENT loop (em_persist.h): far-slot tables, the exchange buffer (control words + granule arrays), the plan's verdict
    uint32_t *esc_far = nullptr, *far_pos = nullptr, *far_xi = nullptr, *ft_list = nullptr; uint2* ftgt = nullptr;
    uint4* recs = nullptr; uint16_t* ovc = nullptr; TilePack* tp = nullptr; uint32_t *cnt8 = nullptr, *cpos = nullptr, *esc_cls_p = nullptr;      // phase A's class records (k_pack_build)
    uint4* cscp = nullptr;                                  // ... and the transcript-major copy with every chunk one slot's (k_cscp_build)
    uint32_t *kv_tmp = nullptr, *idx_tmp = nullptr, *tin_tmp = nullptr;      // the plan's sorted nonzeros, kept until the persistent loop's tables are made
    unsigned char* xbuf = nullptr; size_t xbuf_bytes = 0;   // [control words | status | part0 | part1 | far0 | far1 | xpub]
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_side = nullptr, ev_td = nullptr;      // the plan's side stream (sfgpu_em_create)
    bool side_live = false;                                 // ... forked and not yet joined
    bool xbuf_uncached = false;                             // ... in UNCACHED device memory (the default; SFGPU_EM_XBUF=pool: an ordinary pool block)
    uint32_t* pflags = nullptr;                             // device: [0] plan flags (!= 0: not eligible), [1] most far slots of a tile
    int persist_ok = -1;                                    // -1: not looked at yet; 0: this plan (or this device) does not run persistent
    uint32_t far_cap = 0, esc_ln = 0;                       // LDS of the persistent loop: far slots of a tile at most; far members of a tile kept on chip
    bool persist = false;                                   // this optimize() runs as one launch
    bool no_persist = false;                                // set while several bootstrap lanes run (see sfgpu_bootstrap)
    int sharded_fused = 0;                                  // sfgpu_em_set_sharded_fused: the sharded loop runs one sweep kernel per iteration (every rank agreed)
    uint64_t* bs_prefix = nullptr; uint32_t* bs_base = nullptr;   // bootstrap: prefix sums / copy of the observed counts
    uint32_t *bs_scratch_a = nullptr, *bs_scratch_b = nullptr; uint64_t bs_total = 0;
    EmState* d_state = nullptr;
    EmState* h_state = nullptr;            // pinned
    sfgpu_em_opts opts{};
    bool begun = false;
    hipGraphExec_t graph = nullptr;
    sfgpu_em_opts graph_opts{};
    uint32_t graph_iters = 0;
    std::vector<sfgpu_em*> bs_clones;      // extra bootstrap lanes (sfgpu_bootstrap)
    bool lenc_dirty = false;               // sfgpu_em_rebase replaced the lengths: begin() restores the problem's own
};

static void em_free(sfgpu_em* em) {
    if (!em) return;
    for (sfgpu_em* c : em->bs_clones) em_free(c);
    em->bs_clones.clear();
    if (em->stream) (void)hipStreamSynchronize(em->stream);
    if (em->side) { (void)hipStreamSynchronize(em->side); stream_release(em->side); }
    if (em->ev_fork) (void)hipEventDestroy(em->ev_fork);
    if (em->ev_side) (void)hipEventDestroy(em->ev_side);
    if (em->ev_td) (void)hipEventDestroy(em->ev_td);
    if (em->graph) (void)hipGraphExecDestroy(em->graph);
    void* bufs[] = {em->alpha, em->alpha_out, em->x, em->lenc, em->partials, em->sum_partials, em->scratch,
                    em->counts32, em->d_state, em->tile_lo, em->tile_c0, em->tile_span, em->tile_off, em->partial,
                    em->cov_ptr, em->cov_pos, em->pub_pos, em->bs_prefix, em->bs_base, em->bs_scratch_a, em->bs_scratch_b, em->lstream, em->esc_id, em->esc_cls, em->tile_s0, em->tile_esc0, em->chdr, em->csc, em->csc_slot0, em->tile_qb, em->tile_np, em->tile_pr,
                    em->blkmax, em->tsum, em->inv, em->cperm, em->esc_far, em->far_pos, em->far_xi, em->ft_list, em->ftgt, em->recs, em->ovc, em->tp, em->cnt8, em->cpos, em->esc_cls_p, em->cscp, em->kv_tmp, em->idx_tmp, em->tin_tmp, em->pflags, em->partial_b, em->aout_b, em->aout_c, em->tmax, em->td, em->unc, em->partial_a, em->esc_slots, em->cov2, em->esc_pos, em->alphaP, em->lencP};
    for (void* b : bufs) if (b) pool_free(b);
    if (em->xbuf) { if (em->xbuf_uncached) uncached_free(em->xbuf); else pool_free(em->xbuf); }
    if (em->h_state) pinned_free(em->h_state);
    if (em->h_blkmax) pinned_free(em->h_blkmax);
    if (em->ev_a) (void)hipEventDestroy(em->ev_a);
    if (em->ev_b) (void)hipEventDestroy(em->ev_b);
    if (em->ev_join) (void)hipEventDestroy(em->ev_join);
    if (em->ev_plan) (void)hipEventDestroy(em->ev_plan);
    for (hipEvent_t e : em->ev_poll) if (e) (void)hipEventDestroy(e);
    if (em->h_mirror) pinned_free(em->h_mirror);
    if (em->h_plan) pinned_free(em->h_plan);
    if (em->stream) stream_release(em->stream);    // synchronised above
    delete em;
}

static int em_fill_opts(sfgpu_em* em, const sfgpu_em_opts* o) {
    SF_REQUIRE(o, SFGPU_ERR_INVALID, "null sfgpu_em_opts");
    SF_REQUIRE(o->tol >= 0.0, SFGPU_ERR_INVALID, "tol must be >= 0");
    em->opts = *o;
    if (em->opts.iters_per_launch == 0) {
        static const uint32_t dflt = []() { const char* e = SF_DEV_ENV("SFGPU_EM_CHUNK"); long v = e ? atol(e) : 0; return (uint32_t)(v >= 1 && v <= 4096 ? v : 32); }();
        em->opts.iters_per_launch = dflt;                     // (tuning: iterations per graph launch / poll)
    }
    return SFGPU_OK;
}

// Where an iteration's kernels go: straight onto a stream, or into a graph under construction as a chain
// of kernel nodes.  The graph is built with explicit nodes, not by stream capture: on ROCm 7.2 a capture
// in one host thread is invalidated by unrelated calls of other threads (allocations, synchronisations --
// the bootstrap lanes, or any multi-threaded host), whatever the capture mode.
struct Launcher {
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphNode_t last = nullptr;
    hipError_t launch(const void* func, dim3 grid, dim3 block, void** args) {
        if (!graph) return hipLaunchKernel(func, grid, block, args, 0, stream);
        hipKernelNodeParams kp{};
        kp.func = const_cast<void*>(func); kp.gridDim = grid; kp.blockDim = block;
        kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
        hipGraphNode_t node = nullptr;
        hipError_t e = hipGraphAddKernelNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &kp);
        if (e == hipSuccess) last = node;
        return e;
    }
};

static SweepArgs em_sweep_args(sfgpu_em* em) {
    const sfgpu_problem& p = em->prob;
    SweepArgs a{};
    a.null_cls = em->null_cls;
    a.td = em->td; a.st = em->d_state; a.min_iter = em->opts.min_iter; a.max_iter = em->opts.max_iter;
    a.stream = em->lstream; a.chdr = em->chdr; a.x = em->x; a.counts = em->counts32;
    a.part_a = em->partial_a; a.part_b = em->partial_b; a.aout_a = em->alpha_out; a.aout_b = em->aout_b; a.aout_c = em->aout_c;
    a.lenc = em->lenc; a.alpha = em->alpha; a.inv = em->inv; a.esc_id = em->esc_id; a.esc_cls = em->esc_cls; a.esc_slots = em->esc_slots;
    a.csc = em->csc; a.csc_slot0 = em->csc_slot0; a.pub_pos = em->pub_pos; a.alpha_out = em->alpha_out; a.partial = em->partial;
    a.cov_ptr = em->cov_ptr; a.cov2 = em->cov2; a.cov_pos = em->cov_pos; a.unc = em->unc; a.unc_n = em->unc ? em->unc + p.M : nullptr; a.check_mode = em->opts.check_mode;
    a.tmax = em->tmax; a.tol = em->opts.tol; a.log_norm = em->vb_log_norm; a.M = p.M; a.dbg = em->dbg;
    return a;
}

// one sweep of the current iteration
static int em_enqueue_sweep(sfgpu_em* em, Launcher& L) {
    const sfgpu_problem& p = em->prob;
    if (p.C == 0) return SFGPU_OK;
    SweepArgs a = em_sweep_args(em);
    a.tsum = (em->opts.use_vbem && em->in_optimize && !em->const_norm) ? em->tsum : nullptr;
    void* args[] = {&a};
    const void* f = em->gather ? (em->opts.use_vbem ? reinterpret_cast<const void*>(&k_sweep_lds<true, true>)
                                                    : reinterpret_cast<const void*>(&k_sweep_lds<false, true>))
                               : (em->opts.use_vbem ? reinterpret_cast<const void*>(&k_sweep_lds<true, false>)
                                                    : reinterpret_cast<const void*>(&k_sweep_lds<false, false>));
    SF_HIP(L.launch(f, dim3(em->n_tiles), dim3(kSweepBlock), args));
    return SFGPU_OK;
}
static int em_enqueue_sweep(sfgpu_em* em) { Launcher L; L.stream = em->cur; return em_enqueue_sweep(em, L); }

// one FUSED launch: the update of the iteration before + the sweep of this one (k_sweep_lds<.., true, true>).  `first`: nothing to
// update yet -- x comes from the x vector that init made.  The launch's parity is a kernel argument (a graph bakes it: chunks hold
// an even number of launches).
static int em_enqueue_fused(sfgpu_em* em, Launcher& L, bool first, bool sharded = false) {
    SweepArgs a = em_sweep_args(em);
    a.par = em->par; a.first = first ? 1u : 0u; a.sharded = sharded ? 1u : 0u;
    a.post = em->streamed ? em->h_mirror : nullptr; a.post_tag = (1ull << 33) | ((unsigned long long)(em->run_no & 0xFFFFFu) << 34);
    if (em->inv) { a.alpha = em->alphaP; a.lenc = em->lencP; a.esc_id = em->esc_pos; }      // (per-transcript arrays in the plan's order)
    void* args[] = {&a};
    const void* f = em->opts.use_vbem ? reinterpret_cast<const void*>(&k_sweep_lds<true, true, true>)
                                      : reinterpret_cast<const void*>(&k_sweep_lds<false, true, true>);
    SF_HIP(L.launch(f, dim3(em->n_tiles), dim3(kSweepBlock), args));
    em->par ^= 1u;
    return SFGPU_OK;
}
static int em_enqueue_fused(sfgpu_em* em, bool first) { Launcher L; L.stream = em->cur; return em_enqueue_fused(em, L, first); }

// per-wavefront maxima of the fused launches -> the per-block array finish() reads (slot 0 of each parity; the rest "nothing gated")
__global__ void __launch_bounds__(kEmBlock)
k_fused_max(uint64_t n, const double* __restrict__ tmax, double* blkmax, int nb) {
    __shared__ double lmax[kEmBlock / kWave];
    const int par = blockIdx.x;
    double m = -1.0;
    for (uint64_t i = threadIdx.x; i < n; i += kEmBlock) { const double v = tmax[(uint64_t)par * n + i]; if (v > m) m = v; }
    for (int o = kWave / 2; o > 0; o >>= 1) { const double v = __shfl_down(m, o, kWave); if (v > m) m = v; }
    if ((threadIdx.x & (kWave - 1)) == 0) lmax[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) { for (int i = 1; i < kEmBlock / kWave; ++i) if (lmax[i] > m) m = lmax[i]; blkmax[par * kMaxPartials] = m; }
    for (int i = 1 + threadIdx.x; i < nb; i += kEmBlock) blkmax[par * kMaxPartials + i] = -1.0;
}

// `fold`: the sweep's per-tile window sums still have to be folded into alphaOut (true inside
// optimize(); false in the piecewise API, where sfgpu_em_sweep folds before the caller's all-reduce)
static int em_enqueue_update(sfgpu_em* em, bool fold, Launcher& L) {
    const sfgpu_problem& p = em->prob;
    dim3 g(em->nb), b(kEmBlock);
    fold = fold && p.C != 0;
    // inside optimize() the VBEM update gets sum(alpha) from the sweep's per-tile sums and writes the next
    // x itself; the piecewise API (all-reduce between sweep and update) keeps the separate k_vb_prepare pass
    const bool fused = em->opts.use_vbem && fold && em->in_optimize;
    const double* fused_tsum = (fused && !em->const_norm) ? em->tsum : nullptr;
    int use_const = (fused && em->const_norm) ? 1 : 0;
    double const_log_norm = em->vb_log_norm;
    uint64_t M = p.M; double tol = em->opts.tol; int check_mode = em->opts.check_mode; uint32_t n_tiles = em->n_tiles;
    void* args[] = {&M, &em->alpha, &em->alpha_out, &em->x, &em->lenc, &tol, &check_mode, &em->sum_partials,
                    &em->blkmax, &em->d_state, &em->cov_ptr, &em->cov_pos, &em->partial, &fused_tsum, &n_tiles, &const_log_norm, &use_const};
    const void* f;
    if (em->opts.use_vbem) f = fold ? reinterpret_cast<const void*>(&k_update<true, true>) : reinterpret_cast<const void*>(&k_update<true, false>);
    else f = fold ? reinterpret_cast<const void*>(&k_update<false, true>) : reinterpret_cast<const void*>(&k_update<false, false>);
    SF_HIP(L.launch(f, g, b, args));
    if (em->opts.use_vbem && !fused_tsum && !use_const) {
        int nb = em->nb, force = 0;
        void* vargs[] = {&M, &em->alpha, &em->x, &em->lenc, &em->sum_partials, &nb, &em->d_state, &force};
        SF_HIP(L.launch(reinterpret_cast<const void*>(&k_vb_prepare), g, b, vargs));
    }
    return SFGPU_OK;
}
static int em_enqueue_update(sfgpu_em* em, bool fold) { Launcher L; L.stream = em->cur; return em_enqueue_update(em, fold, L); }

static int em_enqueue_fold(sfgpu_em* em) {
    const sfgpu_problem& p = em->prob;
    if (p.C == 0) return SFGPU_OK;
    hipLaunchKernelGGL(k_fold, dim3(blocks_for(p.M)), dim3(kEmBlock), 0, em->cur, p.M, em->alpha_out, em->cov_ptr,
                       em->cov_pos, em->partial, em->d_state);
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

// order the handle's stream after everything already enqueued on the caller's stream
static int em_join_user(sfgpu_em* em) {
    SF_HIP(hipEventRecord(em->ev_join, em->user_stream));
    SF_HIP(hipStreamWaitEvent(em->stream, em->ev_join, 0));
    return SFGPU_OK;
}

static void em_stats_from_state(sfgpu_em* em, sfgpu_em_stats* s) {
    if (!s) return;
    const EmState* h = em->h_state;
    uint32_t it = h->it_a;
    s->iters = it;
    s->fused = em->fused ? 1u : 0u; s->persistent = em->persist ? 1u : 0u;
    s->n_active = h->n_active;
    s->alpha_sum = h->alpha_sum;
    if (it == 0) { s->converged = 0; s->max_rel_diff = -DBL_MAX; return; }
    uint32_t par = (it - 1) & 1;
    s->converged = em->fused ? (h->notconv3[(it - 1) % 3] == 0) : (h->notconv[par] == 0);
    double m = -1.0;
    for (int i = 0; i < em->nb; ++i) { double v = em->h_blkmax[par * kMaxPartials + i]; if (v > m) m = v; }
    s->max_rel_diff = (m >= 0.0) ? m : -DBL_MAX;      // the reference starts from -DBL_MAX (:850)
}

extern "C" {

// Build the plan's own transcript and class order (see the kernels above): em->inv, em->cperm, the class table in that order
// (rowptr2 / vids: members as POSITIONS; freed by the caller once the stream is written) and counts32 in the new class order.
static int em_renumber(sfgpu_em* em, const sfgpu_problem* prob, uint32_t L, uint32_t** rowptr2_out, uint32_t** vids_out) {
    const uint64_t M = prob->M, C = prob->C;
    hipStream_t st = em->cur;
    uint32_t *key = nullptr, *perm = nullptr, *inv = nullptr, *ckey = nullptr, *cperm = nullptr, *lens = nullptr, *vals = nullptr;
    uint32_t *rowptr2 = nullptr, *vids = nullptr;
    uint64_t *k_in = nullptr, *k_out = nullptr, *off64 = nullptr;
    int rc = SFGPU_OK;
    const uint64_t N = std::max(M, C) + 1;
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(st);
        for (void* q : {(void*)key, (void*)perm, (void*)inv, (void*)ckey, (void*)cperm, (void*)lens, (void*)vals, (void*)rowptr2, (void*)vids,
                        (void*)k_in, (void*)k_out, (void*)off64}) if (q) pool_free(q);
        return code;
    };
#define RN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); return fail(SFGPU_ERR_HIP); } } while (0)
    RN_TRY(pool_malloc(&key, M * 4)); RN_TRY(pool_malloc(&perm, M * 4)); RN_TRY(pool_malloc(&inv, M * 4));
    RN_TRY(pool_malloc(&ckey, (C ? C : 1) * 4)); RN_TRY(pool_malloc(&cperm, (C ? C : 1) * 4)); RN_TRY(pool_malloc(&lens, (C + 1) * 4));
    RN_TRY(pool_malloc(&vals, N * 4)); RN_TRY(pool_malloc(&k_in, N * 8)); RN_TRY(pool_malloc(&k_out, N * 8)); RN_TRY(pool_malloc(&off64, (C + 2) * 8));
    RN_TRY(pool_malloc(&rowptr2, (C + 1) * 4)); RN_TRY(pool_malloc(&vids, ((uint64_t)L + 1) * 4));
    // key_t = smallest transcript id within a few class hops of t
    hipLaunchKernelGGL(k_renum_iota, dim3(blocks_for(M)), dim3(kEmBlock), 0, st, M, key);
    int hops = kRenumberHops;
    if (const char* e = SF_DEV_ENV("SFGPU_EM_RENUMBER_HOPS")) { int v = atoi(e); if (v >= 1 && v <= 64) hops = v; }      // tuning
    {
        uint32_t *kin = key, *kout = perm;                  // (perm is filled further down: free until then)
        for (int hop = 0; hop < hops; ++hop) {
            RN_TRY(hipMemcpyAsync(kout, kin, M * 4, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_lp_min, dim3(blocks_for(C)), dim3(kEmBlock), 0, st, C, prob->d_rowptr, prob->d_ids, kin, kout);
            std::swap(kin, kout);
        }
        if (kin != key) RN_TRY(hipMemcpyAsync(key, kin, M * 4, hipMemcpyDeviceToDevice, st));
    }
    // transcripts by (key, id): inv[position] = transcript, perm[transcript] = position
    hipLaunchKernelGGL(k_renum_pair_keys, dim3(blocks_for(M)), dim3(kEmBlock), 0, st, M, key, k_in, vals);
    RN_TRY(hipGetLastError());
    if ((rc = sort_pairs_u64_u32(k_in, k_out, vals, inv, M, st, 64, false))) return fail(rc);
    hipLaunchKernelGGL(k_invert_perm, dim3(blocks_for(M)), dim3(kEmBlock), 0, st, M, inv, perm);
    // classes by their smallest position
    hipLaunchKernelGGL(k_class_min_pos, dim3(blocks_for(C)), dim3(kEmBlock), 0, st, C, prob->d_rowptr, prob->d_ids, perm, ckey);
    hipLaunchKernelGGL(k_renum_pair_keys, dim3(blocks_for(C)), dim3(kEmBlock), 0, st, C, ckey, k_in, vals);
    RN_TRY(hipGetLastError());
    if ((rc = sort_pairs_u64_u32(k_in, k_out, vals, cperm, C, st, 64, false))) return fail(rc);
    hipLaunchKernelGGL(k_renum_lens, dim3(blocks_for(C + 1)), dim3(kEmBlock), 0, st, C, prob->d_rowptr, cperm, lens);
    RN_TRY(hipGetLastError());
    if ((rc = exclusive_scan_u32(lens, off64, C, st, false))) return fail(rc);
    hipLaunchKernelGGL(k_renum_csr, dim3(blocks_for(C + 1)), dim3(kEmBlock), 0, st, C, prob->d_rowptr, prob->d_ids, cperm, perm, off64, rowptr2, vids);
    {
        unsigned int* ovf = reinterpret_cast<unsigned int*>(em->partials);
        RN_TRY(hipMemsetAsync(ovf, 0, 4, st));
        hipLaunchKernelGGL(k_narrow_counts, dim3(blocks_for(C)), dim3(kEmBlock), 0, st, C, prob->d_counts, rowptr2, em->counts32, ovf, cperm);
    }
    RN_TRY(hipGetLastError());
    RN_TRY(hipStreamSynchronize(st));
#undef RN_TRY
    for (void* q : {(void*)key, (void*)perm, (void*)ckey, (void*)lens, (void*)vals, (void*)k_in, (void*)k_out, (void*)off64}) pool_free(q);
    em->inv = inv; em->cperm = cperm;
    *rowptr2_out = rowptr2; *vids_out = vids;
    return SFGPU_OK;
}

// The persistent loop's part of the plan (em_persist.h): far slots per tile, the lists of far slots per target, the exchange buffer.
// Leaves em->pflags on the device ([0] != 0: this plan does not run persistent; [1]: the most far slots a tile has); sfgpu_em_create
// queues their read-back.  Nothing is waited for.
static int em_persist_plan(sfgpu_em* em, uint32_t nt, uint64_t E, uint64_t P, const uint32_t* p_rowptr) {
    const bool off = []() { const char* e = getenv("SFGPU_EM_PERSIST"); return e && atoi(e) == 0; }() || !g_allow_persist.load(std::memory_order_relaxed);
    if (off || 2 * P * 16ull + 3 * E * 16ull >= (1ull << 31)) return SFGPU_OK;                  // (granules are addressed with 32-bit byte offsets)
    if (nt > 4096u) return SFGPU_OK;                            // (a plan of several rounds of tiles never runs persistent: no tables -- phase C's chunk array alone reserves 32 KB per tile)
    const uint64_t M = em->prob.M;
    hipStream_t st = em->cur;
    SF_HIP(pool_malloc(&em->pflags, 16));
    SF_HIP(hipMemsetAsync(em->pflags, 0, 16, st));
    const uint64_t En = E ? E : 1;
    // The class records / the remapped transcript-major copy and the far-slot tables do not depend on each other (k_far_tiles writes f0 / nf
    // of the tiles' records, k_pack_build reads e0 / n_esc): the first pair runs on a side stream next to the far tables' ~17 small kernels
    hipStream_t s2 = em->side_live ? em->side : st;      // (sfgpu_em_create forked it behind k_tile_build and joins it when the plan is complete)
    {   // phase A's class records (k_pack_build: from the compact stream's 16-bit slots, the plan's rowptr and tile table) and phase C's
        // transcript-major copy with the classes' permuted positions (k_csc_remap)
        const uint64_t C = em->prob.C, Lnz = em->L, S8 = Lnz / 8 + 2 * (uint64_t)nt + 2;      // (a tile's stream is padded to whole chunks of 8)
        SF_HIP(pool_malloc(&em->recs, (C + S8 + 4 * (uint64_t)nt + 4) * 16)); SF_HIP(pool_malloc(&em->ovc, S8 * 2));
        SF_HIP(pool_malloc(&em->tp, (size_t)nt * sizeof(TilePack)));
        SF_HIP(pool_malloc(&em->cnt8, (C ? C : 1) * 4)); SF_HIP(pool_malloc(&em->cpos, (C ? C : 1) * 4)); SF_HIP(pool_malloc(&em->esc_cls_p, En * 4));
        {
            // (staging the tile's slots in LDS was measured: 82 -> 98 us on cfg3 -- the 36 KB leave one block per CU where three ran; lds_slots = 0
            //  keeps the reads in memory, the kernel still takes the staged form for a build that asks for it)
            const uint32_t want = (em->tile_nnz + 7u) & ~7u, cap_slots = (160u * 1024u - 50u * 1024u) / 2u;
            const bool stage = []() { const char* e = SF_DEV_ENV("SFGPU_EM_PACK_LDS"); return e && atoi(e) != 0; }();
            const uint32_t lds_slots = (stage && want <= cap_slots) ? want : 0u;
            if (lds_slots) SF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pack_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_slots * 2u + 16u)));
            hipLaunchKernelGGL(k_pack_build, dim3(nt), dim3(kSweepBlock), lds_slots ? lds_slots * 2u + 16u : 0u, s2, p_rowptr, em->tile_c0, em->tile_s0,
                               reinterpret_cast<const uint16_t*>(em->lstream), em->td, em->esc_cls, em->recs, em->ovc, em->tp, em->cpos, em->esc_cls_p, lds_slots);
        }
        SF_HIP(pool_malloc(&em->cscp, (Lnz / 8 + 2112ull * nt + (uint64_t)nt + 8) * 16));      // (a tile: at most n8 / 8 + 2046 chunks, one partial chunk per key)
        hipLaunchKernelGGL(k_cscp_build, dim3(nt), dim3(kSweepBlock), 0, s2, em->td, em->tp, em->cpos, em->kv_tmp, em->idx_tmp, em->tin_tmp, em->tile_s0, em->cscp, em->null_cls);
        SF_CHECK_LAUNCH();
    }
    if (E) {
        uint64_t *k_in = nullptr, *k_out = nullptr, *k2_in = nullptr, *k2_out = nullptr, *gsum = nullptr;
        uint32_t *v_in = nullptr, *v_out = nullptr, *head = nullptr, *esc_g = nullptr;
        struct Scratch {
            void** slots[9]; hipStream_t st;
            ~Scratch() { void* ps[9]; for (int i = 0; i < 9; ++i) ps[i] = *slots[i]; pool_free_on_many(ps, 9, st); }
        } scratch{{(void**)&k_in, (void**)&k_out, (void**)&k2_in, (void**)&k2_out, (void**)&gsum, (void**)&v_in, (void**)&v_out, (void**)&head, (void**)&esc_g}, st};
        SF_HIP(pool_malloc(&k_in, E * 8)); SF_HIP(pool_malloc(&k_out, E * 8)); SF_HIP(pool_malloc(&k2_in, E * 8)); SF_HIP(pool_malloc(&k2_out, E * 8));
        SF_HIP(pool_malloc(&gsum, (E + 2) * 8)); SF_HIP(pool_malloc(&v_in, E * 4)); SF_HIP(pool_malloc(&v_out, E * 4));
        SF_HIP(pool_malloc(&head, (E + 2) * 4)); SF_HIP(pool_malloc(&esc_g, E * 4));
        SF_HIP(pool_malloc(&em->esc_far, E * 4)); SF_HIP(pool_malloc(&em->far_pos, E * 4)); SF_HIP(pool_malloc(&em->far_xi, E * 4));
        SF_HIP(pool_malloc(&em->ft_list, E * 4)); SF_HIP(pool_malloc(&em->ftgt, M * 8));
        SF_HIP(hipMemsetAsync(em->ftgt, 0, M * 8, st));
        int tbits = 1; while (tbits < 31 && (1u << tbits) < nt) ++tbits;
        int pbits = 1; while (pbits < 32 && (1ull << pbits) < M) ++pbits;
        hipLaunchKernelGGL(k_far_keys, dim3(nt), dim3(kEmBlock), 0, st, em->td, em->inv ? em->esc_pos : em->esc_id, k_in, v_in);
        int rc = sort_pairs_u64_u32(k_in, k_out, v_in, v_out, E, st, 32 + tbits, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_far_heads, dim3(blocks_for(E + 1)), dim3(kEmBlock), 0, st, E, k_out, head);
        if ((rc = exclusive_scan_u32(head, gsum, E, st, false))) return rc;
        hipLaunchKernelGGL(k_far_assign, dim3(blocks_for(E)), dim3(kEmBlock), 0, st, E, k_out, v_out, head, gsum, esc_g, em->far_pos, k2_in);
        hipLaunchKernelGGL(k_far_tiles, dim3((nt + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, st, nt, E, k_out, gsum, em->td, em->pflags);
        hipLaunchKernelGGL(k_far_local, dim3(nt), dim3(kEmBlock), 0, st, em->td, esc_g, em->esc_far);
        if ((rc = sort_pairs_u64_u32(k2_in, k2_out, v_in, v_out, E, st, 32 + pbits, false))) return rc;     // (values unused)
        hipLaunchKernelGGL(k_ft_ranges, dim3(blocks_for(E)), dim3(kEmBlock), 0, st, E, gsum, k2_out, em->ftgt, em->ft_list);
        hipLaunchKernelGGL(k_far_xi, dim3(blocks_for(E)), dim3(kEmBlock), 0, st, E, gsum, em->far_pos, em->ftgt, em->cov2, em->far_xi, em->pflags);
        SF_CHECK_LAUNCH();
    }
    // [control words + status | part0 | part1 | far0 | far1 | xpub], every piece 256-byte aligned
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    em->xbuf_bytes = up((size_t)kCtlWords * 8 + 64 + sizeof(PersistCold)) + 2 * up((size_t)(P ? P : 1) * 16) + 3 * up((size_t)En * 16);
    // The exchange buffer lives in UNCACHED device memory (round 6): what the tiles hand each other must never be served from an XCD's L2.
    // In ordinary (coarse-grained) pool memory that was a property of the sc1 forms observed on one stream -- and with other streams'
    // kernels in flight tiles did poll stale lines (profiles/r5_em_notes.md 5).  The sc1 forms stay (they also skip the L1).
    {
        const bool in_pool = []() { const char* e = getenv("SFGPU_EM_XBUF"); return e && strcmp(e, "pool") == 0; }();
        if (!in_pool && uncached_malloc(&em->xbuf, em->xbuf_bytes) == hipSuccess) em->xbuf_uncached = true;
        else { (void)hipGetLastError(); em->xbuf = nullptr; SF_HIP(pool_malloc(&em->xbuf, em->xbuf_bytes)); }
    }
    return SFGPU_OK;
}

int sfgpu_em_create(sfgpu_em** out, const sfgpu_problem* prob, sfgpu_stream stream) {
    SF_REQUIRE(out && prob, SFGPU_ERR_INVALID, "sfgpu_em_create: null pointer");
    SF_REQUIRE(prob->M > 0 && prob->d_len, SFGPU_ERR_INVALID, "sfgpu_em_create: need M > 0 and d_len");
    SF_REQUIRE(prob->M <= (1ull << 31), SFGPU_ERR_RANGE, "sfgpu_em_create: transcript ids must fit 31 bits");
    SF_REQUIRE(prob->C == 0 || (prob->d_rowptr && prob->d_ids && prob->d_counts), SFGPU_ERR_INVALID,
               "sfgpu_em_create: null CSR pointer");
    sfgpu_em* em = new sfgpu_em();
    em->prob = *prob;
    em->user_stream = as_stream(stream);
    const uint64_t M = prob->M, C = prob->C;
    int nb = (int)((M + kEmBlock - 1) / kEmBlock);
    em->nb = nb < 1 ? 1 : (nb > kMaxPartials ? kMaxPartials : nb);
#define EM_TRY(expr)                                                                                 \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                              \
        set_error("%s failed: %s", #expr, hipGetErrorString(_e)); em_free(em); return SFGPU_ERR_HIP; } } while (0)
    EM_TRY(stream_acquire(&em->stream));
    em->cur = em->stream;
    EM_TRY(hipEventCreate(&em->ev_a)); EM_TRY(hipEventCreate(&em->ev_b));
    EM_TRY(hipEventCreateWithFlags(&em->ev_join, hipEventDisableTiming));
    EM_TRY(hipEventCreateWithFlags(&em->ev_plan, hipEventDisableTiming));
    for (hipEvent_t& e : em->ev_poll) EM_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    EM_TRY(pinned_malloc(&em->h_mirror, 128));
    memset(em->h_mirror, 0, 128);
    EM_TRY(pinned_malloc(&em->h_plan, 128));
    memset(em->h_plan, 0, 128);
    EM_TRY(pool_malloc(&em->alpha, M * 8)); EM_TRY(pool_malloc(&em->alpha_out, M * 8));
    EM_TRY(pool_malloc(&em->x, M * 8)); EM_TRY(pool_malloc(&em->lenc, M * 8)); EM_TRY(pool_malloc(&em->scratch, M * 8));
    EM_TRY(pool_malloc(&em->partials, kMaxPartials * 8)); EM_TRY(pool_malloc(&em->sum_partials, kMaxPartials * 8));
    EM_TRY(pool_malloc(&em->counts32, (C ? C : 1) * 4));
    EM_TRY(pool_malloc(&em->d_state, sizeof(EmState)));
    EM_TRY(pool_malloc(&em->blkmax, 2 * kMaxPartials * 8));
    EM_TRY(pinned_malloc(&em->h_blkmax, 2 * kMaxPartials * 8));
    EM_TRY(pool_malloc(&em->tile_lo, 4));   // sized once nnz is known (below)
    EM_TRY(pinned_malloc(&em->h_state, sizeof(EmState)));
    memset(em->h_state, 0, sizeof(EmState));
    EM_TRY(hipMemsetAsync(em->d_state, 0, sizeof(EmState), em->cur));
    int rc = em_join_user(em);
    if (rc) { em_free(em); return rc; }
    hipLaunchKernelGGL(k_clamp_len, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, prob->d_len, em->lenc);
    uint32_t rp_end = 0;
    if (C) {
        unsigned int* ovf = reinterpret_cast<unsigned int*>(em->partials);
        EM_TRY(hipMemsetAsync(ovf, 0, 4, em->cur));
        hipLaunchKernelGGL(k_narrow_counts, dim3(blocks_for(C)), dim3(kEmBlock), 0, em->cur, C, prob->d_counts,
                           prob->d_rowptr, em->counts32, ovf, (const uint32_t*)nullptr);
        // (pinned destinations: the two copies are queued behind the kernel and cost ONE wait)
        unsigned int* hp = reinterpret_cast<unsigned int*>(em->h_plan);
        EM_TRY(hipMemcpyAsync(hp, ovf, 4, hipMemcpyDeviceToHost, em->cur));
        EM_TRY(hipMemcpyAsync(hp + 1, prob->d_rowptr + C, 4, hipMemcpyDeviceToHost, em->cur));
        EM_TRY(hipStreamSynchronize(em->cur));
        const unsigned int h_ovf = hp[0];
        rp_end = hp[1];
        if (h_ovf & 2u) { set_error("sfgpu_em_create: rowptr not strictly ascending (a class without members)"); em_free(em); return SFGPU_ERR_INVALID; }
        if (h_ovf) { set_error("sfgpu_em_create: a class count >= 2^31"); em_free(em); return SFGPU_ERR_RANGE; }
    } else {
        EM_TRY(hipStreamSynchronize(em->cur));
    }
    em->L = rp_end;
    if (C) {   // nnz-balanced tile plan of the sweep, its label stream, and the cover lists of the fold
        // Two 1024-thread blocks are resident per CU (LDS), so the tiles come in rounds of 2 x #CU, and a round costs
        // about the same whether its tiles hold 5 000 or 18 000 nonzeros (it is latency, not volume: cfg3's 9.3 M
        // nonzeros take 26.1 us as 2.3 rounds of 8 000 and 21.9 us as one round of 18 300).  So: as few rounds as
        // possible, tiles of equal size.  What bounds a tile is its number of CLASSES (the 13-bit class field of the
        // stream words, the size of den[]), not its nonzeros: the plan is checked and, if a tile holds more than
        // kTileNnz classes, redone with one more round -- down to kTileNnz nonzeros per tile, which cannot fail.
        int n_cu = 256;
        { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); }
        const uint64_t slots = 2ull * (uint64_t)(n_cu > 0 ? n_cu : 256);
        auto tile_for = [&](uint64_t rounds) -> uint32_t {
            uint64_t want = ((uint64_t)rp_end + slots * rounds - 1) / (slots * rounds);
            if (want < 2048) want = 2048;
            if (want > (uint64_t)kTileNnzMax) want = kTileNnzMax;
            return (uint32_t)want;
        };
        // first guess: rounds such that a tile holds at most kTileNnzMax nonzeros and about 3/4 kTileNnz classes
        uint64_t rounds = std::max<uint64_t>(1, ((uint64_t)rp_end + slots * kTileNnzMax - 1) / (slots * kTileNnzMax));
        rounds = std::max<uint64_t>(rounds, (C + slots * (kTileNnz * 3 / 4) - 1) / (slots * (kTileNnz * 3 / 4)));
        uint32_t tile_nnz = tile_for(rounds);
        if (const char* e = SF_DEV_ENV("SFGPU_EM_TILE")) { long v = atol(e); if (v >= 2048 && v <= kTileNnzMax) tile_nnz = (uint32_t)v; }   // tuning
        em->n_tiles = (uint32_t)std::max<uint64_t>(1, ((uint64_t)rp_end + tile_nnz - 1) / tile_nnz);
        const uint32_t nt_small = (uint32_t)std::max<uint64_t>(1, ((uint64_t)rp_end + kTileNnz - 1) / kTileNnz);
        const uint32_t nt_cap = std::max(em->n_tiles, nt_small);
        uint32_t nt = em->n_tiles;
        uint32_t *t_len8 = nullptr, *t_nesc = nullptr;
        // the class table the plan is made from: the caller's, or (plan_state 1) the renumbered copy of em_renumber
        const uint32_t* p_rowptr = prob->d_rowptr; const uint32_t* p_ids = prob->d_ids;
        uint32_t *rowptr2 = nullptr, *vids = nullptr;
        int plan_state = 0;                  // 0: caller's order, 1: renumbered (trial), 2: caller's order after a trial that did not pay
        uint64_t E_first = 0;
        const uint64_t rounds0 = rounds; const uint32_t tile_nnz0 = tile_nnz, nt0 = nt;
        uint64_t P = 0, S = 0, E = 0;
        bool lo_monotone = false;                            // the tiles' `lo` never decreases (k_tile_mono): the cover lists need no sort
        pool_free(em->tile_lo); em->tile_lo = nullptr;
        EM_TRY(pool_malloc(&em->tile_lo, (size_t)nt_cap * 4));
        EM_TRY(pool_malloc(&em->tile_span, ((size_t)nt_cap + 1) * 4));
        EM_TRY(pool_malloc(&em->tile_c0, ((size_t)nt_cap + 1) * 4));
        EM_TRY(pool_malloc(&em->tile_off, ((size_t)nt_cap + 1) * 8));
        EM_TRY(pool_malloc(&em->tile_s0, ((size_t)nt_cap + 1) * 8));
        EM_TRY(pool_malloc(&em->tile_esc0, ((size_t)nt_cap + 1) * 8));
        EM_TRY(pool_malloc(&em->cov_ptr, ((size_t)M + 1) * 4));
        EM_TRY(pool_malloc(&em->tsum, (size_t)nt_cap * 8));
        EM_TRY(hipMemsetAsync(em->tsum, 0, (size_t)nt_cap * 8, em->cur));      // empty tiles never write theirs
      for (;;) {                             // (the plan; twice or three times when the transcripts are renumbered)
        rounds = rounds0; tile_nnz = tile_nnz0; nt = nt0; em->n_tiles = nt0;
        EM_TRY(pool_malloc(&t_len8, ((size_t)nt_cap + 1) * 4)); EM_TRY(pool_malloc(&t_nesc, ((size_t)nt_cap + 1) * 4));
        // (the check of the tiles' class counts rides on the read-back of the plan's sizes: the window pass and the scans below run on a
        //  plan that may have to be redone with one more round -- rare, and harmless: they do not depend on the class counts)
        for (;;) {
            hipLaunchKernelGGL(k_tile_plan, dim3((nt + 1 + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, em->cur, C, nt, tile_nnz,
                               p_rowptr, em->tile_c0);
            hipLaunchKernelGGL(k_tile_window, dim3(nt), dim3(kEmBlock), 0, em->cur, p_rowptr, p_ids, em->tile_c0,
                               em->tile_lo, em->tile_span, t_len8, t_nesc);
            // (one block: the three scans over the tiles, the plan's checks, and everything the host reads posted into pinned memory)
            hipLaunchKernelGGL(k_tile_scans, dim3(1), dim3(1024), 0, em->cur, nt, em->tile_c0, em->tile_lo, em->tile_span, t_len8, t_nesc,
                               em->tile_off, em->tile_s0, em->tile_esc0, em->h_plan);
            EM_TRY(hipGetLastError());
            EM_TRY(hipStreamSynchronize(em->cur));
            const uint32_t most = *reinterpret_cast<const uint32_t*>(em->h_plan + 5);
            lo_monotone = reinterpret_cast<const uint32_t*>(em->h_plan + 5)[1] == 0u;
            em->null_cls = most;                             // (<= kTileNnz when the loop ends: a tile holds no more classes than nonzeros)
            if (tile_nnz <= (uint32_t)kTileNnz || most <= (uint32_t)kTileNnz) break;
            ++rounds;
            tile_nnz = tile_for(rounds);
            if (tile_nnz < (uint32_t)kTileNnz) tile_nnz = kTileNnz;
            nt = (uint32_t)std::max<uint64_t>(1, ((uint64_t)rp_end + tile_nnz - 1) / tile_nnz);
            if (nt > nt_cap) { nt = nt_small; tile_nnz = kTileNnz; }
            em->n_tiles = nt;
        }
        { void* ps[2] = {t_len8, t_nesc}; pool_free_on_many(ps, 2, em->cur); }
        P = em->h_plan[1]; S = em->h_plan[2]; E = em->h_plan[3];
        // Many members outside their tile's window (an index whose isoforms are not adjacent): let the plan order the transcripts
        // itself, and keep that order if it removes at least 40 % of the escapes.
        if (plan_state == 0 && C >= kRenumberMinClasses && E * 8 > (uint64_t)rp_end && getenv("SFGPU_EM_NO_RENUMBER") == nullptr) {
            E_first = E;
            if (em_renumber(em, prob, rp_end, &rowptr2, &vids) == SFGPU_OK) { p_rowptr = rowptr2; p_ids = vids; plan_state = 1; continue; }
            (void)hipGetLastError();
        } else if (plan_state == 1 && E * 10 > E_first * 6) {
            (void)hipStreamSynchronize(em->cur);
            pool_free(rowptr2); pool_free(vids); pool_free(em->inv); pool_free(em->cperm);
            rowptr2 = vids = nullptr; em->inv = em->cperm = nullptr;
            unsigned int* ovf = reinterpret_cast<unsigned int*>(em->partials);
            hipLaunchKernelGGL(k_narrow_counts, dim3(blocks_for(C)), dim3(kEmBlock), 0, em->cur, C, prob->d_counts, prob->d_rowptr, em->counts32, ovf,
                               (const uint32_t*)nullptr);
            p_rowptr = prob->d_rowptr; p_ids = prob->d_ids; plan_state = 2;
            continue;
        }
        break;
      }
        if (plan_state == 1) log_msg(0, "EM plan: transcripts renumbered by co-occurrence, %llu -> %llu of %u members outside their window",
                                     (unsigned long long)E_first, (unsigned long long)E, rp_end);
        if (P >= (1ull << 32)) { set_error("sfgpu_em_create: window slots exceed 2^32"); em_free(em); return SFGPU_ERR_RANGE; }
        em->P = P; em->E = E; em->tile_nnz = tile_nnz;
        if (env_timing()) fprintf(stderr, "em plan: %u tiles (%u nnz each), P = %llu window slots (%.2f per transcript), %llu escapes of %llu nonzeros\n",
                                            nt, tile_nnz, (unsigned long long)P, (double)P / (double)M, (unsigned long long)E, (unsigned long long)rp_end);
        EM_TRY(pool_malloc(&em->lstream, (S ? S : 1) * 4 + 32));
        EM_TRY(pool_malloc(&em->esc_id, (E ? E : 1) * 4));
        EM_TRY(pool_malloc(&em->esc_cls, (E ? E : 1) * 4));
        {
            const char* eg = getenv("SFGPU_EM_GATHER");
            em->gather = (!eg || atoi(eg) != 0) && (uint64_t)rp_end > E;
        }
        if (!em->gather) {
            hipLaunchKernelGGL(k_fill_stream, dim3(nt), dim3(kEmBlock), 0, em->cur, p_rowptr, p_ids, em->tile_c0,
                               em->tile_lo, em->tile_s0, em->tile_esc0, em->lstream, em->esc_id, em->esc_cls, em->inv);
            EM_TRY(hipGetLastError());
        } else {
            // the compact class-major stream and the transcript-major copy for phase C: k_tile_build sorts every tile's nonzeros
            // by window slot, the rest lays them out in chunks of 8 sixteen-bit entries, pure chunks first.  SFGPU_EM_GATHER=0
            // keeps the scatter form.
            const uint64_t Lnz = rp_end;
            uint32_t *tmp = nullptr, *kv = nullptr, *idx = nullptr, *tin = nullptr, *chunks = nullptr, *pure = nullptr;
            uint64_t *cb = nullptr, *ps = nullptr;
            struct Scratch {                                  // the plan's temporaries go back to the pool on every way out
                void** slots[5]; hipStream_t st;
                ~Scratch() { void* ps[5]; for (int i = 0; i < 5; ++i) ps[i] = *slots[i]; pool_free_on_many(ps, 5, st); }
            } scratch{{(void**)&tmp, (void**)&chunks, (void**)&pure, (void**)&cb, (void**)&ps}, em->cur};
            // (kv / idx / tin -- the tiles' nonzeros sorted by window slot -- live until em_persist_plan has made phase C's chunks from them)
            EM_TRY(pool_malloc(&em->chdr, (S / 8 + 1) * 4));
            // G: the number of chunks is only known on the device (cb[nt]); its bound -- every tile ends in a partial chunk --
            // sizes the arrays, and the flags behind the last real chunk stay 0, so no readback holds the plan up
            const uint64_t G = Lnz / 8 + nt + 1;
            EM_TRY(pool_malloc(&tmp, Lnz * 4)); EM_TRY(pool_malloc(&em->kv_tmp, Lnz * 4));
            EM_TRY(pool_malloc(&em->idx_tmp, ((size_t)nt + 2) * 4)); EM_TRY(pool_malloc(&em->tin_tmp, ((size_t)nt + 2) * 4));
            kv = em->kv_tmp; idx = em->idx_tmp; tin = em->tin_tmp;
            EM_TRY(pool_malloc(&chunks, ((size_t)nt + 2) * 4)); EM_TRY(pool_malloc(&cb, ((size_t)nt + 3) * 8));
            EM_TRY(pool_malloc(&em->tile_qb, ((size_t)nt + 2) * 8)); EM_TRY(pool_malloc(&em->tile_np, ((size_t)nt + 2) * 4)); EM_TRY(pool_malloc(&em->tile_pr, ((size_t)nt + 2) * 4));
            constexpr size_t kBuildLds = (size_t)kBuildWaves * kBuildBins * 4;
            EM_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildLds));
            hipLaunchKernelGGL(k_tile_build, dim3(nt), dim3(kBuildBlock), kBuildLds, em->cur, p_rowptr, p_ids, em->tile_c0, em->tile_lo, em->tile_s0, em->tile_esc0,
                               reinterpret_cast<uint16_t*>(em->lstream), em->chdr, em->esc_id, em->esc_cls, em->inv, tmp, kv, idx, tin, chunks);
            EM_TRY(hipGetLastError());
            // From here on the plan runs on TWO streams (round 6): the transcript-major copy of the sweep kernels (two scans, three kernels) and,
            // in em_persist_plan, the persistent loop's records and chunks go to a side stream; the cover lists, the tile records, the
            // overlap tables and the far tables stay on this one.  The two join when the plan is complete (below).
            hipStream_t cs = em->cur;
            {
                bool ok = em->side || stream_acquire(&em->side) == hipSuccess;
                ok = ok && (em->ev_fork || hipEventCreateWithFlags(&em->ev_fork, hipEventDisableTiming) == hipSuccess);
                ok = ok && (em->ev_side || hipEventCreateWithFlags(&em->ev_side, hipEventDisableTiming) == hipSuccess);
                ok = ok && (em->ev_td || hipEventCreateWithFlags(&em->ev_td, hipEventDisableTiming) == hipSuccess);
                ok = ok && hipEventRecord(em->ev_fork, em->cur) == hipSuccess && hipStreamWaitEvent(em->side, em->ev_fork, 0) == hipSuccess;
                if (ok) { cs = em->side; em->side_live = true; } else (void)hipGetLastError();
            }
            scratch.st = cs;
            int cr = exclusive_scan_u32(chunks, cb, nt, cs, false);
            if (!cr) {
                EM_TRY(pool_malloc(&pure, (G + 2) * 4)); EM_TRY(pool_malloc(&ps, (G + 3) * 8));
                EM_TRY(hipMemsetAsync(pure, 0, (G + 2) * 4, cs));
                hipLaunchKernelGGL(k_csc_pure, dim3(nt), dim3(kEmBlock), 0, cs, kv, idx, tin, cb, pure);
                cr = exclusive_scan_u32(pure, ps, G, cs, false);
            }
            if (!cr) {
                hipLaunchKernelGGL(k_csc_offsets, dim3((nt + 1 + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, cs, nt, cb, ps, em->tile_qb, em->tile_np, em->tile_pr);
                EM_TRY(pool_malloc(&em->csc, 32 * (G ? G : 1) + 32));                  // (every chunk mixed: the upper bound)
                EM_TRY(pool_malloc(&em->csc_slot0, (G + 1) * 2));
                hipLaunchKernelGGL(k_csc_write, dim3(nt), dim3(kEmBlock), 0, cs, kv, idx, tin, cb, ps, em->tile_qb, em->tile_np, em->csc, em->csc_slot0, em->null_cls);
                EM_TRY(hipGetLastError());
            }
            if (cr) { em_free(em); return cr; }
        }
        // (rowptr2 / vids, a renumbered plan's class table, are freed at the end: the persistent loop's plan reads the plan's rowptr)
        EM_TRY(pool_malloc(&em->partial, (P ? P : 1) * 8));
        EM_TRY(pool_malloc(&em->cov_pos, (P ? P : 1) * 4));
        EM_TRY(pool_malloc(&em->pub_pos, (P ? P : 1) * 4));
        EM_TRY(hipMemsetAsync(em->partial, 0, (P ? P : 1) * 8, em->cur));
        const bool cover_by_sort = getenv("SFGPU_EM_COVER_SORT") != nullptr;             // (dev / tests: the sort form whatever the plan)
        if (P && lo_monotone && !cover_by_sort) {
            uint32_t* cnt = nullptr;
            EM_TRY(pool_malloc(&cnt, ((size_t)M + 2) * 4));
            EM_TRY(hipMemsetAsync(cnt, 0, ((size_t)M + 2) * 4, em->cur));
            hipLaunchKernelGGL(k_cov_count, dim3(nt), dim3(kEmBlock), 0, em->cur, em->tile_lo, em->tile_span, em->inv, cnt);
            int cs = exclusive_scan_u32_u32(cnt, em->cov_ptr, M, em->cur);
            if (!cs) hipLaunchKernelGGL(k_cov_fill, dim3(nt), dim3(kEmBlock), 0, em->cur, em->tile_lo, em->tile_span, em->tile_off, em->inv, em->cov_ptr,
                                        em->cov_pos, em->pub_pos);
            pool_free_on(cnt, em->cur);
            if (cs) { em_free(em); return cs; }
            if (getenv("SFGPU_EM_COVER_CHECK")) {
                // dev / tests: the sorted form next to it, compared word for word on the host
                uint64_t *k_in = nullptr, *k_out = nullptr; uint32_t *v_in = nullptr, *pos2 = nullptr, *ptr2 = nullptr;
                EM_TRY(pool_malloc(&k_in, P * 8)); EM_TRY(pool_malloc(&k_out, P * 8)); EM_TRY(pool_malloc(&v_in, P * 4));
                EM_TRY(pool_malloc(&pos2, P * 4)); EM_TRY(pool_malloc(&ptr2, ((size_t)M + 1) * 4));
                hipLaunchKernelGGL(k_cover_pairs, dim3(nt), dim3(kEmBlock), 0, em->cur, em->tile_lo, em->tile_span, em->tile_off, k_in, v_in, em->inv);
                int bits = 1; while (bits < 32 && (1ull << bits) <= M) ++bits;
                int src = sort_pairs_u64_u32(k_in, k_out, v_in, pos2, P, em->cur, bits, true);
                if (!src) hipLaunchKernelGGL(k_cover_ptr, dim3(blocks_for(M + 1)), dim3(kEmBlock), 0, em->cur, M, P, k_out, ptr2);
                std::vector<uint32_t> a(P), b(P), c((size_t)M + 1), d((size_t)M + 1);
                (void)hipStreamSynchronize(em->cur);
                (void)hipMemcpy(a.data(), em->cov_pos, P * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), pos2, P * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(c.data(), em->cov_ptr, ((size_t)M + 1) * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(d.data(), ptr2, ((size_t)M + 1) * 4, hipMemcpyDeviceToHost);
                pool_free(k_in); pool_free(k_out); pool_free(v_in); pool_free(pos2); pool_free(ptr2);
                if (src || a != b || c != d) { set_error("sfgpu_em_create: the cover lists built without a sort differ from the sorted ones"); em_free(em); return SFGPU_ERR_STATE; }
            }
        } else if (P) {
            uint64_t *k_in = nullptr, *k_out = nullptr; uint32_t* v_in = nullptr;
            EM_TRY(pool_malloc(&k_in, P * 8)); EM_TRY(pool_malloc(&k_out, P * 8)); EM_TRY(pool_malloc(&v_in, P * 4));
            hipLaunchKernelGGL(k_cover_pairs, dim3(nt), dim3(kEmBlock), 0, em->cur, em->tile_lo, em->tile_span, em->tile_off,
                               k_in, v_in, em->inv);
            int bits = 1; while (bits < 32 && (1ull << bits) <= M) ++bits;
            int src = sort_pairs_u64_u32(k_in, k_out, v_in, em->cov_pos, P, em->cur, bits, false);     // (synchronised below, before the frees)
            if (!src) {
                hipLaunchKernelGGL(k_invert_perm, dim3(blocks_for(P)), dim3(kEmBlock), 0, em->cur, P, em->cov_pos, em->pub_pos);
                hipLaunchKernelGGL(k_cover_ptr, dim3(blocks_for(M + 1)), dim3(kEmBlock), 0, em->cur, M, P, k_out, em->cov_ptr);
            }
            // (no host wait: the scratch goes back to the pool when the stream has passed this point)
            { void* ps[3] = {k_in, k_out, v_in}; pool_free_on_many(ps, 3, em->cur); }
            if (src) { em_free(em); return src; }
        } else {
            EM_TRY(hipMemsetAsync(em->cov_ptr, 0, ((size_t)M + 1) * 4, em->cur));
        }
        EM_TRY(pool_malloc(&em->td, (size_t)nt * sizeof(TileDesc)));
        hipLaunchKernelGGL(k_tile_desc, dim3((nt + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, em->cur, nt, em->tile_c0, em->tile_lo, em->tile_span,
                           em->tile_s0, em->tile_esc0, em->tile_off, (const uint64_t*)nullptr, em->tile_np, em->tile_pr, em->td);
        if (em->gather) {      // (the copy's fields: behind the records on this stream, behind the copy's offsets on the side stream)
            hipStream_t cs = em->cur;
            if (em->side_live) { EM_TRY(hipEventRecord(em->ev_td, em->cur)); EM_TRY(hipStreamWaitEvent(em->side, em->ev_td, 0)); cs = em->side; }
            hipLaunchKernelGGL(k_tile_desc_csc, dim3((nt + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, cs, nt, em->tile_qb, em->tile_np, em->tile_pr, em->td);
        }
        EM_TRY(hipGetLastError());
        if (em->gather) {
            // what the FUSED iteration (k_sweep_lds<.., .., true>) needs besides: two slot-major arrays of window sums, two more escape
            // accumulators, the wavefronts' maxima, the tiles' overlap tables, the cover lists by position + the list of positions no
            // window holds, the far members' inline cover slots; for a plan with an order of its own also the far members as positions
            // and room for alpha / effLen in that order.  The tables' verdict (flags) is read back behind them without a wait;
            // optimize() looks at it.
            const uint64_t Pn = P ? P : 1;
            EM_TRY(pool_malloc(&em->partial_a, Pn * 8)); EM_TRY(pool_malloc(&em->partial_b, Pn * 8));
            EM_TRY(pool_malloc(&em->aout_b, M * 8)); EM_TRY(pool_malloc(&em->aout_c, M * 8));
            EM_TRY(pool_malloc(&em->tmax, 2ull * nt * (kSweepBlock / kWave) * 8));
            EM_TRY(pool_malloc(&em->unc, (M + 2) * 4)); EM_TRY(pool_malloc(&em->cov2, M * 8));
            EM_TRY(hipMemsetAsync(em->unc + M, 0, 8, em->cur));
            hipLaunchKernelGGL(k_nb_table, dim3((nt + kEmBlock - 1) / kEmBlock), dim3(kEmBlock), 0, em->cur, nt, em->tile_lo, em->tile_span,
                               em->tile_off, em->td, em->unc + M + 1);
            hipLaunchKernelGGL(k_cov_by_pos, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, em->inv, em->cov_ptr, em->cov2, em->unc, em->unc + M);
            EM_TRY(pool_malloc(&em->esc_slots, (E ? E : 1) * 8));
            if (E) hipLaunchKernelGGL(k_esc_slots, dim3(blocks_for(E)), dim3(kEmBlock), 0, em->cur, E, em->esc_id, em->cov_ptr, em->cov_pos, em->esc_slots);
            if (em->inv) {
                uint32_t* pos_of = nullptr;
                EM_TRY(pool_malloc(&pos_of, M * 4)); EM_TRY(pool_malloc(&em->esc_pos, (E ? E : 1) * 4));
                EM_TRY(pool_malloc(&em->alphaP, M * 8)); EM_TRY(pool_malloc(&em->lencP, M * 8));
                hipLaunchKernelGGL(k_invert_perm, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, em->inv, pos_of);
                if (E) hipLaunchKernelGGL(k_map_u32, dim3(blocks_for(E)), dim3(kEmBlock), 0, em->cur, E, em->esc_id, pos_of, em->esc_pos);
                pool_free_on(pos_of, em->cur);
            }
            EM_TRY(hipGetLastError());
            {   // the persistent loop's tables (em_persist.h); its verdict rides on the read-back below
                const int pr = em_persist_plan(em, nt, E, P, p_rowptr);
                if (pr) { em_free(em); return pr; }
                if (em->side_live) { EM_TRY(hipEventRecord(em->ev_side, em->side)); EM_TRY(hipStreamWaitEvent(em->cur, em->ev_side, 0)); em->side_live = false; }
                { void* ps3[3] = {em->kv_tmp, em->idx_tmp, em->tin_tmp}; pool_free_on_many(ps3, 3, em->cur); em->kv_tmp = em->idx_tmp = em->tin_tmp = nullptr; }
            }
            EM_TRY(hipMemcpyAsync(em->h_plan + 4, em->unc + M + 1, 4, hipMemcpyDeviceToHost, em->cur));
            if (em->pflags) EM_TRY(hipMemcpyAsync(em->h_plan + 8, em->pflags, 16, hipMemcpyDeviceToHost, em->cur));
            EM_TRY(hipEventRecord(em->ev_plan, em->cur));
            if (env_timing()) {                     // dev: how many tiles go by the cover list, how many neighbours the others have
                std::vector<TileDesc> h(nt);
                (void)hipStreamSynchronize(em->cur);
                (void)hipMemcpy(h.data(), em->td, (size_t)nt * sizeof(TileDesc), hipMemcpyDeviceToHost);
                uint32_t by_list = 0; uint64_t nb_sum = 0, span_sum = 0;
                for (const TileDesc& t : h) { if (t.nb_n == kNbByList) ++by_list; else nb_sum += t.nb_n; span_sum += t.span; }
                uint32_t hist[kNbMax + 1] = {};
                for (const TileDesc& t : h) if (t.nb_n <= (uint32_t)kNbMax) ++hist[t.nb_n];
                fprintf(stderr, "em fused plan: %u of %u tiles by the cover list, %.2f overlapping tiles on average for the others (0..6: %u %u %u %u %u %u %u), mean span %.0f\n", by_list, nt,
                        nt > by_list ? (double)nb_sum / (nt - by_list) : 0.0, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], (double)span_sum / nt);
            }
        }
        if (rowptr2) { pool_free_on(rowptr2, em->cur); pool_free_on(vids, em->cur); rowptr2 = vids = nullptr; }
    }
#undef EM_TRY
    *out = em;
    return SFGPU_OK;
}

int sfgpu_em_destroy(sfgpu_em* em) { em_free(em); return SFGPU_OK; }

double* sfgpu_em_alpha_out(sfgpu_em* em) { return em ? em->alpha_out : nullptr; }

static int em_begin_on(sfgpu_em* em, const sfgpu_em_opts* opts, hipStream_t work) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_begin: null handle");
    int rc = em_fill_opts(em, opts);
    if (rc) return rc;
    em->cur = work;
    em->fused = false; em->streamed = false;                 // (em_run decides)
    em->const_norm = getenv("SFGPU_EM_EXACT_NORM") == nullptr;
    em->vb_log_norm = digamma_pos((double)em->prob.M * kPriorAlpha + (double)em->prob.num_mapped);
    if (em->lenc_dirty) {
        hipLaunchKernelGGL(k_clamp_len, dim3(blocks_for(em->prob.M)), dim3(kEmBlock), 0, em->cur, em->prob.M, em->prob.d_len, em->lenc);
        SF_CHECK_LAUNCH();
        em->lenc_dirty = false;
    }
    SF_HIP(hipMemsetAsync(em->alpha_out, 0, em->prob.M * 8, em->cur));
    if (em->L) {
        hipLaunchKernelGGL(k_mark_active, dim3(blocks_for(em->L)), dim3(kEmBlock), 0, em->cur, em->L, em->prob.d_ids,
                           em->alpha_out);
        SF_CHECK_LAUNCH();
    }
    em->begun = true;
    return SFGPU_OK;
}

int sfgpu_em_begin(sfgpu_em* em, const sfgpu_em_opts* opts) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_begin: null handle");
    (void)hipStreamSynchronize(em->stream);    // nothing of a previous optimize() may be in flight
    em->in_optimize = false;
    return em_begin_on(em, opts, em->user_stream);
}

static int sfgpu_em_init_impl(sfgpu_em* em) {
    SF_REQUIRE(em && em->begun, SFGPU_ERR_STATE, "sfgpu_em_init: call begin first");
    const sfgpu_problem& p = em->prob;
    dim3 g(em->nb), b(kEmBlock);
    hipLaunchKernelGGL(k_count_active, g, b, 0, em->cur, p.M, em->alpha_out, em->partials);
    SF_CHECK_LAUNCH();
    double total = (double)p.num_mapped;   // :792
    if (em->opts.use_vbem) {
        hipLaunchKernelGGL(k_init_alpha<true>, g, b, 0, em->cur, p.M, em->alpha, em->alpha_out, em->x, em->lenc, total,
                           em->partials, em->sum_partials, em->nb, em->d_state);
        SF_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_vb_prepare, g, b, 0, em->cur, p.M, em->alpha, em->x, em->lenc, em->sum_partials, em->nb,
                           em->d_state, 1);
    } else {
        hipLaunchKernelGGL(k_init_alpha<false>, g, b, 0, em->cur, p.M, em->alpha, em->alpha_out, em->x, em->lenc, total,
                           em->partials, em->sum_partials, em->nb, em->d_state);
    }
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

int sfgpu_em_init(sfgpu_em* em) { return sfgpu_em_init_impl(em); }

double* sfgpu_em_alpha(sfgpu_em* em) { return em ? em->alpha : nullptr; }
double* sfgpu_em_lengths(sfgpu_em* em) { return em ? em->lenc : nullptr; }

int sfgpu_em_allow_persistent(int on) { g_allow_persist.store(on != 0, std::memory_order_relaxed); return SFGPU_OK; }

int sfgpu_em_set_bounds(sfgpu_em* em, uint32_t min_iter, uint32_t max_iter) {
    SF_REQUIRE(em && em->begun && !em->in_optimize, SFGPU_ERR_STATE, "sfgpu_em_set_bounds: piecewise API only, after begin");
    em->opts.min_iter = min_iter; em->opts.max_iter = max_iter;
    return SFGPU_OK;
}

// updateEqClassWeights (src/CollapsedEMOptimizer.cpp:527-555) for the piecewise loop: new lengths, x rebuilt from alpha
int sfgpu_em_rebase(sfgpu_em* em, const double* d_len) {
    SF_REQUIRE(em && em->begun && !em->in_optimize && d_len, SFGPU_ERR_STATE, "sfgpu_em_rebase: piecewise API only, after init");
    const sfgpu_problem& p = em->prob;
    dim3 g(em->nb), b(kEmBlock);
    hipLaunchKernelGGL(k_clamp_len, dim3(blocks_for(p.M)), b, 0, em->cur, p.M, d_len, em->lenc);
    em->lenc_dirty = true;
    if (em->opts.use_vbem) {
        hipLaunchKernelGGL(k_alpha_partials, g, b, 0, em->cur, p.M, em->alpha, em->sum_partials);
        hipLaunchKernelGGL(k_vb_prepare, g, b, 0, em->cur, p.M, em->alpha, em->x, em->lenc, em->sum_partials, em->nb, em->d_state, 1);
    } else {
        hipLaunchKernelGGL(k_x_from_alpha, dim3(blocks_for(p.M)), b, 0, em->cur, p.M, em->alpha, em->lenc, em->x);
    }
    SF_CHECK_LAUNCH();
    return SFGPU_OK;
}

int sfgpu_em_sweep(sfgpu_em* em) {
    SF_REQUIRE(em && em->begun, SFGPU_ERR_STATE, "sfgpu_em_sweep: call begin/init first");
    int rc = em_enqueue_sweep(em);
    return rc ? rc : em_enqueue_fold(em);
}

int sfgpu_em_update(sfgpu_em* em) {
    SF_REQUIRE(em && em->begun, SFGPU_ERR_STATE, "sfgpu_em_update: call begin/init first");
    return em_enqueue_update(em, false);
}

// `with_max`: also fetch the per-block maxima behind stats->max_rel_diff.  The loop of optimize() only needs the
// stop decision between chunks (one 48-byte copy); finish() fetches the maxima once at the end.
static int em_poll_impl(sfgpu_em* em, int* done, sfgpu_em_stats* stats, bool with_max) {
    SF_HIP(hipMemcpyAsync(em->h_state, em->d_state, sizeof(EmState), hipMemcpyDeviceToHost, em->cur));
    if (with_max) SF_HIP(hipMemcpyAsync(em->h_blkmax, em->blkmax, 2 * kMaxPartials * 8, hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    const EmState* h = em->h_state;
    uint32_t it = h->it_a;
    bool stop = false;
    if (it >= em->opts.min_iter) stop = (it >= em->opts.max_iter) || (it > 0 && h->notconv[(it - 1) & 1] == 0);
    if (done) *done = stop ? 1 : 0;
    em_stats_from_state(em, stats);
    return SFGPU_OK;
}

int sfgpu_em_poll(sfgpu_em* em, int* done, sfgpu_em_stats* stats) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_poll: null handle");
    return em_poll_impl(em, done, stats, true);
}

// The stop test of optimize()'s loop costs the device no idle time: a one-thread kernel behind chunk k (part of the chunk's
// graph) posts "iterations done | ended" into pinned host memory, an event marks the end of the chunk, and the host looks at
// chunk k's word only once chunk k + 1 is on the stream.  A device-to-host copy command per chunk plus a host that waits for
// it before launching again left the device idle ~22 us per chunk of 16 iterations (0.3 ms of cfg3's 232 iterations, 1.1 ms of
// cfg2's 818).  The chunk that is already enqueued when the end is seen consists of no-ops (the latch lives in device
// memory): ~80 us, once.
// Two words, one per event: the word of chunk k is only rewritten by chunk k + 2, which is enqueued after the host has read
// it -- every rank of a sharded run reads the same value at the same point of the loop, whatever its timing (a single word
// could already hold the next chunk's state on one rank and not on the other: they would leave the loop at different chunks,
// one of them inside an all-reduce).  The graph of em_run bakes its arguments and always posts into word 0: there a newer
// value only means the end is seen a chunk earlier.
static int em_enqueue_post(sfgpu_em* em, Launcher& L, int slot) {
    const EmState* st = em->d_state; uint32_t mn = em->opts.min_iter, mx = em->opts.max_iter; unsigned long long* m = em->h_mirror + 8 * slot;
    int fused = em->fused ? 1 : 0;
    void* args[] = {&st, &mn, &mx, &m, &fused};
    SF_HIP(L.launch(reinterpret_cast<const void*>(&k_post_state), dim3(1), dim3(1), args));
    return SFGPU_OK;
}
static int em_poll_start(sfgpu_em* em, int slot, bool post) {
    if (post) { Launcher L; L.stream = em->cur; int rc = em_enqueue_post(em, L, slot); if (rc) return rc; }
    SF_HIP(hipEventRecord(em->ev_poll[slot], em->cur));
    return SFGPU_OK;
}
static int em_poll_wait(sfgpu_em* em, int slot, bool posted_by_graph, int* done) {
    SF_HIP(hipEventSynchronize(em->ev_poll[slot]));
    const unsigned long long v = *reinterpret_cast<volatile unsigned long long*>(em->h_mirror + (posted_by_graph ? 0 : 8 * slot));
    *done = (int)((v >> 32) & 1ull);
    return SFGPU_OK;
}

int sfgpu_em_finish(sfgpu_em* em, double* d_alpha_out, double* d_mass_out, sfgpu_em_stats* stats) {
    SF_REQUIRE(em && d_alpha_out, SFGPU_ERR_INVALID, "sfgpu_em_finish: null pointer");
    const sfgpu_problem& p = em->prob;
    double cutoff = em->opts.use_vbem ? (kPriorAlpha + kMinAlpha) : kMinAlpha;   // :812
    dim3 g(em->nb), b(kEmBlock);
    hipLaunchKernelGGL(k_truncate, g, b, 0, em->cur, p.M, em->alpha, cutoff, d_alpha_out, em->partials);
    SF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_mass, g, b, 0, em->cur, p.M, d_alpha_out, d_mass_out, em->partials, em->nb, em->d_state);
    SF_CHECK_LAUNCH();
    if (em->fused) {
        hipLaunchKernelGGL(k_fused_max, dim3(2), dim3(kEmBlock), 0, em->cur, (uint64_t)em->n_tiles * (kSweepBlock / kWave), em->tmax, em->blkmax, em->nb);
        SF_CHECK_LAUNCH();
    }
    SF_HIP(hipMemcpyAsync(em->h_state, em->d_state, sizeof(EmState), hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipMemcpyAsync(em->h_blkmax, em->blkmax, 2 * kMaxPartials * 8, hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    em_stats_from_state(em, stats);
    if (em->h_state->alpha_sum < kTiny) {                                       // :877-881
        set_error("Total alpha weight was too small! Make sure you ran sailfish correctly.");
        return SFGPU_ERR_ALPHA_SUM;
    }
    return SFGPU_OK;
}

sfgpu_stream sfgpu_em_stream(sfgpu_em* em) { return em ? reinterpret_cast<sfgpu_stream>(em->cur) : nullptr; }

// Can this handle run the sharded loop with one sweep kernel per iteration?  (its plan has the fused kernel's tables, the caller's
// transcript order -- a plan with an order of its own indexes its vectors by position, which the other ranks' plans do not share --
// and VBEM's normaliser is the run's constant.)  A multi-rank host asks every rank, takes the minimum (a collective of its own) and
// tells every rank the answer with sfgpu_em_set_sharded_fused: all ranks must run the same form.
int sfgpu_em_sharded_fused_ok(sfgpu_em* em) {
    if (!em || !em->gather || !em->partial_a || em->inv || em->prob.C == 0) return 0;
    if (const char* fe = getenv("SFGPU_EM_FUSED")) if (atoi(fe) == 0) return 0;
    if (getenv("SFGPU_EM_EXACT_NORM") != nullptr) return 0;
    if (em->fused_ok < 0) {
        if (hipEventSynchronize(em->ev_plan) != hipSuccess) return 0;
        em->fused_ok = ((*reinterpret_cast<const uint32_t*>(em->h_plan + 4) & 2u) == 0u && em->partial_a) ? 1 : 0;
    }
    return em->fused_ok == 1 ? 1 : 0;
}
int sfgpu_em_set_sharded_fused(sfgpu_em* em, int on) {
    SF_REQUIRE(em, SFGPU_ERR_INVALID, "sfgpu_em_set_sharded_fused: null handle");
    em->sharded_fused = on ? 1 : 0;
    return SFGPU_OK;
}

// classes partitioned over the ranks, alpha replicated, one SUM all-reduce of alphaOut per iteration; the
// transport is the caller's (RCCL in a multi-GPU host), the loop is the piecewise one
int sfgpu_em_optimize_sharded(sfgpu_em* em, const sfgpu_em_opts* opts, sfgpu_allreduce_fn allreduce, void* user, uint32_t poll_every,
                              double* d_alpha_out, double* d_mass_out, sfgpu_em_stats* stats) {
    SF_REQUIRE(em && opts && allreduce && d_alpha_out, SFGPU_ERR_INVALID, "sfgpu_em_optimize_sharded: null pointer");
    int rc;
    if ((rc = sfgpu_em_begin(em, opts))) return rc;
    const uint64_t M = em->prob.M;
    auto reduce = [&]() -> int {
        const int cr = allreduce(em->alpha_out, M, user, reinterpret_cast<sfgpu_stream>(em->cur));
        if (cr) { set_error("sfgpu_em_optimize_sharded: the all-reduce callback returned %d", cr); return SFGPU_ERR_STATE; }
        return SFGPU_OK;
    };
    if ((rc = reduce())) return rc;                                             // union of the ranks' active sets
    if ((rc = sfgpu_em_init(em))) return rc;
    int done = 0;
    sfgpu_em_stats st{};
    if ((rc = sfgpu_em_poll(em, &done, &st))) return rc;
    if (st.n_active == 0) {                                                      // :794-798
        set_error("It seems that no transcripts are expressed; something is likely wrong!");
        if (stats) *stats = st;
        return SFGPU_ERR_NO_ACTIVE;
    }
    if (poll_every == 0) poll_every = 16;
    em->h_mirror[0] = em->h_mirror[8] = 0ull;
    if (em->sharded_fused && sfgpu_em_sharded_fused_ok(em) == 1) {
        // ONE sweep kernel per iteration (round 5): the update of iteration it - 1 runs at the head of sweep `it` from the all-reduced
        // vector alone (k_sweep_lds<.., .., FUSED> with `sharded`: no neighbour sums to add -- the fold below ran before the
        // all-reduce), so an iteration is sweep + fold + all-reduce instead of sweep + fold + all-reduce + update.  The accumulators
        // rotate through three buffers as in the fused loop (read L - 1's sum, add into L's, zero L + 1's); the stop test lags one
        // launch, and launches past the stop are no-ops whose all-reduce sums a vector nobody reads.  Every rank runs this form or
        // none does (sfgpu_em_set_sharded_fused after a collective agreement): the two forms notice the stop one iteration apart.
        em->fused = true; em->par = 0;
        double* bufs[3] = {em->alpha_out, em->aout_b, em->aout_c};
        uint32_t L = 0;
        for (uint32_t k = 0; !done; ++k) {
            for (uint32_t i = 0; i < poll_every; ++i, ++L) {
                Launcher Ln; Ln.stream = em->cur;
                if ((rc = em_enqueue_fused(em, Ln, L == 0, true))) return rc;
                hipLaunchKernelGGL(k_fold_slots, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, em->cov2, em->cov_pos, em->partial_a, em->partial_b,
                                   em->alpha_out, em->aout_b, em->aout_c, em->d_state);
                SF_CHECK_LAUNCH();
                const int cr = allreduce(bufs[L % 3u], M, user, reinterpret_cast<sfgpu_stream>(em->cur));
                if (cr) { set_error("sfgpu_em_optimize_sharded: the all-reduce callback returned %d", cr); return SFGPU_ERR_STATE; }
            }
            if ((rc = em_poll_start(em, (int)(k & 1u), true))) return rc;
            if (k > 0 && (rc = em_poll_wait(em, (int)((k - 1u) & 1u), false, &done))) return rc;
        }
        return sfgpu_em_finish(em, d_alpha_out, d_mass_out, stats);
    }
    // (Iterations past the stop -- up to 2 poll_every of them -- still run their all-reduce.  What it sums then is all zeros: the last
    //  update zeroed alphaOut, and past the stop the sweep, k_fold and the update return at once, so nothing can grow; finish() and
    //  the statistics never read alphaOut.  The collective itself is the price of not having the host in the loop.)
    for (uint32_t k = 0; !done; ++k) {                                           // (stop test pipelined as in em_run: every rank sees
        for (uint32_t i = 0; i < poll_every; ++i) {                              //  the same state, iterations past the stop are no-ops)
            if ((rc = sfgpu_em_sweep(em)) || (rc = reduce()) || (rc = sfgpu_em_update(em))) return rc;
        }
        if ((rc = em_poll_start(em, (int)(k & 1u), true))) return rc;
        if (k > 0 && (rc = em_poll_wait(em, (int)((k - 1u) & 1u), false, &done))) return rc;
    }
    return sfgpu_em_finish(em, d_alpha_out, d_mass_out, stats);
}

static bool same_opts(const sfgpu_em_opts& a, const sfgpu_em_opts& b) {
    return a.use_vbem == b.use_vbem && a.tol == b.tol && a.min_iter == b.min_iter && a.max_iter == b.max_iter &&
           a.check_mode == b.check_mode && a.iters_per_launch == b.iters_per_launch;
}

constexpr uint32_t kPreLaunched = 8;      // iterations enqueued directly while the host builds the graph (em_run)

// `n` iterations as an executable graph (kernel arguments are baked, the iteration index and the stop
// latch live in device memory)
static bool em_graph_ready(const sfgpu_em* em, uint32_t n) {
    return em->graph && same_opts(em->graph_opts, em->opts) && em->graph_iters == n && em->graph_fused == em->fused &&
           em->graph_const_norm == em->const_norm && em->graph_log_norm == em->vb_log_norm;      // (baked kernel arguments, like the bounds)
}
static int em_build_graph(sfgpu_em* em, uint32_t n) {
    if (em_graph_ready(em, n)) return SFGPU_OK;
    if (em->graph) { (void)hipGraphExecDestroy(em->graph); em->graph = nullptr; }
    Launcher L;
    SF_HIP(hipGraphCreate(&L.graph, 0));
    int rc = SFGPU_OK;
    const uint32_t par0 = em->par;                       // (fused: the graph starts on parity 0 and, n being even, ends on it)
    em->par = 0;
    for (uint32_t i = 0; i < n && rc == SFGPU_OK; ++i) {
        if (em->fused) { rc = em_enqueue_fused(em, L, false); continue; }
        rc = em_enqueue_sweep(em, L);
        if (rc == SFGPU_OK) rc = em_enqueue_update(em, true, L);
    }
    em->par = par0;
    if (rc == SFGPU_OK) rc = em_enqueue_post(em, L, 0);
    if (rc) { (void)hipGraphDestroy(L.graph); return rc; }
    hipError_t ei = hipGraphInstantiate(&em->graph, L.graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(L.graph);
    SF_HIP(ei);
    em->graph_opts = em->opts; em->graph_iters = n; em->graph_fused = em->fused;
    em->graph_const_norm = em->const_norm; em->graph_log_norm = em->vb_log_norm;
    return SFGPU_OK;
}

// ---- the persistent loop (em_persist.h): eligibility, and the launch ----
static std::mutex g_persist_mu[16];          // per device: two persistent launches of this process never share the chip (each needs ALL its blocks resident)
// xs | acc | den | facc | fxs | wmax | sctl, hprev | cntl | far_xi_l | ftg_l | esc_l | (dev stamps)
static size_t em_persist_lds_base(const sfgpu_em* em) {
    return ((size_t)2 * (kWin + 2) + (em->null_cls + 2) + 2 * (size_t)em->far_cap + 2 * (kSweepBlock / kWave)) * 8 + 32 + 4 * kShards * 4
           + ((size_t)(em->null_cls + 2) + (size_t)em->far_cap + (em->ftgt ? (size_t)kWin : 0) + 1) * 4;
}
static size_t em_persist_lds(const sfgpu_em* em) { return em_persist_lds_base(em) + (size_t)em->esc_ln * 8 + 64; }
static const void* em_persist_func(bool vb) {
    return vb ? reinterpret_cast<const void*>(&k_em_persist<true>) : reinterpret_cast<const void*>(&k_em_persist<false>);
}
// the plan's verdict (read back behind the tables), the LDS carve and the chip's residency: every block of the launch must be resident
static void em_persist_check(sfgpu_em* em) {
    em->persist_ok = 0;
    const bool say = env_timing();
    auto no = [&](const char* why) { if (say) fprintf(stderr, "em persistent: not eligible -- %s\n", why); };
    if (!em->xbuf || !em->pflags || !em->partial_a) return no("no tables (SFGPU_EM_PERSIST=0 at create, or the exchange buffer would pass 2 GB)");
    if (hipEventSynchronize(em->ev_plan) != hipSuccess) return no("plan event");
    const uint32_t* pf = reinterpret_cast<const uint32_t*>(em->h_plan + 8);
    if (pf[0] & 2u) return no("a far member's transcript lies in no window (no home thread)");
    if (pf[0] & 4u) return no("a transcript is fed by more far slots than its home thread should walk");
    if ((*reinterpret_cast<const uint32_t*>(em->h_plan + 4) & 1u) != 0u) return no("a tile that more than kNbMax tiles overlap (it goes by the cover list)");
    em->far_cap = pf[1];
    constexpr size_t kLdsPerBlock = 81920;                   // half a CU's LDS: two blocks per CU, like the sweep
    {   // the tiles' far members on chip: all of the plan's largest list, or what is left of the block's LDS
        const size_t base = em_persist_lds_base(em) + 64;
        const size_t room = base < kLdsPerBlock ? (kLdsPerBlock - base) / 8 : 0;
        em->esc_ln = (uint32_t)std::min<size_t>(pf[2], room);
    }
    const size_t lds = em_persist_lds(em);
    if (lds > kLdsPerBlock) return no("the tile's classes + far slots do not fit the LDS");
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return no("device query");
    for (int vb = 0; vb < 2; ++vb) {
        int nb = 0;
        if (hipFuncSetAttribute(em_persist_func(vb != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerBlock) != hipSuccess) { (void)hipGetLastError(); return no("hipFuncSetAttribute"); }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, em_persist_func(vb != 0), kPB, lds) != hipSuccess) { (void)hipGetLastError(); return no("occupancy query"); }
        if ((uint64_t)nb * (uint64_t)n_cu < em->n_tiles) return no("more tiles than resident blocks (a multi-round plan)");
    }
    if (say) fprintf(stderr, "em persistent: eligible (%u tiles, %zu bytes of LDS, %u far slots at most per tile)\n", em->n_tiles, lds, em->far_cap);
    em->persist_ok = 1;
}
static int em_launch_persist(sfgpu_em* em, int ablate) {
    const uint64_t P = em->P ? em->P : 1, En = em->E ? em->E : 1;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    unsigned char* q = em->xbuf;
    PersistArgs a{}; PersistCold c{};
    a.ctl = reinterpret_cast<unsigned long long*>(q); c.status = reinterpret_cast<uint32_t*>(q + (size_t)kCtlWords * 8);
    PersistCold* d_cold = reinterpret_cast<PersistCold*>(q + (size_t)kCtlWords * 8 + 64);
    a.cold = d_cold;
    a.xbuf = q; a.xbuf_bytes = (uint32_t)em->xbuf_bytes;
    size_t o = up((size_t)kCtlWords * 8 + 64 + sizeof(PersistCold));
    a.part_off[0] = (uint32_t)o; o += up(P * 16); a.part_off[1] = (uint32_t)o; o += up(P * 16);
    a.far_off0 = (uint32_t)o; a.far_stride = (uint32_t)up(En * 16);         // (far slots by parity, then the far targets' x)
    a.tiles = em->td; c.st = em->d_state; a.min_iter = em->opts.min_iter; a.max_iter = em->opts.max_iter; a.n_tiles = em->n_tiles; a.check_mode = em->opts.check_mode;
    a.tp = em->tp; a.recs = em->recs; a.ovc = em->ovc; a.counts = em->cnt8; a.cscp = em->cscp;
    c.x = em->x; c.inv = em->inv;
    a.lenc = em->inv ? em->lencP : em->lenc; a.alpha = em->inv ? em->alphaP : em->alpha;
    c.esc_cls = em->esc_cls_p; c.esc_far = em->esc_far; c.far_pos = em->far_pos; c.far_xi = em->far_xi; a.ftgt = em->ftgt; c.ft_list = em->ft_list;
    c.unc = em->unc; c.unc_n = em->unc + em->prob.M;
    c.tmax = em->tmax; a.tol = em->opts.tol; a.log_norm = em->vb_log_norm;
    a.den_cap = em->null_cls; a.far_cap = em->far_cap; a.esc_ln = em->esc_ln; a.ablate = ablate;
    {   // the launch's epoch: 1..255 in rotation over all persistent launches of the process (exchange buffers are recycled between handles)
        static std::atomic<uint32_t> g_epoch{0};
        a.tag0 = (g_epoch.fetch_add(1u, std::memory_order_relaxed) % 255u + 1u) << 24;
    }
#if defined(SFGPU_P_STAMP) || defined(SFGPU_P_PROGRESS)
    if (!em->dbg) SF_HIP(pool_malloc(&em->dbg, (size_t)em->n_tiles * 16 * 8));
    SF_HIP(hipMemsetAsync(em->dbg, 0, (size_t)em->n_tiles * 16 * 8, em->cur));
    c.dbg = em->dbg;
#endif
    // tags, counters, abort word, status: zeroed before EVERY launch; the cold block written behind the control words
    {
        static_assert(((size_t)kCtlWords * 8 + 64) % 16 == 0, "the cold block starts on a 16-byte boundary");
        const uint32_t c0 = (uint32_t)(((size_t)kCtlWords * 8 + 64) / 16), cn = (uint32_t)((sizeof(PersistCold) + 15) / 16);
        const unsigned nb = (unsigned)std::min<uint64_t>((em->xbuf_bytes / 16 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_persist_init, dim3(nb), dim3(256), 0, em->cur, (void*)em->xbuf, (uint32_t)em->xbuf_bytes, c0, cn, d_cold, c,
                           (uint64_t)em->prob.C, (const uint32_t*)em->counts32, (const uint32_t*)em->cpos, em->cnt8);
        SF_CHECK_LAUNCH();
    }
    void* args[] = {&a};
    {
        // (SFGPU_EM_COOP=1: a cooperative launch -- the runtime checks the grid against the occupancy query at launch time; residency on
        //  the chip is the same as a plain launch's, see profiles/r6_em_notes.md)
        const bool coop = []() { const char* e = getenv("SFGPU_EM_COOP"); return e && atoi(e) != 0; }();
        if (coop) SF_HIP(hipLaunchCooperativeKernel(em_persist_func(em->opts.use_vbem != 0), dim3(em->n_tiles), dim3(kPB), args, (unsigned int)em_persist_lds(em), em->cur));
        else SF_HIP(hipLaunchKernel(em_persist_func(em->opts.use_vbem != 0), dim3(em->n_tiles), dim3(kPB), args, em_persist_lds(em), em->cur));
    }
    // (the launch's verdict, next to the plan's words in pinned memory; read behind finish()'s wait)
    SF_HIP(hipMemcpyAsync(reinterpret_cast<uint32_t*>(em->h_plan + 7), c.status, 4, hipMemcpyDeviceToHost, em->cur));
    return SFGPU_OK;
}

// begin -> init -> iterate to the stop latch -> finish, on the handle's own stream
static int em_run(sfgpu_em* em, const sfgpu_em_opts* opts, double* d_alpha_out, double* d_mass_out,
                  sfgpu_em_stats* stats, bool quiet) {
    int rc;
    em->in_optimize = true;
    if ((rc = em_begin_on(em, opts, em->stream))) return rc;
    // The fused iteration: GATHER plans, EM or VBEM with the constant normaliser.  It pays when the tiles are large: its head adds ~3 us
    // of dependent round trips to every launch, which a co-resident block's phases hide when those are long, and saves k_update's
    // launch + one kernel boundary.  Measured (profiles/r4_em_notes.md): 18 000-nonzero tiles (cfg3) - 2.3 us (EM) / - 3.0 (VBEM) per
    // iteration, 12 900 - 3.1 / - 2.5, 5 300 (cfg2) + 1.25 / + 2.0.  So: fused from kFusedMinTileNnz nonzeros per tile up;
    // SFGPU_EM_FUSED=1 forces it wherever it can run, =0 keeps sweep + k_update.
    {
        constexpr uint32_t kFusedMinTileNnz = 9000;
        const char* fe = getenv("SFGPU_EM_FUSED");              // (read per run: tests switch it)
        const bool fused_off = fe && atoi(fe) == 0, fused_forced = fe && atoi(fe) != 0;
        em->fused = !fused_off && (fused_forced || em->tile_nnz >= kFusedMinTileNnz) && em->gather && em->fused_ok != 0 && em->prob.C != 0 &&
                    (!em->opts.use_vbem || em->const_norm);
    }
    // The PERSISTENT loop (em_persist.h; round 5): the whole loop as one launch, wherever the fused iteration could run and the plan fits
    // the chip in one round of blocks (whatever the tile size: it has no launch to amortise).  SFGPU_EM_PERSIST=0 keeps one kernel
    // per iteration (read when the plan is made too: no tables then).
    int persist_ablate = 0;
    {
        const char* fe = getenv("SFGPU_EM_FUSED"); const char* pe = getenv("SFGPU_EM_PERSIST");
        const bool family = !(fe && atoi(fe) == 0) && !(pe && atoi(pe) == 0) && em->gather && em->fused_ok != 0 && em->prob.C != 0 &&
                            g_allow_persist.load(std::memory_order_relaxed) && (!em->opts.use_vbem || em->const_norm) && em->opts.max_iter >= 1u && em->opts.max_iter < (1u << 24) - 2u && em->persist_ok != 0 && em->xbuf && !em->no_persist;
        em->persist = family;
        if (family) em->fused = true;
        if (pe && atoi(pe) == 2) persist_ablate = 1;         // dev: no tag checks (timing only)
        if (pe && atoi(pe) == 4) persist_ablate = 4;         // tests: tile 0 starts late
        if (pe && atoi(pe) == 3) persist_ablate = 3;         // tests: a tile gives up in step 2 (the run is repeated with one kernel per iteration)
    }
    if (em->fused && em->fused_ok < 0) {
        // the plan's verdict on the fused kernel's tables (sfgpu_em_create queued its read-back behind them; long done by now)
        SF_HIP(hipEventSynchronize(em->ev_plan));
        em->fused_ok = ((*reinterpret_cast<const uint32_t*>(em->h_plan + 4) & 2u) == 0u && em->partial_a) ? 1 : 0;
        if (!em->fused_ok) em->fused = false;                // (the classes are not in canonical order)
    }
    if (em->persist && !em->fused) em->persist = false;
    if (em->persist && em->persist_ok < 0) em_persist_check(em);
    if (em->persist && em->persist_ok != 1) {
        em->persist = false;
        const char* fe = getenv("SFGPU_EM_FUSED");           // (back to the rule of the fused iteration)
        em->fused = ((fe && atoi(fe) != 0) || em->tile_nnz >= 9000u);
    }
    std::unique_lock<std::mutex> persist_lock;
    if (em->persist) { int dev = 0; (void)hipGetDevice(&dev); persist_lock = std::unique_lock<std::mutex>(g_persist_mu[dev & 15]); }
    if ((rc = sfgpu_em_init_impl(em))) return rc;
    if (em->fused && em->inv) {                               // alpha and effLen in the plan's order (the fused kernel's index space)
        const uint64_t M = em->prob.M;
        hipLaunchKernelGGL(k_gather_f64, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, em->alpha, em->inv, em->alphaP);
        hipLaunchKernelGGL(k_gather_f64, dim3(blocks_for(M)), dim3(kEmBlock), 0, em->cur, M, em->lenc, em->inv, em->lencP);
        SF_CHECK_LAUNCH();
    }
    int done = 0;
    sfgpu_em_stats st{};
    if (!em->fused) {
        if ((rc = sfgpu_em_poll(em, &done, &st))) return rc;
        if (st.n_active == 0) {                                                      // :794-798
            set_error("It seems that no transcripts are expressed; something is likely wrong!");
            if (stats) *stats = st;
            return SFGPU_ERR_NO_ACTIVE;
        }
    }
    // (fused: no wait here -- the first launches go out behind init at once and the number of active transcripts is looked at when
    //  the loop has ended: a job without any runs minIter iterations over zeros before it reports so)
    if (!quiet) log_msg(0, "Optimizing over %llu equivalence classes", (unsigned long long)em->prob.C);   // :790
    const bool use_graph = SF_DEV_ENV("SFGPU_EM_NOGRAPH") == nullptr;
    uint32_t chunk = em->opts.iters_per_launch;
    if (em->fused) {
        chunk += chunk & 1u;                                  // (a graph bakes the launches' parities)
        em->par = 0;
    }
#ifdef SFGPU_X_STAMP
    if (!em->dbg) { SF_HIP(pool_malloc(&em->dbg, (size_t)em->n_tiles * 16 * 8)); }
    SF_HIP(hipMemsetAsync(em->dbg, 0, (size_t)em->n_tiles * 16 * 8, em->cur));
#endif
    auto iteration = [&](bool first) -> int {
        if (em->fused) return em_enqueue_fused(em, first);
        int r = em_enqueue_sweep(em);
        return r ? r : em_enqueue_update(em, true);
    };
    // The fused loop is STREAMED by default (SFGPU_EM_STREAMED=0: graph chunks as below): an iteration is one kernel of ~20 us and a
    // launch costs the host ~5, so the host simply stays eight launches ahead of the device (SFGPU_EM_AHEAD; 4 .. 20 measure alike).  Every launch writes "updates done |
    // ended" into a word of pinned host memory at its head; the host enqueues while it is less than kAhead launches ahead and
    // stops the moment the word says ended.  Against graph chunks of 32 iterations: no 14 us between graph launches (7 per cfg3
    // run), ~8 no-op launches past the stop instead of ~48, no graph to build: cfg3 22.5 -> 21.3 us per iteration, EM phase 5.5 -> 5.2 ms.
    {
        const char* se = SF_DEV_ENV("SFGPU_EM_STREAMED");
        em->streamed = em->fused && !em->persist && !(se && atoi(se) == 0);
    }
    SF_HIP(hipEventRecord(em->ev_a, em->cur));
    if (em->persist) {
        if ((rc = em_launch_persist(em, persist_ablate))) return rc;
        done = 1;
    } else if (em->streamed) {
        static const uint32_t kAhead = []() { const char* e = SF_DEV_ENV("SFGPU_EM_AHEAD"); long v = e ? atol(e) : 0; return (uint32_t)(v >= 1 && v <= 256 ? v : 8); }();
        volatile unsigned long long* mir = em->h_mirror;
        *mir = 0ull;
        ++em->run_no;                                         // (a word some launch of an earlier, failed run may still write is not this run's)
        const unsigned long long tag = (1ull << 33) | ((unsigned long long)(em->run_no & 0xFFFFFu) << 34);
        uint32_t launched = 0;
        if ((rc = iteration(true))) return rc;
        ++launched;
        // (the host follows the device through a word of pinned memory: a device that has stopped answering must not spin it for ever --
        //  every few thousand polls the stream is asked how it is, and two minutes without progress end the run)
        auto give_up = [&](const char* why) -> int {
            (void)hipStreamSynchronize(em->cur);              // nothing of this run stays queued behind the error
            em->in_optimize = false;
            set_error("sfgpu_em_optimize: %s", why);
            return SFGPU_ERR_HIP;
        };
        const auto t_start = std::chrono::steady_clock::now();
        unsigned long long last_v = ~0ull; auto t_progress = t_start;
        for (uint32_t spins = 0, polls = 0;; ++polls) {
            const unsigned long long v = *mir;
            if ((polls & 4095u) == 4095u) {
                const hipError_t q = hipStreamQuery(em->cur);
                if (q != hipSuccess && q != hipErrorNotReady) return give_up(hipGetErrorString(q));
                const auto now = std::chrono::steady_clock::now();
                if (v != last_v) { last_v = v; t_progress = now; }
                else if (std::chrono::duration<double>(now - t_progress).count() > 120.0) return give_up("the EM loop made no progress for two minutes");
            }
            const bool any = (v >> 33) == (tag >> 33);
            if (any && ((v >> 32) & 1ull)) break;             // a launch found the loop ended: everything behind it is a no-op
            // launch n posts n - 1 updates done at its head; with `launched` enqueued, launched - 2 - posted wait behind the running one
            const uint32_t posted = any ? (uint32_t)v : 0u;
            const uint32_t queued = any ? (launched >= posted + 2u ? launched - posted - 2u : 0u) : launched;
            if (queued < kAhead) { if ((rc = iteration(false))) { (void)hipStreamSynchronize(em->cur); em->in_optimize = false; return rc; } ++launched; spins = 0; }
            else if (++spins > 64u) { std::this_thread::yield(); spins = 0; }
        }
        done = 1;
    } else {
    if (em->fused) {
        // the first launch has nothing to update; a second one keeps the parity even for the graph
        if ((rc = iteration(true)) || (rc = iteration(false))) return rc;
    }
    if (use_graph && !em_graph_ready(em, chunk)) {
        // building the graph takes the host ~170 us: the first iterations go straight onto the stream and run meanwhile
        // (iterations past the stop are no-ops, so it does not matter how many of them there are)
        for (uint32_t i = 0; i < kPreLaunched; ++i) if ((rc = iteration(false))) return rc;
    }
    if (use_graph && (rc = em_build_graph(em, chunk))) return rc;
    em->h_mirror[0] = em->h_mirror[8] = 0ull;                // (nothing of an earlier run is in flight: finish() waited for it)
    for (uint32_t k = 0; !done; ++k) {
        if (use_graph) {
            SF_HIP(hipGraphLaunch(em->graph, em->cur));
        } else {
            for (uint32_t i = 0; i < chunk; ++i) if ((rc = iteration(false))) return rc;
        }
        if ((rc = em_poll_start(em, (int)(k & 1u), !use_graph))) return rc;
        if (k > 0 && (rc = em_poll_wait(em, (int)((k - 1u) & 1u), use_graph, &done))) return rc;       // the chunk before this one
    }
    }
    SF_HIP(hipEventRecord(em->ev_b, em->cur));
    if (em->fused && em->inv) {                               // back to the caller's order
        hipLaunchKernelGGL(k_scatter_f64, dim3(blocks_for(em->prob.M)), dim3(kEmBlock), 0, em->cur, em->prob.M, em->alphaP, em->inv, em->alpha);
        SF_CHECK_LAUNCH();
    }
    rc = sfgpu_em_finish(em, d_alpha_out, d_mass_out, &st);
    if (em->persist && *reinterpret_cast<const volatile uint32_t*>(em->h_plan + 7) != 0u) {
        // a tile gave up waiting (its neighbours never became resident: the chip is shared with another process' kernels): this
        // handle goes back to one kernel per iteration, and the run is repeated from its start
        persist_lock.unlock();
        em->persist_ok = 0;
        {
            unsigned long long who[4] = {0, 0, 0, 0};
            (void)hipMemcpy(who, em->xbuf + (size_t)(kCtlAbort + 1) * kCtlStride * 8, sizeof(who), hipMemcpyDeviceToHost);
#ifdef SFGPU_P_PROGRESS
            {
                std::vector<unsigned long long> h(em->n_tiles);
                (void)hipMemcpy(h.data(), em->dbg, h.size() * 8, hipMemcpyDeviceToHost);
                std::string line;
                for (uint32_t b = 0; b < em->n_tiles; ++b) { line += (char)('0' + (h[b] > 9 ? 9 : (int)h[b])); }
                fprintf(stderr, "persist progress (step + 1 per tile, 0 = never ran): %s\n", line.c_str());
                std::vector<unsigned long long> w(em->n_tiles);
                (void)hipMemcpy(w.data(), em->dbg + em->n_tiles, w.size() * 8, hipMemcpyDeviceToHost);
                fprintf(stderr, "persist waits at the give-up (tile: why * 1000 + step; tiles at steps < 2 only):");
                for (uint32_t b = 0; b < em->n_tiles; ++b) if (h[b] < 3) fprintf(stderr, " %u:%llu", b, w[b]);
                fprintf(stderr, "\n");
                {   // when and where every block started (100 MHz clock): the late ones, and a histogram of the waits' reasons
                    std::vector<unsigned long long> t0(em->n_tiles), hw(em->n_tiles);
                    (void)hipMemcpy(t0.data(), em->dbg + 2 * em->n_tiles, t0.size() * 8, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(hw.data(), em->dbg + 3 * em->n_tiles, hw.size() * 8, hipMemcpyDeviceToHost);
                    unsigned long long tmin = ~0ull; uint32_t never = 0;
                    for (uint32_t b = 0; b < em->n_tiles; ++b) { if (!t0[b]) ++never; else tmin = std::min(tmin, t0[b]); }
                    std::vector<uint32_t> ord(em->n_tiles);
                    for (uint32_t b = 0; b < em->n_tiles; ++b) ord[b] = b;
                    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return t0[x] > t0[y]; });
                    fprintf(stderr, "persist starts: %u blocks never started; the latest (tile: us after the first, xcc, hw_id):", never);
                    for (uint32_t i = 0; i < 12 && i < em->n_tiles; ++i) { const uint32_t b = ord[i]; fprintf(stderr, " %u:%.1f,x%llu,%05llx", b, t0[b] ? (double)(t0[b] - tmin) * 0.01 : -1.0, hw[b] >> 32, hw[b] & 0xFFFFFull); }
                    fprintf(stderr, "\n  why histogram (why * 1000 + step -> tiles):");
                    std::map<unsigned long long, uint32_t> hist;
                    for (uint32_t b = 0; b < em->n_tiles; ++b) ++hist[w[b]];
                    for (auto& kv : hist) fprintf(stderr, " %llu->%u", kv.first, kv.second);
                    fprintf(stderr, "\n  steps histogram (step + 1 -> tiles):");
                    std::map<unsigned long long, uint32_t> hs;
                    for (uint32_t b = 0; b < em->n_tiles; ++b) ++hs[h[b]];
                    for (auto& kv : hs) fprintf(stderr, " %llu->%u", kv.first, kv.second);
                    // the arrival counters as memory holds them now
                    unsigned long long ctlw[4 * kShards];
                    for (uint32_t k = 0; k < 4 * kShards; ++k) (void)hipMemcpy(&ctlw[k], em->xbuf + (size_t)(kCtlArrive + k) * kCtlStride * 8, 8, hipMemcpyDeviceToHost);
                    fprintf(stderr, "\n  arrival counters [slot][shard] (low word):");
                    for (uint32_t k = 0; k < 4 * kShards; ++k) fprintf(stderr, "%s%llu", (k % kShards) ? " " : " | ", ctlw[k] & 0xFFFFFFFFull);
                    fprintf(stderr, "\n");
                }
                // the first stuck tile's view: what memory holds NOW where it polled (its neighbours' pieces, parity 1 = tags 1, 3, ...)
                for (uint32_t b = 0; b < em->n_tiles; ++b) if (h[b] == 2 && w[b] / 1000 == 2) {
                    TileDesc t; (void)hipMemcpy(&t, em->td + b, sizeof(t), hipMemcpyDeviceToHost);
                    auto up2 = [](size_t x) { return (x + 255) & ~(size_t)255; };
                    const size_t o1 = up2((size_t)kCtlWords * 8 + 64 + sizeof(PersistCold)) + up2((size_t)(em->P ? em->P : 1) * 16);
                    fprintf(stderr, "tile %u (lo %u span %u nb %u): tags in memory of its neighbours' first / last overlapping slots (parity 1):", b, t.lo, t.span, t.nb_n);
                    for (uint32_t j2 = 0; j2 < t.nb_n && j2 < 6; ++j2) {
                        const uint32_t lo2 = t.e[j2].x, sp2 = t.e[j2].y, off2 = t.e[j2].z;
                        const uint32_t p0 = std::max(lo2, t.lo), p1 = std::min(lo2 + sp2, t.lo + t.span) - 1;
                        uint32_t g0[4], g1[4];
                        (void)hipMemcpy(g0, em->xbuf + o1 + (size_t)(off2 + (p0 - lo2)) * 16, 16, hipMemcpyDeviceToHost);
                        (void)hipMemcpy(g1, em->xbuf + o1 + (size_t)(off2 + (p1 - lo2)) * 16, 16, hipMemcpyDeviceToHost);
                        fprintf(stderr, " [tile %u: %u/%u .. %u/%u]", t.e[j2].w, g0[1], g0[3], g1[1], g1[3]);
                    }
                    fprintf(stderr, "\n");
                    break;
                }

            }
#endif
            log_msg(1, "EM: the persistent loop gave up waiting for a tile (is the device shared?); running one kernel per iteration [tile %llu of %u, thread %llu, wait %llu, step %llu]",
                    who[0] - 1ull, em->n_tiles, who[1], who[2], who[3]);
        }
        return em_run(em, opts, d_alpha_out, d_mass_out, stats, quiet);
    }
    if (em->fused && st.n_active == 0) {                                             // :794-798 (see above)
        set_error("It seems that no transcripts are expressed; something is likely wrong!");
        if (stats) *stats = st;
        return SFGPU_ERR_NO_ACTIVE;
    }
#ifdef SFGPU_P_STAMP
    if (em->persist && em->dbg && st.iters > 2) {          // dev: where a persistent step goes, per tile (100 MHz clock)
        std::vector<unsigned long long> h((size_t)em->n_tiles * 8);
        (void)hipMemcpy(h.data(), em->dbg, h.size() * 8, hipMemcpyDeviceToHost);
        static const char* nm[7] = {"operands", "x+update", "barrier", "A", "B", "C", "D"};
        const double steps = (double)st.iters + 1.0;
        fprintf(stderr, "persist stamps (%s, %u tiles, %u steps; us per step and tile, mean / min / max over the tiles):", em->opts.use_vbem ? "VBEM" : "EM", em->n_tiles, st.iters + 1);
        for (int k = 0; k < 7; ++k) {
            double sum = 0, mn = 1e30, mx = 0;
            for (uint32_t b = 0; b < em->n_tiles; ++b) { const double v = (double)h[b * 8 + k] * 0.01 / steps; sum += v; mn = std::min(mn, v); mx = std::max(mx, v); }
            fprintf(stderr, " %s %.2f/%.2f/%.2f", nm[k], sum / em->n_tiles, mn, mx);
        }
        fprintf(stderr, "\n");
        // the tiles that wait least for their operands set the pace: what are they made of?
        std::vector<TileDesc> htd(em->n_tiles);
        (void)hipMemcpy(htd.data(), em->td, htd.size() * sizeof(TileDesc), hipMemcpyDeviceToHost);
        std::vector<uint32_t> order(em->n_tiles);
        for (uint32_t b = 0; b < em->n_tiles; ++b) order[b] = b;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return h[x * 8] < h[y * 8]; });
        std::vector<TilePack> htp(em->n_tiles);
        (void)hipMemcpy(htp.data(), em->tp, htp.size() * sizeof(TilePack), hipMemcpyDeviceToHost);
        double mean_nc = 0, mean_np = 0, mean_nm = 0, mean_ov = 0, mean_n[4] = {0, 0, 0, 0};
        for (const TileDesc& t : htd) { mean_nc += t.nc; mean_np += t.np; mean_nm += t.nm; }
        for (const TilePack& t : htp) { mean_ov += t.n_ov; mean_n[0] += t.n1; mean_n[1] += t.n2; mean_n[2] += t.n3; mean_n[3] += t.n4; }
        fprintf(stderr, "  tile means: classes %.0f (records of 4 / 8 / 16 bytes / long: %.0f %.0f %.0f %.0f), pure chunks %.0f, mixed chunks %.0f, overflow chunks %.0f\n", mean_nc / em->n_tiles,
                mean_n[0] / em->n_tiles, mean_n[1] / em->n_tiles, mean_n[2] / em->n_tiles, mean_n[3] / em->n_tiles, mean_np / em->n_tiles, mean_nm / em->n_tiles, mean_ov / em->n_tiles);
        for (uint32_t i = 0; i < 6 && i < em->n_tiles; ++i) {
            const uint32_t b = order[i]; const TileDesc& t = htd[b];
            fprintf(stderr, "  tile %4u:", b);
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.2f", nm[k], (double)h[b * 8 + k] * 0.01 / steps);
            fprintf(stderr, " | span %u classes %u pure %u mixed %u overflow %u far members %u far slots %u neighbours %u\n", t.span, t.nc, t.np, t.nm, htp[b].n_ov, t.n_esc, t.nf, t.nb_n);
        }
    }
#endif
#ifdef SFGPU_X_STAMP
    if (em->dbg) {                                          // dev: phase stamps of the last launch that ran (100 MHz clock)
        std::vector<unsigned long long> h((size_t)em->n_tiles * 16);
        (void)hipMemcpy(h.data(), em->dbg, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0min = ~0ull, tend = 0; double sum[16] = {0}; double ramp = 0; uint32_t n = 0;
        for (uint32_t b = 0; b < em->n_tiles; ++b) if (h[b * 16] && h[b * 16 + 10]) { t0min = std::min(t0min, h[b * 16]); tend = std::max(tend, h[b * 16 + 10]); }
        for (uint32_t b = 0; b < em->n_tiles; ++b) if (h[b * 16] && h[b * 16 + 10]) {
            ++n; ramp += (double)(h[b * 16] - t0min);
            for (int k = 1; k <= 10; ++k) sum[k] += h[b * 16 + k] ? (double)(h[b * 16 + k] - h[b * 16]) : 0.0;
        }
        fprintf(stderr, "stamps (%s, %u tiles; us after the tile's entry): entry %.2f after the first |", em->fused ? "fused" : "unfused", n, ramp / n * 0.01);
        for (int k = 1; k <= 10; ++k) fprintf(stderr, " s%d %.2f", k, sum[k] / n * 0.01);
        fprintf(stderr, " | first entry -> last end %.2f us\n", (double)(tend - t0min) * 0.01);
        // the slowest tiles: duration of each phase for the five tiles with the latest end
        std::vector<uint32_t> order;
        for (uint32_t b = 0; b < em->n_tiles; ++b) if (h[b * 16] && h[b * 16 + 10]) order.push_back(b);
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return h[x * 16 + 10] - h[x * 16] > h[y * 16 + 10] - h[y * 16]; });
        std::vector<TileDesc> htd(em->n_tiles);
        (void)hipMemcpy(htd.data(), em->td, htd.size() * sizeof(TileDesc), hipMemcpyDeviceToHost);
        for (size_t q = 0; q < order.size() && q < 5; ++q) {
            const uint32_t b = order[q];
            fprintf(stderr, "  slow tile %u (nc %u span %u n8 %u n_esc %u np %u nm %u nb %u; entry +%.2f):", b, htd[b].nc, htd[b].span, htd[b].n8, htd[b].n_esc, htd[b].np, htd[b].nm,
                    htd[b].nb_n, (double)(h[b * 16] - t0min) * 0.01);
            for (int k = 1; k <= 10; ++k) fprintf(stderr, " %.2f", h[b * 16 + k] ? (double)(h[b * 16 + k] - h[b * 16]) * 0.01 : 0.0);
            fprintf(stderr, "\n");
        }
        if (order.size() > 5) { const uint32_t b = order[order.size() / 2]; fprintf(stderr, "  median tile %u: total %.2f\n", b, (double)(h[b * 16 + 10] - h[b * 16]) * 0.01); }
    }
#endif
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, em->ev_a, em->ev_b);
    st.loop_ms = ms;
    if (!quiet) log_msg(0, "iteration = %u | max rel diff. = %g", st.iters, st.max_rel_diff);   // :871-872
    if (stats) *stats = st;
    return rc;
}

int sfgpu_em_optimize(sfgpu_em* em, const sfgpu_em_opts* opts, double* d_alpha_out, double* d_mass_out,
                      sfgpu_em_stats* stats) {
    SF_REQUIRE(em && d_alpha_out, SFGPU_ERR_INVALID, "sfgpu_em_optimize: null pointer");
    int rc;
    if ((rc = em_join_user(em))) return rc;
    return em_run(em, opts, d_alpha_out, d_mass_out, stats, false);
}

// optimize() with doBiasCorrect (src/CollapsedEMOptimizer.cpp:717, :814-840, :888).  The loop runs in segments
// that end at the recompute iterations: a segment is the ordinary on-device loop with max_iter set to the
// next hook (the stop latch is re-evaluated by the first sweep of the next segment, so a segment that ended
// on its hook simply continues).  At a hook that the reference would reach (its while condition still true)
// the lengths are recomputed from the current alpha, and x -- alpha / effLen, or expTheta / effLen -- is
// rebuilt with them (updateEqClassWeights :527-555: the weights are not stored here, x carries 1 / effLen).
static int em_run_bias(sfgpu_em* em, const sfgpu_em_opts* opts, sfgpu_bias* bias, double* d_alpha_out, double* d_mass_out,
                       double* d_eff_len_out, uint32_t* n_recomputes, sfgpu_em_stats* stats) {
    static const uint32_t kHooks[3] = {50, 500, 1000};                         // recomputeIt :814
    int rc;
    em->in_optimize = true;
    if ((rc = em_begin_on(em, opts, em->stream))) return rc;
    const sfgpu_em_opts user = em->opts;
    if ((rc = sfgpu_em_init_impl(em))) return rc;
    int done = 0;
    sfgpu_em_stats st{};
    if ((rc = sfgpu_em_poll(em, &done, &st))) return rc;
    if (st.n_active == 0) {
        set_error("It seems that no transcripts are expressed; something is likely wrong!");
        if (stats) *stats = st;
        return SFGPU_ERR_NO_ACTIVE;
    }
    log_msg(0, "Optimizing over %llu equivalence classes", (unsigned long long)em->prob.C);
    const sfgpu_problem& p = em->prob;
    const bool use_graph = SF_DEV_ENV("SFGPU_EM_NOGRAPH") == nullptr;
    const uint32_t chunk = user.iters_per_launch;
    uint32_t recomputes = 0;
    SF_HIP(hipEventRecord(em->ev_a, em->cur));
    for (;;) {
        const EmState* h = em->h_state;
        const uint32_t it = h->it_a;
        const bool conv = it > 0 && h->notconv[(it - 1) & 1] == 0;
        if (it >= user.min_iter && (it >= user.max_iter || conv)) break;       // the while condition of :820 is false
        uint32_t next = user.max_iter > user.min_iter ? user.max_iter : user.min_iter;   // the loop cannot end before either
        for (uint32_t hk : kHooks) {
            if (hk == it) {
                log_msg(0, "iteration %u, recomputing effective lengths", it);    // :827
                if ((rc = sfgpu_bias_update(bias, em->lenc, em->alpha, em->lenc, nullptr, reinterpret_cast<sfgpu_stream>(em->cur)))) return rc;
                dim3 g(em->nb), b(kEmBlock);
                if (user.use_vbem) {
                    hipLaunchKernelGGL(k_alpha_partials, g, b, 0, em->cur, p.M, em->alpha, em->sum_partials);
                    hipLaunchKernelGGL(k_vb_prepare, g, b, 0, em->cur, p.M, em->alpha, em->x, em->lenc, em->sum_partials,
                                       em->nb, em->d_state, 1);
                } else {
                    hipLaunchKernelGGL(k_x_from_alpha, dim3(blocks_for(p.M)), b, 0, em->cur, p.M, em->alpha, em->lenc, em->x);
                }
                SF_CHECK_LAUNCH();
                ++recomputes;
            }
            if (hk > it && hk < next) next = hk;
        }
        em->opts = user;
        em->opts.max_iter = next;
        if (em->opts.min_iter > next) em->opts.min_iter = next;
        if (use_graph && (rc = em_build_graph(em, chunk))) return rc;
        done = 0;
        while (!done) {
            if (use_graph) {
                SF_HIP(hipGraphLaunch(em->graph, em->cur));
            } else {
                for (uint32_t i = 0; i < chunk; ++i) {
                    if ((rc = em_enqueue_sweep(em))) return rc;
                    if ((rc = em_enqueue_update(em, true))) return rc;
                }
            }
            if ((rc = em_poll_impl(em, &done, &st, false))) return rc;
        }
    }
    em->opts = user;
    SF_HIP(hipEventRecord(em->ev_b, em->cur));
    if (d_eff_len_out) SF_HIP(hipMemcpyAsync(d_eff_len_out, em->lenc, p.M * 8, hipMemcpyDeviceToDevice, em->cur));   // :888
    rc = sfgpu_em_finish(em, d_alpha_out, d_mass_out, &st);
    // the handle's clamped lengths go back to the problem's own (a later optimize / bootstrap starts from them)
    hipLaunchKernelGGL(k_clamp_len, dim3(blocks_for(p.M)), dim3(kEmBlock), 0, em->cur, p.M, p.d_len, em->lenc);
    SF_HIP(hipStreamSynchronize(em->cur));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, em->ev_a, em->ev_b);
    st.loop_ms = ms;
    log_msg(0, "iteration = %u | max rel diff. = %g", st.iters, st.max_rel_diff);
    if (n_recomputes) *n_recomputes = recomputes;
    if (stats) *stats = st;
    return rc;
}

int sfgpu_em_optimize_bias(sfgpu_em* em, const sfgpu_em_opts* opts, sfgpu_bias* bias, double* d_alpha_out,
                           double* d_mass_out, double* d_eff_len_out, uint32_t* n_recomputes, sfgpu_em_stats* stats) {
    SF_REQUIRE(em && bias && d_alpha_out, SFGPU_ERR_INVALID, "sfgpu_em_optimize_bias: null pointer");
    SF_REQUIRE(bias_num_transcripts(bias) == em->prob.M, SFGPU_ERR_INVALID,
               "sfgpu_em_optimize_bias: the bias handle and the problem disagree on the number of transcripts");
    SF_REQUIRE(em->prob.M > 0, SFGPU_ERR_INVALID, "sfgpu_em_optimize_bias: no transcripts");
    int rc;
    if ((rc = em_join_user(em))) return rc;
    return em_run_bias(em, opts, bias, d_alpha_out, d_mass_out, d_eff_len_out, n_recomputes, stats);
}

// ---- bootstrap (a15): gatherBootstraps / doBootstrap, src/CollapsedEMOptimizer.cpp:438-525, 557-709 ----
__global__ void k_mask_counts(uint64_t C, const uint32_t* __restrict__ c32, uint32_t* __restrict__ out) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[c] = c32[c] & 0x7FFFFFFFu; else if (c == C) out[c] = 0;
}

static int em_bootstrap_prepare(sfgpu_em* em) {
    if (em->bs_prefix) return SFGPU_OK;
    const uint64_t C = em->prob.C;
    uint32_t* masked = nullptr;
    SF_HIP(pool_malloc(&masked, (C + 1) * 4));
    SF_HIP(pool_malloc(&em->bs_prefix, (C + 1) * 8));
    SF_HIP(pool_malloc(&em->bs_base, (C ? C : 1) * 4));
    const uint64_t W = multinomial_tree_width(C ? C : 1);
    SF_HIP(pool_malloc(&em->bs_scratch_a, W * 4)); SF_HIP(pool_malloc(&em->bs_scratch_b, W * 4));
    hipLaunchKernelGGL(k_mask_counts, dim3(blocks_for(C + 1)), dim3(kEmBlock), 0, em->stream, C, em->counts32, masked);
    int rc = exclusive_scan_u32(masked, em->bs_prefix, C, em->stream);
    pool_free(masked);
    if (rc) return rc;
    SF_HIP(hipMemcpyAsync(em->bs_base, em->counts32, (C ? C : 1) * 4, hipMemcpyDeviceToDevice, em->stream));
    SF_HIP(hipMemcpyAsync(&em->bs_total, em->bs_prefix + C, 8, hipMemcpyDeviceToHost, em->stream));
    SF_HIP(hipStreamSynchronize(em->stream));
    return SFGPU_OK;
}

int sfgpu_bootstrap_counts(sfgpu_em* em, uint64_t seed, uint64_t draw, uint32_t* d_counts_out) {
    SF_REQUIRE(em && d_counts_out, SFGPU_ERR_INVALID, "sfgpu_bootstrap_counts: null pointer");
    int rc;
    if ((rc = em_join_user(em))) return rc;
    if ((rc = em_bootstrap_prepare(em))) return rc;
    // n is a uint32_t in MultinomialSampler::operator() (MultinomialSampler.hpp:15): totals wrap
    uint32_t* dst = d_counts_out;
    if (em->cperm) SF_HIP(pool_malloc(&dst, (em->prob.C ? em->prob.C : 1) * 4));       // the plan's class order -> the caller's
    rc = multinomial_tree(em->bs_prefix, em->prob.C, (uint32_t)em->bs_total, seed, draw, nullptr, dst,
                          em->bs_scratch_a, em->bs_scratch_b, em->stream);
    if (!rc && em->cperm) {
        hipLaunchKernelGGL(k_renum_scatter, dim3(blocks_for(em->prob.C)), dim3(kEmBlock), 0, em->stream, em->prob.C, dst, em->cperm, d_counts_out);
    }
    hipError_t e = hipStreamSynchronize(em->stream);
    if (em->cperm) pool_free(dst);
    if (rc) return rc;
    SF_HIP(e);
    return SFGPU_OK;
}

struct BsOrder { std::mutex mu; std::condition_variable cv; uint32_t next = 0; bool abort = false; };

// replicates lane, lane + n_lanes, ... on handle h (its own stream, counts, state)
static int bootstrap_lane(sfgpu_em* h, const sfgpu_em_opts& o, uint32_t lane, uint32_t n_lanes, uint32_t n_bootstraps,
                          uint64_t seed, double* d_out, sfgpu_sample_cb cb, void* user, uint32_t* h_iters, BsOrder* order) {
    const uint64_t M = h->prob.M, C = h->prob.C;
    double* d_tmp = nullptr; double* h_tmp = nullptr;
    if (!d_out) SF_HIP(pool_malloc(&d_tmp, M * 8));
    if (cb) SF_HIP(pinned_malloc(&h_tmp, M * 8));
    const uint64_t keep_mapped = h->prob.num_mapped;
    h->prob.num_mapped = h->bs_total;                   // alpha init uses totalNumFrags = sum of counts (:470-474, :696)
    int rc = SFGPU_OK;
    for (uint32_t b = lane; b < n_bootstraps && rc == SFGPU_OK; b += n_lanes) {
        rc = multinomial_tree(h->bs_prefix, C, (uint32_t)h->bs_total, seed, b, h->bs_base, h->counts32,
                              h->bs_scratch_a, h->bs_scratch_b, h->stream);                       // :468
        if (rc) break;
        double* dst = d_out ? d_out + (uint64_t)b * M : d_tmp;
        sfgpu_em_stats st{};
        rc = em_run(h, &o, dst, nullptr, &st, true);                                               // :486-514
        if (h_iters) h_iters[b] = st.iters;
        if (rc) break;
        if (cb) {
            hipError_t e = hipMemcpyAsync(h_tmp, dst, M * 8, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { set_error("bootstrap copy failed: %s", hipGetErrorString(e)); rc = SFGPU_ERR_HIP; break; }
            // the writer hook (:522) sees the replicates in draw order, one at a time, whichever lane made them
            std::unique_lock<std::mutex> lk(order->mu);
            order->cv.wait(lk, [&] { return order->next == b || order->abort; });
            if (order->abort) { rc = SFGPU_ERR_INVALID; break; }
            const bool ok = cb(h_tmp, M, user) != 0;
            if (!ok) { set_error("bootstrap writer callback failed"); rc = SFGPU_ERR_INVALID; }
            order->next = b + 1;
            if (!ok) order->abort = true;
            lk.unlock();
            order->cv.notify_all();
            if (!ok) break;
        }
    }
    if (rc) { { std::lock_guard<std::mutex> lk(order->mu); order->abort = true; } order->cv.notify_all(); }
    h->prob.num_mapped = keep_mapped;
    (void)hipMemcpyAsync(h->counts32, h->bs_base, C * 4, hipMemcpyDeviceToDevice, h->stream);      // observed counts back
    (void)hipStreamSynchronize(h->stream);
    if (d_tmp) pool_free(d_tmp);
    if (h_tmp) pinned_free(h_tmp);
    return rc;
}

// Replicates are independent EM runs of ~750 two-kernel iterations each; a single run leaves the GPU idle
// at every kernel boundary (~1.7 us of each ~6.6 us).  Up to three LANES -- the handle itself plus
// clones of it with their own streams and state -- run replicates concurrently from their own host threads,
// so one lane's kernels fill the other lanes' boundaries ([MI355X] cfg2 classes: 10.6 -> 6.4 ms per replicate
// with three lanes, 7.4 with four).  Draw b uses the Philox stream (seed, b) whichever lane runs it.
int sfgpu_bootstrap(sfgpu_em* em, const sfgpu_em_opts* opts, uint32_t n_bootstraps, uint64_t seed,
                    double* d_out, sfgpu_sample_cb cb, void* user, uint32_t* h_iters) {
    SF_REQUIRE(em && opts, SFGPU_ERR_INVALID, "sfgpu_bootstrap: null pointer");
    SF_REQUIRE(em->prob.C > 0, SFGPU_ERR_NO_ACTIVE, "It seems that no transcripts are expressed; something is likely wrong!");
    int rc;
    if ((rc = em_join_user(em))) return rc;
    if ((rc = em_bootstrap_prepare(em))) return rc;
    log_msg(0, "Will draw %u bootstrap samples", n_bootstraps);                                   // :601
    log_msg(0, "Optimizing over %llu equivalence classes", (unsigned long long)em->prob.C);       // :602
    sfgpu_em_opts o = *opts;
    o.min_iter = 0;            // doBootstrap has no 50-iteration floor (:486)
    o.check_mode = 1;          // and gates on alphas > 1e-2 (:499)
    // Lanes.  Where the plan runs as ONE persistent launch per replicate, one lane is the fastest form (cfg3: 4.15 ms per replicate against
    // 4.98 with three lanes of one kernel per iteration -- the persistent loop needs the chip to itself, see below); everywhere else
    // three lanes fill each other's kernel boundaries.  SFGPU_BS_LANES overrides.
    uint32_t n_lanes = 3;
    {
        const char* fe = getenv("SFGPU_EM_FUSED"); const char* pe = getenv("SFGPU_EM_PERSIST");
        bool may = g_allow_persist.load(std::memory_order_relaxed) && !(fe && atoi(fe) == 0) && !(pe && atoi(pe) == 0) && em->gather && em->prob.C != 0 && em->xbuf && getenv("SFGPU_EM_EXACT_NORM") == nullptr &&
                   o.max_iter >= 1u && o.max_iter < (1u << 24) - 2u && em->persist_ok != 0 && em->fused_ok != 0;
        if (may && em->fused_ok < 0 && hipEventSynchronize(em->ev_plan) == hipSuccess)
            em->fused_ok = ((*reinterpret_cast<const uint32_t*>(em->h_plan + 4) & 2u) == 0u && em->partial_a) ? 1 : 0;
        if (may && em->fused_ok == 1 && em->persist_ok < 0) em_persist_check(em);
        if (may && em->fused_ok == 1 && em->persist_ok == 1) n_lanes = 1;
    }
    if (const char* e = getenv("SFGPU_BS_LANES")) { long v = atol(e); if (v >= 1 && v <= 8) n_lanes = (uint32_t)v; }
    if (n_lanes > n_bootstraps) n_lanes = n_bootstraps ? n_bootstraps : 1;
    while (em->bs_clones.size() + 1 < n_lanes) {        // clones are kept with the handle for the next call
        sfgpu_em* c = nullptr;
        if ((rc = sfgpu_em_create(&c, &em->prob, reinterpret_cast<sfgpu_stream>(em->user_stream)))) return rc;
        em->bs_clones.push_back(c);
    }
    for (uint32_t l = 1; l < n_lanes; ++l) {
        sfgpu_em* c = em->bs_clones[l - 1];
        if ((rc = em_join_user(c))) return rc;
        if ((rc = em_bootstrap_prepare(c))) return rc;
    }
    // Several lanes keep one kernel per iteration.  The persistent loop needs the chip to itself: launched while another stream's kernels
    // are in flight, the last blocks of one or two XCDs never become resident -- blocks of the other kernel that came and went while a CU's
    // first persistent block was placed leave its registers / LDS allocated in pieces, and the second 1024-thread block does not fit while
    // the first one lives (profiles/r6_em_notes.md 4: start times of every block, kernel trace) -- the run then gives up after ~50 ms
    // and falls back, correct but late.  One lane runs persistent.
    const bool lanes_persist = []() { const char* e = getenv("SFGPU_BS_PERSIST"); return e && atoi(e) != 0; }();
    em->no_persist = n_lanes > 1 && !lanes_persist;
    for (sfgpu_em* c : em->bs_clones) c->no_persist = n_lanes > 1 && !lanes_persist;
    struct LaneGuard { sfgpu_em* e; ~LaneGuard() { e->no_persist = false; for (sfgpu_em* c : e->bs_clones) c->no_persist = false; } } lane_guard{em};
    BsOrder order;
    std::vector<int> lane_rc(n_lanes, SFGPU_OK);
    std::vector<std::string> lane_err(n_lanes);
    int dev = 0; (void)hipGetDevice(&dev);
    std::vector<std::thread> threads;
    for (uint32_t l = 1; l < n_lanes; ++l)
        threads.emplace_back([&, l] {
            (void)hipSetDevice(dev);
            lane_rc[l] = bootstrap_lane(em->bs_clones[l - 1], o, l, n_lanes, n_bootstraps, seed, d_out, cb, user, h_iters, &order);
            if (lane_rc[l]) lane_err[l] = sfgpu_last_error();          // the error slot is per thread
        });
    lane_rc[0] = bootstrap_lane(em, o, 0, n_lanes, n_bootstraps, seed, d_out, cb, user, h_iters, &order);
    for (auto& t : threads) t.join();
    for (uint32_t l = 0; l < n_lanes; ++l)
        if (lane_rc[l]) { if (l) set_error("%s", lane_err[l].c_str()); return lane_rc[l]; }
    return SFGPU_OK;
}

int sfgpu_em_time_sweep(sfgpu_em* em, const sfgpu_em_opts* opts, uint32_t n, double* avg_ms) {
    SF_REQUIRE(em && avg_ms && n > 0, SFGPU_ERR_INVALID, "sfgpu_em_time_sweep: bad argument");
    int rc;
    if (!em->begun) {
        if ((rc = sfgpu_em_begin(em, opts))) return rc;
        if ((rc = sfgpu_em_init(em))) return rc;
    } else if ((rc = em_fill_opts(em, opts))) return rc;
    // keep alphaOut and the state block intact: time on copies
    EmState saved;
    SF_HIP(hipMemcpyAsync(em->scratch, em->alpha_out, em->prob.M * 8, hipMemcpyDeviceToDevice, em->cur));
    SF_HIP(hipMemcpyAsync(&saved, em->d_state, sizeof(EmState), hipMemcpyDeviceToHost, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    EmState run = saved; run.it_a = 0;
    uint32_t keep_min = em->opts.min_iter, keep_max = em->opts.max_iter;
    em->opts.min_iter = 1; em->opts.max_iter = 2;   // it_a = 0 never satisfies the stop test
    SF_HIP(hipMemcpyAsync(em->d_state, &run, sizeof(EmState), hipMemcpyHostToDevice, em->cur));
    for (int w = 0; w < 3; ++w) if ((rc = em_enqueue_sweep(em))) return rc;
    SF_HIP(hipEventRecord(em->ev_a, em->cur));
    for (uint32_t i = 0; i < n; ++i) if ((rc = em_enqueue_sweep(em))) return rc;
    SF_HIP(hipEventRecord(em->ev_b, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    float ms = 0.f;
    SF_HIP(hipEventElapsedTime(&ms, em->ev_a, em->ev_b));
    *avg_ms = (double)ms / n;
    em->opts.min_iter = keep_min; em->opts.max_iter = keep_max;
    SF_HIP(hipMemcpyAsync(em->d_state, &saved, sizeof(EmState), hipMemcpyHostToDevice, em->cur));
    SF_HIP(hipMemcpyAsync(em->alpha_out, em->scratch, em->prob.M * 8, hipMemcpyDeviceToDevice, em->cur));
    SF_HIP(hipStreamSynchronize(em->cur));
    return SFGPU_OK;
}

}  // extern "C"
