// rng.h -- counter-based random numbers and an exact binomial sampler, usable from device code
// (hipcc) and from host code (g++: tests/test_sampling_cpu.py compiles this header as plain C++ to
// check the samplers' distributions without a GPU).
//
// Why not the reference's generator: Sailfish seeds std::mt19937 from std::random_device in every
// sampling routine (src/CollapsedEMOptimizer.cpp:463-464, src/CollapsedGibbsSampler.cpp:104-105,
// 227-228), so its draws are not reproducible and only distributional parity is definable.  A
// sequential Mersenne Twister has no place on a GPU; Philox4x32-10 (Salmon et al., SC'11) is
// counter-based: every (key, counter) pair is an independent stream, so any lane can draw without
// shared state and results are reproducible from the seed.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SF_HD __host__ __device__ __forceinline__
#else
#define SF_HD inline
#endif

namespace sfgpu {

#if !defined(SFGPU_X_PHILOX_ROUNDS)        // (dev, timing only: tools/gibbs_variants.sh builds the samplers with fewer rounds / a cut walk to see what each part costs)
#define SFGPU_X_PHILOX_ROUNDS 10
#endif
#if !defined(SFGPU_X_WALK_MAX)
#define SFGPU_X_WALK_MAX (kBinvSwitch - 1u)
#endif

struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t out[4];
    int have;   // unread words in out[]

    // (seed, stream, substream) name one sequence; the BLOCK counter has a counter word of its own (ctr[0]), so that no
    // two sequences ever share a block.  (Round 1 advanced the word that also held the substream: block k of substream
    // d was block 0 of substream d + k -- consecutive bootstrap draws, and consecutive classes of a Gibbs chain, shared
    // random numbers.  Found by the two-sample tests against the reference's MultinomialSampler, tests/test_ref_units.py.)
    SF_HD void init(uint64_t seed, uint64_t stream, uint64_t substream) {
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32) ^ ((uint32_t)(substream >> 32) * 0x9E3779B1u);
        ctr[0] = 0u; ctr[1] = (uint32_t)substream;
        ctr[2] = (uint32_t)stream; ctr[3] = (uint32_t)(stream >> 32) ^ 0x5AF15u;
        have = 0;
    }
    SF_HD static void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
        uint64_t p = (uint64_t)a * b; hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
    }
    SF_HD void block() {
        uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
        for (int r = 0; r < SFGPU_X_PHILOX_ROUNDS; ++r) {
            uint32_t hi0, lo0, hi1, lo1;
            mulhilo(0xD2511F53u, c0, hi0, lo0);
            mulhilo(0xCD9E8D57u, c2, hi1, lo1);
            uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
        ++ctr[0];                                   // 2^32 blocks per sequence; the sequence id lives in the other words
        have = 4;
    }
    // (selects, not out[--have]: a dynamically indexed array of a device thread lives in LDS or scratch -- a dependent round trip per word)
    SF_HD uint32_t next32() {
        if (have == 0) block();
        --have;
        return have == 3 ? out[3] : (have == 2 ? out[2] : (have == 1 ? out[1] : out[0]));
    }
    // uniform in the open interval (0, 1), 53 random bits
    // (a PAIR of words per draw, one refill test: uniform() is the only consumer in the samplers, `have` is 0, 2 or 4 there; the words
    //  come in next32()'s order)
    SF_HD double uniform() {
        if (have < 2) block();
        have -= 2;
        const uint64_t hi = have == 2 ? out[3] : out[1], lo = have == 2 ? out[2] : out[0];
        uint64_t x = ((hi << 32) | lo) >> 11;
        return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
    }
};

// The BINV walk below, f_x = f_(x-1) * (a / x - s) against u_x = u_(x-1) - f_(x-1), costs an f64 division per step.  Both
// sides scaled by x! need none:  F_x = x! f_x = F_(x-1) * (a - s x),  U_x = x! u_x = x * (U_(x-1) - F_(x-1)),  and
// u_x > f_x  <=>  U_x > F_x.  170! is the last factorial a double holds: from step kBinvSwitch - 4 ... - 1 on the walk continues
// unscaled.  BINV is used up to a mean of kBinvMaxMean = 110 (the customary 30 dates from scalar machines: in a wavefront of 64
// chains BTPE costs every lane the slowest lane's rejection path, the walk costs mean + ~3 sd cheap steps -- 7 vector instructions
// each since round 5; measured on the Gibbs sampler over cfg3's classes: 30 -> 60 was -14 % in round 3, 60 -> 100 is -4 % with the
// round-5 walk, 120 no better).  At the bound the walk passes step 164 with probability < 1e-6 (5 sd), and then only gets slower.
#if !defined(SFGPU_BINV_MAX_MEAN)
#define SFGPU_BINV_MAX_MEAN 110.0
#endif
constexpr double kBinvMaxMean = SFGPU_BINV_MAX_MEAN;
#if !defined(SFGPU_BINV_SWITCH)
#define SFGPU_BINV_SWITCH 168          // (tests/test_sampling_cpu.py builds the header with 8 as well, to walk through the switch)
#endif
constexpr uint32_t kBinvSwitch = SFGPU_BINV_SWITCH;
static_assert(kBinvSwitch >= 8u && kBinvSwitch <= 170u, "the scaled walk holds x! in a double and leaves it four steps before the switch");
constexpr uint32_t kBinvPowMax = 1024;
constexpr double binv_unscale(uint32_t n) { double f = 1.0; for (uint32_t i = 2; i <= n; ++i) f *= (double)i; return 1.0 / f; }
constexpr double kBinvUnscale = binv_unscale(kBinvSwitch - 1u);      // ~ 1 / 167! (U and F get the same factor: its value does not matter, only that neither leaves the range)

// a / b for the samplers' ratios (b > 0, finite): on the device the reciprocal instruction and two Newton steps instead of the IEEE
// division sequence (~6 instead of ~15 instructions; the last bit may differ -- the ratio of two weights that are sums of products)
SF_HD double ratio_of(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SFGPU_X_IEEE_DIV)
    double y = __builtin_amdgcn_rcp(b);
    y = fma(fma(-b, y, 1.0), y, y);
    y = fma(fma(-b, y, 1.0), y, y);
    return a * y;
#else
    return a / b;
#endif
}

// BINV: Binomial(n, r) for r <= 1/2 and a mean n r < kBinvMaxMean, by walking the CDF from 0 (q = 1 - r)
SF_HD uint32_t binv(Philox& g, uint32_t n, double r, double q) {
    const double dn = (double)n;
    uint32_t y;
    {
        const double s = ratio_of(r, q), a = (dn + 1.0) * s;
        // f_0 = q^n: by squaring for n <= kBinvPowMax (<= 2 log2 n multiplications against ~110 instructions of log1p + exp; the error
        // grows like n ulp / 2 -- 1e-13 at the bound), exp(n log1p(-r)) beyond.  q^n >= e^(-1.39 * 110): no underflow of the result.
        double f0;
#if defined(SFGPU_X_NOPOW)              // (dev, timing only: what does q^n cost?)
        if (n != 0xFFFFFFFFu) f0 = q; else
#endif
        if (n <= kBinvPowMax) {
            f0 = 1.0;
            double b = q;
            for (uint32_t e = n; e != 0u; e >>= 1) { if (e & 1u) f0 *= b; b *= b; }
        } else f0 = exp(dn * log1p(-r));
        // (the walk is the sampler's hot loop.  Four steps per trip, each behind its own test and nested in the one before: a lane
        //  that is done drops out of the rest of the trip with one mask instruction, and the trip's bookkeeping -- the bound, the
        //  branch -- is paid once per four steps.  With the end-of-range and the switch tests inside every step the compiler spent 25
        //  scalar instructions per step on lane masks next to 10 vector ones; this form has 7 vector and ~3 scalar.  A lane whose
        //  walk runs past n -- rounding: F is 0 or noise from there on -- walks on to the switch and redraws.)
        for (;;) {
            double F = f0, U = g.uniform();                  // x! f_x and x! u_x
            uint32_t x = 0;
            double dx = 0.0;
#define SFGPU_BINV_STEP ++x; dx += 1.0; U = dx * (U - F); F *= fma(-s, dx, a);      /* scaled by x!: no division */
            while (U > F && x < SFGPU_X_WALK_MAX - 3u) {
                SFGPU_BINV_STEP
                if (U > F) { SFGPU_BINV_STEP
                    if (U > F) { SFGPU_BINV_STEP
                        if (U > F) { SFGPU_BINV_STEP } } }
            }
#undef SFGPU_BINV_STEP
#if defined(SFGPU_X_WALK_CUT)
            y = x; break;
#endif
            if (x > n) continue;                              // rounding ran off the end: redraw
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(U), "+v"(F));              // (the exit state is compared HERE, once -- not tracked in a lane mask through every step)
#endif
            if (!(U > F)) { y = x; break; }
            // kBinvSwitch - 4 <= x <= n, x < kBinvSwitch (probability < 1e-40 for the means BINV is used for): on unscaled
            bool ok = true;
            U *= kBinvUnscale; F *= kBinvUnscale;
            while (U > F) { U -= F; ++x; if (x > n) { ok = false; break; } F *= (a / (double)x - s); }
            if (ok) { y = x; break; }
        }
    }
    return y;
}

// Binomial(n, p), exact: inversion (BINV) for small means, BTPE (Kachitvichyanukul & Schmeiser, 1988)
// otherwise.  n < 2^32.
SF_HD uint32_t binomial(Philox& g, uint32_t n, double p) {
    if (n == 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double r = flip ? 1.0 - p : p;       // r <= 0.5
    const double q = 1.0 - r;
    const double dn = (double)n;
    double y;
    if (dn * r < kBinvMaxMean) {
        y = (double)binv(g, n, r, q);
    } else {
#if defined(SFGPU_X_NO_BTPE)
        y = 0.0;
#else
        // ---- BTPE
        const double fm = dn * r + r;
        const double m = floor(fm);
        const double nrq = dn * r * q;
        const double p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
        const double xm = m + 0.5, xl = xm - p1, xr = xm + p1;
        const double c = 0.134 + 20.5 / (15.3 + m);
        double al = (fm - xl) / (fm - xl * r);
        const double laml = al * (1.0 + al / 2.0);
        al = (xr - fm) / (xr * q);
        const double lamr = al * (1.0 + al / 2.0);
        const double p2 = p1 * (1.0 + 2.0 * c), p3 = p2 + c / laml, p4 = p3 + c / lamr;
        for (;;) {
            const double u = g.uniform() * p4;
            double v = g.uniform();
            if (u <= p1) { y = floor(xm - p1 * v + u); break; }            // triangular centre: accept
            if (u <= p2) {                                                   // parallelograms
                const double x = xl + (u - p1) / c;
                v = v * c + 1.0 - fabs(m - x + 0.5) / p1;
                if (v > 1.0) continue;
                y = floor(x);
            } else if (u <= p3) {                                            // left exponential tail
                y = floor(xl + log(v) / laml);
                if (y < 0.0) continue;
                v = v * (u - p2) * laml;
            } else {                                                         // right exponential tail
                y = floor(xr - log(v) / lamr);
                if (y > dn) continue;
                v = v * (u - p3) * lamr;
            }
            const double k = fabs(y - m);
            if (k <= 20.0 || k >= nrq / 2.0 - 1.0) {
                // explicit evaluation of f(y)/f(m) by the recurrence
                const double s = r / q, a = s * (dn + 1.0);
                if (k <= 20.0) {
                    // f(y)/f(m) = prod (a - s i) / prod i over the <= 20 steps between y and m: ONE comparison of products instead
                    // of a division per step (every factor is <= n + 1 < 2^32.1, so 20 of them stay below 1e193).  In a
                    // wavefront of 64 chains some lane takes this path in nearly every trial: -18 % on the Gibbs sampler.
                    double num = 1.0, den = 1.0;
                    const double i0 = (m < y ? m : y) + 1.0, i1 = m < y ? y : m;
                    for (double i = i0; i <= i1; i += 1.0) { num *= (a - s * i); den *= i; }
                    if (m < y ? (v * den > num) : (v * num > den)) continue;
                    break;
                }
                double F = 1.0;                      // far from the mode (only when n r q is small): the factors decay, F may underflow to 0
                if (m < y) { for (double i = m + 1.0; i <= y; i += 1.0) F *= (a / i - s); }
                else if (m > y) { for (double i = y + 1.0; i <= m; i += 1.0) F /= (a / i - s); }
                if (v > F) continue;
                break;
            }
            // squeeze, then the Stirling-series bound
            const double rho = (k / nrq) * ((k * (k / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
            const double t = -k * k / (2.0 * nrq);
            const double A = log(v);
            if (A < t - rho) break;
            if (A > t + rho) continue;
            const double x1 = y + 1.0, f1 = m + 1.0, z = dn + 1.0 - m, w = dn - y + 1.0;
            const double x2 = x1 * x1, f2 = f1 * f1, z2 = z * z, w2 = w * w;
            const double bound = xm * log(f1 / x1) + (dn - m + 0.5) * log(z / w) + (y - m) * log(w * r / (x1 * q))
                + (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / f2) / f2) / f2) / f2) / f1 / 166320.0
                + (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / z2) / z2) / z2) / z2) / z / 166320.0
                + (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / x2) / x2) / x2) / x2) / x1 / 166320.0
                + (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / w2) / w2) / w2) / w2) / w / 166320.0;
            if (A > bound) continue;
            break;
        }
#endif
    }
    if (y < 0.0) y = 0.0;
    if (y > dn) y = dn;
    uint32_t k = (uint32_t)y;
    return flip ? n - k : k;
}

// The same distribution from BINV alone: a sum of independent Binomial(n_i, r) with sum n_i = n IS Binomial(n, r), so a large mean is
// drawn as equal parts of mean < kBinvMaxMean each.  For kernels that cannot afford BTPE's registers (52 VGPRs more in the Gibbs phase
// kernel: four wavefronts per SIMD instead of six) and know their n to be moderate -- the cost is ~ n r walk steps.
SF_HD uint32_t binomial_by_inversion(Philox& g, uint32_t n, double p) {
    if (n == 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double r = flip ? 1.0 - p : p;       // r <= 0.5
    const double q = 1.0 - r;
    const double mean = (double)n * r;
    uint32_t k;
    if (mean < kBinvMaxMean) k = binv(g, n, r, q);
    else {
        const uint32_t parts = (uint32_t)(mean / kBinvMaxMean) + 1u;             // <= n / 220 + 1
        const uint32_t base = n / parts, extra = n % parts;                      // `extra` parts of base + 1, the others of base
        k = 0;
        for (uint32_t i = 0; i < parts; ++i) k += binv(g, base + (i < extra ? 1u : 0u), r, q);
    }
    return flip ? n - k : k;
}

}  // namespace sfgpu
