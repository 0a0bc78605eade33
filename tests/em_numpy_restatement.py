"""An INDEPENDENT restatement of CollapsedEMOptimizer::optimize for the tests: vectorised numpy over whole arrays,
written from the arithmetic contract of SURVEY.md Appendix A (which cites src/CollapsedEMOptimizer.cpp:734-891) and
NOT from oracle/sf_oracle.c -- a second reading of the reference, so that a misreading shared by the C oracle and the
HIP kernels (same author) would show up as a disagreement with this file.

Differences in kind from the C oracle, on purpose: the stored, normalised auxiliary weights w_{c,i} are materialised
as one array and used exactly as the reference uses them (the HIP path folds them away); sums run over numpy
reductions (pairwise), not serial loops; digamma comes from scipy (Boost in the reference, a series in the C oracle).
An mpmath variant (50 digits) of the same recurrences arbitrates on small problems."""
import numpy as np
from scipy.special import digamma

DENORM_MIN = 4.9406564584124654e-324          # minEQClassWeight = minWeight (:33-34)
PRIOR_ALPHA = 0.01                            # :786
ALPHA_CHECK_CUTOFF = 1e-2
MIN_ALPHA = 1e-8


class Problem:
    def __init__(self, eff_len, rowptr, ids, counts, num_mapped):
        self.M = len(eff_len)
        self.rp = np.asarray(rowptr, np.int64); self.ids = np.asarray(ids, np.int64)
        self.cnt = np.asarray(counts, np.float64)                      # uint64 -> double at every use (:264)
        self.N = float(num_mapped)
        self.k = np.diff(self.rp)
        self.row = np.repeat(np.arange(len(self.k)), self.k)
        # :734-740  effLens(i) = noCorr ? RefLength : EffectiveLength; "if (effLens(i) <= 1.0) effLens(i) = 1.0"
        eff = np.asarray(eff_len, np.float64).copy()
        eff[eff <= 1.0] = 1.0
        self.eff = eff
        # :760-769  weights[i] = count / effLens(txp_i); wnorm = 1 / sum; weights[i] *= wnorm
        v = self.cnt[self.row] / eff[self.ids]
        self.w = v * (1.0 / np.add.reduceat(v, self.rp[:-1]))[self.row] if len(self.k) else v
        self.single = self.k[self.row] == 1 if len(self.k) else np.zeros(0, bool)
        # :774-803
        self.active = np.zeros(self.M, bool); self.active[self.ids] = True
        self.n_active = int(self.active.sum())

    def alpha0(self):
        return np.where(self.active, (1.0 / self.n_active) * self.N, 0.0)           # :800-803 uniformPrior * totalNumFrags

    def e_step(self, weight_vec, base):
        """EMUpdate_ (:236-277) with `weight_vec` in the role of alphaIn (EM) or expTheta (VBEM); `base` is what alphaOut
        holds before the classes are added (0, or the prior)."""
        out = np.full(self.M, base)
        if len(self.k) == 0:
            return out
        a = weight_vec[self.ids]
        v = a * self.w
        if base != 0.0:                                                   # VBEM: only expTheta > 0 terms take part (:344, :356)
            v = np.where(a > 0.0, v, 0.0)
        denom = np.add.reduceat(np.where(self.single, 0.0, v), self.rp[:-1])
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(denom > DENORM_MIN, self.cnt / denom, 0.0)       # invDenom = count / denom (:264)
        with np.errstate(invalid="ignore"):
            contrib = np.where(np.isnan(v), 0.0, v * inv[self.row])        # "if (!std::isnan(v))" (:269)
        contrib = np.where(self.single, self.cnt[self.row], contrib)        # :275 singletons add the count
        np.add.at(out, self.ids, contrib)
        return out

    def step(self, alpha, vb):
        if not vb:
            return self.e_step(alpha, 0.0)
        log_norm = digamma(alpha.sum())                                    # :300-303
        with np.errstate(divide="ignore", invalid="ignore"):
            exp_theta = np.where(alpha > DENORM_MIN, np.exp(digamma(np.where(alpha > DENORM_MIN, alpha, 1.0)) - log_norm), 0.0)
        return self.e_step(exp_theta, PRIOR_ALPHA)                         # alphaOut = priorAlpha for EVERY transcript (:318)

    def optimize(self, vb=False, tol=0.01, min_iter=50, max_iter=10000):
        """-> (alpha after truncation, mass, iterations, converged)   (:809-891)"""
        alpha = self.alpha0()
        it, conv = 0, False
        while it < min_iter or (it < max_iter and not conv):               # :820
            new = self.step(alpha, vb)
            gate = new > ALPHA_CHECK_CUTOFF                                 # :852
            rel = np.abs(alpha[gate] - new[gate]) / new[gate]
            conv = bool(np.all(rel <= tol)) if rel.size else True           # :853-857 (converged starts true each round)
            alpha = new
            it += 1
        cutoff = (PRIOR_ALPHA + MIN_ALPHA) if vb else MIN_ALPHA             # :810-812
        alpha = np.where(alpha <= cutoff, 0.0, alpha)                       # truncateCountVector (:36-44)
        s = alpha.sum()
        return alpha, (alpha / s if s > 0 else alpha), it, conv


def tpm(alpha, length, num_mapped):
    """GZipWriter.cpp:216-245"""
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (alpha / num_mapped) / length
    d = x.sum()
    return x / d * 1e6


def optimize_mp(eff_len, rowptr, ids, counts, num_mapped, vb=False, n_iter=50, digits=50):
    """the same recurrences in mpmath (small problems only): -> alpha after n_iter rounds, untruncated (floats)"""
    import mpmath as mp
    mp.mp.dps = digits
    M = len(eff_len)
    eff = [mp.mpf(max(float(e), 1.0)) for e in eff_len]
    C = len(rowptr) - 1
    w = []
    for c in range(C):
        ts = [int(t) for t in ids[rowptr[c]:rowptr[c + 1]]]
        v = [mp.mpf(int(counts[c])) / eff[t] for t in ts]
        s = mp.fsum(v)
        w.append((ts, [x / s for x in v]))
    act = sorted({t for ts, _ in w for t in ts})
    alpha = [mp.mpf(0)] * M
    for t in act:
        alpha[t] = mp.mpf(num_mapped) / len(act)
    for _ in range(n_iter):
        if vb:
            ln = mp.digamma(mp.fsum(alpha))
            src = [mp.e ** (mp.digamma(a) - ln) if a > 0 else mp.mpf(0) for a in alpha]
            out = [mp.mpf(PRIOR_ALPHA)] * M
        else:
            src = alpha; out = [mp.mpf(0)] * M
        for c, (ts, ws) in enumerate(w):
            if len(ts) == 1:
                out[ts[0]] += int(counts[c]); continue
            v = [src[t] * x for t, x in zip(ts, ws)]
            d = mp.fsum(v)
            if d > 0:
                for t, x in zip(ts, v):
                    out[t] += x * (mp.mpf(int(counts[c])) / d)
        alpha = out
    return np.array([float(a) for a in alpha])
