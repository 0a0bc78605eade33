"""Pins against the REFERENCE's own code: oracle/_ref/libsailfish_ref.so = oracle/ref_glue.cpp around reference units
compiled unmodified from /root/reference (src/LibraryFormat.cpp, include/MultinomialSampler.hpp,
include/cuckoohash_map.hh, src/xxhash.c), and tests/golden/ref_unit_vectors.json generated from it
(tests/golden/make_ref_unit_vectors.py).  The fixture is what the GPU box checks against; when the .so is present
(it travels with gpurun) the live library is checked too."""
import json
import os

import numpy as np
import pytest
from scipy import stats

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def vec(built):
    return json.load(open(os.path.join(GOLD, "ref_unit_vectors.json")))


# ---- LibraryFormat -------------------------------------------------------------------------------
def test_library_format_matches_reference(vec):
    from sailfish_amd import hits as H
    lf = vec["library_format"]
    assert H.MAX_LIB_TYPE_ID == lf["max_id"] == 39
    L = O.ref_sailfish()
    for f in lf["formats"]:
        t = tuple(f["format"])
        assert H.format_id(t) == f["id"] and H.format_check(t) == f["check"] and H.format_str(t) == f["str"], f
        if L is not None:
            assert int(L.ref_format_id(*t)) == f["id"] and bool(L.ref_format_check(*t)) == f["check"]
    for e in lf["from_id"]:
        assert H.format_from_id(e["id"]) == tuple(e["format"])
    # every name of parseLibraryFormatStringNew's table is a valid combination with a distinct id
    ids = {n: H.format_id(f) for n, f in H.LIBRARY_FORMATS.items()}
    assert len(set(ids.values())) == len(ids) == 12
    assert all(H.format_check(f) for f in H.LIBRARY_FORMATS.values())
    assert all(H.format_from_id(i) == H.LIBRARY_FORMATS[n] for n, i in ids.items())


# ---- the class table: libcuckoo upsert + XXH64 (addGroup) ------------------------------------------
def _gold_table(vec):
    e = vec["eq_build"]
    ids = np.array(e["ids"], np.uint32); off = np.array(e["off"], np.uint64)
    tab = {tuple(r["label"]): (r["count"], int(r["hash"], 16)) for r in e["table"]}
    return ids, off, tab


def test_oracle_builder_matches_reference_table(vec):
    ids, off, tab = _gold_table(vec)
    b = O.EqBuilder(); b.add_batch(ids, off)
    rp, ii, cc, hh = b.finish()
    got = {tuple(ii[rp[c]:rp[c + 1]].tolist()): (int(cc[c]), int(hh[c])) for c in range(b.n_classes)}
    assert got == tab
    assert b.total_reads == sum(v[0] for v in tab.values()) == 4000          # the 5 empty lists never reach addGroup
    assert (7, 8) in tab and (8, 7) in tab                                  # the ORDERED list is the key
    if O.ref_sailfish() is not None:                                         # the live reference, concurrent upserts
        assert O.ref_eq_build(ids, off, n_threads=3) == tab
        rng = np.random.default_rng(3)
        n = 3000
        lens = rng.integers(0, 12, n)
        off2 = np.zeros(n + 1, np.uint64); off2[1:] = np.cumsum(lens)
        ids2 = rng.integers(0, 6, int(off2[-1])).astype(np.uint32)           # few ids: many repeated labels, repeated ids inside labels
        b2 = O.EqBuilder(); b2.add_batch(ids2, off2)
        rp, ii, cc, hh = b2.finish()
        got2 = {tuple(ii[rp[c]:rp[c + 1]].tolist()): (int(cc[c]), int(hh[c])) for c in range(b2.n_classes)}
        assert got2 == O.ref_eq_build(ids2, off2, n_threads=4)


@pytest.mark.gpu
def test_device_builder_matches_reference_table(vec, gpu):
    import torch
    import sailfish_amd as sf
    ids, off, tab = _gold_table(vec)
    eq = sf.EquivalenceClassBuilder(device=gpu)
    eq.start()
    eq.add_batch(torch.from_numpy(ids.view(np.int32).copy()).to(gpu), torch.from_numpy(off.astype(np.uint32).view(np.int32).copy()).to(gpu))
    eq.finish()
    rp, ii, cc, hh = eq.eqVec().to_numpy()
    got = {tuple(ii[rp[c]:rp[c + 1]].tolist()): (int(cc[c]), int(hh[c])) for c in range(eq.n_classes)}
    assert got == tab and eq.total_reads == 4000


# ---- MultinomialSampler ----------------------------------------------------------------------------
def _chi2_p(counts, p, n):
    exp = np.asarray(p) * n
    return float(stats.chisquare(np.asarray(counts, float), exp * (np.sum(counts) / exp.sum())).pvalue)


@pytest.mark.parametrize("key", ["k7", "k150"])
def test_oracle_multinomial_matches_reference_sampler(vec, key):
    """same (n, p): the reference's draws (fixture; live when the .so is there) and the restatement's are samples of one
    distribution -- goodness of fit of each against n p, and a two-sample test of the pooled draws"""
    m = vec["multinomial"][key]
    n, p = m["n"], np.array(m["p"])
    ref = np.array(m["draws"], np.int64)
    assert np.all(ref.sum(1) == n)                                           # the reference drops no draw at these sizes
    mine = np.stack([O.multinomial(n, p, seed=100 + i).astype(np.int64) for i in range(len(ref))])
    assert np.all(mine.sum(1) == n)
    assert _chi2_p(ref.sum(0), p, n) > 1e-4 and _chi2_p(mine.sum(0), p, n) > 1e-4
    # per draw: the chi-square statistics of both samplers follow chi2(k-1)
    k = len(p)
    for d in (ref, mine):
        stat = ((d - n * p) ** 2 / (n * p)).sum(1)
        assert stats.kstest(stat, "chi2", args=(k - 1,)).pvalue > 1e-4
    # two-sample: contingency of the pooled counts
    assert stats.chi2_contingency(np.stack([ref.sum(0), mine.sum(0)]))[1] > 1e-4
    # per-category variance n p (1 - p): ratio of the samplers' variances within the F-test band
    v_ref, v_mine = ref.var(0, ddof=1), mine.var(0, ddof=1)
    big = n * p > 50
    f = v_mine[big] / v_ref[big]
    lo, hi = stats.f.ppf([1e-5, 1 - 1e-5], len(ref) - 1, len(ref) - 1)
    assert np.all((f > lo) & (f < hi))
    L = O.ref_sailfish()
    if L is not None:
        live = np.stack([O.ref_multinomial(n, p).astype(np.int64) for _ in range(8)])
        assert np.all(live.sum(1) == n) and stats.chi2_contingency(np.stack([live.sum(0), mine.sum(0)]))[1] > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["k7", "k150"])
def test_device_multinomial_matches_reference_sampler(vec, gpu, key):
    """the bootstrap's resample (binomial-tree multinomial, csrc/sampling.hip) against the reference sampler's draws:
    classes with counts proportional to p -> sampCounts(N = sum count, p = count / N) (src/CollapsedEMOptimizer.cpp:460-468)"""
    import torch
    import sailfish_amd as sf
    m = vec["multinomial"][key]
    n, p = m["n"], np.array(m["p"])
    ref = np.array(m["draws"], np.int64)
    cnt = np.floor(p * n + 0.5).astype(np.int64)
    cnt[np.argmax(cnt)] += n - cnt.sum()                                     # counts sum to n exactly
    q = cnt / n
    k = len(p)
    rowptr = torch.arange(k + 1, dtype=torch.int32, device=gpu)             # k singleton classes over k transcripts
    ids = torch.arange(k, dtype=torch.int32, device=gpu)
    prob = sf.EMProblem(torch.full((k,), 1000.0, dtype=torch.float64, device=gpu), rowptr, ids, torch.from_numpy(cnt).to(gpu), n)
    draws = np.stack([prob.bootstrap_counts(seed=9, draw=i).cpu().numpy() for i in range(len(ref))])
    prob.close()
    assert np.all(draws.sum(1) == n)
    assert _chi2_p(draws.sum(0), q, n) > 1e-4
    stat = ((draws - n * q) ** 2 / (n * q)).sum(1)
    assert stats.kstest(stat, "chi2", args=(k - 1,)).pvalue > 1e-4
    # against the reference's sample (p and q differ by rounding to whole reads: < 1/(2n) per category)
    assert stats.chi2_contingency(np.stack([ref.sum(0), draws.sum(0)]))[1] > 1e-4
    v_ref, v_dev = ref.var(0, ddof=1), draws.var(0, ddof=1)
    big = n * p > 50
    f = v_dev[big] / v_ref[big]
    lo, hi = stats.f.ppf([1e-5, 1 - 1e-5], len(ref) - 1, len(ref) - 1)
    assert np.all((f > lo) & (f < hi))
