"""CPU tests: the oracle against every golden / known-answer vector we hold for the hot path."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _csr(classes):
    rowptr = np.zeros(len(classes) + 1, np.uint64)
    rowptr[1:] = np.cumsum([len(c) for c in classes])
    ids = np.array([t for c in classes for t in c], np.uint32)
    return rowptr, ids


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(GOLD, "survey_kat.json")))


def test_xxh64_golden_vectors(built):
    """fixture generated from the reference's own xxhash.c (tests/golden/make_xxh64_golden.py)"""
    g = json.load(open(os.path.join(GOLD, "xxh64_vectors.json")))
    assert len(g["vectors"]) >= 40
    for v in g["vectors"]:
        assert "%016x" % O.xxh64(np.array(v["ids"], np.uint32).tobytes()) == v["xxh64"], v["ids"]


def test_xxh64_survey_kat(built):
    for ids, want in [([], "ef46db3751d8e999"), ([5], "c3d48b2f79d2b939"), ([2, 9], "b6e5dcf465f8e9bc"),
                      ([1, 2, 3], "b5148cb100a911fc"), (list(range(9)), "05df6b7adb49d27f")]:
        assert "%016x" % O.xxh64(np.array(ids, np.uint32).tobytes()) == want


def test_xxh64_vs_compiled_reference_and_pypi(built):
    """oracle == oracle/_ref (reference xxhash.c, when built) == python-xxhash, all byte lengths 0..300"""
    import xxhash
    R = O.ref_xxhash()
    rng = np.random.default_rng(1)
    for n in range(0, 301):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        h = O.xxh64(b)
        assert h == xxhash.xxh64(b, seed=0).intdigest()
        if R is not None:
            buf = C.create_string_buffer(b, len(b))
            assert h == R.XXH64(C.cast(buf, C.c_void_p), len(b), 0)
    assert O.xxh64(b"abc", seed=7) == xxhash.xxh64(b"abc", seed=7).intdigest()


def test_builder_survey_kat(built, kat):
    k = kat["builder"]
    b = O.EqBuilder()
    for r in k["reads"]:
        b.add_batch(np.array(r, np.uint32), np.array([0, len(r)], np.uint64))
    rowptr, ids, counts, hashes = b.finish()
    assert b.n_classes == k["n_classes"] and b.total_reads == k["total"]
    got = {",".join(map(str, ids[rowptr[c]:rowptr[c + 1]])): int(counts[c]) for c in range(b.n_classes)}
    assert got == k["counts_by_label"]


def test_builder_semantics_vs_dict(built):
    """ordered lists are the key: permutations and duplicate ids are distinct classes; empties skipped"""
    rng = np.random.default_rng(3)
    reads = []
    for _ in range(5000):
        n = int(rng.choice([0, 1, 1, 2, 3, 7, 8, 9, 40, 200]))
        reads.append(rng.integers(0, 30, n).astype(np.uint32))
    reads += [np.array([1, 2], np.uint32), np.array([2, 1], np.uint32), np.array([1, 1, 2], np.uint32)] * 3
    off = np.zeros(len(reads) + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in reads])
    b = O.EqBuilder(); b.add_batch(np.concatenate(reads), off)
    rowptr, ids, counts, hashes = b.finish()
    want = {}
    for r in reads:
        if len(r):
            want[tuple(r.tolist())] = want.get(tuple(r.tolist()), 0) + 1
    got = {tuple(ids[rowptr[c]:rowptr[c + 1]].tolist()): int(counts[c]) for c in range(b.n_classes)}
    assert got == want and b.total_reads == sum(want.values())
    # canonical order: (first id, hash, len, label)
    keys = [(int(ids[rowptr[c]]), int(hashes[c]), int(rowptr[c + 1] - rowptr[c])) for c in range(b.n_classes)]
    assert keys == sorted(keys)
    for c in range(b.n_classes):
        assert int(hashes[c]) == O.xxh64(ids[rowptr[c]:rowptr[c + 1]].tobytes())


def test_em_survey_kat_toy5(built, kat):
    k = kat["em_toy5"]
    eff = np.array(k["ref_len"], float) - k["eff_len_minus"]
    rowptr, ids = _csr(k["classes"]); cnt = np.array(k["counts"], np.uint64)
    rc, a, m, st = O.em_optimize(eff, rowptr, ids, cnt, k["num_mapped"], tol=k["tol"], max_iter=k["max_iter"])
    assert rc == 0 and st["iters"] == k["stop_iter"]
    np.testing.assert_allclose(a, k["em_est_count"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(m, k["em_mass"], rtol=1e-14, atol=0)
    rc, a, m, st = O.em_optimize(eff, rowptr, ids, cnt, k["num_mapped"], use_vbem=True, tol=k["tol"], max_iter=k["max_iter"])
    assert rc == 0
    # VBEM ran through a stand-in digamma in the survey build too; agreement is to ~1e-13
    np.testing.assert_allclose(a, k["vbem_est_count"], rtol=1e-12, atol=0)


def test_em_survey_kat_toy7_and_sampling(built, kat):
    k = kat["em_toy7"]
    eff = np.array(k["ref_len"], float) - k["eff_len_minus"]
    rowptr, ids = _csr(k["classes"]); cnt = np.array(k["counts"], np.uint64)
    rc, a, m, st = O.em_optimize(eff, rowptr, ids, cnt, k["num_mapped"])
    assert rc == 0
    np.testing.assert_allclose(a, k["em_est_count_6dp"], atol=6e-7)
    # sampling paths are random_device seeded in the reference: distributional agreement only
    rc, bs, _ = O.bootstrap(eff, rowptr, ids, cnt, 400, seed=11)
    assert rc == 0 and np.all(np.abs(bs.mean(0) - np.array(k["bootstrap_mean_200"])) < 4 * bs.std(0) / np.sqrt(200) + 0.5)
    rc, gs = O.gibbs(eff, m, rowptr, ids, cnt, k["num_mapped"], 2000, seed=5)
    assert rc == 0 and np.all(gs.sum(1) == k["num_mapped"])
    assert np.all(np.abs(gs.mean(0) - np.array(k["gibbs_mean_200"])) < 12.0)


def test_em_error_paths(built):
    eff = np.array([10.0, 20.0])
    rc, *_ = O.em_optimize(eff, np.array([0], np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint64), 0)
    assert rc == 1          # "no transcripts expressed"
    rc, a, m, st = O.em_optimize(eff, np.array([0, 1], np.uint64), np.array([1], np.uint32), np.array([0], np.uint64), 0)
    assert rc == 2          # alpha sum too small


def test_digamma_vs_scipy(built):
    from scipy.special import digamma
    xs = np.concatenate([np.logspace(-8, 8, 400), [0.01, 1.0, 1.4616321449683623, 2.0, 10.0, 485.0]])
    got = np.array([O.digamma(x) for x in xs])
    want = digamma(xs)
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 5e-15


def test_efflen_tables(built):
    cf = O.cf_gaussian(1000, 200, 80)
    assert cf[0] == 0.0 and np.all(np.diff(cf) >= 0) and abs(cf[999] - 202.0) < 3.0
    i = np.arange(1000.0); d = np.exp(-0.5 * ((i - 200) / 80.0) ** 2) / 80.0
    np.testing.assert_allclose(cf[1:], (np.cumsum(i * d) / np.cumsum(d))[1:], rtol=1e-12)
    fld = O.fld_gaussian_counts()
    assert abs(int(fld.sum()) - 10000) <= 40 and fld[200] == fld.max()
    cfc = O.cf_counts(fld.astype(np.uint32))
    assert abs(cfc[999] - cf[999]) < 0.5
    ref_len = np.array([1, 50, 199, 200, 201, 999, 1000, 1001, 100000], np.uint32)
    eff = O.efflen_smoothed(ref_len, cf)
    for L, e in zip(ref_len, eff):
        c = cf[min(int(L), 999)]
        want = float(L) - c + 1.0
        assert e == (want if want >= 1.0 else float(L))


def _empirical_efflen_numpy(fl, ref_len):
    """independent restatement of --unsmoothedFLD (src/SailfishQuantify.cpp:717-767 over
    src/EmpiricalDistribution.cpp:29-118) with numpy scalars: float32 pdf table, double accumulation"""
    fl = np.asarray(fl, np.uint32); n = len(fl)
    valsum = float(np.sum(fl.astype(np.float64)))          # counts are integers: any order gives the same double
    cum, last, maxval = 0.0, 0, 1
    with np.errstate(invalid="ignore", divide="ignore"):
        while last < n:
            cum += np.float64(fl[last]) / np.float64(valsum); maxval = last
            if cum > 1.0 - 1e-6:
                break
            last += 1
        kept = float(np.sum(fl[:last].astype(np.float64)))
        pdf = (fl[:maxval].astype(np.float64) / np.float64(kept)).astype(np.float32) if maxval else np.zeros(0, np.float32)
    i, j = 0, n - 1
    u, v = int(fl[0]), int(fl[n - 1])
    while i < j:
        if u <= v:
            v = (v - u) & 0xFFFFFFFF; i += 1; u = int(fl[i])
        else:
            u = (u - v) & 0xFFFFFFFF; j -= 1; v = int(fl[j])
    med = np.float32(i) if maxval else np.float32(np.nan)
    out = np.zeros(len(ref_len))
    for t, L in enumerate(np.asarray(ref_len, np.uint32)):
        L = int(L)
        if float(L) <= float(med) or not (n - 1 > 0):
            out[t] = float(L); continue
        e = np.float64(0.0)
        for l in range(0, min(L, n - 1) + 1):
            p = np.float64(pdf[l]) if l < maxval else np.float64(0.0)
            e = e + p * (np.float64(L - l) + 1.0)
        out[t] = float(e)
    return out


def test_efflen_unsmoothed_fld(built):
    """--unsmoothedFLD branch of a14: oracle vs an independent numpy restatement, incl. the odd corners
    (all mass in one bin, empty pdf table -> NaN median -> effective length 0, two-bin distributions)"""
    rng = np.random.default_rng(4)
    ref_len = np.array([1, 2, 50, 150, 199, 200, 201, 250, 999, 1000, 1001, 5000, 100000], np.uint32)
    cases = [O.fld_gaussian_counts(1000, 200, 80, 10000).astype(np.uint32),
             O.fld_gaussian_counts(1000, 250, 60, 20000).astype(np.uint32),
             rng.integers(0, 50, 1000).astype(np.uint32),
             np.r_[np.zeros(300), 7, np.zeros(699)].astype(np.uint32),          # one bin
             np.r_[5, np.zeros(999)].astype(np.uint32),                         # all mass at length 0: empty table
             np.r_[np.zeros(10), 3, np.zeros(20), 9, np.zeros(968)].astype(np.uint32),
             np.array([0, 4], np.uint32), np.array([4, 0], np.uint32)]
    for fl in cases:
        got = O.efflen_empirical(fl, ref_len)
        want = _empirical_efflen_numpy(fl, ref_len)
        np.testing.assert_array_equal(got, want)
    # sanity on the Gaussian case: long transcripts lose about the mean fragment length
    eff = O.efflen_empirical(cases[0], ref_len)
    assert abs((100000 - eff[-1]) - 199.0) < 3.0 and eff[0] == 1.0 and eff[3] == 150.0


def test_tpm_columns(built):
    rng = np.random.default_rng(0)
    a = rng.random(100) * 50; a[::7] = 0; ln = rng.integers(200, 5000, 100).astype(float)
    t = O.tpm(a, ln, a.sum())
    assert abs(t.sum() - 1e6) < 1e-6 and np.all(t[::7] == 0)
    np.testing.assert_allclose(t, (a / ln) / (a / ln).sum() * 1e6, rtol=1e-12)


def test_all_cores_legs_agree_with_one_thread(built):
    """the multi-threaded timing legs of bench.py's cpu_baseline compute what the serial restatement computes"""
    rng = np.random.default_rng(12)
    M, R = 3000, 120_000
    pool = [np.sort(rng.choice(M, k, replace=False)) for k in rng.integers(1, 7, 9000)]
    pick = rng.integers(0, len(pool), R)
    ids = np.concatenate([pool[p] for p in pick]).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum([len(pool[p]) for p in pick])]).astype(np.uint64)
    b1 = O.EqBuilder(); b1.add_batch(ids, off); r1 = b1.finish()
    for T in (1, 3, 8):
        b = O.EqBuilder(); assert b.add_batch_mt(ids, off, T) > 0.0
        r = b.finish()
        assert b.total_reads == b1.total_reads and all(np.array_equal(x, y) for x, y in zip(r, r1))
    rp, ii, cc, _ = r1
    eff = rng.integers(200, 3000, M).astype(np.float64)
    for vb in (False, True):
        rc, a, m, st = O.em_optimize(eff, rp, ii, cc, R, use_vbem=vb, tol=0.0, min_iter=25, max_iter=25)
        for T in (1, 4):
            sec, am = O.em_iterations_mt(eff, rp, ii, cc, R, 25, T, use_vbem=vb)
            nz = a > 0
            assert sec > 0 and np.max(np.abs(am[nz] - a[nz]) / a[nz]) < 1e-11
