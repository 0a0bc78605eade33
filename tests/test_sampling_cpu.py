"""CPU tests of the sampling primitives the HIP bootstrap / Gibbs kernels are built from
(sailfish_amd/csrc/rng.h compiled as plain C++): Philox4x32-10 known answers, uniform range,
and the exact binomial sampler (BINV + BTPE) against scipy's binomial distribution."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))


def _harness(tmp_path_factory, *defs):
    so = tmp_path_factory.mktemp("h") / "libharness.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", *defs, "-o", str(so), os.path.join(HERE, "sampling_harness.cpp")])
    L = C.CDLL(str(so))
    L.draw_binomial.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p]
    L.draw_binomial_by_inversion.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p]
    L.draw_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    L.philox_block.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
    return L


@pytest.fixture(scope="module")
def H(tmp_path_factory):
    return _harness(tmp_path_factory)


@pytest.fixture(scope="module")
def H8(tmp_path_factory):
    """the same header with the BINV walk leaving its factorial-scaled form after 4 - 7 steps instead of 164 - 167"""
    return _harness(tmp_path_factory, "-DSFGPU_BINV_SWITCH=8")


def test_philox_known_answers(H):
    """Random123 kat_vectors for philox4x32-10"""
    out = np.zeros(4, np.uint32)
    H.philox_block(0, 0, 0, 0, 0, 0, out.ctypes.data)
    assert [hex(x) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    H.philox_block(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, out.ctypes.data)
    assert [hex(x) for x in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    H.philox_block(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, out.ctypes.data)
    assert [hex(x) for x in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_uniform_open_interval_and_flat(H):
    u = np.zeros(200000)
    H.draw_uniform(42, 7, len(u), u.ctypes.data)
    assert u.min() > 0.0 and u.max() < 1.0
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    v = np.zeros(1000); H.draw_uniform(42, 8, len(v), v.ctypes.data)
    assert not np.array_equal(u[:1000], v)              # streams differ


def test_binv_switch_point_is_invisible(H, H8):
    """both forms of the walk (scaled by x!, and unscaled behind the switch) follow the same recurrence: the same seed gives
    the same draws wherever the switch lies -- up to the last bits of the running products, which move a draw that falls
    on a boundary of the CDF by one (rare)"""
    N = 20000
    for n, p in ((100, 0.2), (1000, 0.05), (59, 0.5), (130, 0.45), (100000, 0.0004), (215, 0.5), (1000, 0.105)):
        a = np.zeros(N, np.uint32); b = np.zeros(N, np.uint32)
        H.draw_binomial(99, n, p, N, a.ctypes.data); H8.draw_binomial(99, n, p, N, b.ctypes.data)
        assert b.max() > 8                                      # the walk did pass the switch
        d = a.astype(np.int64) - b.astype(np.int64)
        assert np.abs(d).max() <= 1 and (d != 0).mean() < 1e-3, (n, p, (d != 0).mean())


@pytest.mark.parametrize("n,p", [(1, 0.3), (10, 0.5), (100, 0.01), (1000, 0.02), (50, 0.9), (200, 0.16), (1000, 0.5),
                                 (100000, 0.001), (100000, 0.3), (3000000, 0.7), (4000000000, 1e-9), (4000000000, 0.25),
                                 (59, 0.5), (61, 0.5), (400, 0.075),
                                 # around the BINV / BTPE boundary (mean 110; 60 until round 5) and BTPE's explicit-product path
                                 (119, 0.5), (121, 0.5), (1000, 0.055), (600000, 0.0001), (130, 0.45), (250, 0.45), (700, 0.1),
                                 (219, 0.5), (221, 0.5), (1000, 0.109), (1000, 0.111), (1100000, 0.0001), (260, 0.45), (1200, 0.1)])
def test_binomial_matches_scipy(H, n, p):
    _check_against_scipy(H.draw_binomial, n, p)


@pytest.mark.parametrize("n,p", [(1, 0.3), (100, 0.01), (200, 0.16), (219, 0.5), (221, 0.5), (440, 0.5), (441, 0.5), (300, 0.45), (1000, 0.3),
                                 (1000, 0.7), (5000, 0.5), (100000, 0.001), (100000, 0.0025), (333, 0.34)])
def test_binomial_by_inversion_matches_scipy(H, n, p):
    """the BINV-only sampler of the Gibbs kernel's light form: one walk below a mean of 110, a sum of equal parts above"""
    _check_against_scipy(H.draw_binomial_by_inversion, n, p)


def _check_against_scipy(draw, n, p):
    N = 60000
    x = np.zeros(N, np.uint32)
    draw(12345 + n, n, p, N, x.ctypes.data)
    x = x.astype(np.float64)
    mean, var = n * p, n * p * (1 - p)
    assert x.min() >= 0 and x.max() <= n
    assert abs(x.mean() - mean) < 5 * np.sqrt(var / N) + 1e-12
    assert abs(x.var() - var) < 6 * var * np.sqrt(2.0 / N) + 1e-9 + 5 * np.sqrt(var * max(1 - 6 * p * (1 - p), 0) / N)
    # chi-square on equiprobable-ish bins of the exact pmf
    qs = np.unique(stats.binom.ppf(np.linspace(0.02, 0.98, 25), n, p))
    edges = np.concatenate([[-0.5], qs + 0.5, [n + 0.5]])
    obs, _ = np.histogram(x, edges)
    cdf = stats.binom.cdf(np.floor(edges[1:]), n, p) - stats.binom.cdf(np.floor(edges[:-1]), n, p)
    cdf[0] = stats.binom.cdf(np.floor(edges[1]), n, p)
    keep = cdf * N > 5
    if keep.sum() >= 3:
        chi = ((obs[keep] - cdf[keep] * N) ** 2 / (cdf[keep] * N)).sum()
        assert stats.chi2.sf(chi, keep.sum() - 1) > 1e-4, (chi, keep.sum())


def test_binomial_edges(H):
    x = np.zeros(10, np.uint32)
    H.draw_binomial(1, 0, 0.5, 10, x.ctypes.data); assert np.all(x == 0)
    H.draw_binomial(1, 77, 0.0, 10, x.ctypes.data); assert np.all(x == 0)
    H.draw_binomial(1, 77, 1.0, 10, x.ctypes.data); assert np.all(x == 77)


def test_sequences_share_no_block(H):
    """(seed, stream, substream) sequences are disjoint: neighbouring substreams (consecutive bootstrap draws, consecutive
    classes of a Gibbs chain) and neighbouring streams must not run into each other when a sequence needs many blocks"""
    H.draw_uniform_sub.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    seqs = {}
    for stream in (5, 6, (1 << 32) | 5):
        for sub in (0, 1, 2, 3, 1 << 32, (1 << 32) + 1):
            u = np.zeros(4096)
            H.draw_uniform_sub(77, stream, sub, len(u), u.ctypes.data)
            seqs[(stream, sub)] = u
    keys = list(seqs)
    for i, a in enumerate(keys):
        for b in keys[i + 1:]:
            assert len(np.intersect1d(seqs[a], seqs[b])) == 0, (a, b)


def test_multinomial_tree_draws_are_independent(H):
    """the bootstrap's resample (csrc/sampling.hip, restated node for node in the harness): pooled over the draws of one
    seed the category totals must fit n p -- correlated draws (round 1: consecutive draws shared random numbers) inflate
    the pooled statistic, which shows as a non-uniform p-value distribution over seeds"""
    H.tree_multinomial.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(4)
    k, n = 150, 60000
    p = rng.random(k); p /= p.sum()
    cnt = np.floor(p * n + 0.5).astype(np.int64); cnt[np.argmax(cnt)] += n - cnt.sum()
    q = cnt / n
    prefix = np.zeros(k + 1, np.uint64); prefix[1:] = np.cumsum(cnt)
    ps = []
    for seed in range(200, 240):
        tot = np.zeros(k, np.int64)
        for d in range(24):
            out = np.zeros(k, np.uint32)
            H.tree_multinomial(seed, d, k, prefix.ctypes.data, n, out.ctypes.data)
            assert out.sum() == n
            tot += out
        ps.append(stats.chisquare(tot, q * tot.sum()).pvalue)
    assert stats.kstest(ps, "uniform").pvalue > 1e-3 and min(ps) > 1e-5, (min(ps),)


def test_gibbs_colouring_of_classes_that_share_transcripts(tmp_path):
    """sailfish_amd/csrc/colour.h (host side of the Gibbs plan): first-fit colouring of classes over their transcripts.  No two
    classes of a colour share a transcript; a transcript shared by k classes forces >= k colours and first fit needs no more
    than (largest number of classes conflicting with one class) + 1; colours beyond 63 (the overflow words) work; the result
    is a function of the input alone."""
    import ctypes, subprocess
    so = tmp_path / "colour_harness.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(so), os.path.join(os.path.dirname(__file__), "colour_harness.cpp")])
    lib = ctypes.CDLL(str(so))
    lib.colour_classes.restype = ctypes.c_uint32
    P = ctypes.c_void_p
    lib.colour_classes.argtypes = [P, ctypes.c_uint64, P, ctypes.c_uint64, P, ctypes.c_uint64, ctypes.c_uint64, P]
    rng = np.random.default_rng(12)
    M = 5000
    labels = []
    for c in range(6000):
        k = int(rng.integers(1, 7))
        lab = set((int(rng.integers(0, 3000)) + 3 * np.arange(k)).tolist())
        if c % 3 == 0: lab.add(4000 + c // 900)            # a far transcript shared by ~300 classes: > 64 colours
        if c % 50 == 0: lab.add(4999)                      # ... and one shared by 120 classes spread over the whole list
        labels.append(np.array(sorted(lab), np.uint32))
    rowptr = np.zeros(len(labels) + 1, np.uint32); rowptr[1:] = np.cumsum([len(l) for l in labels])
    ids = np.concatenate(labels)
    wl = np.sort(rng.choice(len(labels), 4500, replace=False)).astype(np.uint32)     # the "wide" subset, in class order
    col = np.zeros(len(wl), np.uint32)
    ptr = lambda a: a.ctypes.data_as(P)
    k = lib.colour_classes(ptr(wl), len(wl), ptr(rowptr), len(labels), ptr(ids), len(ids), M, ptr(col))
    assert k == int(col.max()) + 1 and k > 64
    # no two classes of a colour share a transcript
    for c in range(k):
        members = np.concatenate([labels[i] for i in wl[col == c]])
        assert len(members) == len(np.unique(members)), c
    # lower bound: the most shared transcript; upper bound of first fit: max conflict degree + 1
    per_t = np.bincount(np.concatenate([labels[i] for i in wl]), minlength=M)
    assert k >= int(per_t.max())
    deg = max(int(sum(per_t[t] - 1 for t in labels[i])) for i in wl)
    assert k <= deg + 1
    col2 = np.zeros_like(col)
    assert lib.colour_classes(ptr(wl), len(wl), ptr(rowptr), len(labels), ptr(ids), len(ids), M, ptr(col2)) == k and np.array_equal(col, col2)


def test_gibbs_components_of_classes_that_share_transcripts(tmp_path):
    """colour.h, components_of_wide_classes: classes that share a transcript -- directly or through other listed classes -- get
    one component number, classes of different components share nothing; numbered in order of first appearance.  Checked
    against scipy's connected components of the class/transcript bipartite graph."""
    import ctypes, subprocess
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    so = tmp_path / "colour_harness.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(so), os.path.join(os.path.dirname(__file__), "colour_harness.cpp")])
    lib = ctypes.CDLL(str(so))
    lib.class_components.restype = ctypes.c_uint32
    P = ctypes.c_void_p
    lib.class_components.argtypes = [P, ctypes.c_uint64, P, ctypes.c_uint64, P, ctypes.c_uint64, ctypes.c_uint64, P]
    rng = np.random.default_rng(5)
    M = 4000
    labels = []
    for c in range(3000):
        base = int(rng.integers(0, 3500))
        lab = set((base + np.arange(int(rng.integers(1, 4)))).tolist())
        if c % 4 == 0: lab.add(3600 + (base // 500))           # a far transcript per neighbourhood of 500 ids
        labels.append(np.array(sorted(lab), np.uint32))
    rowptr = np.zeros(len(labels) + 1, np.uint32); rowptr[1:] = np.cumsum([len(l) for l in labels])
    ids = np.concatenate(labels)
    wl = np.sort(rng.choice(len(labels), 2000, replace=False)).astype(np.uint32)
    comp = np.zeros(len(wl), np.uint32)
    ptr = lambda a: a.ctypes.data_as(P)
    k = lib.class_components(ptr(wl), len(wl), ptr(rowptr), len(labels), ptr(ids), len(ids), M, ptr(comp))
    rows = np.concatenate([np.full(len(labels[c]), i) for i, c in enumerate(wl)])
    cols = np.concatenate([labels[c] for c in wl]).astype(np.int64) + len(wl)
    n = len(wl) + M
    g = coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    _, ref = connected_components(g, directed=False)
    ref = ref[:len(wl)]
    assert k == len(np.unique(ref)) == int(comp.max()) + 1 and 1 < k < len(wl)
    # the same partition ...
    pairs = set(zip(comp.tolist(), ref.tolist()))
    assert len(pairs) == k
    # ... numbered in order of first appearance
    first = [int(np.argmax(comp == c)) for c in range(k)]
    assert first == sorted(first)
