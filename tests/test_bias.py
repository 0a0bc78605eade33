"""Bias-aware effective lengths: sailfish::utils::updateEffectiveLengths and the optimize() hook.

CPU: the oracle's pieces against hand-computed values and against an independent numpy restatement of the same
sums (the algebraic form the device uses).  -m gpu: the HIP path against the oracle.

Tolerance: the reference accumulates in f64 serially; the device sums the same terms in another order (and the
4096-bin expectation through atomics), so lengths and expectations are compared at 1e-9 relative."""
import numpy as np
import pytest

from oracle import oracle as O

RTOL = 1e-9


def make_txome(rng, lens, alphabet=b"ACGT", sep=b"$"):
    """Concatenated sequences with a separator byte between transcripts, like RapMapSAIndex::seq."""
    lens = np.asarray(lens, dtype=np.uint32)
    off = np.zeros(len(lens), np.uint64)
    parts = []
    pos = 0
    for t, L in enumerate(lens):
        off[t] = pos
        parts.append(bytes(rng.choice(np.frombuffer(alphabet, np.uint8), int(L)).astype(np.uint8)))
        parts.append(sep)
        pos += int(L) + len(sep)
    return b"".join(parts), off, lens


def gaussian_fld(n=1000, mean=200.0, sd=60.0, total=1e6):
    x = np.arange(n)
    return np.round(total * np.exp(-0.5 * ((x - mean) / sd) ** 2) / (sd * np.sqrt(2 * np.pi))).astype(np.uint32)


def workload(seed, M=60, lo=5, hi=2500, **fld_kw):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi, M)
    lens[:4] = [3, 6, 7, 12]                                   # at / around K = 6
    seq, off, lens = make_txome(rng, lens, alphabet=b"ACGTacgtUu")
    fl = gaussian_fld(**fld_kw)
    txp_eff = np.maximum(lens - 180.0, 1.0)
    txp_eff[5] = float(lens[5])                                # unprocessedLen == 0
    txp_eff[6] = float(lens[6]) + 10.0                         # negative -> clamped to 0
    alphas = rng.random(M) * 200
    alphas[7] = 0.0; alphas[8] = 5e-9                          # below minAlpha
    eff_in = np.maximum(txp_eff * (0.9 + 0.2 * rng.random(M)), 1.0)
    rb = rng.integers(1, 400, 4096).astype(np.uint32)
    og = rng.integers(1, 3000, 101).astype(np.uint32)
    return dict(seq=seq, off=off, lens=lens, fl=fl, txp_eff=txp_eff, alphas=alphas, eff_in=eff_in, rb=rb, og=og)


# ---------------------------------------------------------------------------------------------- CPU
def test_kmer_index_known_answers(built):
    """indexForKmer / nextKmerIndex (include/UtilityFunctions.hpp:40-148): 2 bits per base, A=0 C=1 G=2 T=3"""
    assert O.index_for_kmer(b"AAAAAA") == 0 and O.index_for_kmer(b"TTTTTT") == 4095
    assert O.index_for_kmer(b"ACGTAC") == 0b000110110001                 # A C G T A C
    assert O.index_for_kmer(b"ACGTAC", rc=True) == 0b101100011011         # revcomp = G T A C G T
    assert O.index_for_kmer(b"acgUac") == O.index_for_kmer(b"ACGTAC")
    assert O.index_for_kmer(b"AAAAAA", rc=True) == 4095 and O.index_for_kmer(b"TTTTTT", rc=True) == 0
    assert O.index_for_kmer(b"ACGNAC") == 0xFFFFFFFF and O.index_for_kmer(b"ACGNAC", rc=True) == 0xFFFFFFFF
    rng = np.random.default_rng(3)
    s = bytes(rng.choice(np.frombuffer(b"ACGTacgu", np.uint8), 200).astype(np.uint8))
    idx = O.index_for_kmer(s[:6])                                         # forward: append on the right
    for i in range(1, len(s) - 5):
        idx = O.next_kmer_index(idx, s[i + 5:i + 6])
        assert idx == O.index_for_kmer(s[i:i + 6])
    i0 = len(s) - 6                                                       # reverse complement: walk to the left
    idx = O.index_for_kmer(s[i0:i0 + 6], rc=True)
    for i in range(i0 - 1, -1, -1):
        idx = O.next_kmer_index(idx, s[i:i + 1], rc=True)
        assert idx == O.index_for_kmer(s[i:i + 6], rc=True)
    assert O.next_kmer_index(0b11, b"N") == 0b1100                        # an unknown byte shifts in 0


def test_gc_frac_known_answers(built):
    """Transcript::gcFrac (include/Transcript.hpp:85-95): GCCount[e] - GCCount[s] over e - s + 1, lrint (half-even)"""
    s = b"GGGGCCCCAT"
    assert O.gc_frac(s, 0, 3) == 75          # bases 1..3 counted, length 4: the count at s itself drops out
    assert O.gc_frac(s, 0, 9) == 70
    assert O.gc_frac(s, 8, 9) == 0
    assert O.gc_frac(s, 4, 4) == 0
    assert O.gc_frac(b"AGAAAAAA", 0, 7) == 12    # 12.5 -> 12 (even)
    assert O.gc_frac(b"AGGGAAAA", 0, 7) == 38    # 37.5 -> 38 (even)
    assert O.gc_frac(b"agcgAAAA", 0, 7) == 38    # case-insensitive


def test_device_gc_rounding_formula_is_exact():
    """gc_bin of sailfish_amd/csrc/bias.hip (f32 product + exact f32 residual for the ties) == lrint((100.0 * d) / fl)
    for every fragment length the library accepts and every count"""
    for fl in range(1, 16000):
        d = np.arange(fl + 1, dtype=np.int64)
        n200 = (200 * d).astype(np.float32)
        h = np.float32(0.5) / np.float32(fl)
        r = np.rint(n200 * h)
        rem2 = r.astype(np.float64) * (-2.0 * fl) + n200.astype(np.float64)          # the fma: exact, integers < 2^24
        assert np.all(np.abs(rem2) < 2 ** 24)
        even = np.rint(n200 * (np.float32(0.5) * h))
        q = np.where(np.abs(rem2) == fl, even + even, r).astype(np.int64)
        ref = np.rint((100.0 * d) / fl).astype(np.int64)
        assert np.array_equal(q, ref), fl


def test_fld_cdf_is_float_accumulation(built):
    """EmpiricalDistribution (src/EmpiricalDistribution.cpp:29-77, :121-124): float tables, cut at 1 - 1e-6"""
    fl = gaussian_fld()
    cdf, size = O.fld_cdf(fl)
    tot = float(fl.sum())
    cum = np.cumsum(fl / tot)
    cut = int(np.argmax(cum > 1.0 - 1e-6))
    assert size == cut
    kept = float(fl[:cut].sum())
    pdf = (fl[:cut] / kept).astype(np.float32)
    run = np.float32(0.0); exp = np.zeros(cut, np.float32)
    for i in range(cut):
        run = pdf[i] if i == 0 else np.float32(run + pdf[i])
        exp[i] = run
    assert np.array_equal(cdf[:cut], exp) and np.all(cdf[cut:] == 1.0)


def _codes(seq_bytes):
    lut = np.full(256, -1, np.int64)
    for ch, v in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"TtUu", 3)):
        for c in ch:
            lut[c] = v
    return lut[np.frombuffer(seq_bytes, np.uint8)]


def _numpy_update(w, mode, gc_speed_samp=1, prob=(0.6, 0.4)):
    """The same quantities as vectorised sums: independent of the oracle's loops (used to cross-check it)."""
    cdf, size = O.fld_cdf(w["fl"])
    cdf = cdf.astype(np.float64)
    cdf_at = lambda x: np.where(x < len(cdf), cdf[np.minimum(x, len(cdf) - 1)], 1.0)
    pf, pr = prob
    M = len(w["lens"])
    out = w["eff_in"].copy()
    live = [t for t in range(M) if w["alphas"][t] >= 1e-8 and int(w["lens"][t]) - int(w["txp_eff"][t]) > 0]
    seqs = {t: w["seq"][int(w["off"][t]):int(w["off"][t]) + int(w["lens"][t])] for t in live}
    if mode == "seq":
        exp = np.ones(4096)
        per = {}
        for t in live:
            L = int(w["lens"][t]); c = _codes(seqs[t]); n = L - 6
            if n <= 0:
                continue
            win = np.lib.stride_tricks.sliding_window_view(c, 6)[:n]
            fwd = (win * (4 ** np.arange(5, -1, -1))).sum(1)
            rc = ((3 - win) * (4 ** np.arange(6))).sum(1)
            i = np.arange(n)
            per[t] = (fwd, rc, cdf_at(L - i - 1), cdf_at(i + 5))
            contrib = w["alphas"][t] / w["eff_in"][t]
            np.add.at(exp, rc, pf * contrib * per[t][2])
            np.add.at(exp, fwd, pr * contrib * per[t][3])
        txome = exp.sum(); read_norm = float(w["rb"].astype(np.uint64).sum() & 0xFFFFFFFF)
        prior = ((4096.0 / (read_norm - 4096.0)) * txome) / 4096.0
        ratio = w["rb"] / (exp + prior)
        for t, (fwd, rc, cf, cr) in per.items():
            e = ((pf * ratio[rc] * cf).sum() + (pr * ratio[fwd] * cr).sum()) * (txome / read_norm)
            unproc = int(w["lens"][t]) - int(w["txp_eff"][t])
            if e > unproc:
                out[t] = e
        return out, exp
    lo = int(np.argmax(cdf.astype(np.float32) >= 0.005)); hi = int(np.argmax(cdf.astype(np.float32) >= 0.995))
    fls = np.arange(lo, hi + 1, gc_speed_samp)
    wk = np.diff(np.concatenate([[cdf[0]], cdf_at(fls)]))
    wk[0] = cdf_at(fls[:1])[0] - cdf[0]
    S = {}
    for t in live:
        L = int(w["lens"][t]); n = L - 6
        row = np.zeros(101)
        if n > 0:
            isgc = np.isin(np.frombuffer(seqs[t], np.uint8), np.frombuffer(b"GCgc", np.uint8)).astype(np.int64)
            G = np.cumsum(isgc)
            for k, fl in enumerate(fls):
                i = np.arange(min(n, L - fl + 1)) if L - fl + 1 > 0 else np.arange(0)
                if len(i) == 0:
                    break
                g = np.rint(100.0 * (G[i + fl - 1] - G[i]) / fl).astype(np.int64)
                row += wk[k] * np.bincount(g, minlength=101)
        S[t] = row
    exp = np.ones(101)
    for t in live:
        exp += (w["alphas"][t] / w["eff_in"][t]) * S[t]
    txome = exp.sum(); gc_norm = float(w["og"].sum())
    prior = ((101.0 / (gc_norm - 101.0)) * txome) / 101.0
    ratio = w["og"] / (prior + exp)
    for t in live:
        e = (ratio * S[t]).sum() * (pf + pr) * (txome / gc_norm)
        unproc = int(w["lens"][t]) - int(w["txp_eff"][t])
        if e > unproc:
            out[t] = e
    return out, exp


@pytest.mark.parametrize("mode,samp", [("seq", 1), ("gc", 1), ("gc", 7)])
def test_oracle_matches_vectorised_restatement(built, mode, samp):
    w = workload(11, M=40, hi=1200)
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], num_fwd=600, num_rc=400,
                           seq_bias=mode == "seq", gc_bias=mode == "gc", gc_speed_samp=samp)
    rc, out, es, eg, nc = O.update_efflens(bm, w["eff_in"], w["alphas"])
    assert rc == 0 and 0 < nc < 40
    ref_out, ref_exp = _numpy_update(w, mode, samp)
    np.testing.assert_allclose(out, ref_out, rtol=1e-10)
    np.testing.assert_allclose(es if mode == "seq" else eg, ref_exp, rtol=1e-10)
    untouched = (w["alphas"] < 1e-8) | (w["lens"].astype(np.int64) - w["txp_eff"].astype(np.int64) <= 0)
    assert np.array_equal(out[untouched], w["eff_in"][untouched])
    if mode == "gc":
        assert np.all(es == 1.0)                                          # reset even when unused (:652-653)


def test_oracle_skip_rules(built):
    w = workload(5, M=12, hi=400)
    for kw, code in ((dict(seq_bias=True, num_fwd=0, num_rc=0), 1), (dict(seq_bias=True, gc_bias=True, num_fwd=5, num_rc=5), 2)):
        bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], **kw)
        rc, out, es, eg, nc = O.update_efflens(bm, w["eff_in"], w["alphas"])
        assert rc == code and nc == 0 and np.array_equal(out, w["eff_in"])


def _classes(rng, M, C, num_reads):
    sizes = rng.integers(1, 5, C)
    rowptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ids = np.concatenate([np.sort(rng.choice(M, s, replace=False)) for s in sizes]).astype(np.uint32)
    counts = rng.multinomial(num_reads, rng.dirichlet(np.ones(C) * 0.5)).astype(np.uint64) + 1
    return rowptr, ids, counts


def test_oracle_optimize_hooks(built):
    """the recompute hook fires at 50 / 500 / 1000 only while the loop is still running (:820-840)"""
    w = workload(21, M=50, hi=900)
    rng = np.random.default_rng(2)
    rowptr, ids, counts = _classes(rng, 50, 300, 20000)
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], num_fwd=7, num_rc=3, seq_bias=True)
    n = int(counts.sum())
    for (lo, hi), hooks in (((40, 40), 0), ((50, 50), 0), ((51, 51), 1), ((500, 500), 1), ((501, 501), 2), ((1001, 1001), 3)):
        rc, a, m, eff, es, eg, nr, st = O.em_optimize_bias(bm, w["txp_eff"], rowptr, ids, counts, n, min_iter=lo, max_iter=hi)
        assert rc == 0 and nr == hooks and st["iters"] == hi
        if hooks == 0:
            rc0, a0, _, _ = O.em_optimize(w["txp_eff"], rowptr, ids, counts, n, min_iter=lo, max_iter=hi)
            assert np.array_equal(a, a0) and np.array_equal(eff, np.maximum(w["txp_eff"], 1.0))
        else:
            assert not np.array_equal(eff, np.maximum(w["txp_eff"], 1.0))


# ---------------------------------------------------------------------------------------------- GPU
def _device_model(w, dev, **kw):
    import torch
    import sailfish_amd as sf
    seq = torch.from_numpy(np.frombuffer(w["seq"], np.uint8).copy()).to(dev)
    off = torch.from_numpy(w["off"].astype(np.int64)).to(dev)
    rl = torch.from_numpy(w["lens"].astype(np.uint32).view(np.int32).copy()).to(dev)
    te = torch.from_numpy(w["txp_eff"]).to(dev)
    return sf.bias.BiasModel(seq, off, rl, te, w["fl"], w["rb"], w["og"], **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,samp,fld", [("seq", 1, {}), ("gc", 1, {}), ("gc", 5, {}),
                                           ("gc", 1, dict(n=6000, mean=2500.0, sd=900.0)),      # several position chunks
                                           ("gc", 3, dict(n=400, mean=60.0, sd=15.0))])
def test_update_matches_oracle(built, mode, samp, fld):
    import torch
    dev = torch.device("cuda:0")
    big = "mean" in fld and fld["mean"] > 1000
    w = workload(101 + samp, M=40 if big else 300, hi=16000 if big else 5000, **fld)      # (big: the oracle's two passes took 36 s of the suite at M = 120)
    kw = dict(num_fwd=611, num_rc=389, seq_bias=mode == "seq", gc_bias=mode == "gc", gc_speed_samp=samp)
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], **kw)
    rc, out, es, eg, nc = O.update_efflens(bm, w["eff_in"], w["alphas"])
    assert rc == 0 and nc > 0
    model = _device_model(w, dev, **kw)
    got, st = model.update(torch.from_numpy(w["eff_in"]).to(dev), torch.from_numpy(w["alphas"]).to(dev))
    assert st["status"] == 0 and st["n_corrected"] == nc and st["n_uncorrected"] == len(out) - nc
    np.testing.assert_allclose(got.cpu().numpy(), out, rtol=RTOL)
    ges, geg = model.expected()
    np.testing.assert_allclose(ges, es, rtol=RTOL)
    np.testing.assert_allclose(geg, eg, rtol=RTOL)
    if big: model.close(); return
    # a second update with other abundances reuses the handle (GC profile built once)
    a2 = w["alphas"][::-1].copy()
    rc, out2, *_ = O.update_efflens(bm, out, a2)
    got2, _ = model.update(got, torch.from_numpy(a2).to(dev))
    np.testing.assert_allclose(got2.cpu().numpy(), out2, rtol=RTOL)
    model.close()


@pytest.mark.gpu
@pytest.mark.parametrize("step,samp", [(2, 1), (4, 3), (7, 1), (25, 1)])
def test_gc_model_with_sampled_gc_counts(built, step, samp):
    """--gcSizeSamp > 1: Transcript::gcCountInterp_'s interpolated counts (include/Transcript.hpp:133-162), bins clamped
    to [0,100] where the reference would index out of range"""
    import torch
    dev = torch.device("cuda:0")
    w = workload(300 + step, M=120, hi=2500)
    kw = dict(num_fwd=7, num_rc=5, gc_bias=True, gc_speed_samp=samp, gc_size_samp=step)
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], **kw)
    rc, out, es, eg, nc = O.update_efflens(bm, w["eff_in"], w["alphas"])
    exact = O.update_efflens(O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], **{**kw, "gc_size_samp": 1}),
                             w["eff_in"], w["alphas"])
    assert rc == 0 and nc > 0 and not np.array_equal(out, exact[1])          # the interpolation does change the answer
    model = _device_model(w, dev, **kw)
    got, st = model.update(torch.from_numpy(w["eff_in"]).to(dev), torch.from_numpy(w["alphas"]).to(dev))
    assert st["n_corrected"] == nc
    np.testing.assert_allclose(got.cpu().numpy(), out, rtol=RTOL)
    np.testing.assert_allclose(model.expected()[1], eg, rtol=RTOL)
    model.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["seq", "gc"])
def test_update_with_unknown_bases_inside(built, mode):
    """a byte outside ACGTU away from a transcript's first k-mers: nextKmerIndex shifts in 0 for it in both directions
    (include/UtilityFunctions.hpp:40-90) and it is not G/C; the device's direct 6-mer evaluation agrees"""
    import torch
    dev = torch.device("cuda:0")
    w = workload(55, M=150, lo=40, hi=3000)
    w["lens"][:4] = np.maximum(w["lens"][:4], 40)          # workload() plants lengths 3, 6, 7, 12 there: regenerate those
    rng = np.random.default_rng(1)
    seq, off, lens = make_txome(rng, np.maximum(w["lens"], 40))
    sq = bytearray(seq)
    for t in range(len(lens)):
        L = int(lens[t])
        for p in rng.integers(6, L - 7, max(1, L // 50)):  # never inside [0,6) or [L-7, L): the two first k-mers
            sq[int(off[t]) + int(p)] = ord(rng.choice(list("NnRY-")))
    w.update(seq=bytes(sq), off=off, lens=lens, txp_eff=np.maximum(lens - 180.0, 1.0))
    w["eff_in"] = np.maximum(w["txp_eff"], 1.0)
    kw = dict(num_fwd=3, num_rc=7, seq_bias=mode == "seq", gc_bias=mode == "gc")
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], **kw)
    rc, out, es, eg, nc = O.update_efflens(bm, w["eff_in"], w["alphas"])
    assert rc == 0 and nc > 0
    model = _device_model(w, dev, **kw)
    got, st = model.update(torch.from_numpy(w["eff_in"]).to(dev), torch.from_numpy(w["alphas"]).to(dev))
    assert st["n_corrected"] == nc
    np.testing.assert_allclose(got.cpu().numpy(), out, rtol=RTOL)
    ges, geg = model.expected()
    np.testing.assert_allclose(ges, es, rtol=RTOL); np.testing.assert_allclose(geg, eg, rtol=RTOL)
    model.close()


@pytest.mark.gpu
def test_update_skip_and_invalid(built):
    import torch
    import sailfish_amd as sf
    dev = torch.device("cuda:0")
    w = workload(9, M=30, hi=600)
    e = torch.from_numpy(w["eff_in"]).to(dev); a = torch.from_numpy(w["alphas"]).to(dev)
    for kw, code in ((dict(seq_bias=True, num_fwd=0, num_rc=0), 1), (dict(seq_bias=True, gc_bias=True, num_fwd=5, num_rc=5), 2)):
        m = _device_model(w, dev, **kw)
        got, st = m.update(e, a)
        assert st["status"] == code and torch.equal(got, e)
        m.close()
    with pytest.raises(sf._lib.SfgpuError):                               # 0.005 quantile at length 0
        w0 = dict(w); w0["fl"] = np.array([50, 30, 10, 5, 5], np.uint32)
        _device_model(w0, dev, gc_bias=True, num_fwd=1, num_rc=1)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,vb,iters", [("seq", False, (60, 60)), ("gc", False, (520, 520)), ("seq", True, (60, 60)),
                                           ("gc", True, (50, 10000)), ("seq", False, (50, 10000)),
                                           ("seq", False, (70, 10))])          # maxIter below minIter: runs to minIter (:820)
def test_optimize_with_bias_matches_oracle(built, mode, vb, iters):
    import torch
    import sailfish_amd as sf
    dev = torch.device("cuda:0")
    M = 400
    w = workload(77, M=M, hi=3000)
    rng = np.random.default_rng(8)
    rowptr, ids, counts = _classes(rng, M, 3000, 400000)
    n = int(counts.sum())
    kw = dict(num_fwd=52, num_rc=48, seq_bias=mode == "seq", gc_bias=mode == "gc")
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], w["og"], **kw)
    fixed = iters[0] >= iters[1]
    tol = 0.01 if fixed else 1e-5                                         # the convergent runs go past the first hook
    rc, a, m, eff, es, eg, nr, st = O.em_optimize_bias(bm, w["txp_eff"], rowptr, ids, counts, n, use_vbem=vb, tol=tol,
                                                       min_iter=iters[0], max_iter=iters[1])
    assert rc == 0 and nr >= 1
    model = _device_model(w, dev, **kw)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).astype(dt)).to(dev)
    prob = sf.EMProblem(t(w["txp_eff"], np.float64), t(rowptr.astype(np.uint32).view(np.int32), np.int32),
                        t(ids.view(np.int32), np.int32), t(counts, np.int64), n)
    grc, gst, geff, gnr = prob.optimize_bias(model, use_vbem=vb, tol=tol, min_iter=iters[0], max_iter=iters[1])
    assert grc == 0 and gnr == nr
    if fixed:
        assert gst["iters"] == st["iters"] == max(iters)
    else:
        assert abs(int(gst["iters"]) - int(st["iters"])) <= 1
    np.testing.assert_allclose(geff.cpu().numpy(), eff, rtol=1e-6)
    ga = prob.alpha.cpu().numpy()
    tol = 1e-6 if fixed else 1e-4
    big = a > 1e-3
    np.testing.assert_allclose(ga[big], a[big], rtol=tol)
    ges, geg = model.expected()
    np.testing.assert_allclose(ges if mode == "seq" else geg, es if mode == "seq" else eg, rtol=1e-6)
    # the handle's lengths are the problem's own again: a plain optimize() matches the plain oracle
    rc0, a0, _, st0 = O.em_optimize(w["txp_eff"], rowptr, ids, counts, n, use_vbem=vb, min_iter=60, max_iter=60)
    prob.optimize(use_vbem=vb, min_iter=60, max_iter=60)
    np.testing.assert_allclose(prob.alpha.cpu().numpy()[a0 > 1e-3], a0[a0 > 1e-3], rtol=1e-9)
    prob.close(); model.close()


@pytest.mark.gpu
def test_reference_style_driver_with_bias(built):
    """CollapsedEMOptimizer.optimize(readExp, sopt) with sopt.biasCorrect, the way the quant driver calls it"""
    import torch
    import sailfish_amd as sf
    dev = torch.device("cuda:0")
    M = 200
    w = workload(31, M=M, hi=2000)
    rng = np.random.default_rng(4)
    rowptr, ids, counts = _classes(rng, M, 1500, 100000)
    n = int(counts.sum())
    sopt = sf.SailfishOpts(biasCorrect=True)
    txps = sf.Transcripts([f"t{i}" for i in range(M)], w["lens"], device=dev)
    txps.EffectiveLength.copy_(torch.from_numpy(w["txp_eff"]))
    exp = sf.ReadExperiment(txps, sopt)
    exp.setSequences(w["seq"], w["off"]); exp.setFragLengthDist(w["fl"].astype(np.int32))
    exp.readBias()[:] = w["rb"]; exp.addNumFwd(70); exp.addNumRC(30)
    eq = exp.equivalenceClassBuilder(); eq.start()
    eq.insertGroups(torch.from_numpy(ids.view(np.int32).copy()).to(dev), torch.from_numpy(rowptr.astype(np.int64)).to(dev),
                    torch.from_numpy(counts.astype(np.int64)).to(dev))
    eq.finish()
    exp.setNumMappedFragments(n)
    opt = sf.CollapsedEMOptimizer()
    assert opt.optimize(exp, sopt, 0.01, 10000)
    bm = O.make_bias_model(w["seq"], w["off"], w["lens"], w["txp_eff"], w["fl"], w["rb"], None, num_fwd=70, num_rc=30, seq_bias=True)
    v = eq.eqVec()
    rc, a, m, eff, es, eg, nr, st = O.em_optimize_bias(bm, w["txp_eff"], v.rowptr.cpu().numpy().astype(np.uint64),
                                                       v.ids.cpu().numpy().view(np.uint32), v.counts.cpu().numpy().astype(np.uint64), n,
                                                       max_iter=10000)
    assert rc == 0 and opt.last_recomputes == nr
    np.testing.assert_allclose(txps.EffectiveLength.cpu().numpy(), eff, rtol=1e-6)
    np.testing.assert_allclose(exp.expectedSeqBias(), es, rtol=1e-6)
    big = a > 1e-3
    np.testing.assert_allclose(txps.estCount.cpu().numpy()[big], a[big], rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("vb", [False, True])
def test_piecewise_loop_with_rebase_matches_optimize_bias(built, vb):
    """the multi-GPU driver's form of the hook (set_bounds / sfgpu_bias_update / rebase between polls) gives what
    sfgpu_em_optimize_bias gives"""
    import torch
    import sailfish_amd as sf
    dev = torch.device("cuda:0")
    M = 400
    w = workload(78, M=M, hi=3000)
    rng = np.random.default_rng(9)
    rowptr, ids, counts = _classes(rng, M, 3000, 400000)
    n = int(counts.sum())
    model = _device_model(w, dev, num_fwd=52, num_rc=48, seq_bias=True)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).astype(dt)).to(dev)
    prob = sf.EMProblem(t(w["txp_eff"], np.float64), t(rowptr.astype(np.uint32).view(np.int32), np.int32),
                        t(ids.view(np.int32), np.int32), t(counts, np.int64), n)
    rc, st, eff, hooks = prob.optimize_bias(model, use_vbem=vb, tol=1e-5, min_iter=50, max_iter=10000)
    assert rc == 0 and hooks >= 1
    a_ref = prob.alpha.clone()
    prob.begin(use_vbem=vb, tol=1e-5, min_iter=50, max_iter=10000)
    prob.init()
    it, conv, taken = 0, False, 0
    while not (it >= 50 and (it >= 10000 or conv)):
        if it in (50, 500, 1000):
            new_len, _ = model.update(prob.length_view(), prob.alpha_view())
            prob.rebase(new_len); taken += 1
        nxt = min([h for h in (50, 500, 1000) if h > it] + [10000])
        prob.set_bounds(min(50, nxt), nxt)
        done = False
        while not done:
            for _ in range(16):
                prob.sweep(); prob.update()
            done, seg = prob.poll()
        it, conv = seg["iters"], seg["converged"]
    eff2 = prob.length_view().clone()
    rc2, st2 = prob.finish()
    assert rc2 == 0 and taken == hooks and abs(int(st2["iters"]) - int(st["iters"])) <= 1
    np.testing.assert_allclose(eff2.cpu().numpy(), eff.cpu().numpy(), rtol=1e-9)
    big = a_ref.cpu().numpy() > 1e-3
    np.testing.assert_allclose(prob.alpha.cpu().numpy()[big], a_ref.cpu().numpy()[big], rtol=1e-5)
    # begin() restores the problem's own lengths
    prob.begin(use_vbem=vb, min_iter=60, max_iter=60); prob.init()
    np.testing.assert_array_equal(prob.length_view().cpu().numpy(), np.maximum(w["txp_eff"], 1.0))
    prob.close(); model.close()
