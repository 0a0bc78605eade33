"""`sailfish quant` after the mapper, end to end (sailfish_amd/quant.py; src/SailfishQuantify.cpp:1160-1440): hit records
in, quant.sf / aux / gene-level files out, against the same chain assembled from the oracle's pieces."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from test_filter import FORMATS, _random_reads, _txome


def _read_quant(path):
    rows = [l.split("\t") for l in open(path).read().strip().split("\n")]
    assert rows[0] == ["Name", "Length", "EffectiveLength", "TPM", "NumReads"]
    return {r[0]: tuple(float(x) for x in r[1:]) for r in rows[1:]}


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["plain", "seq", "gc"])
def test_quantify_matches_the_oracle_chain(built, gpu, tmp_path, which):
    import sailfish_amd as sf
    rng = np.random.default_rng(41)
    M, R = 300, 60_000
    seq, so, rl = _txome(rng, M, lo=400, hi=3000)
    names = [f"tx{i:04d}" for i in range(M)]
    batches = []
    for b in range(3):                                                     # three "parser jobs"
        # pairs of transcripts (2j, 2j+1) sharing 94 % of their fragments, the rest unique to 2j: 2j+1 is explained away at a
        # rate near 1, so the EM needs far more than 50 iterations (the bias recompute is reached)
        n = R // 3
        j = rng.integers(0, M // 2, n)
        shared = rng.random(n) < 0.94
        k = np.where(shared, 2, 1)
        off = np.zeros(n + 1, np.uint32); off[1:] = np.cumsum(k)
        h = np.zeros(int(off[-1]), O.HIT_DTYPE)
        first = off[:-1].astype(np.int64)
        pick = np.zeros(n, np.int64)     # unique fragments only ever hit the first member: the second decays slowly to zero
        h["tid"][first] = np.where(shared, 2 * j, 2 * j + pick)
        h["tid"][first[shared] + 1] = 2 * j[shared] + 1
        h["mate_status"] = 3
        L = rl[h["tid"]].astype(np.int64)
        left = (rng.random(len(h)) * np.maximum(L - 260, 1)).astype(np.int32)
        h["frag_len"] = rng.integers(120, 260, len(h))
        fw = np.repeat(rng.integers(0, 2, n), k)                           # one orientation per fragment: every hit is IU-compatible
        h["pos"] = np.where(fw == 1, left, left + h["frag_len"] - 50); h["mate_pos"] = np.where(fw == 1, left + h["frag_len"] - 50, left)
        h["read_len"] = 50; h["mate_len"] = 50; h["fwd"] = fw; h["mate_fwd"] = 1 - fw
        batches.append((h, off))
    sopt = sf.SailfishOpts(biasCorrect=which == "seq", gcBiasCorrect=which == "gc", dumpEq=True, numBootstraps=3, numFragSamples=2000)
    (tmp_path / "map.tsv").write_text("".join(f"{n} g{i // 3}\n" for i, n in enumerate(names)))
    out = str(tmp_path / "out")
    rc, exp = sf.quant.quantify(names, rl, batches, "IU", out, sopt, seq=seq, seq_off=so, allow_orphans=True, num_bias_samples=5000,
                                gene_map=str(tmp_path / "map.tsv"), cmd_options={"libType": "IU", "output": out}, seed=3, device=gpu)
    assert rc == 0
    # ---- the same chain from the oracle's pieces
    ob = O.EqBuilder()
    fl = np.zeros(1000, np.uint32); rem_fl, rem_b = 2000, 5000
    rb = np.ones(4096, np.uint32) if which == "seq" else None
    og = np.ones(101, np.uint32) if which == "gc" else None
    tot = dict(n_observed=0, n_mapped=0, n_fwd=0, n_rc=0)
    for h, off in batches:
        (ids, oo, fl, rem_fl, st), rb, rem_b, og, _, _ = O.filter_hits_bias(h, off, FORMATS["IU"], True, seq, so, rl, read_bias=rb,
                                                                            remaining_bias_samples=rem_b, observed_gc=og,
                                                                            discard_orphans=False, fl_counts=fl, remaining_fl_ops=rem_fl)
        ob.add_batch(ids, oo.astype(np.uint64))
        for k in tot:
            tot[k] += st[k]
    rp, ii, cc, hh = ob.finish()
    assert rem_fl == 0                                                      # enough unique pairs: the empirical branch
    eff0 = O.efflen_smoothed(rl, O.cf_counts(fl))
    if which == "plain":
        rc_o, a, m, st_o = O.em_optimize(eff0, rp, ii, cc, tot["n_mapped"], max_iter=10000)
        eff = np.maximum(eff0, 0) if False else eff0
    else:
        bm = O.make_bias_model(seq, so, rl, eff0, fl, rb, og, num_fwd=tot["n_fwd"], num_rc=tot["n_rc"], seq_bias=which == "seq", gc_bias=which == "gc")
        rc_o, a, m, eff, es, eg, nr, st_o = O.em_optimize_bias(bm, eff0, rp, ii, cc, tot["n_mapped"], max_iter=10000)
        assert nr >= 1, st_o
    tpm = O.tpm(a, eff, tot["n_mapped"])
    q = _read_quant(os.path.join(out, "quant.sf"))
    assert list(q) == names
    got = np.array([q[n] for n in names])
    np.testing.assert_array_equal(got[:, 0], rl)
    np.testing.assert_allclose(got[:, 1], eff, rtol=2e-5)
    big = a > 1e-2
    np.testing.assert_allclose(got[big, 3], a[big], rtol=1e-4)
    np.testing.assert_allclose(got[big, 2], tpm[big], rtol=1e-4)
    assert exp.numMappedFragments() == tot["n_mapped"] and exp.numObservedFragments() == tot["n_observed"] == R
    assert (exp.numFwd(), exp.numRC()) == (tot["n_fwd"], tot["n_rc"])
    # ---- the other files of the run
    aux = os.path.join(out, "aux")
    assert json.load(open(os.path.join(out, "cmd_info.json"))) == {"sf_version": "0.10.0", "libType": "IU", "output": out}
    meta = json.load(open(os.path.join(aux, "meta_info.json")))
    assert meta["num_processed"] == R and meta["num_mapped"] == tot["n_mapped"] and meta["samp_type"] == "bootstrap"
    eqf = open(os.path.join(aux, "eq_classes.txt")).read().split("\n")
    assert int(eqf[0]) == M and int(eqf[1]) == ob.n_classes
    if which == "seq":
        np.testing.assert_array_equal(np.frombuffer(gzip.open(os.path.join(aux, "observed_bias.gz")).read(), np.int32), rb)
        np.testing.assert_allclose(np.frombuffer(gzip.open(os.path.join(aux, "expected_bias.gz")).read(), np.float64), es, rtol=1e-6)
    if which == "gc":
        np.testing.assert_array_equal(np.frombuffer(gzip.open(os.path.join(aux, "observed_gc.gz")).read(), np.int32), og)
        np.testing.assert_allclose(np.frombuffer(gzip.open(os.path.join(aux, "expected_gc.gz")).read(), np.float64), eg, rtol=1e-6)
    bs = np.frombuffer(gzip.open(os.path.join(aux, "bootstrap", "bootstraps.gz")).read(), np.float64).reshape(3, M)
    assert np.all(np.abs(bs.sum(1) - tot["n_mapped"]) < 1e-6 * tot["n_mapped"])
    genes = open(os.path.join(out, "quant.genes.sf")).read().strip().split("\n")
    assert genes[0].startswith("Name") and len(genes) == 1 + (M + 2) // 3
    tot_reads = sum(float(l.split("\t")[4]) for l in genes[1:])
    assert abs(tot_reads - got[:, 3].sum()) < 1e-3 * tot["n_mapped"]


@pytest.mark.gpu
def test_quantify_option_checks(built, gpu, tmp_path):
    import sailfish_amd as sf
    msgs = []
    sopt = sf.SailfishOpts(numBootstraps=2, numGibbsSamples=2, jointLog=lambda lvl, m: msgs.append(m))
    rc, _ = sf.quant.quantify(["a"], np.array([100], np.uint32), [], "U", str(tmp_path / "o1"), sopt, device=gpu)
    assert rc == 1 and "cannot perform both" in msgs[-1] and os.path.exists(tmp_path / "o1" / "cmd_info.json")
    sopt = sf.SailfishOpts(biasCorrect=True, gcBiasCorrect=True, jointLog=lambda lvl, m: msgs.append(m))
    rc, _ = sf.quant.quantify(["a"], np.array([100], np.uint32), [], "IU", str(tmp_path / "o2"), sopt, device=gpu)
    assert rc == 1 and "not yet supported" in msgs[-1]
    # no reads at all: "no transcripts expressed" -> 1, as the reference
    rc, _ = sf.quant.quantify(["a", "b"], np.array([100, 200], np.uint32), [], "U", str(tmp_path / "o3"), sf.SailfishOpts(), device=gpu)
    assert rc == 1


GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_sample_data_fixture_is_consistent(built):
    """tests/golden/sample_data_hits.npz (BASELINE config 1's bundled data through the stand-in mapper): every record
    names the transcript its read was simulated from among its hits"""
    d = np.load(os.path.join(GOLD, "sample_data_hits.npz"))
    hits = d["hits"].view(O.HIT_DTYPE); off = d["offsets"]
    assert len(d["names"]) == 15 and len(off) == 10001 and off[-1] == len(hits)
    assert all(d["truth"][r] in hits["tid"][off[r]:off[r + 1]] for r in range(10000))
    assert np.all(hits["frag_len"] < 1000) and np.all(hits["tid"] < 15)


@pytest.mark.gpu
def test_quantify_bundled_sample_data(built, gpu, tmp_path):
    """BASELINE config 1 ("sailfish quant on the bundled sample_data", plumbing / correctness): the reference's own
    15-transcript sample through quantify(), against the oracle chain on the same hit records and against the
    simulator's truth carried by the read names"""
    import sailfish_amd as sf
    d = np.load(os.path.join(GOLD, "sample_data_hits.npz"))
    names = [str(n) for n in d["names"]]; rl = d["ref_len"]
    hits = d["hits"].view(O.HIT_DTYPE); off = d["offsets"]
    cuts = [0, 1000, 2000, 5000, 10000]                                    # parser jobs of different sizes
    batches = [(hits[off[a]:off[b]], (off[a:b + 1] - off[a]).astype(np.uint32)) for a, b in zip(cuts, cuts[1:])]
    out = str(tmp_path / "out")
    sopt = sf.SailfishOpts(numFragSamples=5000)
    rc, exp = sf.quant.quantify(names, rl, batches, "IU", out, sopt, cmd_options={"libType": "IU"}, device=gpu)
    assert rc == 0
    (ids, oo, fl, rem, st) = O.filter_hits(hits, off, FORMATS["IU"], True, fl_counts=np.zeros(1000, np.uint32), remaining_fl_ops=5000)
    assert st["n_mapped"] == 10000 and rem == 0 and exp.numMappedFragments() == 10000
    ob = O.EqBuilder(); ob.add_batch(ids, oo.astype(np.uint64)); rp, ii, cc, hh = ob.finish()
    eff = O.efflen_smoothed(rl, O.cf_counts(fl))
    rc_o, a, m, st_o = O.em_optimize(eff, rp, ii, cc, 10000, max_iter=10000)
    tpm = O.tpm(a, eff, 10000)
    q = _read_quant(os.path.join(out, "quant.sf"))
    got = np.array([q[n] for n in names])
    np.testing.assert_allclose(got[:, 1], eff, rtol=2e-5)
    big = a > 1e-2
    np.testing.assert_allclose(got[big, 3], a[big], rtol=1e-4); np.testing.assert_allclose(got[big, 2], tpm[big], rtol=1e-4)
    truth = np.bincount(d["truth"], minlength=15).astype(np.float64)
    assert abs(got[:, 3].sum() - 10000) < 1e-6 * 10000
    assert np.abs(got[:, 3] - truth).sum() / 10000 < 0.1, (got[:, 3], truth)     # the EM recovers the simulated abundances (isoforms 0-2 are hard)
