"""Per-read hit filtering: the oracle against the reference's own LibraryTypeTests expectations
and hand-built reads (CPU); the HIP path against the oracle on random hit lists (-m gpu), chained into the
class builder."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SINGLE, LEFT, RIGHT, PAIRED = 0, 1, 2, 3
SAME, AWAY, TOWARD, NONE = 0, 1, 2, 3
SA, AS, S, A, U = 0, 1, 2, 3, 4
FORMATS = {"IU": (1, TOWARD, U), "ISF": (1, TOWARD, SA), "ISR": (1, TOWARD, AS), "OU": (1, AWAY, U), "OSF": (1, AWAY, SA),
           "OSR": (1, AWAY, AS), "MU": (1, SAME, U), "MSF": (1, SAME, S), "MSR": (1, SAME, A), "U": (0, NONE, U),
           "SF": (0, NONE, S), "SR": (0, NONE, A)}


def _hits(rows):
    """rows: (tid, pos, fwd, mate_status[, mate_pos, mate_fwd, frag_len, read_len, mate_len])"""
    h = np.zeros(len(rows), O.HIT_DTYPE)
    for i, r in enumerate(rows):
        r = tuple(r) + (0, 0, 0, 50, 50)[len(r) - 4:]
        h[i] = (r[0], r[1], r[4], r[6], r[7], r[8], int(r[2]), int(r[5]), r[3], 0)
    return h


def test_library_type_vectors(built):
    """tests/LibraryTypeTests.cpp, as data: compatibleHit (both overloads) and the format id encoding"""
    g = json.load(open(os.path.join(GOLD, "library_type_vectors.json")))
    assert len(g["paired"]) == 72 and len(g["single"]) == 72
    for v in g["paired"]:
        assert O.compatible_pair(tuple(v["expected"]), tuple(v["observed"])) == v["compatible"], v
    for v in g["single"]:
        assert O.compatible_single(tuple(v["expected"]), v["is_forward"], v["mate_status"]) == v["compatible"], v
    for v in g["format_ids"]:
        t, o, s = v["format"]
        assert v["id"] == (t & 1) | ((o & 3) << 1) | ((s & 7) << 3) and v["id"] <= (1 | (3 << 1) | (4 << 3))


def test_hit_type_cases(built):
    """hitType (src/SailfishUtils.cpp:232-281): the six outcomes its comments name, and the dovetail stretch"""
    assert O.hit_type(100, True, 50, 300, False, 50, False) == (1, TOWARD, SA)       # ISF
    assert O.hit_type(300, True, 50, 100, False, 50, False) == (1, AWAY, SA)         # OSF
    assert O.hit_type(300, False, 50, 100, True, 50, False) == (1, TOWARD, AS)       # ISR
    assert O.hit_type(100, False, 50, 300, True, 50, False) == (1, AWAY, AS)         # OSR
    assert O.hit_type(100, True, 50, 300, True, 50, False) == (1, SAME, S)           # MSF
    assert O.hit_type(100, False, 50, 300, False, 50, False) == (1, SAME, A)         # MSR
    assert O.hit_type(120, True, 50, 100, False, 50, False) == (1, AWAY, SA)         # read 1 starts 20 past read 2 ...
    assert O.hit_type(120, True, 50, 100, False, 50, True) == (1, TOWARD, SA)        # ... allowed when dovetailing
    assert O.hit_type(151, True, 50, 100, False, 50, True) == (1, AWAY, SA)          # but not beyond the mate's length


def test_filter_rules_on_hand_built_reads(built):
    fmt = FORMATS["ISF"]
    reads = [
        _hits([(3, 10, True, PAIRED, 200, False, 240), (9, 10, False, PAIRED, 200, True, 240)]),   # 1 compatible of 2
        _hits([(4, 10, False, PAIRED, 200, True, 240)]),                                           # none compatible -> all hits
        _hits([(7, 10, True, PAIRED, 200, False, 300)]),                                           # unique proper pair: FLD sample
        _hits([(8, 5, True, LEFT), (2, 5, True, RIGHT), (8, 5, False, RIGHT)]),                    # orphans: merged by tid
        _hits([(1, 0, True, PAIRED, 100, False, 150)] * 5),                                        # > maxReadOccs
        np.zeros(0, O.HIT_DTYPE),
    ]
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint32)
    hits = np.concatenate(reads)
    ids, o, fl, rem, st = O.filter_hits(hits, off, fmt, True, discard_orphans=False, max_read_occs=4, remaining_fl_ops=10)
    lists = [ids[o[i]:o[i + 1]].tolist() for i in range(len(reads))]
    assert lists == [[3], [4], [7], [2, 8, 8], [], []]
    assert rem == 8 and fl[240] == 1 and fl[300] == 1 and fl.sum() == 2           # reads 1 and 2 are unique pairs; read 0 has two hits
    assert st == dict(n_observed=6, n_mapped=4, total_hits=7, upper_bound_hits=5, n_fwd=1 + 0 + 1 + 2, n_rc=0 + 1 + 0 + 1, fl_sampled=2)   # orphans: left+fwd and right+rc count as forward (:313-320)
    # orphans dropped when they are not allowed; enforceLibCompat drops the incompatible read; ignoreLibCompat keeps everything
    ids, o, *_ = O.filter_hits(hits, off, fmt, True, discard_orphans=True, max_read_occs=4)
    assert [ids[o[i]:o[i + 1]].tolist() for i in range(6)] == [[3], [4], [7], [], [], []]
    ids, o, *_ = O.filter_hits(hits, off, fmt, True, discard_orphans=False, enforce_compat=True, max_read_occs=4)
    assert [ids[o[i]:o[i + 1]].tolist() for i in range(6)] == [[3], [], [7], [], [], []]    # orphans of an ISF library are never strand-compatible (strandedness SA)
    ids, o, *_ = O.filter_hits(hits, off, fmt, True, discard_orphans=False, ignore_compat=True, max_read_occs=4)
    assert [ids[o[i]:o[i + 1]].tolist() for i in range(6)] == [[3, 9], [4], [7], [2, 8, 8], [], []]
    # the sample budget is first come, first served
    _, _, fl, rem, st = O.filter_hits(hits, off, fmt, True, discard_orphans=False, max_read_occs=4, remaining_fl_ops=1)
    assert rem == 0 and fl[240] == 1 and fl.sum() == 1 and st["fl_sampled"] == 1
    # single-end library: strand compatibility only
    se = _hits([(5, 0, True, SINGLE), (6, 0, False, SINGLE), (7, 0, False, SINGLE)])
    ids, o, _, _, st = O.filter_hits(se, np.array([0, 3], np.uint32), FORMATS["SR"], False)
    assert ids.tolist() == [6, 7] and st["n_fwd"] == 0 and st["n_rc"] == 2


def _random_reads(rng, R, M, paired):
    n = rng.choice([0, 1, 1, 1, 2, 3, 5, 9, 40, 250], R, p=[.05, .3, .1, .1, .15, .1, .1, .05, .04, .01])
    off = np.zeros(R + 1, np.uint32); off[1:] = np.cumsum(n)
    H = int(off[-1])
    h = np.zeros(H, O.HIT_DTYPE)
    h["pos"] = rng.integers(-20, 3000, H); h["mate_pos"] = rng.integers(-20, 3000, H)
    h["read_len"] = rng.integers(30, 150, H); h["mate_len"] = rng.integers(30, 150, H)
    h["frag_len"] = rng.integers(0, 1400, H)
    h["fwd"] = rng.integers(0, 2, H); h["mate_fwd"] = rng.integers(0, 2, H)
    kind = rng.integers(0, 3, R)                     # per read: proper pairs / orphans / (single-end library: all SINGLE)
    for r in range(R):
        b, e = off[r], off[r + 1]
        k = e - b
        if k == 0:
            continue
        if not paired:
            h["mate_status"][b:e] = SINGLE
            h["tid"][b:e] = np.sort(rng.integers(0, M, k))
        elif kind[r] == 0:
            h["mate_status"][b:e] = PAIRED
            h["tid"][b:e] = np.sort(rng.integers(0, M, k))
        else:                                        # orphans: left run then right run, each ascending in tid (ties across runs happen)
            nl = int(rng.integers(0, k + 1))
            h["mate_status"][b:b + nl] = LEFT; h["mate_status"][b + nl:e] = RIGHT
            h["tid"][b:b + nl] = np.sort(rng.integers(0, M, nl)); h["tid"][b + nl:e] = np.sort(rng.integers(0, M, k - nl))
    return h, off


@pytest.mark.gpu
def test_filter_hits_device_vs_oracle(built, gpu):
    import torch
    import sailfish_amd as sf
    rng = np.random.default_rng(17)
    for trial in range(14):
        paired = trial % 2 == 0
        name = str(rng.choice(["IU", "ISF", "ISR", "OU", "OSF", "OSR", "MU", "MSF", "MSR"] if paired else ["U", "SF", "SR"]))
        if trial >= 12:
            name = "ISF" if not paired else "SR"     # a library format of the "wrong" kind for the loop: still well defined
        R = int(rng.choice([1, 1000, 70_000]))
        h, off = _random_reads(rng, R, 5000, paired)
        kw = dict(discard_orphans=bool(rng.integers(0, 2)), ignore_compat=bool(rng.integers(0, 4) == 0),
                  enforce_compat=bool(rng.integers(0, 3) == 0), can_dovetail=bool(rng.integers(0, 2)),
                  max_read_occs=int(rng.choice([3, 200])), max_frag_len=1000)
        budget = int(rng.choice([0, 7, 10_000_000]))
        fl0 = rng.integers(0, 5, 1000).astype(np.uint32)
        oi, oo, ofl, orem, ost = O.filter_hits(h, off, FORMATS[name], paired, fl_counts=fl0, remaining_fl_ops=budget, **kw)
        d_fl = torch.from_numpy(fl0.view(np.int32).copy()).to(gpu)
        gi, go, grem, gst = sf.hits.filter_hits(h, off, name, paired_library=paired, allow_orphans=not kw["discard_orphans"],
                                                ignore_lib_compat=kw["ignore_compat"], enforce_lib_compat=kw["enforce_compat"],
                                                allow_dovetail=kw["can_dovetail"], max_read_occs=kw["max_read_occs"],
                                                max_frag_len=1000, fl_counts=d_fl, remaining_fl_ops=budget, device=gpu)
        np.testing.assert_array_equal(go.cpu().numpy().view(np.uint32), oo)
        np.testing.assert_array_equal(gi.cpu().numpy().view(np.uint32), oi)
        np.testing.assert_array_equal(d_fl.cpu().numpy().view(np.uint32), ofl)
        assert grem == orem and gst == ost, (trial, name, kw, gst, ost)
    # reads without any hit at all, and an empty batch
    gi, go, _, gst = sf.hits.filter_hits(np.zeros(0, O.HIT_DTYPE), np.zeros(6, np.uint32), "U", device=gpu)
    assert gi.numel() == 0 and go.cpu().tolist() == [0] * 6 and gst["n_observed"] == 5 and gst["n_mapped"] == 0
    gi, go, _, gst = sf.hits.filter_hits(np.zeros(0, O.HIT_DTYPE), np.zeros(1, np.uint32), "U", device=gpu)
    assert gi.numel() == 0 and go.cpu().tolist() == [0] and gst["n_observed"] == 0
    # the filtered lists feed the class builder without leaving the device
    h, off = _random_reads(rng, 120_000, 300, True)
    oi, oo, *_ = O.filter_hits(h, off, FORMATS["IU"], True, discard_orphans=False)
    gi, go, _, gst = sf.hits.filter_hits(h, off, "IU", allow_orphans=True, device=gpu)
    ob = O.EqBuilder(); ob.add_batch(oi, oo.astype(np.uint64)); orp, oids, ocnt, ohash = ob.finish()
    eq = sf.EquivalenceClassBuilder(device=gpu); eq.start(); eq.add_batch(gi, go); eq.finish()
    rp, ii, cc, hh = eq.eqVec().to_numpy()
    assert eq.total_reads == gst["n_mapped"] == ob.total_reads
    np.testing.assert_array_equal(rp, orp.astype(np.uint32)); np.testing.assert_array_equal(ii, oids)
    np.testing.assert_array_equal(cc, ocnt); np.testing.assert_array_equal(hh, ohash)


# ---- bias / GC samples of the same loop (sfgpu_sample_bias) ----------------------------------------------------
def _txome(rng, M, lo=40, hi=3000, alphabet=b"ACGT"):
    lens = rng.integers(lo, hi, M).astype(np.uint32)
    off = np.zeros(M, np.uint64); parts = []; pos = 0
    for t in range(M):
        off[t] = pos
        parts.append(bytes(rng.choice(np.frombuffer(alphabet, np.uint8), int(lens[t])).astype(np.uint8)) + b"$")
        pos += int(lens[t]) + 1
    return b"".join(parts), off, lens


def test_bias_samples_on_hand_built_reads(built):
    """ReadKmerDist<6>::update (include/ReadKmerDist.hpp:35-73) through the loop (:270-287), and the GC sample (:375-389)"""
    seq = b"AACCGGTTACGTACGTAAAACCCCGGGGTTTTACGT" + b"$"          # one transcript, L = 36
    so, rl = np.array([0], np.uint64), np.array([36], np.uint32)
    base = dict(seq=seq, seq_off=so, ref_len=rl, read_bias=np.ones(4096, np.uint32), remaining_bias_samples=10)
    one = lambda rows, **kw: O.filter_hits_bias(_hits(rows), np.array([0, len(rows)], np.uint32), FORMATS["U"], False, **{**base, **kw})
    # forward read starting at 10: window = seq[8:14] = "ACGTAC", stored reverse-complemented
    _, rb, rem, _, nb, _ = one([(0, 10, True, SINGLE)])
    assert nb == 1 and rem == 9 and rb[O.index_for_kmer(b"ACGTAC", rc=True)] == 2 and rb.sum() == 4097
    # reverse read at pos 5, length 20: start = 25, window = seq[21:27] = "CCCGGG", stored forward
    _, rb, rem, _, nb, _ = one([(0, 5, False, SINGLE, 0, 0, 0, 20, 0)])
    assert nb == 1 and rb[O.index_for_kmer(b"CCCGGG")] == 2
    # no room for the window: start 1 (< 2 before), start 33 (33 - 2 + 6 = 37 > 36), start 0, start at the end
    for rows in ([(0, 1, True, SINGLE)], [(0, 33, True, SINGLE)], [(0, 0, True, SINGLE)], [(0, 36, True, SINGLE)],
                 [(0, 3, False, SINGLE, 0, 0, 0, 0, 0)]):
        _, rb, rem, _, nb, _ = one(rows)
        assert nb == 0 and rem == 10 and rb.sum() == 4096
    # the first hit that yields a sample is the one counted; the budget stops everything
    _, rb, rem, _, nb, _ = one([(0, 1, True, SINGLE), (0, 10, True, SINGLE), (0, 12, True, SINGLE)])
    assert nb == 1 and rb[O.index_for_kmer(b"ACGTAC", rc=True)] == 2
    _, rb, rem, _, nb, _ = one([(0, 10, True, SINGLE)], remaining_bias_samples=0)
    assert nb == 0 and rb.sum() == 4096
    # a byte outside ACGT in the window: no success
    bad = dict(base); bad["seq"] = seq[:9] + b"N" + seq[10:]
    _, rb, rem, _, nb, _ = O.filter_hits_bias(_hits([(0, 10, True, SINGLE)]), np.array([0, 1], np.uint32), FORMATS["U"], False, **bad)
    assert nb == 0
    # fragment GC of a proper pair: start = min(pos, matePos) = 4, stop = 4 + 20: GC of seq[5:25] over 21
    rows = [(0, 4, True, PAIRED, 14, False, 20)]
    (_, _, _, _, st), _, _, og, _, ng = O.filter_hits_bias(_hits(rows), np.array([0, 1], np.uint32), FORMATS["IU"], True, seq=seq, seq_off=so,
                                                           ref_len=rl, observed_gc=np.ones(101, np.uint32))
    gc = sum(c in b"GC" for c in seq[5:25])
    assert ng == 1 and og[int(np.rint(100.0 * gc / 21))] == 2 and og.sum() == 102 and st["n_mapped"] == 1
    # start 0 or stop at the end: no sample
    for rows in ([(0, 0, True, PAIRED, 14, False, 20)], [(0, 4, True, PAIRED, 14, False, 32)]):
        _, _, _, og, _, ng = O.filter_hits_bias(_hits(rows), np.array([0, 1], np.uint32), FORMATS["IU"], True, seq=seq, seq_off=so,
                                                ref_len=rl, observed_gc=np.ones(101, np.uint32))
        assert ng == 0 and og.sum() == 101


@pytest.mark.gpu
def test_sample_bias_device_vs_oracle(built, gpu):
    import torch
    import sailfish_amd as sf
    rng = np.random.default_rng(23)
    M = 400
    seq, so, rl = _txome(rng, M, alphabet=b"ACGTacgtN")               # a few N: windows with one give no sample
    d_seq = torch.from_numpy(np.frombuffer(seq, np.uint8).copy()).to(gpu)
    d_so = torch.from_numpy(so.astype(np.int64)).to(gpu)
    d_rl = torch.from_numpy(rl.view(np.int32).copy()).to(gpu)
    pre = sf.hits.gc_prefix(d_seq, d_so, d_rl)
    # the prefix table is Transcript::GCCount_ per transcript
    hp = pre.cpu().numpy().view(np.uint32)
    for t in (0, 17, M - 1):
        s = np.frombuffer(seq[int(so[t]):int(so[t]) + int(rl[t])], np.uint8)
        np.testing.assert_array_equal(hp[int(so[t]):int(so[t]) + int(rl[t])], np.cumsum(np.isin(s, np.frombuffer(b"GCgc", np.uint8))))
    for trial in range(10):
        paired = trial % 2 == 0
        name = str(rng.choice(["IU", "ISF", "OSR", "MU"] if paired else ["U", "SF", "SR"]))
        R = int(rng.choice([1, 1000, 60_000]))
        h, off = _random_reads(rng, R, M, paired)
        h["pos"] = rng.integers(-5, 1500, len(h)); h["mate_pos"] = rng.integers(-5, 1500, len(h)); h["frag_len"] = rng.integers(0, 900, len(h))
        kw = dict(discard_orphans=bool(rng.integers(0, 2)), max_read_occs=int(rng.choice([3, 200])), max_frag_len=1000)
        budget = int(rng.choice([0, 5, 300, 10_000_000]))
        want_seq, want_gc = bool(trial % 3 != 2), bool(trial % 3 != 1)
        rb0 = rng.integers(1, 9, 4096).astype(np.uint32); og0 = rng.integers(1, 9, 101).astype(np.uint32)
        _, orb, orem, oog, onb, ong = O.filter_hits_bias(h, off, FORMATS[name], paired, seq, so, rl, read_bias=rb0 if want_seq else None,
                                                         remaining_bias_samples=budget, observed_gc=og0 if want_gc else None, **kw)
        d_rb = torch.from_numpy(rb0.view(np.int32).copy()).to(gpu) if want_seq else None
        d_og = torch.from_numpy(og0.view(np.int32).copy()).to(gpu) if want_gc else None
        gstep = int(rng.choice([1, 1, 3, 10]))
        if gstep > 1:                                                  # --gcSizeSamp: the interpolated counts
            _, orb, orem, oog, onb, ong = O.filter_hits_bias(h, off, FORMATS[name], paired, seq, so, rl, read_bias=rb0 if want_seq else None,
                                                             remaining_bias_samples=budget, observed_gc=og0 if want_gc else None,
                                                             gc_size_samp=gstep, **kw)
        grem, gnb, gng = sf.hits.sample_bias(h, off, name, d_seq, d_so, d_rl, read_bias=d_rb, remaining_bias_samples=budget,
                                             observed_gc=d_og, gc_prefix_table=pre, gc_size_samp=gstep, paired_library=paired,
                                             allow_orphans=not kw["discard_orphans"], max_read_occs=kw["max_read_occs"], device=gpu)
        if want_seq:
            np.testing.assert_array_equal(d_rb.cpu().numpy().view(np.uint32), orb)
            assert (grem, gnb) == (orem, onb), (trial, grem, orem, gnb, onb)
        if want_gc:
            np.testing.assert_array_equal(d_og.cpu().numpy().view(np.uint32), oog)
            assert gng == ong and (ong > 0) == (paired and ong > 0)
    # reads without any hit: nothing sampled, nothing dereferenced
    d_rb = torch.ones(4096, dtype=torch.int32, device=gpu)
    assert sf.hits.sample_bias(np.zeros(0, O.HIT_DTYPE), np.zeros(6, np.uint32), "U", d_seq, d_so, d_rl, read_bias=d_rb,
                               remaining_bias_samples=9, device=gpu) == (9, 0, 0) and int(d_rb.sum()) == 4096
