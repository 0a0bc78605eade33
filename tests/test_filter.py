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
