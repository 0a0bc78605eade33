"""The quasi-mapping front end (sailfish_amd/mapper.py, csrc/mapper.hip): record-for-record against the CPU
restatement of its contract (oracle/mapper_oracle.py), against the committed hit records of the reference's bundled
sample_data (tests/golden/sample_data_hits.npz), and -- the only ground truth there is for a mapper here, RapMap being
unavailable -- against the transcript each simulated read names in its header."""
import os

import numpy as np
import pytest

from oracle import mapper_oracle as MO
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sample_reads():
    d = np.load(os.path.join(GOLD, "sample_data_reads.npz"))
    n, L = len(d["truth"]), int(d["read_len"])

    def unpack(p):
        b = np.unpackbits(p).reshape(-1, 2)
        return np.frombuffer(b"ACGT", np.uint8)[(b[:, 0] * 2 + b[:, 1])[: n * L]].reshape(n, L)
    seqs = [bytes(d["seq"][d["seq_off"][t]:d["seq_off"][t + 1]]) for t in range(len(d["names"]))]
    m1, m2 = unpack(d["mate1_2bit"]), unpack(d["mate2_2bit"])
    return [str(x) for x in d["names"]], seqs, [bytes(r) for r in m1], [bytes(r) for r in m2], d["truth"]


def test_contract_restatement_reproduces_the_committed_hit_records(built):
    """the CPU restatement of the mapper's contract, run on the bundled reads, gives the committed fixture (first 1500 pairs)"""
    names, seqs, r1, r2, truth = _sample_reads()
    gold = np.load(os.path.join(GOLD, "sample_data_hits.npz"))
    gh = gold["hits"].view(O.HIT_DTYPE); go = gold["offsets"]
    n = 1500
    hits, off = MO.map_reads(MO.build_index(seqs), r1[:n], r2[:n])
    assert np.array_equal(off, go[: n + 1]) and np.array_equal(hits, gh[: go[n]])
    assert [str(x) for x in gold["names"]] == names and np.array_equal(gold["ref_len"], [len(s) for s in seqs])


def _random_case(rng, M=40, n_reads=3000, read_len=60, k=31):
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), 4000)
    seqs = []
    for t in range(M):                                  # isoform-like: shared segments, so k-mers occur in several transcripts
        a = rng.integers(0, 3000); ln = rng.integers(k - 5, 900)          # a few transcripts shorter than k
        s = base[a:a + ln].copy()
        if t % 7 == 0 and ln > 100:
            s[rng.integers(0, ln, 3)] = ord("N")       # bases that never match
        if t % 5 == 0:
            s = np.frombuffer(s.tobytes().lower(), np.uint8)                                     # case is folded
        seqs.append(bytes(s))
    comp = bytes.maketrans(b"ACGTacgtN", b"TGCAtgcaN")
    r1, r2 = [], []
    for _ in range(n_reads):
        t = rng.integers(0, M); s = seqs[t]
        if len(s) < read_len + 20 or rng.random() < 0.05:
            r1.append(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), rng.integers(10, read_len + 1)))); r2.append(r1[-1][::-1])      # noise / short reads
            continue
        frag = rng.integers(read_len, min(len(s), 300) + 1); p = rng.integers(0, len(s) - frag + 1)
        left = s[p:p + read_len]; right = s[p + frag - read_len:p + frag].translate(comp)[::-1]
        if rng.random() < 0.5:
            left, right = right, left                   # the fragment came from the other strand
        if rng.random() < 0.1:
            left = left[:rng.integers(k, read_len)]      # ragged lengths
        if rng.random() < 0.05:
            b = bytearray(right); b[rng.integers(0, len(b))] = ord("N"); right = bytes(b)
        r1.append(bytes(left)); r2.append(bytes(right))
    return seqs, r1, r2


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [True, False])
def test_device_mapper_matches_its_contract(gpu, paired):
    import sailfish_amd as sf
    rng = np.random.default_rng(17 + paired)
    seqs, r1, r2 = _random_case(rng)
    idx = sf.mapper.QuasiIndex(seqs, k=31, max_occ=1000, device=gpu, seed_len=0)          # the end-seed contract (default: scan mode)
    assert idx.n_positions == sum(max(len(s) - 30, 0) for s in seqs)
    hits, off = idx.map_reads(r1, r2 if paired else None)
    gh, go = sf.mapper.hits_to_numpy(hits, off)
    oh, oo = MO.map_reads(MO.build_index(seqs, 31, 1000), r1, r2 if paired else None)
    assert np.array_equal(go, oo)
    assert np.array_equal(gh, oh)
    assert len(oh) > len(r1) // 2 and (not paired or int((oh["mate_status"] == 3).sum()) > len(r1) // 4)
    # a repeat that exceeds max_occ: the same cut on both sides
    rep = [b"ACGT" * 40] * 6 + seqs[:5]
    idx2 = sf.mapper.QuasiIndex(rep, k=31, max_occ=3, device=gpu, seed_len=0)
    h2, o2 = sf.mapper.hits_to_numpy(*idx2.map_reads([b"ACGT" * 15, b"TTTT" * 15, b"AC"]))
    eh, eo = MO.map_reads(MO.build_index(rep, 31, 3), [b"ACGT" * 15, b"TTTT" * 15, b"AC"])
    assert np.array_equal(o2, eo) and np.array_equal(h2, eh)


def _with_errors(rng, reads, rate):
    """substitutions at `rate` per base (A/C/G/T only; other characters stay)"""
    sub = {65: b"CGT", 67: b"AGT", 71: b"ACT", 84: b"ACG", 97: b"cgt", 99: b"agt", 103: b"act", 116: b"acg"}
    out = []
    for r in reads:
        b = bytearray(r)
        for i in np.nonzero(rng.random(len(b)) < rate)[0]:
            if b[i] in sub:
                b[i] = sub[b[i]][int(rng.integers(0, 3))]
        out.append(bytes(b))
    return out


def test_more_seeds_map_more_reads_with_errors(built):
    """the contract with S > 2 seeds per strand (CPU restatement): on the bundled reads with 2 % substitutions, eight seeds per
    strand place the simulated transcript among the hits of more reads than the two end seeds do (a read maps if ANY of its
    seeds is error free; with k = 31 and 70-base reads the gain is a few percent), and S = 2 is the old contract"""
    names, seqs, r1, r2, truth = _sample_reads()
    rng = np.random.default_rng(3)
    n = 400
    e1 = _with_errors(rng, r1[:n], 0.02)
    index = MO.build_index(seqs)
    found = {}
    for S in (2, 8):
        ok = 0
        for r in range(n):
            ok += int(truth[r] in [t for t, f, p in MO.map_read(index, e1[r], seeds=S)])
        found[S] = ok
    assert found[8] > found[2], found
    assert all(MO.map_read(index, r1[r], seeds=2) == MO.map_read(index, r1[r]) for r in range(50))


@pytest.mark.gpu
@pytest.mark.parametrize("seeds", [3, 5, 8])
def test_device_mapper_with_more_seeds_matches_its_contract(gpu, seeds):
    """S > 2 seeds per strand on the device = the restated contract, record for record: random isoform-like transcriptome (shared
    segments, N, lower case, ragged and short reads), reads with 3 % substitutions, paired and single end; setting the seeds
    back to 2 gives the two-seed records again"""
    import sailfish_amd as sf
    rng = np.random.default_rng(40 + seeds)
    seqs, r1, r2 = _random_case(rng, n_reads=2000, read_len=70)
    r1, r2 = _with_errors(rng, r1, 0.03), _with_errors(rng, r2, 0.03)
    idx = sf.mapper.QuasiIndex(seqs, k=31, max_occ=1000, device=gpu, seeds=seeds)
    oi = MO.build_index(seqs, 31, 1000)
    for paired in (True, False):
        gh, go = sf.mapper.hits_to_numpy(*idx.map_reads(r1, r2 if paired else None))
        oh, oo = MO.map_reads(oi, r1, r2 if paired else None, seeds=seeds)
        assert np.array_equal(go, oo) and np.array_equal(gh, oh)
        assert len(oh) > len(r1) // 3
    idx.set_seeds(2)
    gh, go = sf.mapper.hits_to_numpy(*idx.map_reads(r1, r2))
    oh, oo = MO.map_reads(oi, r1, r2)
    assert np.array_equal(go, oo) and np.array_equal(gh, oh)


@pytest.mark.gpu
def test_bundled_sample_data_from_the_reads(gpu, tmp_path):
    """BASELINE config 1 from the READS on: index the 15 transcripts, map the 10 000 pairs on the device -- the records are
    the committed fixture's, byte for byte -- and quantify; every read's simulated transcript is among its hits and the EM
    recovers the simulated abundances"""
    import sailfish_amd as sf
    names, seqs, r1, r2, truth = _sample_reads()
    idx = sf.mapper.QuasiIndex(seqs, device=gpu)
    for seed_len, fixture in ((None, "sample_data_hits_scan.npz"), (0, "sample_data_hits.npz")):     # default = scan mode; 0 = end seeds
        gold = np.load(os.path.join(GOLD, fixture))
        if seed_len is not None:
            idx.set_scan(seed_len)
        hits, off = sf.mapper.hits_to_numpy(*idx.map_reads(r1, r2))
        assert np.array_equal(off, gold["offsets"]) and np.array_equal(hits, gold["hits"].view(O.HIT_DTYPE)), fixture
        assert all(truth[r] in hits["tid"][off[r]:off[r + 1]] for r in range(len(truth)))
    out = str(tmp_path / "out")
    rc, exp = sf.mapper.quantify_reads(names, seqs, r1, r2, "IU", out, sf.SailfishOpts(numFragSamples=5000), batch_reads=3000,
                                       cmd_options={"libType": "IU"}, device=gpu)
    assert rc == 0 and exp.numMappedFragments() == 10000
    rows = [l.split("\t") for l in open(os.path.join(out, "quant.sf")).read().strip().split("\n")[1:]]
    assert [r[0] for r in rows] == names
    num_reads = np.array([float(r[4]) for r in rows])
    want = np.bincount(truth, minlength=15).astype(np.float64)
    assert abs(num_reads.sum() - 10000) < 1e-2 and np.abs(num_reads - want).sum() / 10000 < 0.1
    print("sample_data from the reads (scan mode): L1 error of NumReads vs the simulator's truth", np.abs(num_reads - want).sum() / 10000)


# ---- scan mode (maximal-match extension; the default since round 3) ---------------------------------------------------------

def test_scan_contract_reproduces_its_committed_hit_records(built):
    """the CPU restatement of the scan contract on the bundled reads gives the committed scan fixture (first 800 pairs)"""
    names, seqs, r1, r2, truth = _sample_reads()
    gold = np.load(os.path.join(GOLD, "sample_data_hits_scan.npz"))
    gh = gold["hits"].view(O.HIT_DTYPE); go = gold["offsets"]
    n = 800
    hits, off = MO.scan_reads(MO.build_scan_index(seqs), r1[:n], r2[:n], s=19)
    assert np.array_equal(off, go[: n + 1]) and np.array_equal(hits, gh[: go[n]])


def test_scan_contract_maps_reads_with_substitutions(built):
    """on the bundled reads (2 x 50 bases) with 2 % substitutions the end-seed contract places the simulated
    transcript among the hits of 274 of 400 reads (287 with eight 31-mer seeds: a single substitution in the middle of a 50-base
    read leaves no clean 31-mer); the scan contract with 19-base seeds and maximal-match extension must reach 380"""
    names, seqs, r1, r2, truth = _sample_reads()
    rng = np.random.default_rng(3)
    n = 400
    e1 = _with_errors(rng, r1[:n], 0.02)
    si = MO.build_scan_index(seqs)
    got = {s: sum(int(truth[r] in [t for t, f, p in MO.scan_read(si, e1[r], s=s)]) for r in range(n)) for s in (31, 19, 15)}
    assert got[19] >= 380 and got[15] >= got[19] >= got[31], got
    # error-free reads: every read carries its transcript, and a whole-read forward match never walks the other strand
    assert all(truth[r] in [t for t, f, p in MO.scan_read(si, r1[r], s=19)] for r in range(n))


@pytest.mark.gpu
@pytest.mark.parametrize("seed_len,rate", [(19, 0.03), (15, 0.05), (31, 0.0), (8, 0.02)])
def test_device_scan_mapper_matches_its_contract(gpu, seed_len, rate):
    """scan mode on the device = the restated contract, record for record: random isoform-like transcriptome (shared segments, N,
    lower case, transcripts shorter than k, ragged and short reads, reads with N), substitutions, paired and single end, several
    seed lengths (31 = whole k-mers, 8 = shorter than the bucket prefix: the range spans buckets), a repeat beyond max_occ"""
    import sailfish_amd as sf
    rng = np.random.default_rng(90 + seed_len)
    seqs, r1, r2 = _random_case(rng, n_reads=1500, read_len=70)
    if rate:
        r1, r2 = _with_errors(rng, r1, rate), _with_errors(rng, r2, rate)
    max_occ = 50 if seed_len == 8 else 1000
    idx = sf.mapper.QuasiIndex(seqs, k=31, max_occ=max_occ, device=gpu, seed_len=seed_len)
    si = MO.build_scan_index(seqs)
    for paired in (True, False):
        gh, go = sf.mapper.hits_to_numpy(*idx.map_reads(r1, r2 if paired else None))
        oh, oo = MO.scan_reads(si, r1, r2 if paired else None, s=seed_len, max_occ=max_occ)
        assert np.array_equal(go, oo) and np.array_equal(gh, oh)
        assert len(oh) > len(r1) // 3
    # a repeat that exceeds max_occ is a miss in scan mode (dropped, not truncated), on both sides
    rep = [b"ACGT" * 40, b"ACGT" * 30 + b"GATTACAGATTACAGATTACAGATTACAGATTACA"]
    idx2 = sf.mapper.QuasiIndex(rep, k=31, max_occ=3, device=gpu, seed_len=19)
    q = [b"ACGT" * 15, b"TACAGATTACAGATTACAGATTACA", b"AC"]
    h2, o2 = sf.mapper.hits_to_numpy(*idx2.map_reads(q))
    eh, eo = MO.scan_reads(MO.build_scan_index(rep), q, s=19, max_occ=3)
    assert np.array_equal(o2, eo) and np.array_equal(h2, eh) and o2[1] == 0 and o2[2] > o2[1]


@pytest.mark.gpu
def test_device_scan_mapper_recovers_reads_with_errors(gpu):
    """the device's default mode on the 400 bundled reads with 2 % substitutions: >= 380 carry their true transcript"""
    import sailfish_amd as sf
    names, seqs, r1, r2, truth = _sample_reads()
    rng = np.random.default_rng(3)
    n = 400
    e1 = _with_errors(rng, r1[:n], 0.02)
    idx = sf.mapper.QuasiIndex(seqs, device=gpu)
    assert idx.seed_len == 19
    hits, off = sf.mapper.hits_to_numpy(*idx.map_reads(e1))
    ok = sum(int(truth[r] in hits["tid"][off[r]:off[r + 1]]) for r in range(n))
    idx.set_scan(0)
    hits0, off0 = sf.mapper.hits_to_numpy(*idx.map_reads(e1))
    ok0 = sum(int(truth[r] in hits0["tid"][off0[r]:off0[r + 1]]) for r in range(n))
    print(f"reads with 2 % substitutions carrying their transcript: scan mode {ok} / {n}, end seeds {ok0} / {n}")
    assert ok >= 380 > ok0
