"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs, against the golden fixtures, and -- at BASELINE.json's full size -- through
size-independent properties.  Bar: bit-exact for hashes / class labels / counts; TPM and NumReads
within 1e-4 relative (north_star), and in practice ~1e-12."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-4          # north_star: TPM / NumReads within 1e-4 relative
TIGHT = 1e-9            # what we actually hold ourselves to on these sizes


def _pack(reads):
    off = np.zeros(len(reads) + 1, np.uint32)
    off[1:] = np.cumsum([len(r) for r in reads])
    ids = np.concatenate([np.asarray(r, np.uint32) for r in reads]) if off[-1] else np.zeros(0, np.uint32)
    return ids.astype(np.uint32), off


def _oracle_classes(batches):
    b = O.EqBuilder()
    for ids, off in batches:
        b.add_batch(ids, off.astype(np.uint64))
    rp, ii, cc, hh = b.finish()
    return b, rp.astype(np.uint32), ii, cc, hh


def _gpu_classes(sf, gpu, batches, device_batches=True, **kw):
    import torch
    eq = sf.EquivalenceClassBuilder(device=gpu, **kw)
    eq.start()
    for ids, off in batches:
        if device_batches:
            eq.add_batch(torch.from_numpy(ids.view(np.int32)).to(gpu), torch.from_numpy(off.view(np.int32)).to(gpu))
        else:
            eq.add_batch(ids, off)
    eq.finish()
    return eq


def _assert_same_classes(eq, ob, orp, oids, ocnt, ohash):
    rp, ii, cc, hh = eq.eqVec().to_numpy()
    assert (eq.n_classes, eq.nnz, eq.total_reads) == (ob.n_classes, ob.nnz, ob.total_reads)
    np.testing.assert_array_equal(rp, orp); np.testing.assert_array_equal(ii, oids)
    np.testing.assert_array_equal(cc, ocnt); np.testing.assert_array_equal(hh, ohash)


@pytest.fixture(scope="module")
def sf(gpu):
    import sailfish_amd
    return sailfish_amd


def _needs_variants():
    """the kernel forms that lost their A/Bs (ring / quad / pipelined class build, graph-replayed EM loops) live in builds made with
    -DSFGPU_VARIANTS only (tools/eq_variants.sh all:"" ; SFGPU_LIB_PATH=sailfish_amd/csrc/variants/libsfgpu_all.so): their tests
    run against such a library and are skipped against the product"""
    from sailfish_amd import _lib
    if not _lib.lib().sfgpu_has_variants():
        pytest.skip("the product library does not contain this variant (build one with tools/eq_variants.sh / tools/em_variants.sh)")


# ---------------------------------------------------------------------------------------- a1
def test_xxh64_kernel_golden_and_random(sf, gpu):
    g = json.load(open(os.path.join(GOLD, "xxh64_vectors.json")))["vectors"]
    reads = [np.array(v["ids"], np.uint32) for v in g]
    rng = np.random.default_rng(5)
    for n in list(range(0, 70)) + [127, 128, 129, 199, 200, 201, 1000]:
        reads.append(rng.integers(0, 2 ** 32, n, dtype=np.uint32))
    ids, off = _pack(reads)
    got = sf.xxh64_labels(ids, off, device=gpu).cpu().numpy().view(np.uint64)
    for i, v in enumerate(g):
        assert "%016x" % int(got[i]) == v["xxh64"]
    np.testing.assert_array_equal(got, O.xxh64_lists(ids, off.astype(np.uint64)))


# ------------------------------------------------------------------------------------- a2-a5
def test_builder_kat_and_edge_cases(sf, gpu):
    rng = np.random.default_rng(9)
    reads = [[1, 2, 3]] * 3 + [[5]] * 2 + [[2, 9]]                       # SURVEY 8c builder KAT
    reads += [[], [], [7], [1, 2], [2, 1], [1, 1, 2], [1, 2, 1]] * 2     # empties, permutations, repeated ids
    reads += [list(range(200)), list(range(200)), list(range(199, -1, -1))]   # maxReadOccs-long labels
    for _ in range(20000):
        n = int(rng.choice([1, 1, 2, 3, 4, 7, 8, 9, 16, 33]))
        reads.append(rng.integers(1000, 1040, n).tolist())
    batch = _pack(reads)
    ob, *oc = _oracle_classes([batch])
    for dev in (True, False):
        eq = _gpu_classes(sf, gpu, [batch], device_batches=dev)
        _assert_same_classes(eq, ob, *oc)
    rp, ii, cc, _ = eq.eqVec().to_numpy()
    got = {tuple(ii[rp[c]:rp[c + 1]]): int(cc[c]) for c in range(eq.n_classes)}
    assert got[(1, 2, 3)] == 3 and got[(5,)] == 2 and got[(2, 9)] == 1
    assert got[(1, 2)] == 2 and got[(2, 1)] == 2 and got[(1, 1, 2)] == 2 and got[tuple(range(200))] == 2


def test_builder_empty_and_reuse(sf, gpu):
    eq = sf.EquivalenceClassBuilder(device=gpu)
    eq.start(); eq.finish()
    assert eq.n_classes == 0 and eq.total_reads == 0 and eq.eqVec().size() == 0
    eq.start()
    eq.add_batch(np.zeros(0, np.uint32), np.zeros(4, np.uint32))        # three empty reads
    eq.addGroup([3, 4]); eq.addGroup([3, 4]); eq.addGroup([9])
    eq.finish()
    rp, ii, cc, _ = eq.eqVec().to_numpy()
    assert eq.n_classes == 2 and eq.total_reads == 3 and sorted(cc.tolist()) == [1, 2]
    eq.start()                                                           # reuse clears
    eq.addGroup([1]); eq.finish()
    assert eq.n_classes == 1 and eq.total_reads == 1


def test_builder_batching_and_order_independence(sf, gpu):
    """same multiset of reads in one batch, many batches, shuffled: identical canonical export"""
    from sailfish_amd import synth
    _, ids, off = synth.workload(5000, 20000, 300_000)
    ids = ids.numpy().view(np.uint32); off = off.numpy().view(np.uint32)
    ob, *oc = _oracle_classes([(ids, off)])
    eq = _gpu_classes(sf, gpu, [(ids, off)]); _assert_same_classes(eq, ob, *oc)
    cuts = [0, 1, 1000, 1001, 150_000, 299_999, 300_000]
    parts = [(ids[off[a]:off[b]].copy(), (off[a:b + 1] - off[a]).astype(np.uint32)) for a, b in zip(cuts[:-1], cuts[1:])]
    eq = _gpu_classes(sf, gpu, parts); _assert_same_classes(eq, ob, *oc)
    eq = _gpu_classes(sf, gpu, parts[::-1], device_batches=False); _assert_same_classes(eq, ob, *oc)


def test_builder_partitioned_path_edge_cases(sf, gpu):
    """a batch big enough for the radix-partitioned kernels (>= 65536 reads) holding everything they
    special-case: empty reads, labels of 1..8, 9..128 and > 128 ids (the last take the generic kernel),
    one very hot label, ids with bit 31 set (the partition stream uses that bit as its label marker), and
    an id array that does not start on a 16-byte boundary (the passes stage ids with 16-byte loads)"""
    import torch
    rng = np.random.default_rng(11)
    pool = []
    for n in [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 64, 127, 128, 129, 130, 255, 256, 300, 1000]:
        for _ in range(3):
            pool.append(np.sort(rng.choice(50_000, n, replace=False)).astype(np.uint32))
    pool.append(np.array([0x80000000, 0x80000001], np.uint32))           # bit 31 in the first id
    pool.append(np.array([5, 0xFFFFFFFF], np.uint32))                    # ... and in a later one
    pool.append(np.array([0x80000000], np.uint32))
    pool.append(np.array([7, 8, 9], np.uint32))                          # the hot label
    hot = len(pool) - 1
    pick = rng.integers(0, len(pool), 200_000)
    pick[rng.random(200_000) < 0.5] = hot
    reads = [pool[i] for i in pick]
    for k in rng.integers(0, len(reads), 500):                           # empty reads in between
        reads[k] = np.zeros(0, np.uint32)
    ids, off = _pack(reads)
    ob, *oc = _oracle_classes([(ids, off)])
    for shift in (0, 1, 2, 3):                                           # ids pointer = 16-byte aligned + 4 * shift
        buf = torch.zeros(ids.size + 8, dtype=torch.int32, device=gpu)
        view = buf[shift:shift + ids.size]
        view.copy_(torch.from_numpy(ids.view(np.int32)))
        assert view.data_ptr() % 16 == (4 * shift) % 16
        eq = sf.EquivalenceClassBuilder(device=gpu)
        eq.start(); eq.add_batch(view, torch.from_numpy(off.view(np.int32)).to(gpu)); eq.finish()
        _assert_same_classes(eq, ob, *oc)
    # the same reads split so that a sub-batch boundary falls inside the batch
    eq = sf.EquivalenceClassBuilder(device=gpu)
    eq.start()
    cut = 100_001
    eq.add_batch(ids[:off[cut]].copy(), off[:cut + 1].copy())
    eq.add_batch(ids[off[cut]:].copy(), (off[cut:] - off[cut]).astype(np.uint32))
    eq.finish()
    _assert_same_classes(eq, ob, *oc)


@pytest.mark.parametrize("sub_batch", ["65536", None])
def test_builder_skewed_stream_hot_classes(sf, gpu, monkeypatch, sub_batch):
    """real RNA-seq is skewed: a few labels hold a large part of the reads.  From the second sub-batch on such classes are
    counted in the route pass itself (k_hot_select / the hot table of k_part_route) instead of overflowing their region's bins
    into the generic kernel; what does spill is added once per wavefront (k_insert).  Hot labels of 1, 3, 4, 8, 40 and 123 ids
    (one, two and many granules), more hot labels than hot-table slots can hold without collisions, near misses of the hottest labels (one id more, another last id), table growth between sub-batches (slots move: the hot table is rebuilt), two add_batch calls.
    sub_batch=65536: ~30 sub-batches; None: one 5 M-read batch, which takes the small scout sub-batch first."""
    import torch
    if sub_batch:
        monkeypatch.setenv("SFGPU_EQ_SUBBATCH", sub_batch)
    rng = np.random.default_rng(5)
    n_reads = 2_000_000 if sub_batch else 5_000_000
    M = 60_000
    hot = [np.sort(rng.choice(M, n, replace=False)).astype(np.uint32) for n in (1, 3, 4, 8, 40, 123)]
    hot += [np.sort(rng.choice(M, int(rng.integers(1, 12)), replace=False)).astype(np.uint32) for _ in range(300)]
    for h in hot[:6]:                                    # near misses of the hottest labels: same head, one id more / another last id
        hot.append(np.concatenate([h, [M + 5]]).astype(np.uint32))
        v = h.copy(); v[-1] = M + 9; hot.append(v)
    cold_n = 400_000
    k = np.minimum(1 + rng.geometric(0.3, cold_n), 60)
    base = rng.integers(0, M, cold_n)
    # cold label i = {base + 7 j}; some of them extend a hot label by one id (same head, different label)
    pick = rng.integers(0, cold_n, n_reads)
    u = rng.random(n_reads)
    hot_pick = np.where(u < 0.45, rng.integers(0, 6, n_reads), rng.integers(0, len(hot), n_reads))
    is_hot = u < 0.6
    lens = np.where(is_hot, np.array([len(h) for h in hot])[hot_pick], k[pick]).astype(np.int64)
    off = np.zeros(n_reads + 1, np.int64); np.cumsum(lens, out=off[1:])
    ids = np.empty(off[-1], np.uint32)
    rr = np.repeat(np.arange(n_reads), lens); j = np.arange(off[-1]) - off[:-1][rr]
    ids[:] = ((base[pick][rr] + 7 * j) % M).astype(np.uint32)
    hot_flat = np.concatenate(hot); hot_off = np.zeros(len(hot) + 1, np.int64); np.cumsum([len(h) for h in hot], out=hot_off[1:])
    hm = is_hot[rr]
    ids[hm] = hot_flat[hot_off[hot_pick[rr[hm]]] + j[hm]]
    # cold labels that wrap around M are not sorted: sort every cold label (the builder wants sorted labels like the mapper's)
    key = rr.astype(np.int64) * (1 << 32) + ids
    cold_rows = ~hm
    order = np.argsort(key[cold_rows], kind="stable")
    ids[cold_rows] = ids[cold_rows][order]
    off32 = off.astype(np.uint32)
    ob, *oc = _oracle_classes([(ids, off32)])
    cut = n_reads // 3 if sub_batch else n_reads - 300_000                      # (None: the first batch is big enough for the scout)
    eq = sf.EquivalenceClassBuilder(device=gpu, expected_classes=1000)          # small table: it grows while the reads arrive
    eq.start()
    eq.add_batch(torch.from_numpy(ids[:off[cut]].view(np.int32)).to(gpu), torch.from_numpy(off32[:cut + 1].view(np.int32)).to(gpu))
    eq.add_batch(torch.from_numpy(ids[off[cut]:].view(np.int32)).to(gpu),
                 torch.from_numpy((off[cut:] - off[cut]).astype(np.uint32).view(np.int32)).to(gpu))
    eq.finish()
    _assert_same_classes(eq, ob, *oc)
    st = eq.stats()
    assert st["insert_launches"] >= (20 if sub_batch else 3)
    # the hot classes hold 60 % of the reads: most of those are counted by the route pass; what spills to the generic kernel is
    # bounded by the sub-batches that ran before the hot classes were known (the first one; the 1 M-read scout)
    assert st["hot_reads"] > 0.3 * n_reads and st["spilled_reads"] < 0.15 * n_reads, st


@pytest.mark.parametrize("sub_batch,expected", [("65536", 1000), (None, 1000), (None, 0)])
def test_builder_clustered_stream_runs(sf, gpu, monkeypatch, sub_batch, expected):
    """reads that arrive CLUSTERED (a position-sorted input: the reads of one label come together).  The route pass folds a
    run of identical labels inside a 64-read step into its first read, which carries the run length in one more granule; the
    insert pass adds that length; a deferred run is replayed with its length as the weight; a spilled run goes to the generic
    kernel read by read.  Runs of 1 ... 300 reads, of labels of 1 ... 130 ids (one, two, many granules; > 123 ids take the
    generic kernel), empty reads inside runs, a fully sorted stretch, a shuffled stretch, runs crossing sub-batch and
    add_batch boundaries; expected=1000 starts from a smaller table (test_builder_deferred_runs_keep_their_length forces the
    deferral of runs)."""
    import torch
    if sub_batch:
        monkeypatch.setenv("SFGPU_EQ_SUBBATCH", sub_batch)
    rng = np.random.default_rng(17)
    M = 70_000
    n_labels = 150_000
    lens = np.minimum(1 + rng.geometric(0.22, n_labels), 130)
    lens[:40] = [1, 2, 3, 4, 7, 8, 9, 11, 12, 15, 16, 31, 64, 100, 123, 124, 125, 130, 5, 6] * 2
    labels = [np.sort(rng.choice(M, int(n), replace=False)).astype(np.uint32) for n in lens[:3000]]
    # the bulk of the labels: arithmetic progressions (cheap to build), sorted by construction
    base = rng.integers(0, M - 130 * 3, n_labels)
    reads = []
    def lab(i):
        return labels[i] if i < 3000 else (base[i] + 3 * np.arange(lens[i])).astype(np.uint32)
    # part 1: runs (run length law: mostly short, some > 64 and > 256)
    i = 0
    while len(reads) < 700_000:
        rl = int(min(300, rng.geometric(0.05))) if rng.random() < 0.9 else int(rng.integers(65, 300))
        l = lab(i % n_labels); i += 1
        for q in range(rl):
            reads.append(l)
            if rng.random() < 0.002: reads.append(np.zeros(0, np.uint32))       # an empty read inside the run
    # part 2: a sorted stretch over few labels (long runs), then a shuffled stretch over the same labels
    pick = np.sort(rng.integers(0, 2000, 400_000))
    reads += [lab(int(k)) for k in pick]
    pick = rng.integers(0, n_labels, 500_000)
    reads += [lab(int(k)) for k in pick]
    ids, off = _pack(reads)
    ob, *oc = _oracle_classes([(ids, off)])
    cut = 333_333
    kw = dict(expected_classes=expected) if expected else {}
    eq = sf.EquivalenceClassBuilder(device=gpu, **kw)
    eq.start()
    eq.add_batch(torch.from_numpy(ids[:off[cut]].view(np.int32)).to(gpu), torch.from_numpy(off[:cut + 1].view(np.int32)).to(gpu))
    eq.add_batch(torch.from_numpy(ids[off[cut]:].view(np.int32)).to(gpu),
                 torch.from_numpy((off[cut:] - off[cut]).astype(np.uint32).view(np.int32)).to(gpu))
    eq.finish()
    _assert_same_classes(eq, ob, *oc)


def test_builder_deferred_runs_keep_their_length(sf, gpu):
    """2.7 M distinct labels, each read three times in a row, into the smallest table (2^22 slots): the sub-batch after the
    scout brings more new classes than its regions hold (2400 of 4096 slots), so labels are deferred -- as runs of three -- and
    replayed after growth with the run length as their weight.  Known answer: every class counts exactly 3."""
    import torch
    n = 2_700_000
    g = torch.Generator(device=gpu); g.manual_seed(4)
    a = torch.randperm(n, generator=g, device=gpu, dtype=torch.int64)
    lab = torch.stack([a, (a * 5 + 1) % 977 + n], 1)                      # all-distinct sorted 2-id labels
    ids = lab.repeat_interleave(3, dim=0).reshape(-1).to(torch.int32)
    off = (torch.arange(3 * n + 1, device=gpu, dtype=torch.int64) * 2).to(torch.int32)
    eq = sf.EquivalenceClassBuilder(device=gpu, expected_classes=1000)
    eq.start(); eq.add_batch(ids, off); eq.finish()
    st = eq.stats()
    assert st["deferred_reads"] > 0 and st["table_grows"] >= 1, st
    v = eq.eqVec()
    assert eq.n_classes == n and eq.total_reads == 3 * n
    assert bool((v.counts == 3).all())
    first = v.ids.view(-1, 2)[:, 0].to(torch.int64)
    assert bool(torch.equal(torch.sort(first).values, torch.arange(n, device=gpu)))


def test_builder_host_batches_from_threads(sf, gpu):
    """the reference-side adaptor's call pattern: several mapper threads hand over
    ~1000-read HOST batches concurrently; the library accumulates them in pinned memory and builds
    2 M reads at a time.  Same classes as the oracle, whatever the interleaving."""
    import threading
    from sailfish_amd import synth
    _, ids, off = synth.workload(20_000, 200_000, 5_000_000, seed=3)
    ids = ids.numpy().view(np.uint32); off = off.numpy().view(np.uint32)
    ob, *oc = _oracle_classes([(ids, off)])
    eq = sf.EquivalenceClassBuilder(device=gpu)
    eq.start()
    T, R = 6, len(off) - 1
    def work(t):
        step = 997 + 13 * t                                  # ragged batch sizes, offsets with a non-zero base
        for r in range(R * t // T, R * (t + 1) // T, step):
            e = min(r + step, R * (t + 1) // T)
            eq.add_batch(ids, off[r:e + 1])
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    eq.finish()
    _assert_same_classes(eq, ob, *oc)


@pytest.mark.parametrize("chunk", ["250000", None])
def test_builder_large_host_batch_is_streamed_in_chunks(sf, gpu, monkeypatch, chunk):
    """ONE large host batch (the pinned-memory entry of SURVEY 8d): it goes through two device staging buffers in
    chunks, the copy of chunk k + 1 overlapping the build of chunk k.  chunk=250000 forces ~20 chunks with ragged
    boundaries and an offsets array that does not start at 0; None is the production chunk size (one chunk here)."""
    import torch
    from sailfish_amd import synth
    if chunk:
        monkeypatch.setenv("SFGPU_EQ_HOST_CHUNK", chunk)
    _, ids, off = synth.workload(20_000, 300_000, 5_000_000, seed=5)
    ids_np = ids.numpy().view(np.uint32); off_np = off.numpy().view(np.uint32)
    ob, *oc = _oracle_classes([(ids_np, off_np)])
    h_ids = torch.empty(ids.shape, dtype=torch.int32, pin_memory=True); h_ids.copy_(ids)
    h_off = torch.empty(off.shape, dtype=torch.int32, pin_memory=True); h_off.copy_(off)
    eq = sf.EquivalenceClassBuilder(device=gpu)
    eq.start(); eq.add_batch(h_ids, h_off); eq.finish()
    _assert_same_classes(eq, ob, *oc)
    # a slice of a larger array: offsets with a non-zero base, handed over in two calls
    eq.start()
    cut = 2_222_223
    eq.add_batch(h_ids, h_off[:cut + 1]); eq.add_batch(h_ids, h_off[cut:])
    eq.finish()
    _assert_same_classes(eq, ob, *oc)


@pytest.mark.parametrize("form", ["SFGPU_EQ_RING", "SFGPU_EQ_QUAD", "SFGPU_EQ_SHARED"])
def test_builder_ring_form_of_the_route_pass(sf, gpu, monkeypatch, form):
    _needs_variants()
    """SFGPU_EQ_RING=1: pass 1 writes its bins through LDS rings (whole 64-byte units from the front of a bin, long labels and
    labels that would have to wait for a ring slot directly at its back; eqclass_part.h) -- off by default, measured no faster;
    the classes must be the oracle's whichever form ran: the benchmark's law, a skewed stream with hot labels, runs of identical
    reads, and labels of up to 123 ids"""
    import torch
    from sailfish_amd import synth
    # (round 4: the same streams through the QUAD form -- single-granule labels written four to a 64-byte unit through per-region
    #  mailboxes and a ticket; measured slower than the direct form, off by default; round 6: the SHARED form -- the blocks of an XCD
    #  share one bin per region and reserve a step's granules with one atomic: whole units leave the L2, the pass is no faster)
    monkeypatch.setenv(form, "1")
    rng = np.random.default_rng(12)
    ref_len, ids, off = synth.workload(20000, 150_000, 2_500_000)
    ids_np, off_np = ids.numpy().view(np.uint32), off.numpy().view(np.uint32)
    lens = np.diff(off_np.astype(np.int64))
    # a skewed copy: 30 % of the reads carry one of 5 labels; and every label 1 .. 4 times in a row
    picks = np.arange(len(lens)); hot = rng.random(len(lens)) < 0.3; picks[hot] = rng.integers(0, 5, int(hot.sum())) * 1013
    picks = np.repeat(picks[: len(picks) // 2], rng.integers(1, 5, len(picks) // 2))[: len(lens)]
    starts = off_np[:-1].astype(np.int64)[picks]; l2 = lens[picks]
    off2 = np.zeros(len(picks) + 1, np.int64); np.cumsum(l2, out=off2[1:])
    ids2 = ids_np[np.repeat(starts, l2) + (np.arange(off2[-1]) - np.repeat(off2[:-1], l2))]
    # long labels: 1 .. 123 ids
    n_long = 100_000                                          # (>= 65 536 reads: the partitioned path)
    ll = rng.integers(1, 124, n_long); base = rng.integers(0, 500, n_long)
    off3 = np.zeros(n_long + 1, np.int64); np.cumsum(ll, out=off3[1:])
    ids3 = (np.repeat(base, ll) + 3 * (np.arange(off3[-1]) - np.repeat(off3[:-1], ll))).astype(np.uint32)
    for ii, oo in ((ids_np, off_np.astype(np.int64)), (ids2, off2), (ids3, off3)):
        ob, *oc = _oracle_classes([(ii, oo.astype(np.uint32))])
        eq = sf.EquivalenceClassBuilder(device=gpu)
        eq.start(); eq.add_batch(torch.from_numpy(ii.view(np.int32)).to(gpu), torch.from_numpy(oo.astype(np.uint32).view(np.int32)).to(gpu)); eq.finish()
        _assert_same_classes(eq, ob, *oc)


@pytest.mark.parametrize("sub_batch", ["65536", None])
def test_builder_growth_and_deferral(sf, gpu, monkeypatch, sub_batch):
    """more distinct classes than the table budget: deferred reads are replayed after growth.
    sub_batch=65536: the table grows between sub-batches; None: one 3.7 M-read sub-batch overflows the
    regions' LDS images in the partitioned pass, so the deferred-label path (copy out, grow, generic
    insert) runs"""
    if sub_batch:
        monkeypatch.setenv("SFGPU_EQ_SUBBATCH", sub_batch)
    rng = np.random.default_rng(2)
    n = 3_500_000                                   # > 2^21 - slack distinct labels
    a = rng.permutation(n).astype(np.uint32)
    ids = np.stack([a, a[::-1] % 7], 1).reshape(-1).astype(np.uint32)   # all-distinct 2-id labels
    off = (np.arange(n + 1, dtype=np.uint64) * 2).astype(np.uint32)
    dup = 200_000
    ids = np.concatenate([ids, ids[:2 * dup]]); off = np.concatenate([off, off[-1] + off[1:dup + 1]])
    eq = _gpu_classes(sf, gpu, [(ids, off)], expected_classes=1000)
    rp, ii, cc, hh = eq.eqVec().to_numpy()
    st = eq.stats()
    if sub_batch is None:
        assert st["deferred_reads"] > 0 and st["table_grows"] >= 1      # the overflow path really ran
    assert eq.n_classes == n and eq.total_reads == n + dup and int(cc.sum()) == n + dup
    lab = ii.reshape(-1, 2)
    assert np.array_equal(np.sort(lab[:, 0]), np.arange(n, dtype=np.uint32))
    two = lab[cc == 2][:, 0]
    assert len(two) == dup and set(two.tolist()) == set(a[:dup].tolist())
    np.testing.assert_array_equal(hh, O.xxh64_lists(ii, rp.astype(np.uint64)))
    first = lab[:, 0].astype(np.uint64)
    assert np.all(np.diff(first) >= 0)              # canonical order: first id ascending (then hash)


@pytest.mark.parametrize("n,dup,min_slots", [(9_000_000, 3_000_000, 1 << 24), (20_000_000, 4_000_000, 1 << 25)])
def test_builder_beyond_partition_limit(sf, gpu, n, dup, min_slots):
    """9 M / 20 M distinct labels: the table passes 16 M slots (4096 regions), the size up to which ONE route + insert pass
    covers every region.  Rounds 1-3 handed larger tables to the generic kernel; since round 4 they are built in groups of 4096
    regions (one route + insert pass over the sub-batch per group, eq_partitioned) -- counts and classes stay exact, and the
    partitioned kernels keep running (the table grows through 2, 4 and 8 groups on the way)"""
    import torch
    g = torch.Generator(device=gpu); g.manual_seed(1)
    a = torch.randperm(n, generator=g, device=gpu, dtype=torch.int64)
    ids = torch.stack([a, (a * 7 + 3) % 1000], 1).reshape(-1).to(torch.int32)
    ids = torch.cat([ids, ids[:2 * dup]])
    off = (torch.arange(n + dup + 1, device=gpu, dtype=torch.int64) * 2).to(torch.int32)
    eq = sf.EquivalenceClassBuilder(device=gpu)
    eq.start(); eq.add_batch(ids, off); eq.finish()
    v = eq.eqVec()
    st = eq.stats()
    assert eq.n_classes == n and eq.total_reads == n + dup and st["table_slots"] > min_slots
    cc = v.counts
    assert int(cc.sum()) == n + dup and int((cc == 2).sum()) == dup and int((cc == 1).sum()) == n - dup
    first = v.ids.view(-1, 2)[:, 0].to(torch.int64)
    assert bool((first[1:] > first[:-1]).all())              # canonical order; every first id is distinct here
    twice = first[cc == 2]
    assert bool(torch.equal(torch.sort(twice).values, torch.sort(a[:dup]).values))
    assert bool((sf.xxh64_labels(v.ids, v.rowptr, device=gpu) == v.hashes).all())


@pytest.mark.parametrize("shape", ["uniform", "hot", "sorted", "many_classes", "long_and_empty"])
def test_builder_pipelined_partition_passes(sf, gpu, monkeypatch, shape):
    _needs_variants()
    """Batches of >= 4 M reads take the PIPELINED partition passes (round 4: route(k + 1) next to insert(k) on a second stream,
    two sets of bins, class ids from a device-side counter, the host one sub-batch behind; eq_pipeline in eqclass.hip).  The
    classes must be the oracle's, and the serial form's (SFGPU_EQ_PIPE=0), on: the benchmark's law; a stream with hot labels
    (counted in the route pass; their table is rebuilt one sub-batch behind); a sorted stream (runs); far more classes than
    the table was sized for (growth and deferral drain the pipeline); labels of up to 200 ids among empty reads (the generic
    kernel's list).  A second batch goes into the same builder (the pipeline starts from a table that holds classes)."""
    import torch
    from sailfish_amd import synth
    rng = np.random.default_rng(31)
    R = 6_500_000
    M, P = (3_000_000, 6_000_000) if shape == "many_classes" else (60_000, 700_000)
    ref_len, ids, off = synth.workload(M, P, R)
    ids_np, off_np = ids.numpy().view(np.uint32), off.numpy().view(np.uint32).astype(np.int64)
    lens = np.diff(off_np)
    picks = np.arange(R)
    if shape == "hot":
        m = rng.random(R) < 0.4; picks[m] = rng.integers(0, 40, int(m.sum())) * 977
    elif shape == "sorted":
        key = ids_np[off_np[:-1]].astype(np.int64) * 256 + np.minimum(lens, 255)
        picks = np.argsort(key, kind="stable")
    elif shape == "long_and_empty":
        m = rng.random(R) < 0.02; picks[m] = -1
    if shape != "uniform":
        l2 = np.where(picks >= 0, lens[np.maximum(picks, 0)], 0)
        o2 = np.zeros(R + 1, np.int64); np.cumsum(l2, out=o2[1:])
        src = np.repeat(off_np[:-1][np.maximum(picks, 0)], l2) + (np.arange(o2[-1]) - np.repeat(o2[:-1], l2))
        ids_np, off_np = ids_np[src], o2
        if shape == "long_and_empty":             # ... and 30 000 labels of 124 .. 200 ids (beyond the partition stream's 123)
            n_long = 30_000
            ll = rng.integers(124, 201, n_long); base = rng.integers(0, 300, n_long)
            o3 = np.zeros(n_long + 1, np.int64); np.cumsum(ll, out=o3[1:])
            i3 = (np.repeat(base, ll) + 2 * (np.arange(o3[-1]) - np.repeat(o3[:-1], ll))).astype(np.uint32)
            ids_np = np.concatenate([ids_np, i3]); off_np = np.concatenate([off_np, off_np[-1] + o3[1:]])
    off32 = off_np.astype(np.uint32)
    cut = len(off32) // 2
    second = (ids_np[: off_np[cut]], off32[: cut + 1])                  # the first half again, as a second batch
    ob, *oc = _oracle_classes([(ids_np, off32), second])
    tables = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("SFGPU_EQ_PIPE", pipe)
        eq = sf.EquivalenceClassBuilder(device=gpu, expected_classes=(1000 if shape == "many_classes" else 0))
        eq.start()
        for ii, oo in ((ids_np, off32), second):
            eq.add_batch(torch.from_numpy(ii.view(np.int32)).to(gpu), torch.from_numpy(oo.view(np.int32)).to(gpu))
        eq.finish()
        _assert_same_classes(eq, ob, *oc)
        st = eq.stats()
        tables.append(st)
        if shape == "hot":
            assert st["hot_reads"] > R // 4
        if shape == "many_classes":
            assert st["table_grows"] >= 1
    assert tables[0]["insert_launches"] >= 2, tables


@pytest.mark.parametrize("n_reads", [40_000, 200_000])
def test_builder_long_labels_that_agree_in_every_sampled_id(sf, gpu, n_reads):
    """the bucket hash of a label of more than 8 ids looks at its length, its first 8 ids and three ids of its
    tail.  6 000 DISTINCT 40-id labels that agree in all of those share one home slot at every table size: more than a region
    holds (3 072).  The builder must notice that growth does not help, hash whole labels from there on, and end with the
    oracle's classes -- not double the table until an allocation fails.  40 000 reads: the generic kernel alone; 200 000: the
    partitioned passes defer the region's labels first."""
    rng = np.random.default_rng(5)
    n_lab, n = 6000, 40
    lab = np.tile(np.arange(0, 3 * n, 3, dtype=np.uint32), (n_lab, 1))
    fixed = {0, 1, 2, 3, 4, 5, 6, 7, n - 1, 8 + (n - 8) // 2, 8 + (n - 8) // 4}
    free = [k for k in range(8, n - 1) if k not in fixed]
    # distinct labels: the label number written into four free positions, in base 10 on top of the ascending pattern
    for d, k in enumerate(free[:4]):
        lab[:, k] += ((np.arange(n_lab) // 10 ** d) % 10).astype(np.uint32) * 1000
    assert len({tuple(r) for r in lab.tolist()}) == n_lab
    picks = np.concatenate([np.arange(n_lab), rng.integers(0, n_lab, n_reads - n_lab)])
    rng.shuffle(picks)
    ids = lab[picks].reshape(-1)
    # ... among ordinary short labels
    short = rng.integers(0, 50_000, (n_reads, 2)).astype(np.uint32)
    ids_all = np.concatenate([ids, short.reshape(-1)])
    off = np.concatenate([np.arange(n_reads + 1, dtype=np.uint64) * n, n_reads * n + np.arange(1, n_reads + 1, dtype=np.uint64) * 2]).astype(np.uint32)
    ob, *oc = _oracle_classes([(ids_all, off)])
    eq = _gpu_classes(sf, gpu, [(ids_all, off)])
    _assert_same_classes(eq, ob, *oc)
    st = eq.stats()
    assert st["deferred_reads"] > 0                       # the cluster really overflowed its region ...
    assert st["table_slots"] <= (1 << 24)                 # ... and the table did not keep doubling


# ------------------------------------------------------------------------------------ a6-a13
@pytest.fixture(scope="module")
def midsize(sf, gpu):
    """M=5000 transcripts, 20000-label pool, 400k reads: classes via the oracle, shared by the EM tests"""
    from sailfish_amd import synth
    ref_len, ids, off = synth.workload(5000, 20000, 400_000)
    ref_len = ref_len.numpy().view(np.uint32)
    ob, rp, ii, cc, hh = _oracle_classes([(ids.numpy().view(np.uint32), off.numpy().view(np.uint32))])
    eff = O.efflen_smoothed(ref_len, O.cf_gaussian())
    return dict(ref_len=ref_len, eff=eff, rowptr=rp, ids=ii, counts=cc, R=400_000)


def _gpu_em(sf, gpu, length, rp, ii, cc, num_mapped):
    import torch
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(gpu)
    return sf.EMProblem(torch.from_numpy(np.ascontiguousarray(length, dtype=np.float64)).to(gpu),
                        t(rp.astype(np.uint32), np.int32), t(ii.astype(np.uint32), np.int32),
                        t(cc.astype(np.uint64), np.int64), num_mapped)


def _rel(a, b):
    nz = b > 0
    assert np.array_equal(a > 0, nz), "support differs"
    return float(np.max(np.abs(a[nz] - b[nz]) / b[nz])) if nz.any() else 0.0


@pytest.mark.parametrize("vb", [False, True])
@pytest.mark.parametrize("n_iter", [1, 2, 50, 200])
def test_em_fixed_iterations(sf, gpu, midsize, vb, n_iter):
    """tol = 0 never converges: exactly max(n_iter, min_iter) iterations on both sides"""
    m = midsize
    rc, oa, om, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"], use_vbem=vb, tol=0.0,
                                    min_iter=0, max_iter=n_iter)
    p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    grc, st = p.optimize(use_vbem=vb, tol=0.0, min_iter=0, max_iter=n_iter, iters_per_launch=7)
    assert rc == 0 and grc == 0 and st["iters"] == ost["iters"] == n_iter
    assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT
    assert _rel(p.mass.cpu().numpy(), om) < TIGHT
    assert abs(st["max_rel_diff"] - ost["max_rel_diff"]) <= 1e-9 * abs(ost["max_rel_diff"])
    assert abs(st["alpha_sum"] - ost["alpha_sum"]) <= 1e-9 * ost["alpha_sum"]


@pytest.mark.parametrize("vb", [False, True])
def test_em_far_members_shared_by_a_neighbourhood(sf, gpu, vb):
    """members far from their class's window ("escapes": a pseudogene / paralog that every read of a gene also hits).  A tile
    adds them up in a small LDS accumulator keyed by transcript and hands each far transcript ONE sum; more distinct far
    transcripts than the accumulator holds go straight to alphaOut.  Classes here: 2..5 local members + (a) one far transcript
    shared by ~1000 neighbouring classes, (b) one of ~600 far transcripts per neighbourhood (more than the 128 slots),
    (c) far singletons; against the oracle after 1, 2 and 40 iterations."""
    rng = np.random.default_rng(31)
    M, C = 60_000, 120_000
    first = np.sort(rng.integers(0, 40_000, C))
    labels, counts = [], []
    for c in range(C):
        k = int(rng.integers(1, 5))
        loc = first[c] + np.sort(rng.choice(200, k, replace=False))
        kind = c % 4
        if kind == 0: far = [M - 1 - first[c] // 1000]                                  # (a) shared by the neighbourhood
        elif kind == 1: far = [45_000 + (first[c] // 1000) * 7 + int(rng.integers(0, 600))]  # (b) many distinct far targets per tile
        elif kind == 2: far = []
        else: far = [50_000 + int(rng.integers(0, 9_000)), 59_500 + int(rng.integers(0, 400))]
        lab = np.unique(np.concatenate([loc, far]).astype(np.uint32))
        labels.append(lab); counts.append(int(rng.integers(1, 50)))
    # far singletons: a class that is ONE far transcript only, in the middle of the sorted class list, cannot exist (a class is
    # sorted by its first id) -- but a singleton class can be the escape target of others; add a few plain singletons
    for t in (M - 1, M - 2, 45_003):
        labels.append(np.array([t], np.uint32)); counts.append(1000)
    key = sorted(range(len(labels)), key=lambda i: (int(labels[i][0]), len(labels[i]), labels[i].tobytes()))
    seen, L2, C2 = set(), [], []
    for i in key:                                               # distinct labels only (a class table has each label once)
        b = labels[i].tobytes()
        if b in seen: continue
        seen.add(b); L2.append(labels[i]); C2.append(counts[i])
    rp = np.zeros(len(L2) + 1, np.uint64); rp[1:] = np.cumsum([len(l) for l in L2])
    ii = np.concatenate(L2).astype(np.uint32); cc = np.asarray(C2, np.uint64)
    R = int(cc.sum())
    eff = np.maximum(rng.lognormal(7.0, 0.7, M), 50.0)
    p = _gpu_em(sf, gpu, eff, rp, ii, cc, R)
    for n_iter in (1, 2, 40):
        rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, R, use_vbem=vb, tol=0.0, min_iter=0, max_iter=n_iter)
        grc, st = p.optimize(use_vbem=vb, tol=0.0, min_iter=0, max_iter=n_iter, iters_per_launch=5)
        assert rc == 0 and grc == 0 and st["iters"] == ost["iters"] == n_iter
        assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT
        assert abs(st["alpha_sum"] - ost["alpha_sum"]) <= 1e-9 * ost["alpha_sum"]


def test_em_with_shuffled_transcript_ids_renumbers_its_plan(sf, gpu):
    """an index whose isoforms are not adjacent (accession order, a shuffled FASTA): every member of a class but the first is
    outside its tile's window.  The plan then orders the transcripts itself (label propagation over the classes) and windows
    its own order; x, alpha and everything the caller sees stay in the caller's order.  Same classes as the midsize problem
    with the transcript ids shuffled: EM / VBEM against the oracle after 1, 2 and 60 iterations and at convergence, the
    renumbering really ran (logger), bootstrap counts come back in the caller's class order."""
    from sailfish_amd import synth, _lib
    M = 30_000
    ref_len, ids, off = synth.workload(M, 120_000, 1_500_000)
    rng = np.random.default_rng(8)
    sigma = rng.permutation(M).astype(np.uint32)
    ids_np, off_np = ids.numpy().view(np.uint32), off.numpy().view(np.uint32)
    sh = sigma[ids_np]
    # members of a label sorted again (a label is a sorted id list)
    rr = np.repeat(np.arange(len(off_np) - 1), np.diff(off_np.astype(np.int64)))
    order = np.lexsort((sh, rr))
    sh = sh[order]
    ob, rp, ii, cc, hh = _oracle_classes([(sh, off_np)])
    eff = O.efflen_smoothed(ref_len.numpy().view(np.uint32), O.cf_gaussian())
    R = 1_500_000
    logs = []
    _lib.set_logger(lambda lvl, msg: logs.append(msg))
    try:
        p = _gpu_em(sf, gpu, eff, rp, ii, cc, R)
    finally:
        _lib.set_logger(None)
    assert any("renumbered by co-occurrence" in m for m in logs), logs
    for vb in (False, True):
        for n_iter in (1, 2, 60):
            rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, R, use_vbem=vb, tol=0.0, min_iter=0, max_iter=n_iter)
            grc, st = p.optimize(use_vbem=vb, tol=0.0, min_iter=0, max_iter=n_iter, iters_per_launch=9)
            assert rc == 0 and grc == 0 and st["iters"] == ost["iters"] == n_iter
            assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT and _rel(p.mass.cpu().numpy(), om) < TIGHT
        rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, R, use_vbem=vb)
        grc, st = p.optimize(use_vbem=vb)
        assert rc == 0 and grc == 0 and st["iters"] == ost["iters"]
        assert _rel(p.alpha.cpu().numpy(), oa) < 1e-7
    # resampled counts are reported per class of the CALLER's table: their mean over draws follows the caller's counts
    D = 60
    draws = np.stack([p.bootstrap_counts(5, d).cpu().numpy() for d in range(D)]).astype(np.float64)
    assert np.all(draws.sum(1) == R)
    r = float(np.corrcoef(draws.mean(0), cc.astype(np.float64))[0, 1])
    assert r > 0.97, r                                     # (a class order mix-up would give ~0)


def _far_member_problem(seed=31, M=60_000, C=120_000):
    """the class table of test_em_far_members_shared_by_a_neighbourhood: local members + far transcripts of three kinds"""
    rng = np.random.default_rng(seed)
    first = np.sort(rng.integers(0, 40_000, C))
    labels, counts = [], []
    for c in range(C):
        k = int(rng.integers(1, 5))
        loc = first[c] + np.sort(rng.choice(200, k, replace=False))
        kind = c % 4
        if kind == 0: far = [M - 1 - first[c] // 1000]
        elif kind == 1: far = [45_000 + (first[c] // 1000) * 7 + int(rng.integers(0, 600))]
        elif kind == 2: far = []
        else: far = [50_000 + int(rng.integers(0, 9_000)), 59_500 + int(rng.integers(0, 400))]
        labels.append(np.unique(np.concatenate([loc, far]).astype(np.uint32))); counts.append(int(rng.integers(1, 50)))
    key = sorted(range(len(labels)), key=lambda i: (int(labels[i][0]), len(labels[i]), labels[i].tobytes()))
    seen, L2, C2 = set(), [], []
    for i in key:
        b = labels[i].tobytes()
        if b in seen: continue
        seen.add(b); L2.append(labels[i]); C2.append(counts[i])
    rp = np.zeros(len(L2) + 1, np.uint64); rp[1:] = np.cumsum([len(l) for l in L2])
    ii = np.concatenate(L2).astype(np.uint32); cc = np.asarray(C2, np.uint64)
    eff = np.maximum(rng.lognormal(7.0, 0.7, M), 50.0)
    return eff, rp, ii, cc, int(cc.sum())


def _stacked_window_problem(M=3000, C=60_000, seed=5):
    """60 000 classes of ~12 members inside ONE band of 900 transcripts: ~20 tiles share a window, more than a tile's overlap
    table holds (6) -- such tiles find the other tiles' sums through the cover list"""
    rng = np.random.default_rng(seed)
    L, cnt, seen = [], [], set()
    while len(L) < C:
        lab = np.unique(rng.integers(100, 1000, int(rng.integers(8, 16)))).astype(np.uint32)
        b = lab.tobytes()
        if b in seen: continue
        seen.add(b); L.append(lab); cnt.append(int(rng.integers(1, 30)))
    key = sorted(range(len(L)), key=lambda i: (int(L[i][0]), len(L[i]), L[i].tobytes()))
    L = [L[i] for i in key]; cnt = [cnt[i] for i in key]
    rp = np.zeros(len(L) + 1, np.uint64); rp[1:] = np.cumsum([len(l) for l in L])
    eff = np.maximum(rng.lognormal(7.0, 0.7, M), 50.0)
    cc = np.asarray(cnt, np.uint64)
    return eff, rp, np.concatenate(L).astype(np.uint32), cc, int(cc.sum())


@pytest.mark.parametrize("vb", [False, True])
@pytest.mark.parametrize("shape", ["midsize", "far_members", "far_members_renumbered", "shuffled_ids", "stacked_windows"])
def test_em_fused_iteration_equals_the_two_kernel_loop(sf, gpu, midsize, monkeypatch, vb, shape):
    """round 4: inside optimize() an iteration is ONE kernel (the update of iteration it - 1 runs at the head of sweep it, the window
    sums go from tile to tile through slot-major arrays and the tiles' overlap tables).  Same stop iteration, same statistics and
    the same alpha (the additions are the same in the same order; the far members' atomics and VBEM's leaner psi / exp differ in the
    last bits: 1e-10) as the sweep + k_update loop (SFGPU_EM_FUSED=0) and the
    oracle, on: the midsize problem; a table with far members of every kind -- shared by a neighbourhood, more distinct ones than a
    tile's accumulator holds, transcripts that are ONLY far members -- in the caller's order (SFGPU_EM_NO_RENUMBER) and as the plan
    sees fit; the midsize table with its transcripts relabelled at random (the plan gives them an order of its own: the fused kernel's
    per-transcript arrays live in that order); and tables whose tiles overlap more than the tables hold (the midsize one too: those
    tiles go by the cover list)."""
    if shape == "midsize":
        m = midsize; eff, rp, ii, cc, R = m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"]
    elif shape == "far_members":
        monkeypatch.setenv("SFGPU_EM_NO_RENUMBER", "1")
        eff, rp, ii, cc, R = _far_member_problem()
    elif shape == "far_members_renumbered":                  # the plan may give the transcripts an order of its own
        eff, rp, ii, cc, R = _far_member_problem()
    elif shape == "shuffled_ids":                            # ... and certainly does here: the midsize table under a random relabelling
        m = midsize
        rng = np.random.default_rng(12)
        perm = rng.permutation(len(m["eff"])).astype(np.uint32)
        labels = [np.sort(perm[m["ids"][int(a):int(b)]]) for a, b in zip(m["rowptr"][:-1], m["rowptr"][1:])]
        key = sorted(range(len(labels)), key=lambda i: (int(labels[i][0]), len(labels[i]), labels[i].tobytes()))
        rp = np.zeros(len(labels) + 1, np.uint64); rp[1:] = np.cumsum([len(labels[i]) for i in key])
        ii = np.concatenate([labels[i] for i in key]).astype(np.uint32); cc = m["counts"][key]
        eff = np.empty_like(m["eff"]); eff[perm] = m["eff"]; R = m["R"]
    else:
        eff, rp, ii, cc, R = _stacked_window_problem()
    runs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SFGPU_EM_FUSED", fused)
        p = _gpu_em(sf, gpu, eff, rp, ii, cc, R)
        out = []
        for kw in (dict(tol=0.0, min_iter=0, max_iter=1), dict(tol=0.0, min_iter=0, max_iter=2), dict(tol=0.0, min_iter=0, max_iter=37),
                   dict(), dict(iters_per_launch=5), dict(check_mode=1)):
            grc, st = p.optimize(use_vbem=vb, **kw)
            assert grc == 0
            out.append((st, p.alpha.cpu().numpy().copy(), p.mass.cpu().numpy().copy()))
        runs[fused] = out
    expect_fused = True
    for (sf_, af, mf), (s0, a0, m0) in zip(runs["1"], runs["0"]):
        assert sf_["fused"] == expect_fused and not s0["fused"]
        assert sf_["iters"] == s0["iters"] and sf_["converged"] == s0["converged"] and sf_["n_active"] == s0["n_active"]
        assert _rel(af, a0) < 1e-10 and _rel(mf, m0) < 1e-10
        assert abs(sf_["max_rel_diff"] - s0["max_rel_diff"]) <= 1e-9 * abs(s0["max_rel_diff"])
        assert abs(sf_["alpha_sum"] - s0["alpha_sum"]) <= 1e-10 * s0["alpha_sum"]
    rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, R, use_vbem=vb)
    st, a, _ = runs["1"][3]
    assert rc == 0 and st["iters"] == ost["iters"] and st["converged"] == ost["converged"] and _rel(a, oa) < TIGHT


def test_em_cover_lists_without_a_sort_equal_the_sorted_ones(sf, gpu, midsize, monkeypatch):
    """round 4: when the tiles' first positions never decrease (every table in canonical order) the plan builds the cover lists --
    which tiles hold a transcript, in tile order -- by counting and ranking instead of sorting (window slot, transcript) pairs.  The
    lists are the same lists: SFGPU_EM_COVER_CHECK makes the plan build both and compare them word for word (sfgpu_em_create fails
    if they differ); and the two-kernel loop, whose fold walks them, gives the same alpha either way.  Shapes: the midsize table, the
    same with its transcripts relabelled at random (a plan with an order of its own), a table with far members of every kind, one
    whose tiles crowd one window."""
    m = midsize
    rng = np.random.default_rng(12)
    perm = rng.permutation(len(m["eff"])).astype(np.uint32)
    labels = [np.sort(perm[m["ids"][int(a):int(b)]]) for a, b in zip(m["rowptr"][:-1], m["rowptr"][1:])]
    key = sorted(range(len(labels)), key=lambda i: (int(labels[i][0]), len(labels[i]), labels[i].tobytes()))
    rp2 = np.zeros(len(labels) + 1, np.uint64); rp2[1:] = np.cumsum([len(labels[i]) for i in key])
    ii2 = np.concatenate([labels[i] for i in key]).astype(np.uint32); cc2 = m["counts"][key]
    eff2 = np.empty_like(m["eff"]); eff2[perm] = m["eff"]
    cases = [("midsize", (m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"]), True),
             ("shuffled", (eff2, rp2, ii2, cc2, m["R"]), False),
             ("far", _far_member_problem(), False), ("stacked", _stacked_window_problem(), False)]
    monkeypatch.setenv("SFGPU_EM_FUSED", "0")
    monkeypatch.setenv("SFGPU_EM_COVER_CHECK", "1")
    for name, (eff, rp, ii, cc, R), exact in cases:
        got = {}
        for sort in (False, True):
            if sort: monkeypatch.setenv("SFGPU_EM_COVER_SORT", "1")
            else: monkeypatch.delenv("SFGPU_EM_COVER_SORT", raising=False)
            p = _gpu_em(sf, gpu, eff, rp, ii, cc, R)
            grc, st = p.optimize(use_vbem=False, tol=0.0, min_iter=0, max_iter=25)
            assert grc == 0 and st["iters"] == 25
            got[sort] = p.alpha.cpu().numpy().copy()
        assert _rel(got[False], got[True]) < 1e-12, name


@pytest.mark.parametrize("vb", [False, True])
def test_em_to_convergence_matches_stop_iteration(sf, gpu, midsize, vb):
    m = midsize
    rc, oa, om, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"], use_vbem=vb)
    p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    for chunk in (0, 1, 13):                         # the stop iteration must not depend on the polling chunk
        grc, st = p.optimize(use_vbem=vb, iters_per_launch=chunk)
        assert grc == 0 and st["iters"] == ost["iters"] and st["converged"] == ost["converged"]
        assert st["n_active"] == ost["n_active"]
        a = p.alpha.cpu().numpy()
        assert _rel(a, oa) < REL_TOL and _rel(a, oa) < TIGHT
    # quant.sf columns
    import torch
    from sailfish_amd import _lib
    tp = torch.zeros_like(p.alpha)
    _lib.check(_lib.lib().sfgpu_tpm(_lib.ptr(p.alpha), _lib.ptr(torch.from_numpy(m["eff"]).to(gpu)), len(oa), float(m["R"]),
                                    _lib.ptr(tp), None))
    torch.cuda.synchronize()
    ot = O.tpm(oa, m["eff"], m["R"])
    assert _rel(tp.cpu().numpy(), ot) < TIGHT and abs(float(tp.sum()) - 1e6) < 1e-3


def test_em_survey_kat(sf, gpu):
    """the only reference-binary EM outputs we hold"""
    k = json.load(open(os.path.join(GOLD, "survey_kat.json")))
    for key in ("em_toy5", "em_toy7"):
        t = k[key]
        eff = np.array(t["ref_len"], float) - t["eff_len_minus"]
        rp = np.zeros(len(t["classes"]) + 1, np.uint32); rp[1:] = np.cumsum([len(c) for c in t["classes"]])
        ii = np.array([x for c in t["classes"] for x in c], np.uint32); cc = np.array(t["counts"], np.uint64)
        p = _gpu_em(sf, gpu, eff, rp, ii, cc, t["num_mapped"])
        rc, st = p.optimize()
        assert rc == 0
        if key == "em_toy5":
            assert st["iters"] == t["stop_iter"]
            np.testing.assert_allclose(p.alpha.cpu().numpy(), t["em_est_count"], rtol=1e-12)
            np.testing.assert_allclose(p.mass.cpu().numpy(), t["em_mass"], rtol=1e-12)
            rc, st = p.optimize(use_vbem=True)
            np.testing.assert_allclose(p.alpha.cpu().numpy(), t["vbem_est_count"], rtol=1e-11)
        else:
            np.testing.assert_allclose(p.alpha.cpu().numpy(), t["em_est_count_6dp"], atol=6e-7)


def test_em_error_paths_and_degenerate_inputs(sf, gpu):
    from sailfish_amd import _lib
    eff = np.array([10.0, 20.0, 0.5])
    p = _gpu_em(sf, gpu, eff, np.array([0], np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint64), 0)
    rc, st = p.optimize()
    assert rc == _lib.ERR_NO_ACTIVE                 # optimize() returns false: "no transcripts expressed"
    p = _gpu_em(sf, gpu, eff, np.array([0, 1], np.uint32), np.array([1], np.uint32), np.array([0], np.uint64), 0)
    rc, st = p.optimize()
    assert rc == _lib.ERR_ALPHA_SUM                 # "total alpha weight was too small"
    # singleton classes only + an effective length below 1 (clamped to 1, :738)
    rp = np.array([0, 1, 2, 4], np.uint32); ii = np.array([0, 2, 1, 2], np.uint32); cc = np.array([5, 7, 11], np.uint64)
    p = _gpu_em(sf, gpu, eff, rp, ii, cc, 23)
    rc, st = p.optimize()
    orc, oa, om, ost = O.em_optimize(eff, rp.astype(np.uint64), ii, cc, 23)
    assert rc == 0 and orc == 0 and st["iters"] == ost["iters"]
    assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT
    with pytest.raises(_lib.SfgpuError):            # counts must fit 32 bits on the device
        _gpu_em(sf, gpu, eff, rp, ii, np.array([5, 2 ** 32, 11], np.uint64), 23)


def test_em_piecewise_equals_optimize(sf, gpu, midsize):
    """the sweep/update pieces driven from the host (the multi-GPU control flow) stop at the same
    iteration with the same numbers as the on-device loop"""
    m = midsize
    p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    rc, st_ref = p.optimize(use_vbem=True)
    want = p.alpha.clone()
    p.begin(use_vbem=True, tol=0.01, min_iter=50, max_iter=10000)
    ao = p.alpha_out_view()
    assert ao.numel() == len(m["eff"]) and float(ao.sum()) == st_ref["n_active"]
    p.init()
    done = False
    while not done:
        for _ in range(10):
            p.sweep(); p.update()
        done, st = p.poll()
    rc, st = p.finish()
    assert rc == 0 and st["iters"] == st_ref["iters"]
    assert _rel(p.alpha.cpu().numpy(), want.cpu().numpy()) < 1e-12


def test_efflen_and_end_to_end_driver(sf, gpu, midsize, tmp_path):
    """the reference's call sequence (mainQuantify): start / addGroup / finish / FLD / optimize / quant.sf"""
    from sailfish_amd import synth
    m = midsize
    _, ids, off = synth.workload(5000, 20000, 400_000)
    names = [f"tx{i}" for i in range(len(m["ref_len"]))]
    for noeff in (False, True):
        sopt = sf.SailfishOpts(noEffectiveLengthCorrection=noeff)
        exp = sf.ReadExperiment(sf.Transcripts(names, m["ref_len"], device=gpu), sopt)
        eq = exp.equivalenceClassBuilder()
        eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); assert eq.finish()
        exp.setNumMappedFragments(eq.total_reads)
        sf.efflen.set_effective_lengths(exp, sopt)
        want_len = m["ref_len"].astype(np.float64) if noeff else m["eff"]
        np.testing.assert_array_equal(exp.transcripts().EffectiveLength.cpu().numpy(), want_len)   # bit exact
        assert sf.CollapsedEMOptimizer().optimize(exp, sopt, 0.01, 10000)
        rc, oa, om, ost = O.em_optimize(want_len, m["rowptr"], m["ids"], m["counts"], m["R"])
        assert _rel(exp.transcripts().estCount.cpu().numpy(), oa) < TIGHT
        out = tmp_path / ("q1" if noeff else "q0")
        sf.writer.write_abundances(str(out), exp, sopt)
        sf.writer.write_equiv_counts(str(out), exp, sopt)
        rows = open(out / "quant.sf").read().splitlines()
        assert rows[0] == "Name\tLength\tEffectiveLength\tTPM\tNumReads" and len(rows) == len(names) + 1
        ot = O.tpm(oa, want_len, m["R"])
        for i in (0, 17, 4999):
            f = rows[i + 1].split("\t")
            assert f[0] == names[i] and int(f[1]) == m["ref_len"][i]
            assert f[3] == "%g" % ot[i] or abs(float(f[3]) - ot[i]) <= 1e-5 * max(ot[i], 1e-300)
            assert f[4] == "%g" % oa[i] or abs(float(f[4]) - oa[i]) <= 1e-5 * max(oa[i], 1e-300)
        eqf = open(out / "aux" / "eq_classes.txt").read().splitlines()
        assert int(eqf[0]) == len(names) and int(eqf[1]) == eq.n_classes and len(eqf) == 2 + len(names) + eq.n_classes
    # empirical FLD branch (>= numFragSamples unique pairs)
    sopt = sf.SailfishOpts()
    exp = sf.ReadExperiment(sf.Transcripts(names, m["ref_len"], device=gpu), sopt)
    fl = O.fld_gaussian_counts(1000, 250, 60, 20000).astype(np.uint32)
    sf.efflen.set_effective_lengths(exp, sopt, fl_counts=fl, remaining_fl_ops=0)
    np.testing.assert_array_equal(exp.transcripts().EffectiveLength.cpu().numpy(), O.efflen_smoothed(m["ref_len"], O.cf_counts(fl)))
    # --unsmoothedFLD: the empirical pdf itself (computeEmpiricalEffectiveLengths), bit-exact incl. odd corners
    rng = np.random.default_rng(8)
    for fld in (fl, rng.integers(0, 50, 1000).astype(np.uint32), np.r_[np.zeros(300), 7, np.zeros(699)].astype(np.uint32),
                np.r_[5, np.zeros(999)].astype(np.uint32), np.array([0, 4], np.uint32)):
        sopt = sf.SailfishOpts(useUnsmoothedFLD=True, maxFragLen=len(fld))
        exp = sf.ReadExperiment(sf.Transcripts(names, m["ref_len"], device=gpu), sopt)
        sf.efflen.set_effective_lengths(exp, sopt, fl_counts=fld, remaining_fl_ops=0)
        np.testing.assert_array_equal(exp.transcripts().EffectiveLength.cpu().numpy(), O.efflen_empirical(fld, m["ref_len"]))


# -------------------------------------------------------------------------------- a15 / a17
@pytest.mark.parametrize("n_classes", [1, 2, 700, 2048, 2049, 40_000, 300_000])
def test_multinomial_tree_in_two_launches_equals_a_launch_per_level(sf, gpu, monkeypatch, n_classes):
    """the resample (include/MultinomialSampler.hpp:13-64 as doBootstrap uses it, :468) as one block for the tree's top levels + one
    block per subtree (round 6) against the form with a launch per level: same nodes, same Philox streams -- the same counts, bit for
    bit, for trees of one block, of exactly one subtree, and of many"""
    import torch
    rng = np.random.default_rng(n_classes)
    M = max(4, min(5000, n_classes + 3))
    ids = rng.integers(0, M, n_classes).astype(np.uint32)
    rowptr = np.arange(n_classes + 1, dtype=np.uint64)
    counts = rng.integers(1, 2000, n_classes).astype(np.uint64)
    counts[rng.integers(0, n_classes)] += 3_000_000                                  # one heavy class: BTPE in the upper levels
    p = _gpu_em(sf, gpu, np.full(M, 1000.0), rowptr, ids, counts, int(counts.sum()))
    for seed, draw in ((1, 0), (1, 1), (77, 5)):
        monkeypatch.setenv("SFGPU_MN_TREE", "levels")
        a = p.bootstrap_counts(seed, draw).cpu().numpy()
        monkeypatch.delenv("SFGPU_MN_TREE")
        b = p.bootstrap_counts(seed, draw).cpu().numpy()
        assert int(a.sum()) == int(counts.sum()) and np.array_equal(a, b)
    p.close()


def test_multinomial_resample_is_exact_in_distribution(sf, gpu, midsize):
    """sampCounts of doBootstrap (:468): every draw sums to N; per-class counts are Binomial(N, p_c)"""
    m = midsize
    p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    N = int(m["counts"].sum()); pc = m["counts"] / N
    D = 400
    draws = np.stack([p.bootstrap_counts(99, d).cpu().numpy() for d in range(D)])
    assert np.all(draws.sum(1) == N)
    assert not np.array_equal(draws[0], draws[1])
    assert np.array_equal(p.bootstrap_counts(99, 3).cpu().numpy(), draws[3])          # reproducible from (seed, draw)
    mean = draws.mean(0); sd = np.sqrt(N * pc * (1 - pc))
    z = (mean - N * pc) / (sd / np.sqrt(D))
    assert np.abs(z).max() < 6.0 and abs(z.mean()) < 0.2 and 0.85 < z.std() < 1.15
    big = np.argsort(-pc)[:20]
    v = draws[:, big].var(0, ddof=1) / (N * pc[big] * (1 - pc[big]))
    assert np.all((v > 0.7) & (v < 1.35))
    # the class counts of the handle are untouched afterwards
    rc, st = p.optimize()
    orc, oa, _, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    assert st["iters"] == ost["iters"] and _rel(p.alpha.cpu().numpy(), oa) < TIGHT


@pytest.mark.parametrize("vb", [False, True])
def test_bootstrap_distribution_vs_oracle(sf, gpu, vb):
    """gatherBootstraps: distributional parity (the reference seeds from random_device) -- replicate
    means and spreads against the oracle's restatement, and against the SURVEY 8c sample means"""
    k = json.load(open(os.path.join(GOLD, "survey_kat.json")))["em_toy7"]
    eff = np.array(k["ref_len"], float) - k["eff_len_minus"]
    rp = np.zeros(len(k["classes"]) + 1, np.uint32); rp[1:] = np.cumsum([len(c) for c in k["classes"]])
    ii = np.array([x for c in k["classes"] for x in c], np.uint32); cc = np.array(k["counts"], np.uint64)
    p = _gpu_em(sf, gpu, eff, rp, ii, cc, k["num_mapped"])
    B = 1500
    got = []
    rc, out, iters = p.bootstrap(B, seed=2024, callback=lambda a: got.append(a) or True, use_vbem=vb)
    assert rc == 0 and len(got) == B and np.array_equal(np.stack(got), out.cpu().numpy())
    g = out.cpu().numpy()
    orc, ob, oit = O.bootstrap(eff, rp.astype(np.uint64), ii, cc, B, use_vbem=vb, seed=77)
    assert orc == 0
    se = np.sqrt(g.var(0) / B + ob.var(0) / B) + 1e-9
    assert np.all(np.abs(g.mean(0) - ob.mean(0)) < 5 * se), (g.mean(0), ob.mean(0))
    assert np.all(np.abs(g.std(0) - ob.std(0)) < 0.15 * ob.std(0) + 0.05)
    assert abs(iters.mean() - oit.mean()) < 0.2 * oit.mean() + 2
    if not vb:
        assert np.all(np.abs(g.mean(0) - np.array(k["bootstrap_mean_200"])) < 6 * g.std(0) / np.sqrt(200) + 0.3)
        np.testing.assert_allclose(g.sum(1), k["num_mapped"], rtol=1e-9)      # EM conserves the resampled mass
    rc2, out2, _ = p.bootstrap(5, seed=2024, use_vbem=vb)
    assert np.array_equal(out2.cpu().numpy(), g[:5])                          # same seed, same replicates


def test_gather_bootstraps_driver(sf, gpu, midsize):
    from sailfish_amd import synth
    m = midsize
    _, ids, off = synth.workload(5000, 20000, 400_000)
    sopt = sf.SailfishOpts(numBootstraps=6)
    exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(5000)], m["ref_len"], device=gpu), sopt)
    eq = exp.equivalenceClassBuilder(); eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); eq.finish()
    exp.setNumMappedFragments(eq.total_reads); sf.efflen.set_effective_lengths(exp, sopt)
    opt = sf.CollapsedEMOptimizer(); assert opt.optimize(exp, sopt, 0.01, 10000)
    point = exp.transcripts().estCount.cpu().numpy()
    rows = []
    assert opt.gatherBootstraps(exp, sopt, lambda a: rows.append(a) or True, 0.01, 10000, seed=5)
    b = np.stack(rows)
    assert b.shape == (6, 5000) and np.all(b >= 0)
    np.testing.assert_allclose(b.sum(1), m["R"], rtol=1e-6)
    top = np.argsort(-point)[:50]
    assert np.all(np.abs(b[:, top].mean(0) - point[top]) < 0.2 * point[top] + 5)


def test_aux_writers(sf, gpu, midsize, tmp_path):
    """writeMeta + writeBootstrap (GZipWriter.cpp:94-192, 249-285): names.tsv.gz, fld.gz, meta_info.json and the raw
    binary bootstraps.gz that downstream tools read back"""
    import gzip
    from sailfish_amd import synth
    m = midsize
    _, ids, off = synth.workload(5000, 20000, 400_000)
    sopt = sf.SailfishOpts(numBootstraps=4)
    names = ["tx%d" % i for i in range(5000)]
    exp = sf.ReadExperiment(sf.Transcripts(names, m["ref_len"], device=gpu), sopt)
    eq = exp.equivalenceClassBuilder(); eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); eq.finish()
    exp.setNumMappedFragments(eq.total_reads); exp.setNumObservedFragments(eq.total_reads + 1000)
    sf.efflen.set_effective_lengths(exp, sopt)
    opt = sf.CollapsedEMOptimizer(); assert opt.optimize(exp, sopt, 0.01, 10000)
    out = str(tmp_path / "q")
    assert sf.writer.write_meta(out, exp, sopt, "Mon Jan  1 00:00:00 2024")
    w = sf.writer.BootstrapWriter(out, sopt)
    kept = []
    assert opt.gatherBootstraps(exp, sopt, lambda a: kept.append(a.copy()) or w(a), 0.01, 10000, seed=5)
    w.close()
    aux = os.path.join(out, "aux")
    assert gzip.open(os.path.join(aux, "bootstrap", "names.tsv.gz")).read().decode() == "\t".join(names) + "\n"
    fld = np.frombuffer(gzip.open(os.path.join(aux, "fld.gz")).read(), np.int32)
    np.testing.assert_array_equal(fld, O.fld_gaussian_counts())
    meta = json.load(open(os.path.join(aux, "meta_info.json")))
    assert list(meta) == ["sf_version", "samp_type", "frag_dist_length", "bias_correct", "num_bias_bins", "num_targets",
                          "num_bootstraps", "num_processed", "num_mapped", "percent_mapped", "call", "start_time"]
    assert meta["samp_type"] == "bootstrap" and meta["num_targets"] == 5000 and meta["num_bootstraps"] == 4
    assert meta["num_mapped"] == eq.total_reads and abs(meta["percent_mapped"] - 100.0 * eq.total_reads / (eq.total_reads + 1000)) < 1e-9
    raw = np.frombuffer(gzip.open(os.path.join(aux, "bootstrap", "bootstraps.gz")).read(), np.float64).reshape(4, 5000)
    np.testing.assert_array_equal(raw, np.stack(kept))
    for name, dt, n in (("expected_bias.gz", np.float64, 4096), ("observed_bias.gz", np.int32, 4096),     # :145-162
                        ("expected_gc.gz", np.float64, 101), ("observed_gc.gz", np.int32, 101)):
        v = np.frombuffer(gzip.open(os.path.join(aux, name)).read(), dt)
        assert v.shape == (n,) and np.all(v == 1)                          # no bias correction: the pseudo-counts


# ---------------------------------------------------------------------------------------- a16
def _toy7():
    k = json.load(open(os.path.join(GOLD, "survey_kat.json")))["em_toy7"]
    eff = np.array(k["ref_len"], float) - k["eff_len_minus"]
    rp = np.zeros(len(k["classes"]) + 1, np.uint32); rp[1:] = np.cumsum([len(c) for c in k["classes"]])
    ii = np.array([x for c in k["classes"] for x in c], np.uint32); cc = np.array(k["counts"], np.uint64)
    return k, eff, rp, ii, cc


def test_gibbs_toy_distribution(sf, gpu):
    """collapsed Gibbs: distributional parity with the oracle's restatement of sampleRound_, and with the
    sample means the reference produced (its posterior mean is NOT the EM point estimate)"""
    import torch
    k, eff, rp, ii, cc = _toy7()
    N = k["num_mapped"]
    orc, oa, om, _ = O.em_optimize(eff, rp.astype(np.uint64), ii, cc, N)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(gpu)
    args = (torch.from_numpy(eff).to(gpu), torch.from_numpy(om).to(gpu), t(rp, np.int32), t(ii, np.int32),
            t(cc.astype(np.uint64), np.int64), N)
    S = 4096
    got = []
    rc, out = sf.gibbs_sample(*args, S, n_chains=64, seed=11, callback=lambda v: got.append(v) or True)
    assert rc == 0 and len(got) == S
    g = out.cpu().numpy()
    assert np.array_equal(np.stack(got), g) and g.dtype == np.int32
    assert np.all(g.sum(1) == N) and np.all(g >= 0)                 # every read stays assigned to some transcript
    # members of no multi-transcript class cannot exceed what their classes hold
    assert np.all(g[:, 3] <= 45) and np.all(g[:, 0] >= 100) and np.all(g[:, 2] >= 10)
    orc, og = O.gibbs(eff, om, rp.astype(np.uint64), ii, cc, N, 6000, seed=3)
    og = og[500:]
    late = g[64 * 8:]                                               # drop each chain's first rounds
    assert np.all(np.abs(late.mean(0) - og.mean(0)) < 0.06 * N / 10), (late.mean(0), og.mean(0))
    assert np.all(np.abs(late.std(0) - og.std(0)) < 0.35 * og.std(0) + 1.0), (late.std(0), og.std(0))
    assert np.all(np.abs(late.mean(0) - np.array(k["gibbs_mean_200"])) < 12.0)
    rc, out2 = sf.gibbs_sample(*args, 128, n_chains=64, seed=11)
    assert np.array_equal(out2.cpu().numpy(), g[:128])              # reproducible from the seed


def test_gibbs_midsize_and_driver(sf, gpu, midsize):
    from sailfish_amd import synth
    m = midsize
    _, ids, off = synth.workload(5000, 20000, 400_000)
    sopt = sf.SailfishOpts(numGibbsSamples=96)
    exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(5000)], m["ref_len"], device=gpu), sopt)
    eq = exp.equivalenceClassBuilder(); eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); eq.finish()
    exp.setNumMappedFragments(eq.total_reads); sf.efflen.set_effective_lengths(exp, sopt)
    assert sf.CollapsedEMOptimizer().optimize(exp, sopt, 0.01, 10000)
    point = exp.transcripts().estCount.cpu().numpy()
    rows = []
    smp = sf.CollapsedGibbsSampler()
    assert smp.sample(exp, sopt, lambda v: rows.append(v) or True, sopt.numGibbsSamples, seed=9, n_chains=64)
    g = np.stack(rows)
    assert g.shape == (96, 5000) and np.all(g.sum(1) == m["R"]) and np.all(g >= 0)
    top = np.argsort(-point)[:40]
    assert np.all(np.abs(g[:, top].mean(0) - point[top]) < 0.25 * point[top] + 10)
    # transcripts in no class never receive reads
    rp, ii, cc, _ = eq.eqVec().to_numpy()
    absent = np.setdiff1d(np.arange(5000), ii)
    assert np.all(g[:, absent] == 0)
