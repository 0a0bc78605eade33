"""GPU tests (-m gpu) of the PERSISTENT EM loop (csrc/em_persist.h, round 5): the whole loop of
src/CollapsedEMOptimizer.cpp:818-861 as one launch, window sums handed from tile to tile as tagged granules.  Against the
oracle (same stop iteration, <= 1e-9) and against the other two loops of the library on the same handle."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


@pytest.fixture(scope="module")
def sf(gpu):
    import sailfish_amd
    return sailfish_amd


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


@pytest.fixture(scope="module")
def local_table(sf, gpu):
    """60 000 transcripts, 200 000-label pool, 2 M reads of the benchmark's law: a few hundred tiles whose windows overlap two or
    three neighbours each -- the shape the persistent loop is made for (classes from the GPU builder: its parity is
    tests/test_gpu_parity.py's business)"""
    from sailfish_amd import synth
    M, R = 60_000, 2_000_000
    ref_len, ids, off = synth.workload(M, 200_000, R)
    eq = sf.EquivalenceClassBuilder(device=gpu); eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); eq.finish()
    rp, ii, cc, hh = eq.eqVec().to_numpy()
    eff = O.efflen_smoothed(ref_len.numpy().view(np.uint32), O.cf_gaussian())
    return dict(eff=eff, rowptr=rp.astype(np.uint64), ids=ii, counts=cc, R=R)


def _far_table_with_homes(seed=3, M=60_000, C=150_000):
    """local members + far members whose targets all have windows of their own: (a) one far transcript shared by the ~1000 classes of
    a neighbourhood, half the transcriptome away; (b) one of 40 far transcripts per neighbourhood; (c) far members in
    singleton-free classes next to plain singletons of the targets"""
    rng = np.random.default_rng(seed)
    first = np.sort(rng.integers(0, M - 300, C))
    labels, counts = [], []
    for c in range(C):
        k = int(rng.integers(1, 5))
        loc = first[c] + np.sort(rng.choice(200, k, replace=False))
        kind = c % 4
        nb = int(first[c]) // 1000
        if kind == 0: far = [(nb * 1000 + 30_000) % (M - 300)]
        elif kind == 1: far = [(nb * 1000 + 20_000 + 5 * int(rng.integers(0, 40))) % (M - 300)]
        else: far = []
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
    return dict(eff=eff, rowptr=rp, ids=ii, counts=cc, R=int(cc.sum()))


@pytest.mark.parametrize("vb", [False, True])
@pytest.mark.parametrize("shape", ["local", "far_with_homes"])
def test_persistent_loop_equals_oracle_and_the_other_loops(sf, gpu, local_table, monkeypatch, vb, shape):
    """fixed iteration counts (tol = 0), the loop to convergence (default bounds), the bootstrap's gate (check_mode = 1): the
    persistent launch stops at the oracle's iteration with the oracle's alpha (<= 1e-9) and statistics, and agrees with one
    kernel per iteration and with sweep + k_update to 1e-10"""
    m = local_table if shape == "local" else _far_table_with_homes()
    if shape == "far_with_homes": monkeypatch.setenv("SFGPU_EM_NO_RENUMBER", "1")
    eff, rp, ii, cc, R = m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"]
    cases = (dict(tol=0.0, min_iter=0, max_iter=1), dict(tol=0.0, min_iter=0, max_iter=2), dict(tol=0.0, min_iter=0, max_iter=37), dict(), dict(check_mode=1))
    runs = {}
    for mode, env in (("persist", dict(SFGPU_EM_FUSED="1", SFGPU_EM_PERSIST="1")), ("fused", dict(SFGPU_EM_FUSED="1", SFGPU_EM_PERSIST="0")),
                      ("two", dict(SFGPU_EM_FUSED="0"))):
        for k in ("SFGPU_EM_FUSED", "SFGPU_EM_PERSIST"): monkeypatch.delenv(k, raising=False)
        for k, v in env.items(): monkeypatch.setenv(k, v)
        p = _gpu_em(sf, gpu, eff, rp, ii, cc, R)
        out = []
        for kw in cases:
            grc, st = p.optimize(use_vbem=vb, **kw)
            assert grc == 0
            out.append((st, p.alpha.cpu().numpy().copy(), p.mass.cpu().numpy().copy()))
        runs[mode] = out
    for (sp, ap, mp), (sf_, af, mf), (s0, a0, m0) in zip(runs["persist"], runs["fused"], runs["two"]):
        assert sp["persistent"] and sp["fused"] and not sf_["persistent"] and sf_["fused"] and not s0["fused"]
        assert sp["iters"] == s0["iters"] == sf_["iters"] and sp["converged"] == s0["converged"] and sp["n_active"] == s0["n_active"]
        assert _rel(ap, a0) < 1e-10 and _rel(mp, m0) < 1e-10 and _rel(ap, af) < 1e-10
        assert abs(sp["max_rel_diff"] - s0["max_rel_diff"]) <= 1e-9 * abs(s0["max_rel_diff"])
        assert abs(sp["alpha_sum"] - s0["alpha_sum"]) <= 1e-10 * s0["alpha_sum"]
    for i, kw in enumerate(cases[:4]):
        rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, R, use_vbem=vb, **kw)
        st, a, _ = runs["persist"][i]
        assert rc == 0 and st["iters"] == ost["iters"] and st["converged"] == ost["converged"] and _rel(a, oa) < TIGHT
        assert abs(st["max_rel_diff"] - ost["max_rel_diff"]) <= 1e-9 * abs(ost["max_rel_diff"])


def test_persistent_loop_gives_up_and_the_run_is_repeated(sf, gpu, local_table, monkeypatch):
    """a tile that waits in vain (its neighbours never became resident: the device is shared with another process' kernels) raises the
    abort word, every tile leaves, and optimize() repeats the run with one kernel per iteration: same answer, `persistent` off, and
    the handle stays off it.  SFGPU_EM_PERSIST=3 makes tile 0 give up in step 2."""
    from sailfish_amd import _lib
    m = local_table
    rc, oa, om, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"], use_vbem=True)
    monkeypatch.setenv("SFGPU_EM_PERSIST", "3")
    logs = []
    _lib.set_logger(lambda lvl, msg: logs.append(msg))
    try:
        p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
        grc, st = p.optimize(use_vbem=True)
    finally:
        _lib.set_logger(None)
    assert grc == 0 and rc == 0 and not st["persistent"] and st["iters"] == ost["iters"]
    assert any("gave up" in x for x in logs), logs
    assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT
    monkeypatch.setenv("SFGPU_EM_PERSIST", "1")
    grc, st = p.optimize(use_vbem=True)                    # (the handle does not try again)
    assert grc == 0 and not st["persistent"] and st["iters"] == ost["iters"]


def test_back_to_back_persistent_runs_with_other_counts_share_nothing(sf, gpu, local_table):
    """two persistent launches behind each other on handles that recycle one exchange buffer, with DIFFERENT counts and the same
    iteration counts (fixed 7 / 7, then to convergence): a granule of the earlier launch with the same step number must never
    validate in the later one -- the tags carry a per-launch epoch (round 6), the buffer is zeroed before every launch and lives in
    uncached memory.  Each run against the oracle on its own counts."""
    m = local_table
    rng = np.random.default_rng(5)
    counts2 = (m["counts"].astype(np.int64) * rng.integers(1, 9, len(m["counts"]))).astype(np.uint64)
    R2 = int(counts2.sum())
    for kw in (dict(tol=0.0, min_iter=7, max_iter=7), dict()):
        for vb in (False, True):
            for cc, R in ((m["counts"], m["R"]), (counts2, R2), (m["counts"], m["R"])):
                p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], cc, R)           # (a new handle: the pool hands it the buffer the last one returned)
                grc, st = p.optimize(use_vbem=vb, **kw)
                rc, oa, om, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], cc, R, use_vbem=vb, **kw)
                assert grc == 0 and rc == 0 and st["persistent"] and st["iters"] == ost["iters"], (kw, vb, st, ost)
                assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT
                grc, st2 = p.optimize(use_vbem=vb, **kw)                                # ... and the same handle again
                assert grc == 0 and st2["iters"] == ost["iters"] and _rel(p.alpha.cpu().numpy(), oa) < TIGHT
                p.close()


def test_bootstrap_on_one_lane_runs_the_persistent_loop_on_the_resampled_counts(sf, gpu, local_table, monkeypatch):
    """a replicate = the class counts resampled + the EM loop over them.  Several lanes keep one kernel per iteration (notes 5);
    ONE lane runs the persistent loop, whose count words are a copy with a flag bit: made again before every launch
    (k_persist_init), or the replicates would all be the observed counts'.  Draw b is the same whichever lane makes it."""
    m = local_table
    p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    monkeypatch.setenv("SFGPU_BS_LANES", "1")
    rc1, out1, it1 = p.bootstrap(3, seed=5)
    monkeypatch.setenv("SFGPU_BS_LANES", "3")
    rc3, out3, it3 = p.bootstrap(3, seed=5)
    assert rc1 == 0 and rc3 == 0 and np.array_equal(it1, it3)
    a1, a3 = out1.cpu().numpy(), out3.cpu().numpy()
    for b in range(3):
        assert _rel(a1[b], a3[b]) < TIGHT
    assert float(np.abs(a1[0] - a1[1]).max()) > 1.0          # (the replicates differ: the counts were resampled)
    grc, st = p.optimize()                                    # the observed counts are back, and the loop reads them
    rc, oa, om, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
    assert grc == 0 and st["persistent"] and st["iters"] == ost["iters"] and _rel(p.alpha.cpu().numpy(), oa) < TIGHT
    p.close()


def test_a_class_of_2_to_the_30_reads_runs_persistent(sf, gpu, local_table):
    """the persistent loop's count words hold 31 bits of count (round 5 kept a flag in bit 30 and sent such a plan to one kernel per
    iteration; the records of round 6 need no flag)"""
    m = local_table
    cc = m["counts"].copy(); cc[len(cc) // 2] = (1 << 30) + 12345
    R = int(cc.sum())
    rc, oa, om, ost = O.em_optimize(m["eff"], m["rowptr"], m["ids"], cc, R, max_iter=20, min_iter=20, tol=0.0)
    p = _gpu_em(sf, gpu, m["eff"], m["rowptr"], m["ids"], cc, R)
    grc, st = p.optimize(max_iter=20, min_iter=20, tol=0.0)
    assert grc == 0 and rc == 0 and st["persistent"] and st["iters"] == ost["iters"]
    assert _rel(p.alpha.cpu().numpy(), oa) < TIGHT
    p.close()
