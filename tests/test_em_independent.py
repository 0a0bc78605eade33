"""EM / VBEM against an independent restatement (tests/em_numpy_restatement.py: vectorised numpy from SURVEY.md
Appendix A / the reference source, with stored aux weights, scipy's digamma) and, on small problems, against the
same recurrences in 50-digit mpmath.  The C oracle and the HIP kernels share an author; this file does not share
their code, so a common misreading would surface here."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
import em_numpy_restatement as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    nz = b > 0
    assert np.array_equal(a > 0, nz), "support differs"
    return float(np.max(np.abs(a[nz] - b[nz]) / b[nz])) if nz.any() else 0.0


@pytest.fixture(scope="module")
def midsize_cpu(built):
    from sailfish_amd import synth
    ref_len, ids, off = synth.workload(5000, 20000, 400_000)
    ref_len = ref_len.numpy().view(np.uint32)
    b = O.EqBuilder(); b.add_batch(ids.numpy().view(np.uint32), off.numpy().view(np.uint32).astype(np.uint64))
    rp, ii, cc, hh = b.finish()
    eff = O.efflen_smoothed(ref_len, O.cf_gaussian())
    return dict(eff=eff, rp=rp, ii=ii, cc=cc, N=400_000, prob=R.Problem(eff, rp, ii, cc, 400_000))


def _toys():
    k = json.load(open(os.path.join(GOLD, "survey_kat.json")))
    out = []
    for name in ("em_toy5", "em_toy7"):
        t = k[name]
        eff = np.array(t["ref_len"], float) - t["eff_len_minus"]
        rp = np.zeros(len(t["classes"]) + 1, np.uint64); rp[1:] = np.cumsum([len(c) for c in t["classes"]])
        ii = np.array([x for c in t["classes"] for x in c], np.uint32)
        out.append((name, eff, rp, ii, np.array(t["counts"], np.uint64), t["num_mapped"], t))
    return out


def test_numpy_restatement_reproduces_the_reference_known_answers(built):
    """the restatement itself is pinned by the values the reference's own optimize() produced"""
    for name, eff, rp, ii, cc, N, t in _toys():
        p = R.Problem(eff, rp, ii, cc, N)
        a, mass, it, conv = p.optimize(vb=False)
        assert it == t.get("stop_iter", it)
        if "em_est_count" in t:
            np.testing.assert_allclose(a, t["em_est_count"], rtol=1e-13, atol=0)
            np.testing.assert_allclose(mass, t["em_mass"], rtol=1e-13, atol=0)
        else:
            np.testing.assert_allclose(a, t["em_est_count_6dp"], rtol=0, atol=5e-7)
        if "vbem_est_count" in t:
            a, mass, it, conv = p.optimize(vb=True)
            np.testing.assert_allclose(a, t["vbem_est_count"], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("vb", [False, True])
def test_oracle_and_numpy_agree_with_mpmath_on_small_problems(built, vb):
    rng = np.random.default_rng(11)
    cases = [(eff, rp, ii, cc, N) for _, eff, rp, ii, cc, N, _ in _toys()]
    M, C = 25, 60                                             # a random problem with tiny and huge lengths, repeated members
    lens = rng.integers(1, 7, C)
    rp = np.zeros(C + 1, np.uint64); rp[1:] = np.cumsum(lens)
    ii = np.concatenate([np.sort(rng.choice(M - 3, l, replace=False)) for l in lens]).astype(np.uint32)
    cc = rng.integers(1, 500, C).astype(np.uint64)
    eff = np.concatenate([rng.uniform(0.2, 3.0, 5), rng.uniform(50, 5000, M - 5)])
    cases.append((eff, rp, ii, cc, int(cc.sum())))
    for eff, rp, ii, cc, N in cases:
        for n_iter in (1, 7, 60):
            ref = R.optimize_mp(eff, rp.astype(np.int64), ii, cc, N, vb=vb, n_iter=n_iter)
            p = R.Problem(eff, rp, ii, cc, N)
            a = p.alpha0()
            for _ in range(n_iter):
                a = p.step(a, vb)
            rc, oa, om, st = O.em_optimize(eff, rp, ii, cc, N, use_vbem=vb, tol=0.0, min_iter=n_iter, max_iter=n_iter)
            cutoff = (0.01 + 1e-8) if vb else 1e-8            # the oracle returns the truncated vector
            keep = ref > cutoff * (1 + 1e-6)
            assert rc == 0
            np.testing.assert_allclose(a[keep], ref[keep], rtol=1e-11)
            np.testing.assert_allclose(oa[keep], ref[keep], rtol=1e-11)


@pytest.mark.parametrize("vb", [False, True])
@pytest.mark.parametrize("n_iter", [1, 2, 50, 200])
def test_oracle_matches_numpy_restatement_fixed_iterations(midsize_cpu, vb, n_iter):
    m = midsize_cpu
    p = m["prob"]
    a = p.alpha0()
    for _ in range(n_iter):
        a = p.step(a, vb)
    cutoff = (0.01 + 1e-8) if vb else 1e-8
    a = np.where(a <= cutoff, 0.0, a)
    rc, oa, om, st = O.em_optimize(m["eff"], m["rp"], m["ii"], m["cc"], m["N"], use_vbem=vb, tol=0.0, min_iter=n_iter, max_iter=n_iter)
    assert rc == 0 and st["iters"] == n_iter and st["n_active"] == p.n_active
    assert _rel(oa, a) < 1e-10


@pytest.mark.parametrize("vb", [False, True])
def test_oracle_matches_numpy_restatement_to_convergence(midsize_cpu, vb):
    m = midsize_cpu
    a, mass, it, conv = m["prob"].optimize(vb=vb)
    rc, oa, om, st = O.em_optimize(m["eff"], m["rp"], m["ii"], m["cc"], m["N"], use_vbem=vb)
    assert rc == 0 and conv and st["iters"] == it and bool(st["converged"])
    assert _rel(oa, a) < 1e-9 and _rel(om, mass) < 1e-9
    t_np, t_or = R.tpm(a, m["eff"], m["N"]), O.tpm(oa, m["eff"], m["N"])
    assert _rel(t_or, t_np) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("vb", [False, True])
@pytest.mark.parametrize("n_iter", [1, 2, 50, 200, 0])
def test_hip_matches_numpy_restatement(midsize_cpu, gpu, vb, n_iter):
    """the HIP path against the independent restatement (n_iter = 0: to convergence, same stop iteration)"""
    import torch
    import sailfish_amd as sf
    m = midsize_cpu
    p = m["prob"]
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(gpu)
    g = sf.EMProblem(torch.from_numpy(m["eff"]).to(gpu), t(m["rp"].astype(np.uint32), np.int32), t(m["ii"], np.int32),
                     t(m["cc"], np.int64), m["N"])
    try:
        if n_iter:
            a = p.alpha0()
            for _ in range(n_iter):
                a = p.step(a, vb)
            cutoff = (0.01 + 1e-8) if vb else 1e-8
            a = np.where(a <= cutoff, 0.0, a)
            rc, st = g.optimize(use_vbem=vb, tol=0.0, min_iter=n_iter, max_iter=n_iter)
            assert rc == 0 and st["iters"] == n_iter and st["n_active"] == p.n_active
        else:
            a, mass, it, conv = p.optimize(vb=vb)
            rc, st = g.optimize(use_vbem=vb)
            assert rc == 0 and st["iters"] == it and bool(st["converged"]) and conv
            assert _rel(g.mass.cpu().numpy(), mass) < 1e-9
        assert _rel(g.alpha.cpu().numpy(), a) < 1e-9
    finally:
        g.close()
