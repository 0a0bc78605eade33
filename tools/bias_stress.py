"""dev: randomised parity of sfgpu_bias_update against the oracle's updateEffectiveLengths restatement.
usage: python tools/bias_stress.py [seed [seconds]]   (transcript counts / lengths / FLD shapes / speed sampling / abundances drawn at random)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sailfish_amd as sf
from oracle import oracle as O
from test_bias import make_txome
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_ok = 0; worst = 0.0
while time.time() < t_end:
    M = int(rng.choice([1, 2, 7, 60, 300]))
    hi = int(rng.choice([40, 300, 3000, 9000]))
    lens = rng.integers(1, hi, M)
    if rng.random() < 0.3:
        lens[rng.integers(0, M)] = int(rng.choice([6, 7, 8, 13, 64, 65, 2048, 2049, 4097]))
    seq, off, lens = make_txome(rng, lens, alphabet=b"ACGTacgtUu")
    n = int(rng.choice([200, 1000, 3000]))
    mean = float(rng.uniform(20, n * 0.7)); sd = float(rng.uniform(3, max(4, n * 0.2)))
    x = np.arange(n)
    fl = np.round(float(rng.choice([50, 1e4, 1e6])) * np.exp(-0.5 * ((x - mean) / sd) ** 2)).astype(np.uint32)
    if fl.sum() == 0:
        continue
    mode = str(rng.choice(["seq", "gc"]))
    samp = int(rng.choice([1, 1, 2, 5, 17, 64, 100]))
    txp_eff = np.maximum(lens - rng.uniform(0, mean * 1.5), 1.0) if rng.random() < 0.8 else lens.astype(np.float64) + rng.integers(-3, 3, M)
    alphas = rng.random(M) * float(rng.choice([1e-9, 1.0, 1e4])); alphas[rng.random(M) < 0.2] = 0.0
    eff_in = np.maximum(txp_eff * (0.8 + 0.4 * rng.random(M)), 1.0)
    rb = rng.integers(1, 1000, 4096).astype(np.uint32); og = rng.integers(1, 5000, 101).astype(np.uint32)
    nf, nr = int(rng.integers(0, 1000)), int(rng.integers(1, 1000))
    step = int(rng.choice([1, 1, 1, 2, 5, 16]))                       # --gcSizeSamp: the interpolated counts
    kw = dict(num_fwd=nf, num_rc=nr, seq_bias=mode == "seq", gc_bias=mode == "gc", gc_speed_samp=samp, gc_size_samp=step)
    bm = O.make_bias_model(seq, off, lens, txp_eff, fl, rb, og, **kw)
    rc, out, es, eg, nc = O.update_efflens(bm, eff_in, alphas)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    try:
        model = sf.bias.BiasModel(t(np.frombuffer(seq, np.uint8), np.uint8), t(off, np.int64), t(lens.astype(np.uint32).view(np.int32), np.int32),
                                  t(txp_eff, np.float64), fl, rb, og, **kw)
    except sf._lib.SfgpuError as e:
        assert rc == -1 or e.code == sf._lib.ERR_RANGE, (rc, str(e))      # fld_low == 0 (the reference divides by zero) / 0.995 quantile too large
        continue
    assert rc == 0, rc
    got, st = model.update(t(eff_in, np.float64), t(alphas, np.float64))
    g = got.cpu().numpy()
    ges, geg = model.expected()
    desc = f"M={M} hi={hi} n={n} mean={mean:.0f} sd={sd:.0f} {mode} samp={samp} step={step} fld=[{st['fld_low']},{st['fld_high']}] corrected={nc}"
    assert st["n_corrected"] == nc, (desc, st)
    for a, b, what in ((g, out, "lengths"), (ges, es, "expected seq"), (geg, eg, "expected gc")):
        rel = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
        worst = max(worst, rel)
        assert rel < 1e-9, (desc, what, rel)
    model.close(); n_ok += 1
print(f"all ok: {n_ok} cases, worst rel {worst:.3g} (seed {seed})")
