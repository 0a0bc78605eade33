"""dev stress: random workloads / batch splits / sub-batch sizes through the builder vs the oracle (bit-exact)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from oracle import oracle as O
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 60
n_ok = 0
while time.time() < t_end:
    M = int(rng.choice([50, 5000, 200_000, 3_000_000]))
    P = int(rng.choice([10, 1000, 100_000, 1_500_000]))
    R = int(rng.choice([70_000, 300_000, 1_200_000, 3_000_000, 5_000_000, 9_000_000]))
    kind = rng.integers(0, 3)
    # label pool with a random length law
    if kind == 0: lens = rng.geometric(0.3, P)
    elif kind == 1: lens = rng.integers(1, 9, P)
    else: lens = np.where(rng.random(P) < 0.01, rng.integers(100, 400, P), rng.integers(1, 20, P))
    lens = np.minimum(lens, M).astype(np.int64)
    poff = np.zeros(P + 1, np.int64); poff[1:] = np.cumsum(lens)
    pids = rng.integers(0, M, poff[-1]).astype(np.uint32)
    pick = np.minimum(rng.integers(0, P, R), rng.integers(0, P, R)) if rng.random() < 0.5 else rng.integers(0, P, R)
    # the order / skew of the stream (round 2: hot classes counted in the route pass, runs folded into one label with a count)
    shape = int(rng.integers(0, 6))
    if shape == 1:                                                 # a few hot labels hold a large part of the reads
        n_hot = int(rng.choice([1, 3, 40, 700, 3000])); frac = float(rng.choice([0.05, 0.3, 0.8]))
        hot = rng.integers(0, P, n_hot)
        m = rng.random(R) < frac
        pick[m] = hot[rng.integers(0, n_hot, int(m.sum()))]
    elif shape == 2:                                               # fully sorted by label
        pick.sort()
    elif shape == 3:                                               # runs of random length (1 .. 400)
        runs = rng.geometric(float(rng.choice([0.5, 0.05, 0.008])), R); runs = np.minimum(runs, 400)
        pick = np.repeat(pick[:R], runs)[:R]
    elif shape == 4:                                               # sorted chunks + hot labels
        c = int(rng.choice([1000, 64, 100_000]))
        for a0 in range(0, R, c): pick[a0:a0 + c].sort()
        m = rng.random(R) < 0.2; pick[m] = pick[0]
    rl = lens[pick].copy()
    rl[rng.random(R) < 0.01] = 0                                  # empty reads
    off = np.zeros(R + 1, np.int64); off[1:] = np.cumsum(rl)
    if off[-1] >= 2 ** 31: continue
    ids = np.empty(off[-1], np.uint32)
    rr = np.repeat(np.arange(R), rl); j = np.arange(off[-1]) - off[:-1][rr]
    ids[:] = pids[poff[pick][rr] + j]
    off32 = off.astype(np.uint32)
    ob = O.EqBuilder(); ob.add_batch(ids, off.astype(np.uint64)); orp, oi, oc, oh = ob.finish()
    sb = rng.choice([None, "65536", "262144", "1048576"])
    if sb: os.environ["SFGPU_EQ_SUBBATCH"] = sb
    else: os.environ.pop("SFGPU_EQ_SUBBATCH", None)
    pipe = rng.choice(["1", "1", "0"])                             # batches of >= 4 M reads: pipelined partition passes (round 4) or the serial form
    os.environ["SFGPU_EQ_PIPE"] = pipe
    # tables of <= 1024 regions: the quad form of pass 1 (round 4), the shared form (round 6: XCD-shared bins) or the direct form (STRESS_FORM fixes it)
    form = os.environ.get("STRESS_FORM") or rng.choice(["quad", "shared", "direct"])
    quad = "1" if form == "quad" else "0"
    os.environ["SFGPU_EQ_QUAD"] = quad
    os.environ["SFGPU_EQ_SHARED"] = "1" if form == "shared" else "0"
    eq = sf.EquivalenceClassBuilder(device=dev, expected_classes=int(rng.choice([0, 1000, 5_000_000])))
    eq.start()
    ncut = int(rng.integers(0, 4)); cuts = sorted(set([0, R] + rng.integers(0, R, ncut).tolist()))
    host = rng.random() < 0.3
    for a, b in zip(cuts[:-1], cuts[1:]):
        bi, bo = ids[off[a]:off[b]], (off[a:b + 1] - off[a]).astype(np.uint32)
        if host: eq.add_batch(bi if bi.size else np.zeros(1, np.uint32), bo)
        else: eq.add_batch(torch.from_numpy(bi.view(np.int32).copy()).to(dev), torch.from_numpy(bo.view(np.int32)).to(dev))
    eq.finish()
    rp, ii, cc, hh = eq.eqVec().to_numpy()
    ok = (eq.n_classes == ob.n_classes and np.array_equal(rp, orp.astype(np.uint32)) and np.array_equal(ii, oi)
          and np.array_equal(cc, oc) and np.array_equal(hh, oh))
    print(f"M={M} P={P} R={R} kind={kind} shape={shape} sb={sb} pipe={pipe} form={form} host={host} cuts={len(cuts)-1}: classes {eq.n_classes} {'ok' if ok else 'MISMATCH'} {eq.stats()}", flush=True)
    if not ok: sys.exit(1)
    n_ok += 1
print("all ok:", n_ok)
