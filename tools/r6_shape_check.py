"""dev check (round 6): class-size shapes that push every record size of the persistent loop past what a thread requests ahead
(more than 2048 records of 4 bytes, more than 1024 of 8 / 16 bytes, overflow-heavy tiles) against the oracle after a few iterations"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from oracle import oracle as O
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
M = 300_000
worst = 0.0
for name, C, klo, khi in (("tiny classes", 1_800_000, 1, 4), ("fives", 900_000, 4, 7), ("tens", 500_000, 8, 13), ("thirties", 160_000, 25, 40), ("mixed", 900_000, 1, 30)):
    k = rng.integers(klo, khi, C).astype(np.int64)
    base = np.sort(rng.integers(0, M - 200, C))                       # canonical order: first id ascending
    rp = np.zeros(C + 1, np.int64); rp[1:] = np.cumsum(k)
    cls = np.repeat(np.arange(C), k); j = np.arange(rp[-1]) - rp[:-1][cls]
    ids = (base[cls] + j * rng.integers(1, 4)).astype(np.uint32)      # ascending, distinct inside a class
    cnt = rng.integers(1, 3000, C).astype(np.uint64)
    eff = np.exp(rng.normal(6.5, 1.0, M)).clip(1.0, 1e5)
    N = int(cnt.sum())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
    p = sf.EMProblem(torch.from_numpy(eff).to(dev), t(rp.astype(np.uint32), np.int32), t(ids, np.int32), t(cnt, np.int64), N)
    for vb in (False, True):
        rc, oa, om, ost = O.em_optimize(eff, rp.astype(np.uint64), ids, cnt, N, use_vbem=vb, tol=0.0, min_iter=4, max_iter=4)
        grc, st = p.optimize(use_vbem=vb, tol=0.0, min_iter=4, max_iter=4)
        ga = p.alpha.cpu().numpy(); nz = oa > 0
        assert rc == 0 and grc == 0 and np.array_equal(ga > 0, nz)
        rel = float(np.max(np.abs(ga[nz] - oa[nz]) / oa[nz])); worst = max(worst, rel)
        print(f"{name}: C {C} nnz {rp[-1]} {'VBEM' if vb else 'EM'} persistent {st.get('persistent')} rel {rel:.2e}", flush=True)
        assert rel < 1e-9
    p.close()
print("all ok, worst", worst)
