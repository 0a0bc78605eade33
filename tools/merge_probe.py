"""dev probe: cost of the multi-GPU class-table merge (weighted upsert of N ranks' tables) on one GPU"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
rp = v.rowptr.to(torch.int64) & 0xFFFFFFFF
lens = (rp[1:] - rp[:-1])
for N in (2, 8):
    ln = lens.repeat(N); ii = v.ids.repeat(N); cc = v.counts.repeat(N)
    m = sf.EquivalenceClassBuilder(device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        o = torch.zeros(ln.numel() + 1, dtype=torch.int64, device=dev); torch.cumsum(ln, 0, out=o[1:])
        m.start(); m.insertGroups(ii, o.to(torch.int32), cc); m.finish(); mv = m.eqVec()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    print(f"N={N}: merge of {ln.numel()} groups -> {m.n_classes} classes: {dt:.2f} ms", m.stats())
