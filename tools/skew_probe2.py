"""dev probe: class build on read streams with LOCALITY (reads of one label arrive together: sorted or clustered input)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
poff, pids = synth.label_pool(M, P, device=dev)
g = torch.Generator(device=dev); g.manual_seed(3)
for mode in ("uniform", "sorted", "runs64", "blocks1M"):
    a = torch.randint(0, P, (R,), generator=g, device=dev); b = torch.randint(0, P, (R,), generator=g, device=dev)
    pick = torch.minimum(a, b)
    if mode == "sorted": pick = torch.sort(pick).values
    if mode == "runs64": pick = pick[::64].repeat_interleave(64)[:R]                       # every label 64 times in a row
    if mode == "blocks1M": pick = torch.sort(pick.view(50, -1), dim=1).values.reshape(-1)   # 1 M-read chunks, each sorted
    k = (poff[1:] - poff[:-1])[pick]
    off = torch.zeros(R + 1, dtype=torch.int64, device=dev); torch.cumsum(k, 0, out=off[1:])
    tot = int(off[-1])
    rr = torch.repeat_interleave(torch.arange(R, device=dev), k, output_size=tot)
    ids = pids[poff[pick][rr] + (torch.arange(tot, device=dev) - off[:-1][rr])].to(torch.int32)
    off32 = off.to(torch.int32)
    del a, b, rr, k
    eq = sf.EquivalenceClassBuilder(device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        eq.start(); eq.add_batch(ids, off32); eq.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = eq.stats()
    print(f"{mode:9s}: {dt*1e3:8.2f} ms  classes {eq.n_classes}  hot {st['hot_reads']}  spilled {st['spilled_reads']}  deferred {st['deferred_reads']}  launches {st['insert_launches']}")
    v = eq.eqVec(); assert int(v.counts.sum()) == R
    del ids, off, off32, eq
