"""dev probe: class build on a SKEWED read stream (a few labels hold a large part of the reads, as in real RNA-seq)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
poff, pids = synth.label_pool(M, P, device=dev)
g = torch.Generator(device=dev); g.manual_seed(3)
for hot_frac, n_hot in ((0.0, 0), (0.1, 1), (0.3, 10), (0.5, 100)):
    a = torch.randint(0, P, (R,), generator=g, device=dev); b = torch.randint(0, P, (R,), generator=g, device=dev)
    pick = torch.minimum(a, b)
    if n_hot:
        u = torch.rand(R, generator=g, device=dev)
        hot = torch.randint(0, n_hot, (R,), generator=g, device=dev) * 7919 % P
        pick = torch.where(u < hot_frac, hot, pick)
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
    print(f"hot {hot_frac:.1f} over {n_hot:3d} labels: {dt*1e3:8.2f} ms  classes {eq.n_classes}  deferred {st['deferred_reads']}  stats {dict((k, v) for k, v in st.items() if 'generic' in k or 'long' in k or 'launch' in k)}")
    v = eq.eqVec(); assert int(v.counts.sum()) == R
    del ids, off, off32, eq
