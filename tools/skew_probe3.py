"""dev probe: class build with FEW distinct labels (every class is hot) and with many medium-hot ones"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, R = 80_000, 50_000_000
g = torch.Generator(device=dev); g.manual_seed(9)
for P in (10, 500, 2000, 8000, 50_000):
    poff, pids = synth.label_pool(M, P, device=dev)
    pick = torch.randint(0, P, (R,), generator=g, device=dev)
    k = (poff[1:] - poff[:-1])[pick]
    off = torch.zeros(R + 1, dtype=torch.int64, device=dev); torch.cumsum(k, 0, out=off[1:])
    tot = int(off[-1])
    rr = torch.repeat_interleave(torch.arange(R, device=dev), k, output_size=tot)
    ids = pids[poff[pick][rr] + (torch.arange(tot, device=dev) - off[:-1][rr])].to(torch.int32)
    off32 = off.to(torch.int32)
    del rr, k
    eq = sf.EquivalenceClassBuilder(device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        eq.start(); eq.add_batch(ids, off32); eq.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = eq.stats()
    print(f"P={P:6d}: {dt*1e3:8.2f} ms  classes {eq.n_classes}  hot {st['hot_reads']}  spilled {st['spilled_reads']}  deferred {st['deferred_reads']}  launches {st['insert_launches']}")
    assert int(eq.eqVec().counts.sum()) == R
    del ids, off, off32, eq
