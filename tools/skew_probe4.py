"""dev probe: class build against the LENGTH of the labels (mean ids per read)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
dev = torch.device("cuda:0")
M, P = 200_000, 500_000
g = torch.Generator(device=dev); g.manual_seed(11)
for mean, R in ((2, 40_000_000), (8, 20_000_000), (30, 8_000_000), (100, 3_000_000), (160, 2_000_000)):
    u = torch.rand(P, generator=g, device=dev).clamp_min(1e-12)
    k = (1 + torch.floor(torch.log(u) / torch.log(torch.tensor(1.0 - 1.0 / mean, device=dev)))).to(torch.int64).clamp(1, 200)
    poff = torch.zeros(P + 1, dtype=torch.int64, device=dev); torch.cumsum(k, 0, out=poff[1:])
    base = torch.randint(0, M - 1400, (P,), generator=g, device=dev)
    cls = torch.repeat_interleave(torch.arange(P, device=dev), k)
    pids = (base[cls] + 7 * (torch.arange(int(poff[-1]), device=dev) - poff[:-1][cls])).to(torch.int32)
    pick = torch.randint(0, P, (R,), generator=g, device=dev)
    kk = k[pick]
    off = torch.zeros(R + 1, dtype=torch.int64, device=dev); torch.cumsum(kk, 0, out=off[1:])
    tot = int(off[-1])
    rr = torch.repeat_interleave(torch.arange(R, device=dev), kk, output_size=tot)
    ids = pids[poff[pick][rr] + (torch.arange(tot, device=dev) - off[:-1][rr])]
    off32 = off.to(torch.int32)
    del rr, kk, cls
    eq = sf.EquivalenceClassBuilder(device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        eq.start(); eq.add_batch(ids, off32); eq.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    st = eq.stats()
    print(f"mean {tot / R:6.1f} ids per read, {R} reads ({tot * 4 / 1e9:.2f} GB of ids): {dt*1e3:8.2f} ms = {tot * 4 / dt / 1e9:7.1f} GB/s of ids, {R / dt / 1e9:.2f} G reads/s  classes {eq.n_classes} spilled {st['spilled_reads']} deferred {st['deferred_reads']}")
    del ids, off, off32, eq
