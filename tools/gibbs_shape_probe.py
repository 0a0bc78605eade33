"""dev probe: collapsed Gibbs sampler (100 draws) when a far transcript is shared by the classes of a neighbourhood"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 1_000_000, 30_000_000
ref_len = synth.transcript_lengths(M, device=dev).to(torch.float64)
logs = []
sf.set_logger(lambda lvl, msg: logs.append(msg)) if hasattr(sf, "set_logger") else None
for block in (0, -4096, -64, 256):
    poff, pids = synth.label_pool(M, P, device=dev)
    pids = pids.to(torch.int64)
    cls = torch.repeat_interleave(torch.arange(P, device=dev), poff[1:] - poff[:-1])
    if block:
        first = pids[poff[:-1]]
        dom = (first // block) * block if block > 0 else (M - 1 - first // (-block))
        key = torch.cat([cls * M + pids, torch.arange(P, device=dev) * M + dom])
    else:
        key = cls * M + pids
    key = torch.unique(key)
    pids2 = (key % M).to(torch.int32); cls2 = key // M
    poff2 = torch.zeros(P + 1, dtype=torch.int64, device=dev); poff2[1:] = torch.cumsum(torch.bincount(cls2, minlength=P), 0)
    ids, off = synth.reads_slice(poff2, pids2, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    p = sf.EMProblem(ref_len, v.rowptr, v.ids, v.counts, eq.total_reads)
    rc, st = p.optimize(use_vbem=True)
    del logs[:]
    torch.cuda.synchronize(); t = time.perf_counter()
    rc, g = sf.gibbs_sample(ref_len, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 100, n_chains=256, seed=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"shared per {block:6d}: EM {st['iters']} iters {st['loop_ms']:.1f} ms | gibbs 100 draws x 256 chains {dt*1e3:8.1f} ms rc={rc}", [m for m in logs if 'gibbs' in m][:1])
    del ids, off, eq, p, v, g
