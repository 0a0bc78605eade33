"""dev probe: bootstrap / Gibbs throughput on cfg2-sized classes"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
length = ref_len.to(torch.float64)
p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
rc, st = p.optimize(); print("EM", st)
torch.cuda.synchronize(); t = time.perf_counter()
rc, out, iters = p.bootstrap(20, seed=1)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"bootstrap: 20 draws {dt*1e3:.1f} ms -> {dt/20*1e3:.2f} ms/draw, iters mean {iters.mean():.0f}")
for nch, ns in ((64, 64), (1024, 1024)):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc, g = sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, ns, n_chains=nch, seed=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"gibbs: {ns} samples / {nch} chains: {dt:.3f} s  rc={rc} sum ok={bool((g.sum(1)==eq.total_reads).all())}")
