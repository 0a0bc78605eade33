"""dev probe: do bootstrap replicates overlap when K handles run them on their own streams from K host threads?"""
import os, sys, time, threading
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
N = 24
for K in (1,):
    ps = [sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads) for _ in range(K)]
    for p in ps: p.bootstrap(1, seed=9)              # warm (plan for the sampler, graph)
    torch.cuda.synchronize(); t = time.perf_counter()
    def work(k):
        with torch.cuda.device(dev):
            ps[k].bootstrap(N // K, seed=100 + k)
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"K={K}: {N} replicates in {dt*1e3:.1f} ms -> {dt/N*1e3:.2f} ms per replicate", flush=True)
    for p in ps: p.close()
