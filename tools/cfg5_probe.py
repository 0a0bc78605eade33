"""dev probe: BASELINE config 5 -- 1000 Gibbs draws (and a few bootstrap replicates) over cfg3-scale classes"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
NCH = int(os.environ.get("GIBBS_CHAINS", "1024"))
M, P, R = 200_000, 4_000_000, 400_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
del ids, off
print("classes", eq.n_classes, "nnz", eq.nnz)
length = ref_len.to(torch.float64)
p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
rc, st = p.optimize(use_vbem=True); print("VBEM", st)
logs = []
from sailfish_amd import _lib
_lib.set_logger(lambda lvl, msg: logs.append(msg))
sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 8, n_chains=NCH, seed=5)    # the first call of a process maps the chain state (~2 s)
torch.cuda.synchronize(); t = time.perf_counter()
rc, g = sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 1000, n_chains=NCH, seed=1)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"gibbs: 1000 samples / {NCH} chains: {dt:.3f} s rc={rc} sums ok={bool((g.sum(1)==eq.total_reads).all())}", [m for m in logs if 'gibbs' in m])
torch.cuda.synchronize(); t = time.perf_counter()
rc, out, iters = p.bootstrap(10, seed=1, use_vbem=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"bootstrap: 10 draws {dt*1e3:.1f} ms -> {dt/10*1e3:.2f} ms/draw, iters mean {iters.mean():.0f}")
