"""dev probe: bootstrap throughput on cfg2 classes; set SFGPU_BS_LANES=1..4 to compare lane counts"""
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
for rep in range(5):
    p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
    torch.cuda.synchronize(); t = time.perf_counter()
    rc, out, iters = p.bootstrap(N, seed=100 + rep)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"rep {rep}: {N} replicates in {dt*1e3:.1f} ms -> {dt/N*1e3:.2f} ms per replicate (lanes: SFGPU_BS_LANES={os.environ.get('SFGPU_BS_LANES', 'default 3')}), rc {rc}, iters {iters.min()}..{iters.max()}", flush=True)
    p.close()
