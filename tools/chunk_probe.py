"""dev probe: optimize() wall / loop time for several graph chunk sizes (iters_per_launch)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
import time
for chunk in (16, 32, 64, 128, 256):
    p = sf.EMProblem(ref_len.to(torch.float64), v.rowptr, v.ids, v.counts, eq.total_reads)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc, st = p.optimize(iters_per_launch=chunk)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    print("chunk", chunk, "optimize wall %.2f ms, loop %.2f ms, iters %d" % (dt, st["loop_ms"], st["iters"]))
    p.close()
