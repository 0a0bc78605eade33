"""dev probe: time the EM sweep kernel on a synthetic problem (not part of the product)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
if len(sys.argv) > 1 and sys.argv[1] == "cfg3":
    M, P, R = 200_000, 4_000_000, 100_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
print("classes", eq.n_classes, "nnz", eq.nnz, eq.stats())
length = ref_len.to(torch.float64)
p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
for vb in (False, True):
    ms = p.time_sweep(300, use_vbem=vb)
    print("vb", vb, "sweep us", ms * 1e3)
rc, st = p.optimize(); print(st)
rc, st = p.optimize(use_vbem=True); print(st)
