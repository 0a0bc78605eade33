"""dev probe: EM loop timing on cfg3's classes (400 M reads) -- sweep us, loop us per iteration, for the library in SFGPU_LIB_PATH"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = (int(os.environ.get(k, d)) for k, d in (('EMP_M', 200_000), ('EMP_P', 4_000_000), ('EMP_R', 400_000_000)))      # (cfg2: 80000 / 1000000 / 50000000)
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
del ids, off
length = ref_len.to(torch.float64)
p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
out = []
for vb in (False, True):
    ms = p.time_sweep(300, use_vbem=vb)
    rc, st = p.optimize(use_vbem=vb)
    rc, st = p.optimize(use_vbem=vb)
    out.append(f"{'VBEM' if vb else 'EM'}: sweep {ms*1e3:.2f} us, loop {st['loop_ms']/st['iters']*1e3:.2f} us/iter x {st['iters']}")
print(os.environ.get("SFGPU_LIB_PATH", "main").split("_")[-1], "classes", eq.n_classes, "nnz", eq.nnz, "|", " | ".join(out))
