"""dev probe: where does a bench step spend host-visible time (cfg2)"""
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
eq = sf.EquivalenceClassBuilder(device=dev)
length = ref_len.to(torch.float64)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = T(); eq.start(); t1 = T(); eq.add_batch(ids, off); t2 = T(); eq.finish(); t3 = T(); v = eq.eqVec(); t4 = T()
    p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads); t5 = T()
    rc, st = p.optimize(); t6 = T()
    p.close(); t7 = T()
print(f"start {1e3*(t1-t0):.2f} add {1e3*(t2-t1):.2f} finish {1e3*(t3-t2):.2f} export {1e3*(t4-t3):.2f} em_create {1e3*(t5-t4):.2f} "
      f"optimize {1e3*(t6-t5):.2f} (loop {st['loop_ms']:.2f}) close {1e3*(t7-t6):.2f}")
