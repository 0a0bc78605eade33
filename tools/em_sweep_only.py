"""dev probe: sweep time only on cfg3's classes for the library in SFGPU_LIB_PATH (ablation variants do not converge)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 4_000_000, 400_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
del ids, off
p = sf.EMProblem(ref_len.to(torch.float64), v.rowptr, v.ids, v.counts, eq.total_reads)
print(os.environ.get("SFGPU_LIB_PATH", "main").split("_")[-1], " ".join(f"{'VB' if vb else 'EM'} {p.time_sweep(300, use_vbem=vb)*1e3:.2f} us" for vb in (False, True)))
