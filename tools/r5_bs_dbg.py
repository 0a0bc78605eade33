import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SFGPU_TIMING"] = "1"
import sailfish_amd as sf
from sailfish_amd import synth, _lib
_lib.set_logger(lambda lvl, msg: print("LOG", lvl, msg, flush=True))
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
length = ref_len.to(torch.float64)
p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
print(p.optimize(use_vbem=True))
for n in (1, 3, 3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc, out, it = p.bootstrap(n, seed=1, use_vbem=True)
    torch.cuda.synchronize(); print("bootstrap", n, (time.perf_counter() - t) * 1e3 / n, "ms per replicate", it, flush=True)
print("=== fresh handle: clones made by a non-persistent bootstrap first", flush=True)
p2 = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
os.environ["SFGPU_EM_PERSIST"] = "0"
rc, out, it = p2.bootstrap(3, seed=1, use_vbem=True)
# (SFGPU_EM_PERSIST=0 at create: the clones have no tables -> they can never go persistent; so make them with tables but run them without)
p3 = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
os.environ["SFGPU_EM_PERSIST"] = "1"; os.environ["SFGPU_EM_FUSED"] = "0"
rc, out, it = p3.bootstrap(3, seed=1, use_vbem=True); print("clones made (two-kernel loop)", it, flush=True)
os.environ.pop("SFGPU_EM_FUSED")
for n in (3, 3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc, out, it = p3.bootstrap(n, seed=1, use_vbem=True)
    torch.cuda.synchronize(); print("bootstrap", n, (time.perf_counter() - t) * 1e3 / n, "ms per replicate", it, flush=True)
