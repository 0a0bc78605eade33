import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth, _lib
_lib.set_logger(lambda lvl, msg: print("LOG", lvl, msg, flush=True) if lvl else None)
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
length = ref_len.to(torch.float64)
p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
for mode in ("1", "4", "4", "1"):
    os.environ["SFGPU_EM_PERSIST"] = mode
    rc, st = p.optimize(use_vbem=True)
    print("mode", mode, rc, st["iters"], st["persistent"], round(st["loop_ms"], 3), flush=True)
