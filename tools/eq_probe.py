"""dev probe: class-build phase timing (cfg2)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
if os.environ.get('EQ_CFG3'): M, P, R = 200_000, 4_000_000, 400_000_000
poff, pids = synth.label_pool(M, P, device=dev, max_k=int(os.environ.get('EQ_MAXK', '200')))     # EQ_MAXK=3: every label one granule
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
expected = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eq = sf.EquivalenceClassBuilder(device=dev, expected_classes=expected)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    eq.start(); eq.add_batch(ids, off)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eq.finish(); v = eq.eqVec()
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"add {1e3*(t1-t):.2f} ms  finish+export {1e3*(t2-t1):.2f} ms  classes {eq.n_classes}", eq.stats())
