"""dev probe: class build of 100 M reads handed over in device batches of different sizes"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 4_000_000, 100_000_000
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
off64 = off.to(torch.int64) & 0xFFFFFFFF
for bs in (R, 16_777_216, 4_000_000, 1_000_000, 250_000):
    eq = sf.EquivalenceClassBuilder(device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        eq.start()
        for a in range(0, R, bs):
            b = min(R, a + bs)
            o = off64[a:b + 1]
            eq.add_batch(ids[int(o[0]):int(o[-1])], (o - o[0]).to(torch.int32))
        eq.finish()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"batches of {bs:>10d} reads: {dt*1e3:8.2f} ms  classes {eq.n_classes} launches {eq.stats()['insert_launches']}")
