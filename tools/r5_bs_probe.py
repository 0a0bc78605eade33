"""dev probe (round 5): bootstrap replicates per second with / without the persistent loop, 1 / 3 lanes (cfg3's classes by default)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
SHAPES = dict(cfg3=(200_000, 4_000_000, 400_000_000), cfg2=(80_000, 1_000_000, 50_000_000))
for shape in os.environ.get("EMP_SHAPES", "cfg3").split(","):
    M, P, R = SHAPES[shape]
    ref_len = synth.transcript_lengths(M, device=dev)
    poff, pids = synth.label_pool(M, P, device=dev)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    del ids, off
    length = ref_len.to(torch.float64)
    for persist in ("1", "0"):
        for lanes in ("1", "3"):
            os.environ["SFGPU_EM_PERSIST"] = persist; os.environ["SFGPU_BS_LANES"] = lanes
            p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
            p.optimize(use_vbem=True)
            ts = []
            for n in (1, 6, 6, 12, 24, 48):
                torch.cuda.synchronize(); t = time.perf_counter()
                rc, out, it = p.bootstrap(n, seed=1, use_vbem=True)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3 / n)
            print(f"{shape} persist {persist} lanes {lanes}: ms per replicate for n = 1, 6, 6, 12, 24, 48: " + " ".join(f"{x:.2f}" for x in ts) + f" | iters {it.mean():.0f}", flush=True)
            p.close()
