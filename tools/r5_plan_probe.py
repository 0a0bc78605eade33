"""dev probe (round 5): what the EM phase of a bench step is made of besides the loop -- handle creation (the plan) and optimize(),
timed apart on cfg3's / cfg2's classes.   EMP_SHAPES=cfg3,cfg2"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
SHAPES = dict(cfg3=(200_000, 4_000_000, 400_000_000), cfg2=(80_000, 1_000_000, 50_000_000))
for shape in os.environ.get("EMP_SHAPES", "cfg3,cfg2").split(","):
    M, P, R = SHAPES[shape]
    ref_len = synth.transcript_lengths(M, device=dev)
    poff, pids = synth.label_pool(M, P, device=dev)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    del ids, off
    length = ref_len.to(torch.float64)
    print(f"== {shape}: classes {eq.n_classes} nnz {eq.nnz}", flush=True)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
        th = time.perf_counter()                                   # (the host is back: the plan's last kernels are still queued)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rc, st = p.optimize(use_vbem=True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"  create {1e3 * (t1 - t0):.3f} ms (host back after {1e3 * (th - t0):.3f}) | optimize {1e3 * (t2 - t1):.3f} ms ({st['iters']} iterations, loop {st['loop_ms']:.3f} ms, persistent {int(st.get('persistent', 0))})", flush=True)
        p.close()
