"""dev probe (round 5): the sharded EM loop on ONE GPU with a communicator of one rank (the sum is the identity): us per iteration of
the two forms -- sweep + fold + all-reduce + update against the fused sweep + fold + all-reduce -- on cfg3's classes"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth, comm
dev = torch.device("cuda:0")
M, P, R = 200_000, 4_000_000, 400_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
del ids, off
p = sf.EMProblem(ref_len.to(torch.float64), v.rowptr, v.ids, v.counts, eq.total_reads)
c = comm.Comm(1, 0, comm.Comm.unique_id(), dev)
for vb in (False, True):
    rc, st0 = p.optimize(use_vbem=vb)
    for fused in (False, True):
        p.set_sharded_fused(fused)
        best = None
        for rep in range(3):
            rc, st = p.optimize_sharded(c, poll_every=16, use_vbem=vb)
            if best is None or st["loop_ms"] < best["loop_ms"]: best = st
        import time
        torch.cuda.synchronize(); t = time.perf_counter(); rc, st = p.optimize_sharded(c, poll_every=16, use_vbem=vb); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{'VBEM' if vb else 'EM'} sharded loop, {'one sweep kernel' if fused else 'sweep + update'} per iteration: iters {st['iters']} (optimize: {st0['iters']}) "
              f"wall {dt * 1e3:.2f} ms = {dt / st['iters'] * 1e6:.2f} us per iteration", flush=True)
c.close()
