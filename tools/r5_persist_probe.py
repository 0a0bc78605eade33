"""dev probe (round 5): the persistent EM loop against one kernel per iteration and the two-kernel loop, on cfg3's / cfg2's classes.
Per mode: stop iteration, loop us per iteration, max relative difference of alpha against the two-kernel loop.
  EMP_SHAPES=cfg3,cfg2   EMP_MODES=two,fused,persist,ablate"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
SHAPES = dict(cfg3=(200_000, 4_000_000, 400_000_000), cfg2=(80_000, 1_000_000, 50_000_000), mid=(400_000, 3_000_000, 100_000_000), tiny=(100_000, 150_000, 3_000_000))
MODES = dict(two=dict(SFGPU_EM_FUSED="0"), fused=dict(SFGPU_EM_FUSED="1", SFGPU_EM_PERSIST="0"), persist=dict(SFGPU_EM_FUSED="1", SFGPU_EM_PERSIST="1"),
             ablate=dict(SFGPU_EM_FUSED="1", SFGPU_EM_PERSIST="2"))
for shape in os.environ.get("EMP_SHAPES", "cfg3,cfg2").split(","):
    M, P, R = SHAPES[shape]
    ref_len = synth.transcript_lengths(M, device=dev)
    poff, pids = synth.label_pool(M, P, device=dev)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    del ids, off
    length = ref_len.to(torch.float64)
    os.environ.pop("SFGPU_EM_PERSIST", None); os.environ.pop("SFGPU_EM_FUSED", None)
    p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
    print(f"== {shape}: classes {eq.n_classes} nnz {eq.nnz}", flush=True)
    for vb in (False, True):
        base = None
        for mode in os.environ.get("EMP_MODES", "two,fused,persist,ablate").split(","):
            for k in ("SFGPU_EM_PERSIST", "SFGPU_EM_FUSED"): os.environ.pop(k, None)
            os.environ.update(MODES[mode])
            best = None
            for rep in range(3):
                rc, st = p.optimize(use_vbem=vb)
                if best is None or st["loop_ms"] < best["loop_ms"]: best = st
            a = p.alpha.cpu().numpy().copy()
            if base is None: base = a
            nz = base > 0
            rel = float(np.max(np.abs(a[nz] - base[nz]) / base[nz])) if nz.any() else 0.0
            sup = bool(np.array_equal(a > 0, nz))
            print(f"  {'VBEM' if vb else 'EM  '} {mode:8s} rc {rc} iters {best['iters']:4d} conv {int(best['converged'])} persistent {int(best.get('persistent', 0))} fused {int(best['fused'])} "
                  f"loop {best['loop_ms']:.3f} ms = {best['loop_ms'] / max(best['iters'], 1) * 1e3:.2f} us/iter | max rel vs two-kernel {rel:.2e} support {'same' if sup else 'DIFFERS'} "
                  f"max_rel_diff {best['max_rel_diff']:.6g}", flush=True)
