"""dev probe: sweep time vs tile size (SFGPU_EM_TILE) on class tables of different sizes (one EMProblem per setting)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
for M, P, R in ((80_000, 1_000_000, 50_000_000), (200_000, 4_000_000, 400_000_000), (200_000, 12_000_000, 400_000_000), (400_000, 30_000_000, 600_000_000)):
    ref_len = synth.transcript_lengths(M, device=dev)
    poff, pids = synth.label_pool(M, P, device=dev)
    ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
    del poff, pids
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    del ids, off
    nnz = eq.nnz
    length = ref_len.to(torch.float64)
    one = (nnz + 511) // 512
    cands = ["default"] + sorted({int(x) for x in (one, (one + 1) // 2 + 1, (one + 2) // 3 + 1, (one + 3) // 4 + 1) if 2048 <= x <= 65536})
    res = []
    for t in cands:
        if t == "default": os.environ.pop("SFGPU_EM_TILE", None)
        else: os.environ["SFGPU_EM_TILE"] = str(t)
        p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
        for vb in (False, True):
            ms = p.time_sweep(200, use_vbem=vb)
            res.append((t, vb, ms * 1e3))
        p.close()
    print(f"M={M} classes={eq.n_classes} nnz={nnz} one-round tile={one}: " + "  ".join(f"{t}/{'vb' if vb else 'em'} {us:.1f}us" for t, vb, us in res), flush=True)
    eq.close(); del v
