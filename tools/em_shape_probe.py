"""dev probe: EM sweep time against the LOCALITY of the labels -- a fraction `far` of every label's members (beyond the first)
is replaced by transcripts drawn uniformly from the whole transcriptome (paralogs / repeats far away in annotation order)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 1_000_000, 30_000_000
ref_len = synth.transcript_lengths(M, device=dev).to(torch.float64)
g = torch.Generator(device=dev); g.manual_seed(5)
for far in (0.0, 0.05, 0.2, 0.5, 1.0):
    poff, pids = synth.label_pool(M, P, device=dev)
    pids = pids.clone().to(torch.int64)
    first = torch.zeros_like(pids, dtype=torch.bool); first[poff[:-1]] = True
    repl = (torch.rand(pids.numel(), generator=g, device=dev) < far) & ~first
    pids[repl] = torch.randint(0, M, (int(repl.sum()),), generator=g, device=dev)
    # sort members inside each label, drop duplicates by nudging (keep it simple: duplicates are harmless for timing? no -- make them distinct)
    cls = torch.repeat_interleave(torch.arange(P, device=dev), poff[1:] - poff[:-1])
    key = torch.sort(cls * M + pids).values
    keep = torch.ones_like(key, dtype=torch.bool); keep[1:] = key[1:] != key[:-1]
    key = key[keep]
    pids2 = (key % M).to(torch.int32); cls2 = key // M
    poff2 = torch.zeros(P + 1, dtype=torch.int64, device=dev); poff2[1:] = torch.cumsum(torch.bincount(cls2, minlength=P), 0)
    ids, off = synth.reads_slice(poff2, pids2, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    p = sf.EMProblem(ref_len, v.rowptr, v.ids, v.counts, eq.total_reads)
    t = p.time_sweep(200, use_vbem=False) * 1e3
    print(f"far {far:4.2f}: classes {eq.n_classes} nnz {eq.nnz}  sweep {t:8.2f} us  ({eq.nnz * 4 / t / 1e6:.2f} TB/s on the stream words)")
    del ids, off, eq, p, v
