"""dev probe: EM sweep time when many classes of a tile share one transcript (a dominant isoform): same-address LDS atomics"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 1_000_000, 30_000_000
ref_len = synth.transcript_lengths(M, device=dev).to(torch.float64)
for block in (0, 256, 32, -4096, -64):
    poff, pids = synth.label_pool(M, P, device=dev)
    pids = pids.to(torch.int64)
    cls = torch.repeat_interleave(torch.arange(P, device=dev), poff[1:] - poff[:-1])
    if block:
        first = pids[poff[:-1]]
        # block > 0: the dominant transcript of the label's neighbourhood (inside the window: same-address LDS atomics);
        # block < 0: a FAR transcript shared by the labels of a neighbourhood (a pseudogene every read of the gene also hits: it escapes)
        dom = (first // block) * block if block > 0 else (M - 1 - first // (-block))
        key = torch.cat([cls * M + pids, torch.arange(P, device=dev) * M + dom])
    else:
        key = cls * M + pids
    key = torch.unique(key)                                # sorted, distinct (class, transcript) pairs
    pids2 = (key % M).to(torch.int32); cls2 = key // M
    poff2 = torch.zeros(P + 1, dtype=torch.int64, device=dev); poff2[1:] = torch.cumsum(torch.bincount(cls2, minlength=P), 0)
    ids, off = synth.reads_slice(poff2, pids2, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    p = sf.EMProblem(ref_len, v.rowptr, v.ids, v.counts, eq.total_reads)
    t = p.time_sweep(200, use_vbem=False) * 1e3
    print(f"shared transcript per {block:5d} ids: classes {eq.n_classes} nnz {eq.nnz}  sweep {t:8.2f} us  ({t * 1e3 / eq.nnz:.2f} ps per nonzero)")
    del ids, off, eq, p, v
