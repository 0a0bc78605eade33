"""dev probe: EM sweep when the transcript ids are SHUFFLED (an index whose isoforms are not adjacent: accession order)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 1_000_000, 30_000_000
ref_len = synth.transcript_lengths(M, device=dev).to(torch.float64)
g = torch.Generator(device=dev); g.manual_seed(2)
for shuffled in (False, True):
    poff, pids = synth.label_pool(M, P, device=dev)
    if shuffled:
        sigma = torch.randperm(M, generator=g, device=dev)
        cls = torch.repeat_interleave(torch.arange(P, device=dev), poff[1:] - poff[:-1])
        key = torch.sort(cls * M + sigma[pids.to(torch.int64)]).values
        pids = (key % M).to(torch.int32)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    for env in ("1", None):
        if env: os.environ["SFGPU_EM_NO_RENUMBER"] = env
        else: os.environ.pop("SFGPU_EM_NO_RENUMBER", None)
        p = sf.EMProblem(ref_len, v.rowptr, v.ids, v.counts, eq.total_reads)
        t = p.time_sweep(200, use_vbem=False) * 1e3
        rc, st = p.optimize(use_vbem=True)
        print(f"shuffled={shuffled} renumber={'off' if env else 'on '}: classes {eq.n_classes} nnz {eq.nnz} sweep {t:8.2f} us | VBEM {st['iters']} iters {st['loop_ms']:.2f} ms alpha_sum {st['alpha_sum']:.6f}")
        del p
