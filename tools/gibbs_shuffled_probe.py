import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 200_000, 1_000_000, 30_000_000
ref_len = synth.transcript_lengths(M, device=dev).to(torch.float64)
g = torch.Generator(device=dev); g.manual_seed(2)
poff, pids = synth.label_pool(M, P, device=dev)
sigma = torch.randperm(M, generator=g, device=dev)
cls = torch.repeat_interleave(torch.arange(P, device=dev), poff[1:] - poff[:-1])
key = torch.sort(cls * M + sigma[pids.to(torch.int64)]).values
pids = (key % M).to(torch.int32)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
p = sf.EMProblem(ref_len, v.rowptr, v.ids, v.counts, eq.total_reads)
rc, st = p.optimize(use_vbem=True)
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc, gs = sf.gibbs_sample(ref_len, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 100, n_chains=256, seed=1)
    torch.cuda.synchronize(); print("gibbs shuffled ids: 100 draws x 256 chains", (time.perf_counter() - t) * 1e3, "ms rc", rc)
