"""dev probe: throughput of sfgpu_filter_hits (per-read hit filtering) on synthetic proper-pair hit lists"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 20_000_000
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
H = ids.numel()
g = torch.Generator(device=dev); g.manual_seed(3)
rec = torch.zeros((H, 6), dtype=torch.int32, device=dev)           # 24-byte records as 6 x int32
rec[:, 0] = ids                                                     # tid
rec[:, 1] = torch.randint(0, 2000, (H,), generator=g, device=dev, dtype=torch.int32)          # pos
rec[:, 2] = rec[:, 1] + 150                                         # mate_pos
rec[:, 3] = torch.randint(100, 400, (H,), generator=g, device=dev, dtype=torch.int32)         # frag_len
rec[:, 4] = 75 | (75 << 16)                                         # read_len, mate_len
fwd = torch.randint(0, 2, (H,), generator=g, device=dev, dtype=torch.int32)
rec[:, 5] = fwd | ((1 - fwd) << 8) | (3 << 16)                      # fwd, mate_fwd, mate_status = PAIRED
hits = rec.view(torch.uint8).reshape(-1)
fl = torch.zeros(1000, dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    out_ids, out_off, rem, st = sf.hits.filter_hits(hits, off, "IU", fl_counts=fl, remaining_fl_ops=10000, device=dev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"filter: {R} reads, {H} hits ({H*24/1e9:.2f} GB of records) in {dt*1e3:.2f} ms = {R/dt/1e9:.2f} G reads/s, "
      f"{(H*24 + out_ids.numel()*4 + R*8)/dt/1e12:.2f} TB/s on the algorithmic bytes (records once + ids out + offsets); stats {st}")
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(out_ids, out_off); eq.finish()
print("classes from the filtered lists:", eq.n_classes, "reads", eq.total_reads)

# the bias / GC samples of the same loop (sfgpu_sample_bias): random transcript sequences of the synth lengths
ref_len = synth.transcript_lengths(M, device=dev)
L = (ref_len.to(torch.int64) & 0xFFFFFFFF)
soff = torch.cumsum(L + 1, 0) - (L + 1)
seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (int((L + 1).sum()),), device=dev, generator=g)]
torch.cuda.synchronize(); t = time.perf_counter()
pre = sf.hits.gc_prefix(seq, soff, ref_len)
torch.cuda.synchronize(); print(f"gc_prefix: {seq.numel()/1e6:.0f} M bases in {(time.perf_counter()-t)*1e3:.2f} ms")
for label, kw in (("6-mer samples (budget 1M)", dict(read_bias=torch.ones(4096, dtype=torch.int32, device=dev), remaining_bias_samples=1_000_000)),
                  ("GC samples", dict(observed_gc=torch.ones(101, dtype=torch.int32, device=dev), gc_prefix_table=pre)),
                  ("both", dict(read_bias=torch.ones(4096, dtype=torch.int32, device=dev), remaining_bias_samples=1_000_000,
                                observed_gc=torch.ones(101, dtype=torch.int32, device=dev), gc_prefix_table=pre))):
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rem, nb, ng = sf.hits.sample_bias(hits, off, "IU", seq, soff, ref_len, device=dev, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"sample_bias {label}: {dt*1e3:.2f} ms for {R} reads / {H} hits; sampled {nb} 6-mers, {ng} fragments")
