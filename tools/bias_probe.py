"""dev probe: bias-aware effective lengths at transcriptome scale (M transcripts, synth lengths, random sequence):
time of the one-off GC profile (sfgpu_bias_create) and of one updateEffectiveLengths per model"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 80_000))
ref_len = synth.transcript_lengths(M, device=dev)
L = (ref_len.to(torch.int64) & 0xFFFFFFFF)
off = torch.cumsum(L + 1, 0) - (L + 1)
total = int((L + 1).sum())
g = torch.Generator(device=dev); g.manual_seed(1)
seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (total,), device=dev, generator=g)]
x = np.arange(1000); fl = np.round(1e6 * np.exp(-0.5 * ((x - 200) / 80.0) ** 2)).astype(np.uint32)
eff = torch.clamp(L.to(torch.float64) - 190.0, min=1.0)
alpha = torch.rand(M, dtype=torch.float64, device=dev, generator=g) * 100
rb = np.random.default_rng(1).integers(1, 500, 4096); og = np.random.default_rng(2).integers(1, 5000, 101)
print(f"M={M} bases={total/1e6:.1f}M mean len {float(L.double().mean()):.0f} max {int(L.max())}", flush=True)
for mode, samp in (("seq", 1), ("gc", 1), ("gc", 5)):
    torch.cuda.synchronize(); t = time.perf_counter()
    m = sf.bias.BiasModel(seq, off, ref_len, eff, fl, rb, og, num_fwd=6, num_rc=4, seq_bias=mode == "seq", gc_bias=mode == "gc",
                          gc_speed_samp=samp)
    torch.cuda.synchronize(); tc = time.perf_counter() - t
    ts = []
    for r in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        out, st = m.update(eff, alpha)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"{mode} samp={samp}: create {tc*1e3:.1f} ms, update {min(ts)*1e3:.2f} ms (first {ts[0]*1e3:.2f}), stats {st}", flush=True)
    m.close()

# optimize() with the recompute hook on cfg2-shaped classes (50 M reads over the same 80 k transcripts)
if os.environ.get("EM", "1") == "1" and M == 80_000:
    P, R = 1_000_000, 50_000_000
    poff, pids = synth.label_pool(M, P, device=dev)
    ids, o2 = synth.reads_from_pool(poff, pids, R, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, o2); eq.finish(); v = eq.eqVec()
    del ids, o2
    prob = sf.EMProblem(eff, v.rowptr, v.ids, v.counts, eq.total_reads)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc, st = prob.optimize()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"optimize (no bias): {dt*1e3:.2f} ms, {st['iters']} iterations", flush=True)
    for mode in ("seq", "gc"):
        m = sf.bias.BiasModel(seq, off, ref_len, eff, fl, rb, og, num_fwd=6, num_rc=4, seq_bias=mode == "seq", gc_bias=mode == "gc")
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            rc, st, eff_out, hooks = prob.optimize_bias(m)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"optimize ({mode} bias): {dt*1e3:.2f} ms, {st['iters']} iterations, {hooks} recomputes, rc {rc}", flush=True)
        m.close()
