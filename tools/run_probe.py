"""dev probe: DistributedQuant.run wall time vs the sum of its phases (cfg2)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth, distributed as sfd
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
sopt = sf.SailfishOpts()
exp = sf.ReadExperiment(sf.Transcripts([f"t{i}" for i in range(M)], ref_len.cpu().numpy().view(np.uint32), device=dev), sopt)
q = sfd.DistributedQuant(exp, sopt)
fl = None
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    info = q.run(ids, off, fl_counts=fl, remaining_fl_ops=1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    parts = info["t_build_ms"] + info["t_efflen_ms"] + info["t_em_ms"] + info["t_tpm_ms"]
    print(f"run {dt:.2f} ms; phases {parts:.2f} (build {info['t_build_ms']:.2f} efflen {info['t_efflen_ms']:.2f} em {info['t_em_ms']:.2f} loop {info['em_stats']['loop_ms']:.2f} tpm {info['t_tpm_ms']:.2f})")
