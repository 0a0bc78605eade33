"""dev probe: throughput of J independent quantification jobs (cfg2 each) running concurrently, each on its own
stream from its own host thread -- kernels of one job fill the kernel boundaries / latency stalls of the others"""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth, distributed as sfd
dev = torch.device("cuda:0")
M, P, R = 80_000, 1_000_000, 50_000_000
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
ref_np = ref_len.cpu().numpy().view(np.uint32)
names = [f"t{i}" for i in range(M)]
STEPS = 6
for J in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(J)]
    quants = []
    for s in streams:
        with torch.cuda.stream(s):
            sopt = sf.SailfishOpts()
            exp = sf.ReadExperiment(sf.Transcripts(names, ref_np, device=dev), sopt)
            quants.append(sfd.DistributedQuant(exp, sopt))
    def work(j, n):
        with torch.cuda.device(dev), torch.cuda.stream(streams[j]):
            for _ in range(n):
                quants[j].run(ids, off, fl_counts=None, remaining_fl_ops=1)
    th = [threading.Thread(target=work, args=(j, 2)) for j in range(J)]; [x.start() for x in th]; [x.join() for x in th]   # warm
    torch.cuda.synchronize(); t = time.perf_counter()
    th = [threading.Thread(target=work, args=(j, STEPS)) for j in range(J)]; [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"J={J}: {J*STEPS} steps in {dt*1e3:.1f} ms -> {dt/(J*STEPS)*1e3:.2f} ms per step, {J*STEPS*R/dt/1e9:.2f} G reads/s", flush=True)
