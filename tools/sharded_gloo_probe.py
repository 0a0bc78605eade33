"""dev probe: time per 16-iteration chunk of the sharded EM loop with W ranks sharing one GPU over gloo
(W = 4: ~22 ms per iteration, all of it the gloo all-reduce through the host, constant over the run.  W = 8 on ONE GPU is
pathological -- eight processes oversubscribe the device's queues; the 8-rank test therefore stops after 80 iterations)"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, port, sizes):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sailfish_amd as sf
    from sailfish_amd import synth
    from sailfish_amd.distributed import nnz_balanced_slices
    M, P, R = sizes
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    ref_len = synth.transcript_lengths(M, device=dev).to(torch.float64)
    poff, pids = synth.label_pool(M, P, device=dev)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
    eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    del ids, off
    rp = (v.rowptr.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    cuts = nnz_balanced_slices(rp, world); c0, c1 = cuts[rank], cuts[rank + 1]; j0, j1 = int(rp[c0]), int(rp[c1])
    rp_loc = ((v.rowptr[c0:c1 + 1].to(torch.int64) & 0xFFFFFFFF) - j0).to(torch.int32)
    p = sf.EMProblem(ref_len, rp_loc, v.ids[j0:j1], v.counts[c0:c1], eq.total_reads)
    p.begin(use_vbem=True, tol=0.01, min_iter=50, max_iter=400)
    ao = p.alpha_out_view()
    dist.all_reduce(ao); p.init()
    done, chunk = False, 0
    while not done and chunk < 12:
        torch.cuda.synchronize(); t = time.perf_counter(); t_ar = 0.0
        for _ in range(16):
            p.sweep()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            dist.all_reduce(ao)
            torch.cuda.synchronize(); t_ar += time.perf_counter() - t1
            p.update()
        done, seg = p.poll()
        dt = time.perf_counter() - t
        if rank == 0:
            a = ao.cpu().numpy()
            sub = int(((np.abs(a) > 0) & (np.abs(a) < 2.3e-308)).sum())
            print(f"chunk {chunk:2d}: {dt*1e3:9.1f} ms for 16 iterations (all-reduce {t_ar*1e3:9.1f} ms)  iters {seg['iters']} denormals in alphaOut {sub}", flush=True)
        chunk += 1
    dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sizes = (200_000, 4_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000)
    mp.spawn(worker, args=(world, 29533, sizes), nprocs=world, join=True)
