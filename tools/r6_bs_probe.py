"""dev probe (round 6): bootstrap lanes x persistent loop x where the exchange buffer lives.  Counts the give-ups (log level 1 lines) per setting.
  BSP_SHAPE=cfg3|cfg2  BSP_N=24"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth, _lib
giveups = [0]
def logger(lvl, msg):
    if "gave up" in msg: giveups[0] += 1; print("LOG", lvl, msg, flush=True)
_lib.set_logger(logger)
dev = torch.device("cuda:0")
SHAPES = dict(cfg3=(200_000, 4_000_000, 400_000_000), cfg2=(80_000, 1_000_000, 50_000_000))
M, P, R = SHAPES[os.environ.get("BSP_SHAPE", "cfg3")]
N = int(os.environ.get("BSP_N", "24"))
ref_len = synth.transcript_lengths(M, device=dev)
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
del ids, off
length = ref_len.to(torch.float64)
ref = None
for xbuf in os.environ.get("BSP_XBUF", "uncached,pool").split(","):
    for coop in os.environ.get("BSP_COOP", "0").split(","):
        LS = [tuple(map(int, x.split(':'))) if ':' in x else (int(x), 1) for x in os.environ['BSP_LANESETS'].split(',')] if 'BSP_LANESETS' in os.environ else ((1, 1), (2, 1), (3, 1), (3, 0))
        for lanes, bsp in LS:
            os.environ["SFGPU_EM_XBUF"] = xbuf; os.environ["SFGPU_EM_COOP"] = coop
            os.environ["SFGPU_BS_LANES"] = str(lanes); os.environ["SFGPU_BS_PERSIST"] = str(bsp)
            p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
            p.optimize(use_vbem=True)
            giveups[0] = 0
            ts = []
            for n in (lanes, N, N):
                torch.cuda.synchronize(); t = time.perf_counter()
                rc, out, it = p.bootstrap(n, seed=1, use_vbem=True)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3 / n)
            o = out.cpu().numpy()
            if ref is None: ref = o
            nz = ref > 0
            rel = float(np.max(np.abs(o[nz] - ref[nz]) / ref[nz]))
            print(f"xbuf {xbuf} coop {coop} lanes {lanes} persist-in-lanes {bsp}: ms per replicate (n = {lanes}, {N}, {N}): " + " ".join(f"{x:.2f}" for x in ts) +
                  f" | iters {it.mean():.0f} | give-ups {giveups[0]} | max rel vs first setting {rel:.1e}", flush=True)
            p.close()
