#!/usr/bin/env python
"""bench.py -- whole-job throughput of the Sailfish quantification hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: packed quasi-mapping hit lists -> equivalence classes (start / addGroup / finish / eqVec)
-> effective lengths -> EM (or VBEM) to convergence -> TPM / NumReads.  Read parsing and the
quasi-mapper are third-party code outside the path and outside every timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|small]

N = 1 : the configuration BASELINE.json's metric is quoted on ("... 200k-txp index"): configs[2], "cfg3" -- 400M
        paired-end fragments, 200k-transcript index, VBEM + empirical fragment-length correction; it fits one GPU
        (8 GB of hit lists).  --workload cfg2 is configs[1] (50M single-end reads, 80k transcripts, EM).
N > 1 : launched by torch.distributed.run, one rank per GPU.  STRONG scaling (BASELINE configs[3], "cfg4": the same
        400M PE reads / 200k transcripts sharded across the GPUs): rank r holds reads [r R/N, (r+1) R/N) of the very
        experiment N = 1 runs (synth.reads_slice: the read stream is defined chunk by chunk, so the union of the
        shards IS the single-GPU input and the merged class table is the single-GPU table); the class tables are merged
        with one exchange and the EM runs on the merged classes (sailfish_amd/distributed.py).  --weak gives every
        rank its own R reads instead (the job grows with N).

Rank 0 prints ONE JSON line (metric, value, roofline, cpu_baseline ...).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (M transcripts, P label pool, R reads per GPU, VBEM?, paired-end?)
    "small": (5_000, 20_000, 2_000_000, False, False),
    "cfg2": (80_000, 1_000_000, 50_000_000, False, False),
    "cfg3": (200_000, 4_000_000, 400_000_000, True, True),
}
HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md): 8 TB/s HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--em-mode", default="auto", choices=["auto", "replicated", "sharded"])
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank gets its own R reads (weak scaling) instead of R/N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-pinned", action="store_true", help="skip the leg that re-runs the step with the hit lists in pinned host memory")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--one-device", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-compare-em-modes", action="store_true",
                    help="N > 1: skip the step in the OTHER EM mode (sharded <-> replicated) that is run and reported after the timed steps")
    ap.add_argument("--no-sampling", action="store_true", help="N = 1: skip the bootstrap / Gibbs (BASELINE config 5) legs after the timed steps")
    ap.add_argument("--gibbs-draws", type=int, default=1000)
    ap.add_argument("--bootstrap-draws", type=int, default=6)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline_full(vec, ref_len_np, ids_t, off_t, n_total, use_vbem, fl_counts):
    """The one-core oracle on the FULL problem's classes instead of a sample's (no extrapolation by an nnz ratio, and the
    class build in the regime the job is in -- lookups into a table that already holds the classes):
      * class build: the oracle's table is first filled with every class of the job (untimed), then 2 M reads from the
        middle of the read stream are timed: lookups + count++, what all but the first few million reads of the job do;
      * EM: 10 iterations over all classes, timed.
    -> (steady-state build reads/s, seconds per EM iteration)"""
    from oracle import oracle as O
    rp, ii, cc, hh = vec.to_numpy()
    b = O.EqBuilder()
    b.add_batch(ii, rp.astype(np.uint64))                                         # every class once (untimed)
    n = min(2_000_000, off_t.numel() - 1)
    r0 = (off_t.numel() - 1 - n) // 2
    off = (off_t[r0: r0 + n + 1].long() & 0xFFFFFFFF).cpu().numpy().astype(np.uint64)
    ids = ids_t[int(off[0]): int(off[-1])].cpu().numpy().view(np.uint32)
    off -= off[0]
    t0 = time.perf_counter(); b.add_batch(ids, off); t_look = time.perf_counter() - t0
    b.finish()
    assert b.n_classes == len(cc), "the sample's reads must all belong to known classes"
    eff = O.efflen_smoothed(ref_len_np, O.cf_counts(fl_counts) if fl_counts is not None else O.cf_gaussian())
    t0 = time.perf_counter()
    rc_o, alpha_o, _, _ = O.em_optimize(eff, rp.astype(np.uint64), ii, cc, n_total, use_vbem=use_vbem, tol=0.0, min_iter=10, max_iter=10)
    em_s = (time.perf_counter() - t0) / 10
    # ... and the alpha of those 10 iterations is the FULL problem's parity check: the HIP loop (the kernel configuration that was
    # timed) over the same classes, 10 iterations, against it
    full = None
    try:
        import sailfish_amd as sf
        prob = sf.EMProblem(torch.from_numpy(eff).to(ids_t.device), vec.rowptr, vec.ids, vec.counts, n_total)
        rc_g, st_g = prob.optimize(use_vbem=use_vbem, tol=0.0, min_iter=10, max_iter=10)
        alpha_g = prob.alpha.cpu().numpy()
        nz = alpha_o > 0
        full = dict(iters=10, rc=[int(rc_o), int(rc_g)], support_identical=bool(np.array_equal(alpha_g > 0, nz)),
                    max_rel_numreads=float(np.max(np.abs(alpha_g[nz] - alpha_o[nz]) / alpha_o[nz])) if nz.any() else 0.0,
                    persistent=bool(st_g.get("persistent")), fused=bool(st_g.get("fused")))
        prob.close()
    except Exception as e:                        # never take the line down
        full = dict(error=repr(e))
    return n / t_look, em_s, full


def cpu_baseline(ref_len_np, ids_t, off_t, target_s, use_vbem):
    """The oracle (C restatement of the reference, 1 thread) timed on a bounded sample of the
    same workload: class build on the first reads of the batch, then EM on the classes of that
    sample for a bounded number of iterations; combined into the metric's unit (reads/s) by
    scaling the EM leg to the iteration count the full problem needs."""
    from oracle import oracle as O
    n_build = 2_000_000
    n_build = min(n_build, off_t.numel() - 1)
    off = (off_t[: n_build + 1].long() & 0xFFFFFFFF).cpu().numpy().astype(np.uint64)
    ids = ids_t[: int(off[-1])].cpu().numpy().view(np.uint32)
    b = O.EqBuilder()
    t0 = time.perf_counter()
    b.add_batch(ids, off)
    rp, ii, cc, hh = b.finish()
    t_build = time.perf_counter() - t0
    build_rate = n_build / t_build
    eff = O.efflen_smoothed(ref_len_np, O.cf_gaussian())
    # time a fixed number of iterations, sized to the remaining budget
    t0 = time.perf_counter()
    O.em_optimize(eff, rp, ii, cc, n_build, use_vbem=use_vbem, tol=0.0, min_iter=0, max_iter=3)
    per_iter = (time.perf_counter() - t0) / 3
    n_it = int(max(5, min(400, (target_s - t_build) / max(per_iter, 1e-6))))
    t0 = time.perf_counter()
    rc_o, alpha_o, _, _ = O.em_optimize(eff, rp, ii, cc, n_build, use_vbem=use_vbem, tol=0.0, min_iter=n_it, max_iter=n_it)
    per_iter = (time.perf_counter() - t0) / n_it
    # "TPM delta vs CPU ref" of the metric: the HIP path on the very same sample (same reads -> classes, same
    # number of iterations), compared with what the oracle just produced
    parity = None
    try:
        import sailfish_amd as sf
        dev = ids_t.device
        eq = sf.EquivalenceClassBuilder(device=dev)
        eq.start(); eq.add_batch(ids_t[: int(off[-1])], off_t[: n_build + 1]); eq.finish()
        v = eq.eqVec()
        grp, gii, gcc, _ = v.to_numpy()
        same_classes = bool(eq.n_classes == b.n_classes and np.array_equal(grp, rp.astype(np.uint32))
                            and np.array_equal(gii, ii) and np.array_equal(gcc, cc))
        prob = sf.EMProblem(torch.from_numpy(eff).to(dev), v.rowptr, v.ids, v.counts, eq.total_reads)
        rc_g, _ = prob.optimize(use_vbem=use_vbem, tol=0.0, min_iter=n_it, max_iter=n_it)
        alpha_g = prob.alpha.cpu().numpy()
        tpm_o, tpm_g = O.tpm(alpha_o, eff, n_build), O.tpm(alpha_g, eff, n_build)
        nz = alpha_o > 0
        rel = lambda a, r: float(np.max(np.abs(a[nz] - r[nz]) / r[nz])) if nz.any() else 0.0
        parity = dict(sample=f"first {n_build} reads, {n_it} {'VBEM' if use_vbem else 'EM'} iterations on both sides",
                      classes_identical=same_classes, support_identical=bool(np.array_equal(alpha_g > 0, nz)),
                      max_rel_numreads=rel(alpha_g, alpha_o), max_rel_tpm=rel(tpm_g, tpm_o),
                      rc=[int(rc_o), int(rc_g)])
        prob.close()
    except Exception as e:            # the checker leg must never take the benchmark line down
        parity = dict(error=repr(e))
    # the same sample on the host's cores: per-thread tables folded into one; classes cut into nnz-balanced
    # ranges adding into one alphaOut with CAS adds, the reference's scheme.  Neither leg scales to hundreds of
    # threads on a sample of this size, so a few thread counts are tried and the best of each leg is reported
    # (`cores` = the threads used by the slower-to-saturate leg, the EM).  A few seconds on top of the one-core leg.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    mt = None
    if avail > 1:
        try:
            cand = sorted({t for t in (4, 8, 16, 32, 64, 128, avail) if t <= avail})
            best_b, best_e = None, None
            for t in cand:
                bm = O.EqBuilder()
                t_mt = bm.add_batch_mt(ids, off, t)
                bm.finish()
                ok = bool(bm.n_classes == b.n_classes and bm.total_reads == b.total_reads)
                if ok and (best_b is None or t_mt < best_b[1]):
                    best_b = (t, t_mt)
                sec_mt, _ = O.em_iterations_mt(eff, rp, ii, cc, n_build, 20, t, use_vbem=use_vbem)
                if best_e is None or sec_mt < best_e[1]:
                    best_e = (t, sec_mt)
            mt = dict(cores=best_e[0], build_threads=best_b[0], build_reads_per_s=n_build / best_b[1], em_ms_per_iter=best_e[1] * 1e3,
                      em_iters=20, host_cores_available=avail, tried=cand)
        except Exception as e:
            mt = dict(error=repr(e))
    return dict(build_reads_per_s=build_rate, em_ms_per_iter=per_iter * 1e3, sample_reads=n_build,
                sample_classes=int(b.n_classes), sample_nnz=int(b.nnz), sample_em_iters=n_it, parity=parity, all_cores=mt)


def info_total_reads(info, quant):
    return int(quant.last_vec.total_reads)


def _relaunch_under_torchrun(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here (one process per GPU, the launch line
    the driver uses: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch_under_torchrun(a)
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if a.one_device:
        local = 0
        if world > 1:
            from sailfish_amd import _lib as _sflib
            _sflib.lib().sfgpu_em_allow_persistent(0)     # several processes on ONE device: no kernel has the chip to itself (the persistent loop needs all its blocks resident)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    import sailfish_amd as sf
    from sailfish_amd import synth
    from sailfish_amd import distributed as sfd

    M, P, R, use_vbem, paired = WORKLOADS[a.workload]
    # ---- synthetic inputs, generated on the device and left resident in HBM (untimed) ----------
    ref_len = synth.transcript_lengths(M, device=dev)
    poff, pids = synth.label_pool(M, P, device=dev)
    if a.weak or world == 1:
        lo, hi = rank * R, (rank + 1) * R                  # weak scaling: N x R reads in all
        R_total = R * world
    else:
        lo, hi = R * rank // world, R * (rank + 1) // world    # strong scaling: the same R reads, sharded
        R_total = R
    ids, off = synth.reads_slice(poff, pids, lo, hi, seed=7, device=dev)                 # this rank's shard
    R_local = hi - lo
    del poff, pids
    ref_len_np = ref_len.cpu().numpy().view(np.uint32)
    n_hits = ids.numel()
    names = [f"T{i}" for i in range(M)]
    sopt = sf.SailfishOpts(useVBOpt=use_vbem)
    # paired-end configs exercise the empirical FLD branch (>= numFragSamples unique pairs)
    fl_counts = None
    if paired:
        i = np.arange(1000.0)
        fl_counts = np.floor(np.exp(-0.5 * ((i - 200.0) / 80.0) ** 2) * 4000 + 0.5).astype(np.uint32)
    exp = sf.ReadExperiment(sf.Transcripts(names, ref_len_np, device=dev), sopt)
    quant = sfd.DistributedQuant(exp, sopt, group=(dist.group.WORLD if dist else None), em_mode=a.em_mode)
    torch.cuda.synchronize()

    def step():
        return quant.run(ids, off, fl_counts=fl_counts, remaining_fl_ops=(0 if paired else 1))

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        info = step()
    barrier()
    t0 = time.perf_counter()
    phase_keys = ("t_build_ms", "t_insert_ms", "t_merge_ms", "t_efflen_ms", "t_em_ms", "t_tpm_ms")
    phase_sum = dict.fromkeys(phase_keys, 0.0); loop_sum = 0.0
    for _ in range(a.steps):
        info = step()
        for k in phase_keys:
            phase_sum[k] += info.get(k, 0.0)
        loop_sum += info["em_stats"]["loop_ms"]
    barrier()
    dt = time.perf_counter() - t0
    for k in phase_keys:                                   # the phases and the EM loop time are reported as MEANS over the timed steps
        info[k] = phase_sum[k] / a.steps
    info["em_stats"] = dict(info["em_stats"], loop_ms=loop_sum / a.steps)
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_reads = R_total * a.steps
    value = total_reads / dt

    # ---- the same step with the hit lists in HOST-PINNED memory (SURVEY 8d / BASELINE.md "Timed region": packed hit lists
    # resident in host-pinned memory -> ... -> TPM).  sfgpu_eq_add_batch_host streams the batch through two device staging
    # buffers (copy of chunk k + 1 on its own stream while chunk k is built), so the step costs its PCIe transfer plus
    # the build of the last chunk plus EM.  PCIe-inclusive: reported next to `value`, never as `value`.
    host_leg = None
    if not a.no_host_pinned:
        # every rank pins ITS shard (its own PCIe link); a rank that cannot (pinned allocation) must not leave the others in a
        # collective: the ok flag is agreed on first, then all ranks run the leg or none does
        h_ids = h_off = None
        ok = 1
        try:
            h_ids = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True); h_ids.copy_(ids)
            h_off = torch.empty(off.shape, dtype=off.dtype, pin_memory=True); h_off.copy_(off)
            torch.cuda.synchronize()
        except Exception as e:
            ok = 0; host_leg = dict(error=repr(e))
        if dist:
            t = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if ok and int(t.item()) == 0:
                host_leg = dict(error="another rank could not pin its shard")
            ok = int(t.item())
        if ok:
            hsteps = max(1, min(a.steps, 3))
            infoh = quant.run(h_ids, h_off, fl_counts=fl_counts, remaining_fl_ops=(0 if paired else 1))      # warm-up (staging buffers)
            barrier()
            th0 = time.perf_counter(); tb = 0.0
            for _ in range(hsteps):
                infoh = quant.run(h_ids, h_off, fl_counts=fl_counts, remaining_fl_ops=(0 if paired else 1))
                tb += infoh["t_build_ms"]
            barrier()
            dth = time.perf_counter() - th0
            if dist:
                t = torch.tensor([dth, tb], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dth, tb = float(t[0]), float(t[1])
            same = bool(infoh["n_classes"] == info["n_classes"] and infoh["nnz"] == info["nnz"] and
                        infoh["em_stats"]["iters"] == info["em_stats"]["iters"])
            nbytes = (ids.numel() + off.numel()) * 4
            host_leg = dict(value=R_total * hsteps / dth, unit="reads/s", steps=hsteps, ms_per_step=dth / hsteps * 1e3,
                            class_build_ms=tb / hsteps, h2d_bytes_per_gpu=nbytes, pcie_gbps_per_gpu=nbytes / (tb / hsteps * 1e-3) / 1e9,
                            em_mode=infoh["em_mode"], same_result_as_hbm_resident_step=same)
        del h_ids, h_off
    st = info["em_stats"]
    C, L = info["n_classes"], info["nnz"]

    # ---- roofline of the dominant kernel, measured live with HIP events ------------------------
    # EM sweep: algorithmic bytes per launch, aux-weight-free variant:
    #   B_iter' = 4L + 8C + 48M   (+16M for VBEM's expTheta vector)
    b_iter = 4 * L + 8 * C + 48 * M + (16 * M if use_vbem else 0)
    sweep_ms = quant.time_sweep(200)
    em_loop_ms_per_iter = st["loop_ms"] / max(st["iters"], 1)
    # class build: B_read = 4*h + 4 (offset) + 16 (one slot probe) bytes per read
    b_read = 4.0 * n_hits / R_local + 20.0
    build_ms = info["t_build_ms"]
    em_ms = info["t_em_ms"]
    # HBM bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 runs of
    # this same command; FETCH_SIZE x2 for the wide streaming loads of the sweep, raw for the random
    # probes of the insert kernel -- see profiles/r1*_pmc_summary.md).  null when no profile matches.
    traffic_em = traffic_em_iter = traffic_ins = traffic_src = None
    try:
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_{a.workload}.json")
        if not os.path.exists(pmc_path):
            pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        pmc = json.load(open(pmc_path))
        if pmc.get("workload") == a.workload and world == 1:
            k = pmc["kernels"]
            bytes_of = lambda e: ((2 if e.get("fetch_x2") else 1) * e["fetch_kib_per_launch"] + e["write_kib_per_launch"]) * 1024
            # (kernel names carry their template arguments: k_sweep_lds<VB, GATHER>, k_part_route<RING>; the instance with the
            #  most launches is the one the step ran)
            pick = lambda prefix: max((v for n, v in k.items() if n == prefix or (n.startswith(prefix) and n[len(prefix)] in "<,>")),
                                      key=lambda v: v.get("launches", 0), default=None)
            # k_sweep_lds<VB, GATHER, FUSED>: the plain sweep (what time_sweep launches) and the fused iteration are different instances
            vbs = "true" if use_vbem else "false"
            plain = [v for n, v in k.items() if n.startswith(f"k_sweep_lds<{vbs}") and (n.count(",") == 1 or n.endswith(", false>"))]
            fusedk = [v for n, v in k.items() if n.startswith(f"k_sweep_lds<{vbs}") and n.count(",") == 2 and n.endswith(", true>")]
            if plain:
                traffic_em = bytes_of(max(plain, key=lambda v: v.get("launches", 0)))
            if fusedk:
                traffic_em_iter = bytes_of(max(fusedk, key=lambda v: v.get("launches", 0)))
            # the persistent loop (round 5): ONE launch holds all iterations -- its counters divided by the steps it ran (the iterations
            # + the one sweep that runs for nothing before the stop is known)
            persk = [v for n, v in k.items() if n.startswith(f"k_em_persist<{vbs}")]
            if persk and st.get("persistent"):
                traffic_em_iter = bytes_of(max(persk, key=lambda v: v.get("launches", 0))) / (st["iters"] + 1)
            # class build = the partition kernels of one sub-batch (k_insert on the generic path)
            parts = [v for v in (pick("k_part_route"), pick("k_part_insert")) if v] or [v for v in (pick("k_insert"),) if v]
            if parts:
                traffic_ins = sum(bytes_of(e) for e in parts)
            # not re-measured by this run: it comes from the committed PMC passes of the same command
            traffic_src = f"{os.path.relpath(pmc_path, ROOT)} ({pmc.get('source', '?')}; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " \
                          f"FETCH x2 for 16-B-per-lane streaming kernels)"
    except (OSError, ValueError, KeyError):
        pass
    n_ins = max(int(info["insert_launches"]), 1)
    ins_ms = info["t_insert_ms"] / n_ins
    # The EM is reported twice, and neither figure borrows from the other (round-3 review: B_iter' over the sweep kernel's time
    # alone over-stated it):
    #   roofline_em_iteration : the WHOLE iteration's algorithmic bytes B_iter' over the time one iteration takes in the loop as it
    #                           runs (sweep + per-transcript update + what a chunk boundary costs) -- the figure to quote;
    #   roofline_em_sweep     : the sweep kernel alone, with the bytes the SWEEP moves: labels 4 L + rowptr / count 8 C + the
    #                           x vector it gathers from (8 M) + one partial sum per transcript it publishes (8 M); the other
    #                           32 M (+ 16 M VBEM) of B_iter' are the update's.
    b_sweep = 4 * L + 8 * C + 16 * M
    fused = bool(st.get("fused"))
    roof_em = dict(bound="hbm", kernel="k_sweep_lds" + (" (the plain sweep, timed outside the loop: the loop itself runs it fused with the update)" if fused else ""), achieved=b_sweep / (sweep_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS,
                   unit="GB/s", frac=b_sweep / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=traffic_em, traffic_source=traffic_src,
                   bytes_per_launch=b_sweep, bytes_formula="4 L + 8 C + 16 M (sweep only: labels, rowptr + count, x gathered, partials published)",
                   avg_launch_ms=sweep_ms, launches_per_step=st["iters"])
    it_s = em_loop_ms_per_iter * 1e-3
    persistent = bool(st.get("persistent"))
    roof_em_iter = dict(bound="hbm", kernel=("one EM iteration inside the persistent loop (k_em_persist: ONE launch per optimize(); the device time of the launch / its iterations)"
                                            if persistent else
                                            "one EM iteration as the loop runs it (ONE kernel: the update of the iteration before at the head of the sweep; + chunk boundaries)"
                                            if fused else "one EM iteration as the loop runs it (sweep + update + chunk boundaries)"),
                        achieved=b_iter / it_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=b_iter / it_s / 1e9 / HBM_PEAK_GBS,
                        traffic=(traffic_em_iter if fused else None), traffic_source=(traffic_src if fused else None),
                        bytes_per_launch=b_iter, bytes_formula="B_iter' = 4 L + 8 C + 48 M (+ 16 M VBEM), SURVEY 8d aux-weight-free variant",
                        avg_launch_ms=em_loop_ms_per_iter, launches_per_step=(1 if persistent else st["iters"]), iterations_per_step=st["iters"], persistent=persistent)
    # class build: SURVEY 8d's B_read (ids + offset + one 16-byte slot probe) and, next to it, the COMPULSORY bytes alone
    # (ids + offset: what any builder must read), so that the probe term cannot flatter the fraction
    b_comp = 4.0 * n_hits / R_local + 4.0
    roof_build = dict(bound="hbm", kernel=info.get("insert_kernels", "k_insert"), achieved=b_read * R_local / n_ins / (ins_ms * 1e-3) / 1e9,
                      peak=HBM_PEAK_GBS, unit="GB/s",
                      frac=b_read * R_local / n_ins / (ins_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=traffic_ins, traffic_source=traffic_src,
                      bytes_per_launch=b_read * R_local / n_ins, avg_launch_ms=ins_ms, launches_per_step=n_ins,
                      frac_compulsory_bytes=b_comp * R_local / n_ins / (ins_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      compulsory_bytes_per_read=b_comp, bytes_per_read=b_read,
                      frac_of_whole_phase=b_read * R_local / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
    dominant = roof_em if sweep_ms * st["iters"] >= info["t_insert_ms"] else roof_build

    out = {
        "metric": "reads quantified/sec, hit lists HBM-resident when the timed region starts (hit lists -> eq-classes -> EM to convergence "
                  "-> TPM); the same step from HOST-PINNED hit lists (BASELINE.md 3's timed region, PCIe-inclusive) is value_host_pinned",
        "value_residency": "hbm",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": ("weak" if (a.weak or world == 1) else "strong"), "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{'cfg4 = ' if (a.workload == 'cfg3' and world > 1 and not a.weak) else ''}{a.workload}: "
                               f"{R_total} reads in all over {world} GPU ({R_local} on rank 0), {M}-transcript index, label pool {P}, "
                               f"{'VBEM' if use_vbem else 'EM'} to convergence (tol 0.01, minIter 50); hit lists HBM-resident "
                               f"(value) / host-pinned (value_host_pinned)",
                   "reads_total": R_total, "reads_per_gpu": R_local, "transcripts": M, "hits": n_hits, "classes": C, "nnz": L,
                   "em_mode": info["em_mode"]},
        "em_iters": st["iters"], "em_iters_per_s": st["iters"] / (em_ms * 1e-3),
        "em_us_per_iter_loop": em_loop_ms_per_iter * 1e3, "em_fused_iteration": fused,
        "phase_ms_is": "mean over the timed steps",
        "phase_ms": {"class_build": build_ms, "insert_kernel": info["t_insert_ms"], "merge": info.get("t_merge_ms", 0.0),
                     "efflen": info["t_efflen_ms"], "em": em_ms, "tpm": info["t_tpm_ms"]},
        "class_build_reads_per_s": R_local / (build_ms * 1e-3),
        "roofline": dominant, "roofline_em_iteration": roof_em_iter, "roofline_em_sweep": roof_em, "roofline_class_build": roof_build,
    }
    if world == 1:
        # ---- what this N = 1 run implies for N = 2 / 4 / 8: a BUDGET made of this run's measured pieces plus
        #      stated constants, so that the first SCALE record of an 8-GPU node can be read against it.  Nothing here was run on > 1 GPU.
        #      build(N)  = insert kernels / N (reads shard with no exchange) + finish / export of the rank's own classes (unchanged)
        #      merge(N)  = the owner all-to-all + disjoint all-gather of the (label, count, hash) table over xGMI, (N - 1) / N of it per rank,
        #                  7 links x ~153 GB/s per GPU (MI355X_MICROARCH.md) at half of peak, + 0.4 ms of launches and packing (cfg4 on one device)
        #      EM        = replicated: this run's EM phase on every rank (no scaling);  sharded: iterations x (the sweep's fixed ~7 us + its
        #                  per-nonzero part / N + one all-reduce of M doubles: 10 us measured with a one-rank RCCL communicator
        #                  (profiles/r5_em_notes.md 6) + ~5 us per doubling of the ranks assumed for xGMI hops) -- `auto` takes the smaller
        tbl_bytes = C * 20 + L * 4
        link_GBs = 7 * 153.0 * 0.5
        sweep_us = em_loop_ms_per_iter * 1e3
        fixed_us = 6.9
        pred = {}
        for n in (2, 4, 8):
            build = info["t_insert_ms"] / n + (build_ms - info["t_insert_ms"])
            merge = 0.4 + tbl_bytes * (n - 1) / n / (link_GBs * 1e9) * 1e3 * 2
            em_rep = em_ms
            ar_us = 10.0 + 5.0 * math.log2(n)
            em_sh = (em_ms - st["iters"] * sweep_us * 1e-3) + st["iters"] * (fixed_us + max(0.0, sweep_us - fixed_us) / n + 10.0 + ar_us) * 1e-3
            step = build + merge + min(em_rep, em_sh) + info["t_efflen_ms"] + info["t_tpm_ms"]
            pred[str(n)] = {"ms_per_step": step, "class_build": build, "merge": merge, "em_replicated": em_rep, "em_sharded": em_sh,
                            "speedup_vs_this_run": (dt / a.steps * 1e3) / step}
        out["multi_gpu"] = {"predicted_ms": pred,
                            "predicted_from": "this N = 1 run's phase_ms + constants stated in bench.py / DESIGN.md 5 (xGMI 7 x 153 GB/s at half of peak, all-reduce of M "
                                              "doubles 10 us with one rank + 5 us per doubling, sharded iteration = one sweep kernel + fold + all-reduce); NOT measured on > 1 GPU"}
    if host_leg is not None:
        # BASELINE.md 3 / SURVEY 8d time the step from host-pinned hit lists: this is the figure that answers them
        out["value_host_pinned"] = host_leg.get("value")
        out["ms_per_step_host_pinned"] = host_leg.get("ms_per_step")
        out["host_pinned"] = host_leg

    # ---- N > 1 only, outside the timed region: what the EM-mode decision rests on, measured on THIS node (the sharded
    # EM pays one all-reduce of M doubles per iteration; `auto` keeps the EM replicated while that costs more than a whole sweep)
    if dist:
        mg = {"em_mode_of_the_timed_steps": info["em_mode"]}
        mg["em_mode"] = info["em_mode"]
        # proof of what the line ran on: the ranks of libsfgpu's OWN RCCL communicator (ncclCommCount; null: the torch.distributed
        # callback summed alphaOut -- gloo, or several ranks on one device), the devices, the backend
        try:
            ar = quant._allreduce() if hasattr(quant, "_allreduce") else None
            mg["rccl_ranks"] = ar.count() if hasattr(ar, "count") else None
        except Exception as e:
            mg["rccl_ranks"] = None; mg["rccl_ranks_error"] = repr(e)
        mg["backend"] = a.backend; mg["one_device"] = bool(a.one_device); mg["world"] = world
        try:
            buf = torch.zeros(M, dtype=torch.float64, device=dev)
            for _ in range(10):
                dist.all_reduce(buf)
            barrier()
            t1 = time.perf_counter()
            for _ in range(100):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            mg["allreduce_f64_M_us"] = (time.perf_counter() - t1) / 100 * 1e6
            mg["allreduce_bytes"] = M * 8
            mg["sweep_us_of_the_timed_problem"] = sweep_ms * 1e3
            if getattr(quant, "auto_measurement", None):
                mg["auto_decision"] = quant.auto_measurement       # what `auto` measured on its first run (libsfgpu's own RCCL communicator)
            mg["em_ms_of_the_timed_steps"] = em_ms
            if a.no_compare_em_modes:
                raise StopIteration
            other = "sharded" if info["em_mode"] == "replicated" else "replicated"
            q2 = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode=other)
            q2.run(ids, off, fl_counts=fl_counts, remaining_fl_ops=(0 if paired else 1))
            barrier()
            t1 = time.perf_counter()
            i2 = q2.run(ids, off, fl_counts=fl_counts, remaining_fl_ops=(0 if paired else 1))
            barrier()
            t2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            mg["other_em_mode"] = {"em_mode": i2["em_mode"], "ms_per_step": float(t2.item()) * 1e3, "em_ms": i2["t_em_ms"],
                                   "em_iters": i2["em_stats"]["iters"], "same_classes": bool(i2["n_classes"] == info["n_classes"])}
            mg["em_ms"] = {info["em_mode"]: em_ms, i2["em_mode"]: i2["t_em_ms"]}
            del q2
        except StopIteration:
            pass
        except Exception as e:                    # never take the headline line down
            mg["error"] = repr(e)
        out["multi_gpu"] = mg

    # ---- N = 1 only, outside the timed region: the posterior-sampling paths over the job's converged classes.  BASELINE
    # config 5 = CollapsedGibbsSampler, 1000 draws; the bootstrap is its sibling (doBootstrap).  SURVEY 8d's byte counts:
    #   Gibbs round, per sample and chain: B_round = 28 L + 8 C;   bootstrap draw: 8 C + iters x B_iter'
    if world == 1 and not a.no_sampling:
        samp = {}
        try:
            p = quant.problem
            nb = max(1, a.bootstrap_draws)
            p.bootstrap(3, seed=3, use_vbem=use_vbem)                               # warm-up: three draws, so that all three lanes' clones are planned here, once
            torch.cuda.synchronize(); t1 = time.perf_counter()
            rc_b, outb, itb = p.bootstrap(nb, seed=1, use_vbem=use_vbem)
            torch.cuda.synchronize(); dtb = time.perf_counter() - t1
            b_bytes = float(sum(8 * C + int(k) * b_iter for k in itb))
            samp["bootstrap"] = dict(draws=nb, rc=int(rc_b), ms_per_replicate=dtb / nb * 1e3, em_iters_mean=float(np.mean(itb)),
                                     roofline=dict(bound="hbm", achieved=b_bytes / dtb / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                                   frac=b_bytes / dtb / 1e9 / HBM_PEAK_GBS, bytes_per_draw=b_bytes / nb,
                                                   formula="8 C + iters x B_iter' per draw (SURVEY 8d)"))
            del outb
            out["bootstrap_ms_per_replicate"] = samp["bootstrap"]["ms_per_replicate"]
        except Exception as e:                    # never take the headline line down
            samp["bootstrap"] = dict(error=repr(e))
        try:
            n_draws = a.gibbs_draws

            def gibbs_leg(n_chains):
                # the first call of a process maps the 4 x nnz x chains bytes of chain state (38 GB for 1024 chains here) for the
                # first time: ~1 s, reported as first_call_seconds; the state then stays in the library's allocator cache
                torch.cuda.synchronize(); t0 = time.perf_counter()
                quant.gibbs(8, seed=5, n_chains=n_chains)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                g = quant.gibbs(n_draws, seed=1, n_chains=n_chains)
                torch.cuda.synchronize(); dtg = time.perf_counter() - t1
                rounds = (n_draws + n_chains - 1) // n_chains
                g_bytes = float(28 * L + 8 * C) * rounds * min(n_chains, n_draws)        # every chain runs `rounds` rounds
                sums_ok = bool((g.sum(1) == info_total_reads(info, quant)).all())
                del g
                return dict(draws=n_draws, chains=n_chains, rounds_per_chain=rounds, seconds=dtg, first_call_seconds=t1 - t0, sums_ok=sums_ok,
                            roofline=dict(bound="hbm", achieved=g_bytes / dtg / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                          frac=g_bytes / dtg / 1e9 / HBM_PEAK_GBS, bytes=g_bytes,
                                          formula="B_round = 28 L + 8 C per sample and chain (SURVEY 8d)"))

            samp["gibbs"] = gibbs_leg(1024)               # one initCountMap_ + one round per draw (rounds 1 and 2's figure)
            out["gibbs_1000_draws_s"] = samp["gibbs"]["seconds"] * (1000.0 / n_draws)
            # fewer, longer chains -- closer to the reference, whose chains are the TBB chunks of the sample range: 512 chains x 2 rounds
            samp["gibbs_512_chains"] = gibbs_leg(512)
        except Exception as e:
            samp["gibbs"] = dict(error=repr(e))
        out["sampling"] = samp

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(ref_len_np, ids, off, a.cpu_seconds, use_vbem)
        # CPU seconds for the full step = build at the sampled rate + EM iterations at the sampled
        # per-iteration cost scaled by nnz (the sweep is linear in nnz)
        cpu_step_s = R / cb["build_reads_per_s"] + st["iters"] * cb["em_ms_per_iter"] * 1e-3 * (L / max(cb["sample_nnz"], 1))
        out["cpu_baseline"] = {
            "value": R / cpu_step_s, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"oracle (C restatement, 1 thread): class build timed on the first {cb['sample_reads']} reads "
                      f"({cb['build_reads_per_s']:.3g} reads/s), EM timed for {cb['sample_em_iters']} iterations on that "
                      f"sample's {cb['sample_classes']} classes ({cb['em_ms_per_iter']:.3g} ms/iter), scaled to the full "
                      f"step ({R} reads, {st['iters']} iterations, nnz ratio {L / max(cb['sample_nnz'], 1):.2f})",
            "class_build_reads_per_s": cb["build_reads_per_s"], "em_ms_per_iter_sample": cb["em_ms_per_iter"],
            "host_cores_available": os.cpu_count(),
        }
        mt = cb.get("all_cores")
        if mt and "error" not in mt:
            mt_step_s = R / mt["build_reads_per_s"] + st["iters"] * mt["em_ms_per_iter"] * 1e-3 * (L / max(cb["sample_nnz"], 1))
            out["cpu_baseline_all_cores"] = {
                "value": R / mt_step_s, "unit": "reads/s", "cores": mt["cores"], "kind": "port",
                "sample": f"the same sample, best of {mt['tried']} threads per leg ({mt['host_cores_available']} available): per-thread "
                          f"class tables folded into one on {mt['build_threads']} threads ({mt['build_reads_per_s']:.3g} reads/s), "
                          f"{mt['em_iters']} EM iterations on {mt['cores']} threads with the classes cut into nnz-balanced ranges and "
                          f"CAS adds as in the reference ({mt['em_ms_per_iter']:.3g} ms/iter), scaled like cpu_baseline",
                "class_build_reads_per_s": mt["build_reads_per_s"], "em_ms_per_iter_sample": mt["em_ms_per_iter"],
                "host_cores_available": mt["host_cores_available"],
            }
        elif mt:
            out["cpu_baseline_all_cores"] = mt
        out["parity_vs_cpu"] = cb["parity"]     # the metric's "TPM delta vs CPU ref", on the baseline's sample
        try:
            look_rate, em_s, full_parity = cpu_baseline_full(quant.last_vec, ref_len_np, ids, off, R, use_vbem, fl_counts)
            if isinstance(out.get("parity_vs_cpu"), dict) and full_parity:
                out["parity_vs_cpu"]["full_problem"] = full_parity
                out["parity_vs_cpu"]["full_problem_max_rel"] = full_parity.get("max_rel_numreads")
            full_s = R / look_rate + st["iters"] * em_s
            out["cpu_baseline"].update({
                "value": R / full_s,
                "sample": f"oracle (C restatement, 1 thread) on the job's own classes: class build = 2000000 reads from the middle of the "
                          f"read stream looked up in a table that already holds all {C} classes ({look_rate:.3g} reads/s; the sample-start, "
                          f"insert-heavy rate was {cb['build_reads_per_s']:.3g}), EM = 10 timed iterations over all {C} classes / {L} nonzeros "
                          f"({em_s * 1e3:.3g} ms each) x the {st['iters']} iterations the job needs; no extrapolation by size ratios",
                "class_build_reads_per_s": look_rate, "em_ms_per_iter_full_problem": em_s * 1e3,
                "value_from_2M_read_sample": R / cpu_step_s})
        except Exception as e:                    # keep the sample-based figure
            out["cpu_baseline"]["full_problem_error"] = repr(e)
    # GPU / CPU ratios for BOTH residencies (reported, not a target: the roofline fractions say how good the kernels are)
    if rank == 0 and "cpu_baseline" in out:
        sp = {}
        for key in ("cpu_baseline", "cpu_baseline_all_cores"):
            cv = out.get(key, {}).get("value") if isinstance(out.get(key), dict) else None
            if cv:
                sp[key] = {"cores": out[key].get("cores"), "hbm_resident": value / cv,
                           "host_pinned": (out["value_host_pinned"] / cv) if out.get("value_host_pinned") else None}
        out["speedup_vs_cpu"] = sp
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
