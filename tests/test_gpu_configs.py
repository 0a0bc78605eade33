"""GPU tests of BASELINE.json's configs[3] and configs[4] at their quoted sizes on ONE GPU.

cfg4 ("8xMI355X: same 400M PE / 200k-txp, eq-classes sharded across GPUs, per-iter all-reduce of alpha"): eight ranks share
the test box's GPU (gloo between them -- RCCL refuses several ranks on one device), rank r holds reads [r R/8, (r+1) R/8) of
the very experiment a single process runs; the merged class table must equal the single-process table and the sharded EM
(classes cut into 8 nnz-balanced slices, SUM all-reduce of alphaOut between sweep and update, every iteration) must give the
single-GPU alpha after the same number of iterations.

cfg5 ("1000 Gibbs draws over the converged eq-classes"): the draws at size through size-independent properties, plus
distributional parity with the oracle's sequential sampleRound_ on a problem whose round needs several phases."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O

pytestmark = pytest.mark.gpu

CFG3 = (200_000, 4_000_000, 400_000_000)
# The eight PROCESSES run only CFG4_ITERS iterations of the sharded loop (both sides: min_iter = max_iter): every iteration costs
# the eight ranks a gloo all-reduce through the host, and nine processes time-slicing one device are erratic -- the same test took
# 82 s on one box and 863 s on the next in round 4 (80 iterations; 20 s to 8 minutes in round 3), which alone would overrun the
# suite's limit.  What the processes are here for is the transport: shards -> owner-partitioned merge over eight ranks -> the table
# of the single-process run, then the all-reduce between sweep and update on eight ranks -> the single-GPU alpha.  That the sharded
# loop stops where the single-GPU loop stops, at full size and run to CONVERGENCE, is asserted without the processes by
# test_cfg4_eight_shards_merged_and_swept_in_one_process below (and on two / three ranks by tests/test_gpu_distributed.py).
# SFGPU_CFG4_FULL=1 lifts the cut for the processes too (212 = 212, ~9 minutes, profiles/r3_cfg4_full.txt).
# tests/conftest.py runs this file's cfg4 test FIRST among the GPU tests (a device that earlier tests have used makes it slower still).
CFG4_FULL = bool(os.environ.get("SFGPU_CFG4_FULL"))
CFG4_MAX_ITER = 10000 if CFG4_FULL else 6
CFG4_MIN_ITER = 50 if CFG4_FULL else 6
CFG4_POLL = 16 if CFG4_FULL else 2


def _fl_counts():
    i = np.arange(1000.0)
    return np.floor(np.exp(-0.5 * ((i - 200.0) / 80.0) ** 2) * 4000 + 0.5).astype(np.uint32)


def _cfg4_worker(rank, world, port, sizes, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sailfish_amd as sf
        from sailfish_amd import distributed as sfd, synth
        M, P, R = sizes
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        ref_len = synth.transcript_lengths(M, device=dev)
        poff, pids = synth.label_pool(M, P, device=dev)
        ids, off = synth.reads_slice(poff, pids, R * rank // world, R * (rank + 1) // world, seed=7, device=dev)
        del poff, pids
        sopt = sf.SailfishOpts(useVBOpt=True)
        exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.cpu().numpy().view(np.uint32), device=dev), sopt)
        q = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode="sharded", poll_every=CFG4_POLL, max_iter=CFG4_MAX_ITER, min_iter=CFG4_MIN_ITER)
        info = q.run(ids, off, fl_counts=_fl_counts(), remaining_fl_ops=0)
        v = q.last_vec
        np.save(os.path.join(outdir, f"alpha{rank}.npy"), exp.transcripts().estCount.cpu().numpy())
        if rank == 0:
            np.savez(os.path.join(outdir, "table.npz"), rowptr=v.rowptr.cpu().numpy(), ids=v.ids.cpu().numpy(), counts=v.counts.cpu().numpy(),
                     meta=np.array([info["n_classes"], info["nnz"], exp.numMappedFragments(), info["em_stats"]["iters"],
                                    int(info["em_stats"]["converged"]), int(info["em_mode"] == "sharded"), q.problem.C]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.skipif(not os.environ.get("SFGPU_CFG4_PROCS"), reason="nine processes time-slicing one device: 82 s, 460 s and 863 s on three boxes in round 4 "
                    "(set SFGPU_CFG4_PROCS=1 to run it); the default suite covers cfg4 at full size in ONE process, see "
                    "test_cfg4_eight_shards_merged_and_swept_in_one_process")
def test_cfg4_eight_ranks_share_the_gpu(gpu):
    import sailfish_amd as sf
    from sailfish_amd import distributed as sfd, synth
    M, P, R = CFG3
    world = 8
    outdir = tempfile.mkdtemp(prefix="cfg4_")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_cfg4_worker, args=(r, world, port, CFG3, outdir)) for r in range(world)]
    for p in procs:
        p.start()
    # meanwhile: the single-process run over all R reads (the children take a while to import torch)
    ref_len = synth.transcript_lengths(M, device=gpu)
    poff, pids = synth.label_pool(M, P, device=gpu)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=gpu)
    del poff, pids
    sopt = sf.SailfishOpts(useVBOpt=True)
    exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.cpu().numpy().view(np.uint32), device=gpu), sopt)
    q1 = sfd.DistributedQuant(exp, sopt, max_iter=CFG4_MAX_ITER, min_iter=CFG4_MIN_ITER)
    info1 = q1.run(ids, off, fl_counts=_fl_counts(), remaining_fl_ops=0)
    v1 = q1.last_vec
    a1 = exp.transcripts().estCount.cpu().numpy()
    del ids, off
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    tab = np.load(os.path.join(outdir, "table.npz"))
    n_classes, nnz, n_mapped, iters, conv, sharded, c_local = [int(x) for x in tab["meta"]]
    # the merged table IS the single-process table: same classes, same canonical order, same counts (integer work: bit exact)
    assert n_mapped == R == exp.numMappedFragments() and n_classes == info1["n_classes"] and nnz == info1["nnz"]
    assert np.array_equal(tab["rowptr"], v1.rowptr.cpu().numpy())
    assert np.array_equal(tab["ids"], v1.ids.cpu().numpy())
    assert np.array_equal(tab["counts"], v1.counts.cpu().numpy())
    # sharded EM: each rank swept ~1/8 of the classes, and the loop stopped where the single-GPU loop stops
    assert sharded == 1 and 0 < c_local < n_classes // 4
    assert iters == info1["em_stats"]["iters"] and conv == int(info1["em_stats"]["converged"]), (iters, info1["em_stats"])
    assert iters == CFG4_MAX_ITER or (CFG4_FULL and conv == 1 and iters > 80), (iters, conv)
    print(f"cfg4: {world} ranks, sharded EM stopped at iteration {iters} (single GPU: {info1['em_stats']['iters']}), converged={conv}")
    alphas = [np.load(os.path.join(outdir, f"alpha{r}.npy")) for r in range(world)]
    for a in alphas[1:]:
        assert np.array_equal(a, alphas[0])                        # the all-reduce leaves every rank with the same bits
    nz = a1 > 0
    assert np.array_equal(alphas[0] > 0, nz)
    assert float(np.max(np.abs(alphas[0][nz] - a1[nz]) / a1[nz])) < 1e-9


def test_cfg4_eight_shards_merged_and_swept_in_one_process(gpu):
    """BASELINE configs[3] at full size, everything the eight ranks do on their devices, in ONE process (nine processes sharing
    the test box's GPU are erratic -- see the skip above; the transport itself is covered by the gloo tests on CPU and by the two-
    and three-rank tests of tests/test_gpu_distributed.py):
      * class tables: cfg3's 400 M reads cut into eight shards, a builder per shard, then the owner-partitioned merge through
        the C ABI pieces the ranks call -- pack_by_owner (8 blocks per shard), fold at the owner (sfgpu_eq_add_block_device),
        export_block, merge_disjoint -- must give the single-process table: same classes, canonical order, counts (bit exact);
      * EM: the merged classes cut into eight nnz-balanced slices (the cuts DistributedQuant makes), one sfgpu_em handle per
        slice, per iteration { sweep on every slice, SUM of the eight alphaOut vectors, update on every slice }, run to
        CONVERGENCE: the sharded loop must stop at the single-GPU loop's iteration with the single-GPU alpha (<= 1e-9), and
        every slice must hold the same bits."""
    import sailfish_amd as sf
    from sailfish_amd import distributed as sfd, synth
    M, P, R = CFG3
    world = 8
    ref_len = synth.transcript_lengths(M, device=gpu)
    poff, pids = synth.label_pool(M, P, device=gpu)
    eng = sfd.HipEngine(gpu)
    # ---- the shards' tables, packed by owner
    packed = []
    for r in range(world):
        ids_r, off_r = synth.reads_slice(poff, pids, R * r // world, R * (r + 1) // world, seed=7, device=gpu)
        b = eng.new_builder(); b.start(); b.add_batch(ids_r, off_r); b.finish()
        packed.append(eng.pack_by_owner(b.eqVec(), world))
        b.close(); del ids_r, off_r
    # ---- every owner folds what the shards sent it; the disjoint partitions are merged into the canonical order
    parts = []
    for o in range(world):
        pb = eng.new_builder(); pb.start()
        for r in range(world):
            buf, sizes = packed[r]
            start = sum(sfd.block_bytes(*sizes[d]) for d in range(o))
            c, l = sizes[o]
            eng.fold_block(pb, buf[start:start + sfd.block_bytes(c, l)], c, l)
        pb.finish()
        parts.append(pb.eqVec()); pb.close()
    del packed
    merged = eng.merge_disjoint([eng.export_block(p) for p in parts], [(int(p.size()), int(p.ids.numel())) for p in parts])
    assert merged is not None
    del parts
    # ---- the single-process run over all reads
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=gpu)
    del poff, pids
    sopt = sf.SailfishOpts(useVBOpt=True)
    exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.cpu().numpy().view(np.uint32), device=gpu), sopt)
    q1 = sfd.DistributedQuant(exp, sopt)
    info1 = q1.run(ids, off, fl_counts=_fl_counts(), remaining_fl_ops=0)
    del ids, off
    v = q1.last_vec
    assert int(merged.size()) == info1["n_classes"] and int(merged.ids.numel()) == info1["nnz"] and merged.total_reads == R
    assert torch.equal(merged.rowptr, v.rowptr) and torch.equal(merged.ids, v.ids) and torch.equal(merged.counts, v.counts)
    a1 = exp.transcripts().estCount.clone()
    it1 = info1["em_stats"]["iters"]
    assert info1["em_stats"]["converged"] and it1 > 80
    length = exp.transcripts().EffectiveLength
    v = merged                              # (the EM below runs on the MERGED table)
    rp_cpu = (v.rowptr.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    cuts = sfd.nnz_balanced_slices(rp_cpu, world)
    probs = []
    for r in range(world):
        c0, c1 = cuts[r], cuts[r + 1]
        j0, j1 = int(rp_cpu[c0]), int(rp_cpu[c1])
        rp_loc = ((v.rowptr[c0:c1 + 1].to(torch.int64) & 0xFFFFFFFF) - j0).to(torch.int32)
        probs.append(sf.EMProblem(length, rp_loc, v.ids[j0:j1], v.counts[c0:c1], exp.numMappedFragments()))
    kw = dict(use_vbem=True, tol=0.01, min_iter=50, max_iter=10000)
    for p in probs:
        p.begin(**kw)
    outs = [p.alpha_out_view() for p in probs]

    def all_reduce():
        tot = outs[0].clone()
        for o in outs[1:]:
            tot += o                      # (rank order: the same sum on every "rank")
        for o in outs:
            o.copy_(tot)

    all_reduce()                          # union of the active sets
    for p in probs:
        p.init()
    done = False
    while not done:
        for _ in range(16):
            for p in probs:
                p.sweep()
            all_reduce()
            for p in probs:
                p.update()
        flags = [p.poll() for p in probs]
        assert len({f[0] for f in flags}) == 1 and len({f[1]["iters"] for f in flags}) == 1      # the ranks leave the loop together
        done = flags[0][0]
    res = [p.finish() for p in probs]
    assert all(rc == 0 for rc, _ in res)
    iters = res[0][1]["iters"]
    print(f"cfg4 in one process: 8 shards merged = the single table; sharded loop stopped at iteration {iters} (single GPU: {it1})")
    assert iters == it1 and all(st["iters"] == it1 and st["converged"] for _, st in res)
    for p in probs[1:]:
        assert torch.equal(p.alpha, probs[0].alpha)
    nz = a1 > 0
    assert torch.equal(probs[0].alpha > 0, nz)
    assert float(((probs[0].alpha[nz] - a1[nz]).abs() / a1[nz]).max()) < 1e-9
    for p in probs:
        p.close()


def test_cfg5_thousand_draws_over_cfg3_classes(gpu):
    """BASELINE configs[4]: 1000 Gibbs draws (1024 chains) over the converged classes of cfg3"""
    import sailfish_amd as sf
    from sailfish_amd import _lib, synth
    M, P, R = CFG3
    ref_len = synth.transcript_lengths(M, device=gpu)
    poff, pids = synth.label_pool(M, P, device=gpu)
    ids, off = synth.reads_slice(poff, pids, 0, R, seed=7, device=gpu)
    del poff, pids
    eq = sf.EquivalenceClassBuilder(device=gpu); eq.start(); eq.add_batch(ids, off); eq.finish(); v = eq.eqVec()
    del ids, off
    length = ref_len.to(torch.float64)
    p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
    rc, st = p.optimize(use_vbem=True)
    assert rc == 0 and st["converged"]
    logs = []
    _lib.set_logger(lambda lvl, msg: logs.append(msg))
    try:
        rc, g = sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 1000, n_chains=1024, seed=1)
        assert rc == 0 and tuple(g.shape) == (1000, M) and g.dtype == torch.int32
        plan = [m for m in logs if "gibbs:" in m][-1]              # "gibbs: 1024 chains, T tiles in K phases, W wide classes"
        K = int(plan.split(" tiles in ")[1].split()[0])
        assert K >= 2, plan                                         # cfg3's classes need a multi-phase round
        # every draw distributes every read: sum over transcripts = numMapped, no negative count
        assert bool((g.sum(1, dtype=torch.int64) == R).all()) and int(g.min()) >= 0
        # transcripts that appear in no class never receive a read
        present = torch.zeros(M, dtype=torch.bool, device=gpu); present[v.ids.long()] = True
        assert int(g[:, ~present].abs().sum()) == 0
        # a transcript's count can never exceed what its classes hold
        cap = torch.zeros(M, dtype=torch.int64, device=gpu)
        rp = v.rowptr.long() & 0xFFFFFFFF
        cap.index_add_(0, v.ids.long(), torch.repeat_interleave(v.counts.long(), rp[1:] - rp[:-1]))
        assert bool((g.max(0).values.long() <= cap).all())
        # the draws scatter around the point estimate for the well-covered transcripts
        a = p.alpha
        top = torch.argsort(a, descending=True)[:2000]
        mean = g[:, top].double().mean(0)
        assert float(((mean - a[top]).abs() / a[top]).median()) < 0.05
        # reproducible from the seed (incl. the order in which the wide classes are visited)
        rc, g2 = sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 8, n_chains=1024, seed=1)
        assert rc == 0 and torch.equal(g2, g[:8])
        rc, g3 = sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, eq.total_reads, 8, n_chains=1024, seed=2)
        assert rc == 0 and not torch.equal(g3, g[:8])
    finally:
        _lib.set_logger(None)
        p.close()


@pytest.mark.parametrize("n_wide_labels,heavy", [(3, False), (150, False), (600, False), (3, True)])
def test_gibbs_multi_phase_round_matches_the_sequential_sampler(gpu, n_wide_labels, heavy):
    """per-transcript mean and spread of the phase-parallel sampler vs the oracle's sequential sampleRound_
    (src/CollapsedGibbsSampler.cpp:113-184) on a problem with thousands of classes, a multi-phase round and wide classes.
    3 wide labels: visited one after another by one launch; 150 wide labels in groups of 30 that share a far transcript (the
    pseudogene every read of a gene also hits) and overlap otherwise: five THIN components (30 classes need 30 colours), each
    walked by one wavefront per 64 chains; 600 wide labels linked into one long chain (label i shares a transcript with label
    i + 1, and a far one with two others): one component, a few colours of ~200 classes, one launch per colour.
    heavy: the first 100 reads 2000 times more -- up to 100 classes of > 1500 reads among the light ones: the phase kernel's two forms
    (BINV only at six wavefronts per SIMD; with BTPE at four) share the tiles."""
    import sailfish_amd as sf
    from sailfish_amd import _lib, synth
    M, P, R = 3000, 6000, 120_000
    ref_len, ids, off = synth.workload(M, P, R)
    # labels spanning the whole transcript range ("wide" classes: visited after the phases)
    rng = np.random.default_rng(5)
    if n_wide_labels == 3:
        wide = [np.sort(rng.choice(M, 6, replace=False)).astype(np.int32) for _ in range(3)]
        reps = 400
    elif n_wide_labels == 150:
        wide = []
        for i in range(n_wide_labels):
            loc = rng.choice(400, 5, replace=False) + 13 * (i // 30)                 # neighbours overlap locally, too
            wide.append(np.unique(np.concatenate([loc, [M - 1 - i // 30]])).astype(np.int32))
        reps = 40
    else:
        wide = [np.array([i, i + 1, 2800 + i % 190], np.int32) for i in range(n_wide_labels)]       # (span > 2048: wide)
        reps = 20
    if heavy:
        n0 = int(off[100])
        ids = torch.cat([ids, ids[:n0].repeat(2000)])
        off = torch.cat([off, int(off[-1]) + (off[1:101].repeat(2000) + n0 * torch.arange(2000, dtype=off.dtype).repeat_interleave(100))])
        R += 200_000
    extra_ids = np.concatenate([np.tile(w, reps) for w in wide])
    extra_len = np.concatenate([np.full(reps, len(w)) for w in wide])
    extra_off = int(off[-1]) + np.cumsum(extra_len)
    ids = torch.cat([ids, torch.from_numpy(extra_ids)])
    off = torch.cat([off, torch.from_numpy(extra_off.astype(np.int32))])
    R += reps * len(wide)
    eq = sf.EquivalenceClassBuilder(device=gpu); eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); eq.finish(); v = eq.eqVec()
    assert eq.total_reads == R and eq.n_classes > 2000
    eff = O.efflen_smoothed(ref_len.numpy().view(np.uint32), O.cf_gaussian())
    length = torch.from_numpy(eff).to(gpu)
    p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, R)
    rc, st = p.optimize()
    assert rc == 0
    logs = []
    _lib.set_logger(lambda lvl, msg: logs.append(msg))
    try:
        n_chains, rounds, burn = 256, 40, 12
        rc, g = sf.gibbs_sample(length, p.mass, v.rowptr, v.ids, v.counts, R, n_chains * rounds, n_chains=n_chains, seed=21)
    finally:
        _lib.set_logger(None)
    assert rc == 0
    plan = [m for m in logs if "gibbs:" in m][-1]
    K = int(plan.split(" tiles in ")[1].split()[0]); n_wide = int(plan.split(" phases, ")[1].split()[0])
    assert K >= 2 and n_wide >= 3, plan
    assert (int(plan.split(" chains, ")[1].split()[0]) > 0) == heavy, plan          # "... chains, H heavy tiles, T tiles in K phases, ..."
    # "... W wide classes: A in B colours, C in D thin components"
    assert ("colours" in plan) == (n_wide_labels > 64), plan
    if n_wide_labels > 64:
        tail = plan.split(": ")[-1]
        n_col, n_colours = int(tail.split(" in ")[0]), int(tail.split(" in ")[1].split()[0])
        n_thin = int(tail.split(", ")[1].split(" in ")[0]) if "thin" in tail else 0
        assert n_col + n_thin == n_wide, plan
        if n_wide_labels == 150:
            assert n_thin >= 150, plan                                  # 30 classes that need 30 colours: walked, not coloured
        else:
            assert n_col >= 600 and 2 <= n_colours <= 12, plan          # a path with a few far links: a handful of colours
    # Chains are sticky: with priorAlpha = 1e-8 a transcript whose count reaches 0 practically never gets a read back, so
    # chains settle into different supports and ONE sequential chain is not comparable with an average over chains.
    # Both samplers are therefore run as many independent chains from the same start (initCountMap_ from the EM's mass)
    # and compared through their per-chain late means: same rounds, z-test with the between-chain variances.
    g = g.cpu().numpy().reshape(rounds, n_chains, M)                # sample s = chain s % n_chains after s // n_chains + 1 rounds
    assert np.all(g.sum(2) == R)
    gm = g[burn:].mean(0)                                           # [n_chains, M]
    rp, ii, cc, _ = v.to_numpy()
    mass = p.mass.cpu().numpy()
    n_oc = 64
    om = []
    for c in range(n_oc):
        orc, og = O.gibbs(eff, mass, rp.astype(np.uint64), ii, cc, R, rounds, seed=1000 + c)
        assert orc == 0 and np.all(og.sum(1) == R)
        om.append(og[burn:].mean(0))
    om = np.stack(om)                                               # [n_oc, M]
    mg, mo = gm.mean(0), om.mean(0)
    vg, vo = gm.var(0, ddof=1), om.var(0, ddof=1)
    # transcripts whose count is the same in every chain of both samplers (only singleton classes, or none): exact
    fixed = (vg == 0) & (vo == 0)
    assert fixed.sum() > 0 and np.array_equal(mg[fixed], mo[fixed])
    z = (mg - mo) / np.sqrt(vg / n_chains + vo / n_oc + 0.02)
    assert float(np.mean(np.abs(z) > 4.0)) < 0.01 and float(np.max(np.abs(z))) < 8.0, (np.sort(np.abs(z))[-10:],)
    assert abs(float(np.mean(z))) < 0.15                            # no systematic shift
    # spread between chains: transcripts that really differ from chain to chain do so alike in both samplers
    mv = vo > 9.0
    assert mv.sum() > 50
    ratio = np.sqrt(vg[mv] / vo[mv])
    assert 0.8 < float(np.median(ratio)) < 1.25 and float(np.mean((ratio > 0.5) & (ratio < 2.0))) > 0.95
    # the share of chains in which a transcript ends without reads (the sticky state) agrees
    wt = np.unique(np.concatenate(wide))                           # members of the wide classes are part of it
    assert float(np.max(np.abs(z[wt]))) < 6.0
