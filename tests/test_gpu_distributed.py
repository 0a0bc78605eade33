"""GPU test of the multi-rank driver with the PRODUCT engine (HipEngine / libsfgpu): two ranks share
the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one device), so the whole
HIP data path of the N > 1 flow runs for real: per-rank class build, all-gather + weighted upsert
merge (sfgpu_eq_add_weighted_device), replicated and sharded EM with the per-iteration all-reduce of
alphaOut -- checked against the oracle over the union of both shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, mode, vb, out, merge_mode="auto"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sailfish_amd as sf
        from sailfish_amd import distributed as sfd, synth
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        M, P, R = 4000, 15000, 150_000
        ref_len = synth.transcript_lengths(M)
        poff, pids = synth.label_pool(M, P)
        ids, off = synth.reads_from_pool(poff, pids, R, seed=7 + 1000 * rank)
        sopt = sf.SailfishOpts(useVBOpt=vb)
        exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.numpy().view(np.uint32), device=dev), sopt)
        q = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode=mode, poll_every=9, merge_mode=merge_mode)
        info = q.run(ids.to(dev), off.to(dev))
        t = exp.transcripts()
        if True:                                       # posterior draws are split over the ranks; after a sharded run
            bs = q.bootstrap(5, seed=3)                # the samplers use a problem over ALL merged classes
            gs = q.gibbs(7, seed=3, n_chains=64)
            N = exp.numMappedFragments()
            assert bs.shape == (5, M) and gs.shape == (7, M)
            # EM conserves the reads; VBEM adds the prior (0.01 per transcript) and the truncation removes at most as much
            assert float((bs.sum(1) - float(N)).abs().max()) <= (0.011 * M if vb else 0.0) + 1e-6 * N
            assert bool((gs.sum(1) == N).all())
        out.put((rank, info["em_mode"], info["n_classes"], info["nnz"], exp.numMappedFragments(), info["em_stats"]["iters"],
                 t.estCount.cpu().numpy(), info["tpm"].cpu().numpy(), ids.numpy().copy(), off.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("mode,vb,merge,world", [("replicated", False, "auto", 2), ("sharded", False, "auto", 2), ("sharded", True, "auto", 2),
                                                 ("sharded", False, "owner", 3), ("replicated", True, "owner", 3)])
def test_two_ranks_on_one_gpu_match_the_oracle(gpu, mode, vb, merge, world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, vb, out, merge)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from sailfish_amd import synth
    M = 4000
    ref_len = synth.transcript_lengths(M).numpy().view(np.uint32)
    b = O.EqBuilder()
    for r in res:
        b.add_batch(r[8].view(np.uint32), r[9].view(np.uint32).astype(np.uint64))
    rp, ii, cc, hh = b.finish()
    eff = O.efflen_smoothed(ref_len, O.cf_gaussian())
    rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, b.total_reads, use_vbem=vb)
    ot = O.tpm(oa, eff, b.total_reads)
    for r in res:
        assert r[1] == mode and r[2] == b.n_classes and r[3] == b.nnz and r[4] == b.total_reads == 150_000 * world
        assert r[5] == ost["iters"]
        nz = oa > 0
        assert np.array_equal(r[6] > 0, nz)
        assert np.max(np.abs(r[6][nz] - oa[nz]) / oa[nz]) < 1e-9
        assert np.max(np.abs(r[7][nz] - ot[nz]) / ot[nz]) < 1e-9


def _bias_worker(rank, world, port, mode, which, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sailfish_amd as sf
        from sailfish_amd import distributed as sfd, synth
        from test_distributed_cpu import _bias_inputs
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        M, P, R = 300, 900, 5000
        ref_len = synth.transcript_lengths(M)
        poff, pids = synth.label_pool(M, P)
        ids, off = synth.reads_from_pool(poff, pids, R, seed=3 + 1000 * rank)
        sopt = sf.SailfishOpts(biasCorrect=which == "seq", gcBiasCorrect=which == "gc")
        exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.numpy().view(np.uint32), device=dev), sopt)
        seq, soff, fl, rb, og = _bias_inputs(M, ref_len.numpy().view(np.uint32))
        exp.setSequences(seq, soff)
        exp.readBias()[:] = rb; exp.observedGC()[:] = og; exp.addNumFwd(55); exp.addNumRC(45)
        q = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode=mode, poll_every=7, tol=1e-4)
        info = q.run(ids.to(dev), off.to(dev))
        t = exp.transcripts()
        out.put((rank, info["em_stats"]["iters"], q.recomputes, t.estCount.cpu().numpy(), t.EffectiveLength.cpu().numpy(),
                 exp.expectedSeqBias().copy(), exp.expectedGCBias().copy(), ids.numpy().copy(), off.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,which", [("sharded", "seq"), ("sharded", "gc"), ("replicated", "seq")])
def test_two_ranks_on_one_gpu_with_the_bias_hook(gpu, mode, which):
    """doBiasCorrect across ranks with the product engine: lowered stop bounds, sfgpu_bias_update on the replicated alpha,
    broadcast, sfgpu_em_rebase (sharded); sfgpu_em_optimize_bias on every rank (replicated)"""
    from test_distributed_cpu import _bias_inputs
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bias_worker, args=(r, 2, port, mode, which, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from sailfish_amd import synth
    M = 300
    ref_len = synth.transcript_lengths(M).numpy().view(np.uint32)
    b = O.EqBuilder()
    for r in res:
        b.add_batch(r[7].view(np.uint32), r[8].view(np.uint32).astype(np.uint64))
    rp, ii, cc, hh = b.finish()
    eff0 = O.efflen_smoothed(ref_len, O.cf_gaussian())
    seq, soff, fl, rb, og = _bias_inputs(M, ref_len)
    # the driver's effective-length stage (single-end synth reads) stores the Gaussian-prior FLD in the experiment
    bm = O.make_bias_model(seq, soff.astype(np.uint64), ref_len, eff0, O.fld_gaussian_counts().astype(np.uint32), rb, og, num_fwd=55, num_rc=45,
                           seq_bias=which == "seq", gc_bias=which == "gc")
    rc, oa, om, oeff, oes, oeg, onr, ost = O.em_optimize_bias(bm, eff0, rp, ii, cc, b.total_reads, tol=1e-4)
    assert rc == 0 and onr >= 1
    for r in res:
        assert abs(int(r[1]) - int(ost["iters"])) <= 1 and r[2] == onr
        np.testing.assert_allclose(r[4], oeff, rtol=1e-6)
        big = oa > 1e-3
        np.testing.assert_allclose(r[3][big], oa[big], rtol=1e-4)
        np.testing.assert_allclose(r[5], oes, rtol=1e-6); np.testing.assert_allclose(r[6], oeg, rtol=1e-6)
    if mode == "sharded":
        assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])     # broadcast lengths: ranks bit-identical


def _a2a_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        send = torch.arange(0, 4096, dtype=torch.int64, device=dev).view(torch.uint8)[: 4096 * 8 - 8]
        recv = torch.empty_like(send)
        n = int(send.numel())
        dist.all_to_all_single(recv, send, output_split_sizes=[n], input_split_sizes=[n])
        torch.cuda.synchronize()
        out.put(bool(torch.equal(recv, send)))
    finally:
        dist.destroy_process_group()


def test_rccl_all_to_all_single_takes_byte_blocks(gpu):
    """the exchange primitive of the owner-partitioned merge (all_to_all_single of uint8 blocks with explicit split
    sizes) on the RCCL backend -- one rank is all this box can offer, which still covers the call and its dtypes"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_a2a_worker, args=(_free_port(), out))
    p.start()
    assert out.get(timeout=240) is True
    p.join(60)
    assert p.exitcode == 0


def test_rccl_comm_drives_the_sharded_loop(gpu):
    """libsfgpu's own RCCL communicator (sailfish_amd/comm.py, csrc/comm.hip: librccl bound at run time) as the all-reduce
    of sfgpu_em_optimize_sharded.  One GPU allows a communicator of ONE rank only (RCCL refuses two ranks on a device): the
    sum over one rank is the identity, so the sharded loop -- piecewise kernels, ncclAllReduce enqueued on the loop's stream
    between sweep and update, no Python between iterations -- must stop where optimize() stops, with its alpha."""
    import sailfish_amd as sf
    from sailfish_amd import comm, synth
    if not comm.available():
        pytest.fail("librccl.so could not be loaded on a GPU box")
    M = 5000
    ref_len, ids, off = synth.workload(M, 20000, 400_000)
    eq = sf.EquivalenceClassBuilder(device=gpu); eq.start(); eq.add_batch(ids.to(gpu), off.to(gpu)); eq.finish(); v = eq.eqVec()
    length = ref_len.to(gpu).to(torch.float64)
    c = comm.Comm(1, 0, comm.Comm.unique_id(), gpu)
    try:
        t = torch.arange(1000, dtype=torch.float64, device=gpu)
        c.all_reduce(t); torch.cuda.synchronize()
        assert torch.equal(t, torch.arange(1000, dtype=torch.float64, device=gpu))
        assert c.time_all_reduce(M, 20) > 0.0
        for vb in (False, True):
            p = sf.EMProblem(length, v.rowptr, v.ids, v.counts, eq.total_reads)
            rc, st = p.optimize(use_vbem=vb)
            want = p.alpha.clone()
            rc2, st2 = p.optimize_sharded(c, poll_every=7, use_vbem=vb)
            assert rc == 0 and rc2 == 0 and st2["iters"] == st["iters"] and st2["converged"] == st["converged"]
            nz = want > 0
            assert torch.equal(p.alpha > 0, nz)
            assert float(((p.alpha[nz] - want[nz]).abs() / want[nz]).max()) < 1e-9
            # round 5: the same loop with ONE sweep kernel per iteration (the update at the head of the next sweep, from the
            # all-reduced vector): this midsize table's tiles crowd one window (cover lists), a local table's do not -- both forms
            # must stop where optimize() stops, with its alpha
            for table in ("midsize", "local"):
                if table == "local":
                    rl2, ids2, off2 = synth.workload(60_000, 200_000, 2_000_000)
                    eq2 = sf.EquivalenceClassBuilder(device=gpu); eq2.start(); eq2.add_batch(ids2.to(gpu), off2.to(gpu)); eq2.finish(); v2 = eq2.eqVec()
                    pf = sf.EMProblem(rl2.to(gpu).to(torch.float64), v2.rowptr, v2.ids, v2.counts, eq2.total_reads)
                    rcw, stw = pf.optimize(use_vbem=vb); want_f = pf.alpha.clone()
                else:
                    pf, stw, want_f = p, st, want
                assert pf.sharded_fused_ok()
                pf.set_sharded_fused(True)
                rc4, st4 = pf.optimize_sharded(c, poll_every=7, use_vbem=vb)
                pf.set_sharded_fused(False)
                assert rc4 == 0 and st4["iters"] == stw["iters"] and st4["converged"] == stw["converged"] and st4["fused"]
                nzf = want_f > 0
                assert torch.equal(pf.alpha > 0, nzf)
                assert float(((pf.alpha[nzf] - want_f[nzf]).abs() / want_f[nzf]).max()) < 1e-9
                assert abs(st4["max_rel_diff"] - stw["max_rel_diff"]) <= 1e-9 * abs(stw["max_rel_diff"])
                if pf is not p: pf.close()
            # and with a Python callable in the callback's place (what the gloo dry runs use)
            calls = []
            rc3, st3 = p.optimize_sharded(lambda buf: calls.append(buf.numel()), poll_every=7, use_vbem=vb)
            assert rc3 == 0 and st3["iters"] == st["iters"] and len(calls) >= st["iters"] + 1 and set(calls) == {M}
            p.close()
    finally:
        c.close()


def test_bench_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2` without a launcher around it starts its two ranks itself (torch.distributed.run, one process per
    rank; here both on the one device, gloo) and prints ONE line that says what it ran on: n_gpus, the EM mode, both modes' times,
    the ranks of the library's RCCL communicator (null on this dry run: gloo sums alphaOut)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo", "--workload", "small",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-pinned", "--no-sampling"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["world"] == 2 and mg["em_mode"] in ("replicated", "sharded") and "rccl_ranks" in mg and mg["one_device"]
    assert set(mg["em_ms"]) == {"replicated", "sharded"}
