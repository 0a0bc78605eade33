"""world_size-2 gloo test of the multi-GPU control flow (sailfish_amd/distributed.py) on CPU.

The product engine (HipEngine) needs GPUs; here the same driver runs with a checker engine backed
by the CPU oracle, so the exchange (all-gather + weighted merge), the nnz-balanced class slicing
and the per-iteration all-reduce of alphaOut are exercised for real across two processes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O


class _Vec:
    def __init__(self, rowptr, ids, counts, total):
        self.rowptr = torch.from_numpy(rowptr.astype(np.uint32).view(np.int32).copy())
        self.ids = torch.from_numpy(ids.astype(np.uint32).view(np.int32).copy())
        self.counts = torch.from_numpy(counts.astype(np.int64))
        self.hashes = None
        self.total_reads = int(total)

    def size(self):
        return self.rowptr.numel() - 1

    @property
    def nnz(self):
        return self.ids.numel()


class _Builder:
    """label -> count dictionary with the canonical export order (first id, XXH64, length, label)"""

    def start(self):
        self.d = {}

    def _add(self, ids, off, w):
        ids = ids.numpy().view(np.uint32); off = off.numpy().view(np.uint32).astype(np.int64)
        for r in range(len(off) - 1):
            lab = tuple(ids[off[r]:off[r + 1]].tolist())
            if lab:
                self.d[lab] = self.d.get(lab, 0) + int(w[r])

    def add_batch(self, ids, off):
        self._add(ids, off, np.ones(len(off) - 1, np.int64))

    def insertGroups(self, ids, off, counts):
        self._add(ids, off, counts.numpy())

    def finish(self):
        key = lambda lab: (lab[0], O.xxh64(np.array(lab, np.uint32).tobytes()), len(lab), lab)
        labs = sorted(self.d, key=key)
        rowptr = np.zeros(len(labs) + 1, np.int64); rowptr[1:] = np.cumsum([len(l) for l in labs])
        ids = np.array([t for l in labs for t in l], np.uint32)
        cnt = np.array([self.d[l] for l in labs], np.int64)
        self._vec = _Vec(rowptr, ids, cnt, cnt.sum())
        self._vec.hashes = torch.from_numpy(np.array([O.xxh64(np.array(l, np.uint32).tobytes()) for l in labs], np.uint64).view(np.int64).copy())
        return True

    def eqVec(self):
        return self._vec

    def stats(self):
        return dict(insert_ms=0.0)


class _EM:
    """optimize() = the oracle; begin/init/sweep/update/poll/finish = the same arithmetic in numpy pieces"""

    def __init__(self, length, rowptr, ids, counts, num_mapped):
        self.len = np.maximum(length.numpy().astype(np.float64), 1.0)
        self.rp = (rowptr.numpy().view(np.uint32)).astype(np.int64)
        self.ids = ids.numpy().view(np.uint32).astype(np.int64)
        self.cnt = counts.numpy().astype(np.float64)
        self.N = float(num_mapped); self.M = len(self.len)
        self.alpha = torch.zeros(self.M, dtype=torch.float64); self.mass = torch.zeros(self.M, dtype=torch.float64)
        self._raw_len = length.numpy().astype(np.float64)

    def optimize(self, use_vbem=False, tol=0.01, min_iter=50, max_iter=10000, **_):
        rc, a, m, st = O.em_optimize(self._raw_len, self.rp.astype(np.uint64), self.ids.astype(np.uint32),
                                     self.cnt.astype(np.uint64), int(self.N), use_vbem=use_vbem, tol=tol,
                                     min_iter=min_iter, max_iter=max_iter)
        self.alpha.copy_(torch.from_numpy(a)); self.mass.copy_(torch.from_numpy(m))
        return rc, st

    def begin(self, use_vbem=False, tol=0.01, min_iter=50, max_iter=10000, **_):
        self.vb, self.tol, self.min_iter, self.max_iter = use_vbem, tol, min_iter, max_iter
        self.ao = torch.zeros(self.M, dtype=torch.float64)
        self.ao[np.unique(self.ids)] = 1.0

    def alpha_out_view(self):
        return self.ao

    def _prep(self):
        from scipy.special import digamma
        a = self.a
        if self.vb:
            ln = digamma(a.sum())
            with np.errstate(divide="ignore"):
                self.x = np.where(a > 0, np.exp(digamma(np.where(a > 0, a, 1.0)) - ln), 0.0) / self.len
        else:
            self.x = a / self.len

    def init(self):
        act = self.ao.numpy() > 0
        self.n_active = int(act.sum())
        self.a = np.where(act, (1.0 / self.n_active) * self.N, 0.0)
        self.ao.zero_(); self.it = 0; self.conv = False; self.done = False
        self._prep()

    def _stop(self):
        return self.it >= self.min_iter and (self.it >= self.max_iter or self.conv)

    def sweep(self):
        self.skip = self._stop()
        if self.skip or len(self.cnt) == 0:
            return
        xv = self.x[self.ids]
        lens = np.diff(self.rp)
        den = np.add.reduceat(xv, self.rp[:-1])
        row = np.repeat(np.arange(len(lens)), lens)
        single = lens[row] == 1
        with np.errstate(divide="ignore", invalid="ignore"):
            contrib = np.where(single, self.cnt[row], np.where(den[row] > 0, xv * (self.cnt[row] / den[row]), 0.0))
        out = self.ao.numpy()
        np.add.at(out, self.ids, contrib)

    def update(self):
        if self.skip:
            return
        ap = self.ao.numpy().copy() + (0.01 if self.vb else 0.0)
        gate = ap > 1e-2
        rel = np.abs(self.a[gate] - ap[gate]) / ap[gate]
        self.conv = bool(np.all(rel <= self.tol)); self.maxrel = float(rel.max()) if rel.size else -1.0
        self.a = ap; self.ao.zero_(); self.it += 1
        self._prep()

    def poll(self):
        return self._stop(), dict(iters=self.it, converged=self.conv, n_active=self.n_active)

    def optimize_sharded(self, allreduce, poll_every=16, **kw):
        """sfgpu_em_optimize_sharded's control flow (csrc/em.hip) over the numpy pieces: begin -> all-reduce -> init ->
        { sweep, all-reduce, update } polled every poll_every iterations -> finish"""
        self.begin(**kw)
        allreduce(self.ao)
        self.init()
        done = self._stop()
        while not done:
            for _ in range(poll_every):
                self.sweep(); allreduce(self.ao); self.update()
            done, _ = self.poll()
        return self.finish()

    # the piecewise loop with the bias hook
    def set_bounds(self, min_iter, max_iter):
        self.min_iter, self.max_iter = min_iter, max_iter

    def alpha_view(self):
        return torch.from_numpy(self.a)

    def length_view(self):
        return torch.from_numpy(self.len)

    def rebase(self, length):
        self.len = np.maximum(length.numpy().astype(np.float64), 1.0)
        self._prep()

    def optimize_bias(self, bias, use_vbem=False, tol=0.01, min_iter=50, max_iter=10000, **_):
        rc, a, m, eff, es, eg, nr, st = O.em_optimize_bias(bias.bm, self._raw_len, self.rp.astype(np.uint64), self.ids.astype(np.uint32),
                                                           self.cnt.astype(np.uint64), int(self.N), use_vbem=use_vbem, tol=tol,
                                                           min_iter=min_iter, max_iter=max_iter)
        self.alpha.copy_(torch.from_numpy(a)); self.mass.copy_(torch.from_numpy(m))
        bias.es, bias.eg = es, eg
        return rc, st, torch.from_numpy(eff), nr

    def bootstrap(self, n, seed=1, use_vbem=False, tol=0.01, max_iter=10000, **_):
        rc, out, iters = O.bootstrap(self._raw_len, self.rp.astype(np.uint64), self.ids.astype(np.uint32), self.cnt.astype(np.uint64),
                                     int(n), use_vbem=use_vbem, tol=tol, max_iter=max_iter, seed=seed & 0x7FFFFFFF)
        return rc, torch.from_numpy(out), iters

    def close(self):
        pass

    def finish(self):
        cutoff = (0.01 + 1e-8) if self.vb else 1e-8
        a = np.where(self.a <= cutoff, 0.0, self.a)
        self.alpha.copy_(torch.from_numpy(a)); self.mass.copy_(torch.from_numpy(a / a.sum()))
        return 0, dict(iters=self.it, converged=self.conv, n_active=self.n_active, alpha_sum=float(a.sum()), loop_ms=0.0,
                       max_rel_diff=self.maxrel)


class _Bias:
    """updateEffectiveLengths = the oracle's"""

    def __init__(self, exp, sopt):
        t = exp.transcripts()
        self.bm = O.make_bias_model(exp._seq.numpy().tobytes(), exp._seq_off.numpy().astype(np.uint64), t.RefLength.numpy().view(np.uint32),
                                    t.EffectiveLength.numpy(), exp.fragLengthDist().astype(np.uint32), exp.readBias(), exp.observedGC(),
                                    num_fwd=exp.numFwd(), num_rc=exp.numRC(), seq_bias=sopt.biasCorrect, gc_bias=sopt.gcBiasCorrect)
        self.es, self.eg = np.ones(4096), np.ones(101)

    def update(self, eff_in, alpha):
        rc, out, self.es, self.eg, nc = O.update_efflens(self.bm, eff_in.numpy(), alpha.numpy())
        return torch.from_numpy(out), dict(status=rc, n_corrected=nc)

    def expected(self):
        return self.es, self.eg


class CheckerEngine:
    device = torch.device("cpu")

    def bias_model(self, exp, sopt):
        return _Bias(exp, sopt)

    def new_builder(self, expected=0):
        return _Builder()

    # ---- the exchange primitives (csrc/merge.hip in the product), restated with numpy for the control-flow tests
    @staticmethod
    def _csr(vec):
        return (vec.rowptr.numpy().view(np.uint32).astype(np.int64), vec.ids.numpy().view(np.uint32), vec.counts.numpy().astype(np.uint64),
                vec.hashes.numpy().view(np.uint64))

    def pack_by_owner(self, vec, n):
        from sailfish_amd.distributed import block_bytes
        rp, ids, cnt, hs = self._csr(vec)
        owner = ((hs >> np.uint64(33)) & np.uint64(0x3FFFFFFF)) % np.uint64(n)
        parts, sizes = [], []
        for d in range(n):
            sel = np.flatnonzero(owner == d)
            lens = (rp[sel + 1] - rp[sel]).astype("<u4")
            idl = np.concatenate([ids[rp[c]:rp[c + 1]] for c in sel]).astype("<u4") if len(sel) else np.zeros(0, "<u4")
            blk = cnt[sel].astype("<u8").tobytes() + lens.tobytes() + idl.tobytes()
            parts.append(blk + b"\0" * (block_bytes(len(sel), len(idl)) - len(blk)))
            sizes.append((len(sel), len(idl)))
        raw = b"".join(parts)
        return torch.frombuffer(bytearray(raw + b"\0" * 8), dtype=torch.uint8)[:len(raw)], sizes

    def fold_block(self, builder, block, c, l):
        if c == 0:
            return
        b = block.numpy().tobytes()
        cnt = np.frombuffer(b[:8 * c], "<u8"); lens = np.frombuffer(b[8 * c:12 * c], "<u4"); ids = np.frombuffer(b[12 * c:12 * c + 4 * l], "<u4")
        off = np.zeros(c + 1, np.int64); off[1:] = np.cumsum(lens)
        builder._add(torch.from_numpy(ids.view(np.int32).copy()), torch.from_numpy(off.astype(np.uint32).view(np.int32).copy()), cnt.astype(np.int64))

    def export_block(self, vec):
        rp, ids, cnt, hs = self._csr(vec)
        raw = cnt.astype("<u8").tobytes() + hs.astype("<u8").tobytes() + np.diff(rp).astype("<u4").tobytes() + ids.astype("<u4").tobytes()
        return torch.frombuffer(bytearray(raw + b"\0" * 8), dtype=torch.uint8)[:len(raw)]

    def merge_disjoint(self, blocks, sizes):
        rows = []
        for blk, (c, l) in zip(blocks, sizes):
            b = blk.numpy().tobytes()
            cnt = np.frombuffer(b[:8 * c], "<u8"); hs = np.frombuffer(b[8 * c:16 * c], "<u8"); lens = np.frombuffer(b[16 * c:20 * c], "<u4")
            ids = np.frombuffer(b[20 * c:20 * c + 4 * l], "<u4")
            o = np.zeros(c + 1, np.int64); o[1:] = np.cumsum(lens)
            rows += [(int(ids[o[k]]), int(hs[k]), tuple(ids[o[k]:o[k + 1]].tolist()), int(cnt[k])) for k in range(c)]
        rows.sort(key=lambda r: (r[0], r[1], len(r[2]), r[2]))
        if any(a[0] == b[0] and a[1] == b[1] for a, b in zip(rows, rows[1:])):
            return None
        rowptr = np.zeros(len(rows) + 1, np.int64); rowptr[1:] = np.cumsum([len(r[2]) for r in rows])
        v = _Vec(rowptr, np.array([t for r in rows for t in r[2]], np.uint32), np.array([r[3] for r in rows], np.int64), sum(r[3] for r in rows))
        v.hashes = torch.from_numpy(np.array([r[1] for r in rows], np.uint64).view(np.int64).copy())
        return v

    def gibbs_sample(self, length, mass, rowptr, ids, counts, num_mapped, n, n_chains=0, seed=1):
        rc, out = O.gibbs(length.numpy(), mass.numpy(), rowptr.numpy().view(np.uint32).astype(np.uint64), ids.numpy().view(np.uint32),
                          counts.numpy().astype(np.uint64), int(num_mapped), int(n), seed=seed & 0x7FFFFFFF)
        return rc, torch.from_numpy(out)

    def em_problem(self, length, rowptr, ids, counts, num_mapped):
        return _EM(length, rowptr, ids, counts, num_mapped)

    def set_effective_lengths(self, exp, sopt, fl_counts, remaining_fl_ops):
        ref = exp.transcripts().RefLength.numpy().view(np.uint32)
        exp.transcripts().EffectiveLength.copy_(torch.from_numpy(O.efflen_smoothed(ref, O.cf_gaussian())))

    def tpm(self, exp, sopt):
        t = exp.transcripts()
        return torch.from_numpy(O.tpm(t.estCount.numpy(), t.EffectiveLength.numpy(), exp.numMappedFragments()))

    def sync(self):
        pass


def _worker(rank, world, port, mode, vb, out, merge_mode="auto"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sailfish_amd as sf
        from sailfish_amd import distributed as sfd, synth
        M, P, R = 600, 1500, 6000
        ref_len = synth.transcript_lengths(M)
        poff, pids = synth.label_pool(M, P)
        ids, off = synth.reads_from_pool(poff, pids, R, seed=7 + 1000 * rank)       # this rank's shard
        sopt = sf.SailfishOpts(useVBOpt=vb)
        exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.numpy().view(np.uint32), device="cpu"), sopt)
        q = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode=mode, engine=CheckerEngine(), poll_every=7,
                                 merge_mode=merge_mode)
        info = q.run(ids, off)
        t = exp.transcripts()
        v = q.last_vec
        out.put((rank, info["em_mode"], info["n_classes"], info["nnz"], exp.numMappedFragments(), info["em_stats"]["iters"],
                 t.estCount.numpy().copy(), info["tpm"].numpy().copy(), ids.numpy().copy(), off.numpy().copy(),
                 v.rowptr.numpy().view(np.uint32).copy(), v.ids.numpy().view(np.uint32).copy(), v.counts.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _sampler_worker(rank, world, port, mode, merge_mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sailfish_amd as sf
        from sailfish_amd import distributed as sfd, synth
        M, P, R = 200, 500, 3000
        ref_len = synth.transcript_lengths(M)
        poff, pids = synth.label_pool(M, P)
        ids, off = synth.reads_from_pool(poff, pids, R, seed=7 + 1000 * rank)
        sopt = sf.SailfishOpts()
        exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.numpy().view(np.uint32), device="cpu"), sopt)
        q = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode=mode, engine=CheckerEngine(), poll_every=7, merge_mode=merge_mode)
        info = q.run(ids, off)
        bs = q.bootstrap(7, seed=5)
        gs = q.gibbs(8, seed=5)
        out.put((rank, info["n_classes"], exp.numMappedFragments(), bs.numpy().copy(), gs.numpy().copy(),
                 exp.transcripts().estCount.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,merge", [("replicated", "owner"), ("sharded", "owner"), ("sharded", "allgather")])
def test_three_rank_samplers_use_the_whole_merged_table(built, mode, merge):
    """gibbs() and bootstrap() after an owner-partitioned merge (no `merged` builder is ever finished) and after a
    sharded EM (self.problem holds one rank's class slice): both must sample over ALL merged classes"""
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sampler_worker, args=(r, world, port, mode, merge, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    N = res[0][2]
    assert N == 3000 * world
    for r in res:
        bs, gs, est = r[3], r[4], r[5]
        assert bs.shape == (7, 200) and gs.shape == (8, 200)
        # every replicate / draw distributes ALL reads of ALL ranks (a rank's slice would hold ~1/3 of them)
        np.testing.assert_allclose(bs.sum(1), N, rtol=1e-6)
        assert np.array_equal(gs.sum(1), np.full(8, N))
        # and tracks the point estimate
        big = est > 0.01 * N / 200
        assert np.all(np.abs(bs.mean(0)[big] - est[big]) < 0.5 * est[big] + 30)
    assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[2][4])     # gathered rows are the same everywhere


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("mode,vb,merge,world", [("replicated", False, "auto", 2), ("sharded", False, "auto", 2), ("sharded", True, "auto", 2),
                                                 ("replicated", False, "owner", 2), ("sharded", False, "auto", 3)])
def test_two_rank_quant_matches_single_process(built, mode, vb, merge, world):
    """world 2 merges by all-gather; merge="owner" / world 3 reduce every class at the rank that owns its hash first"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, vb, out, merge)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process oracle over the union of both shards
    from sailfish_amd import synth
    M = 600
    ref_len = synth.transcript_lengths(M).numpy().view(np.uint32)
    b = O.EqBuilder()
    for r in res:
        b.add_batch(r[8].view(np.uint32), r[9].view(np.uint32).astype(np.uint64))
    rp, ii, cc, hh = b.finish()
    eff = O.efflen_smoothed(ref_len, O.cf_gaussian())
    rc, oa, om, ost = O.em_optimize(eff, rp, ii, cc, b.total_reads, use_vbem=vb)
    ot = O.tpm(oa, eff, b.total_reads)
    for r in res:
        assert r[1] == mode and r[2] == b.n_classes and r[3] == b.nnz and r[4] == b.total_reads == 6000 * world
        assert r[5] == ost["iters"]
        nz = oa > 0
        assert np.array_equal(r[6] > 0, nz)
        assert np.max(np.abs(r[6][nz] - oa[nz]) / oa[nz]) < 1e-9
        assert np.max(np.abs(r[7][nz] - ot[nz]) / ot[nz]) < 1e-9
        # the merged table is the single-process table, class for class, in the canonical order
        assert np.array_equal(r[10], rp.astype(np.uint32)) and np.array_equal(r[11], ii) and np.array_equal(r[12].astype(np.uint64), cc)
    assert np.array_equal(res[0][6], res[1][6])      # both ranks hold the same answer


def test_nnz_balanced_slices():
    from sailfish_amd.distributed import nnz_balanced_slices
    rp = np.concatenate([[0], np.cumsum(np.r_[np.ones(50, int), np.full(5, 100), np.ones(50, int)])])
    for w in (1, 2, 3, 8):
        cuts = nnz_balanced_slices(rp, w)
        assert cuts[0] == 0 and cuts[-1] == len(rp) - 1 and all(a <= b for a, b in zip(cuts, cuts[1:]))
        nn = [rp[b] - rp[a] for a, b in zip(cuts, cuts[1:])]
        assert sum(nn) == rp[-1] and max(nn) <= rp[-1] / w + 100


def _bias_inputs(M, ref_len):
    rng = np.random.default_rng(99)
    off = np.zeros(M, np.int64); parts = []; pos = 0
    for t in range(M):
        off[t] = pos
        parts.append(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(ref_len[t])).astype(np.uint8)) + b"$")
        pos += int(ref_len[t]) + 1
    x = np.arange(1000)
    fl = np.round(1e6 * np.exp(-0.5 * ((x - 200) / 60.0) ** 2)).astype(np.int32)
    return b"".join(parts), off, fl, rng.integers(1, 300, 4096).astype(np.uint32), rng.integers(1, 900, 101).astype(np.uint32)


def _bias_worker(rank, world, port, mode, which, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sailfish_amd as sf
        from sailfish_amd import distributed as sfd, synth
        M, P, R = 300, 900, 5000
        ref_len = synth.transcript_lengths(M)
        poff, pids = synth.label_pool(M, P)
        ids, off = synth.reads_from_pool(poff, pids, R, seed=3 + 1000 * rank)
        sopt = sf.SailfishOpts(biasCorrect=which == "seq", gcBiasCorrect=which == "gc")
        exp = sf.ReadExperiment(sf.Transcripts([str(i) for i in range(M)], ref_len.numpy().view(np.uint32), device="cpu"), sopt)
        seq, soff, fl, rb, og = _bias_inputs(M, ref_len.numpy().view(np.uint32))
        exp.setSequences(seq, soff); exp.setFragLengthDist(fl)
        exp.readBias()[:] = rb; exp.observedGC()[:] = og; exp.addNumFwd(55); exp.addNumRC(45)
        q = sfd.DistributedQuant(exp, sopt, group=dist.group.WORLD, em_mode=mode, engine=CheckerEngine(), poll_every=7, tol=1e-4)
        info = q.run(ids, off)
        t = exp.transcripts()
        out.put((rank, info["em_stats"]["iters"], q.recomputes, t.estCount.numpy().copy(), t.EffectiveLength.numpy().copy(),
                 exp.expectedSeqBias().copy(), exp.expectedGCBias().copy(), ids.numpy().copy(), off.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,which", [("sharded", "seq"), ("sharded", "gc"), ("replicated", "seq")])
def test_two_rank_quant_with_bias_hook(built, mode, which):
    """doBiasCorrect across ranks: the recompute iterations are reached with lowered stop bounds, the lengths are
    recomputed from the replicated alpha and broadcast, the loop continues -- same answer as one process"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bias_worker, args=(r, 2, port, mode, which, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=240) for _ in procs], key=lambda r: r[0])
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
    bm = O.make_bias_model(seq, soff.astype(np.uint64), ref_len, eff0, fl.astype(np.uint32), rb, og, num_fwd=55, num_rc=45,
                           seq_bias=which == "seq", gc_bias=which == "gc")
    rc, oa, om, oeff, oes, oeg, onr, ost = O.em_optimize_bias(bm, eff0, rp, ii, cc, b.total_reads, tol=1e-4)
    assert rc == 0 and onr >= 1
    for r in res:
        assert r[1] == ost["iters"] and r[2] == onr
        nz = oa > 0
        assert np.array_equal(r[3] > 0, nz)
        assert np.max(np.abs(r[3][nz] - oa[nz]) / oa[nz]) < 1e-9
        np.testing.assert_allclose(r[4], oeff, rtol=1e-12)
        np.testing.assert_allclose(r[5], oes, rtol=1e-12); np.testing.assert_allclose(r[6], oeg, rtol=1e-12)
    assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])
