"""Multi-GPU driver of the quantification hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm).

Data flow for N ranks:
  1. reads are sharded over ranks; every rank builds the class table of ITS reads with the HIP
     builder -- no communication (reads are independent);
  2. ONE exchange: the local tables (label lengths, labels, counts: a few tens of MB) are
     all-gathered and every rank folds all of them into one table with the weighted upsert
     (sfgpu_eq_add_weighted_device).  Equal labels from different ranks add their counts, so every
     rank ends with the same canonical class list as a single-GPU run over all reads (integer
     work: bit-exact).  Above two ranks every class is first reduced at the rank that owns its hash
     (one all-to-all), so the all-gather carries each merged class once;
  3. EM over the merged classes, in one of two modes:
       "sharded"   : classes are cut into N contiguous, nnz-balanced slices; each iteration is
                     local sweep -> SUM all-reduce of alphaOut (M doubles, RCCL) -> update.  This is
                     the north-star layout; the all-reduce is latency bound (M = 200k -> 1.6 MB).
       "replicated": every rank runs the whole EM on the merged classes with the on-device loop --
                     no per-iteration collective.
     "auto" MEASURES on the first run: the all-reduce of M doubles on this node's fabric (the communicator of the
     sharded loop) against the sweep of the whole problem; it shards when sweep/N (not below the ~8 us a round of
     tiles costs) plus the all-reduce is shorter than the whole sweep, and keeps the decision for later runs.
     In the sharded mode the loop itself runs in C (sfgpu_em_optimize_sharded) with ncclAllReduce enqueued on
     its stream by libsfgpu's own RCCL communicator (sailfish_amd/comm.py): no Python between two iterations.

The compute engine is injected so that the control flow is covered on CPU (gloo, world_size 2) by
tests that supply their own CPU checker engine; the product engine is HipEngine (libsfgpu only)."""
import time

import numpy as np
import torch

from . import efflen as _efflen
from . import writer as _writer
from .eqclass import EquivalenceClassBuilder
from .experiment import ReadExperiment, SailfishOpts
from .optimizer import EMProblem

kSweepFloorUs = 8.0                    # what one round of tiles costs however small they are
RECOMPUTE_ITERS = (50, 500, 1000)      # recomputeIt, src/CollapsedEMOptimizer.cpp:814


class HipEngine:
    """Everything that touches data goes through libsfgpu (no CPU fallback)."""

    def __init__(self, device):
        self.device = torch.device(device)

    def new_builder(self, expected=0):
        return EquivalenceClassBuilder(expected_classes=expected, device=self.device)

    def em_problem(self, length, rowptr, ids, counts, num_mapped):
        return EMProblem(length, rowptr, ids, counts, num_mapped)

    def set_effective_lengths(self, exp, sopt, fl_counts, remaining_fl_ops):
        return _efflen.set_effective_lengths(exp, sopt, fl_counts=fl_counts, remaining_fl_ops=remaining_fl_ops)

    def bias_model(self, exp, sopt):
        return exp.biasModel(sopt)

    # ---- class-table exchange: the device work of sfgpu_eqvec_* / sfgpu_eq_add_block_device (csrc/merge.hip) ----
    def pack_by_owner(self, vec, n_owners):
        """-> (uint8 device tensor holding the n_owners blocks back to back, [(classes, ids)] per owner)"""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        n = int(vec.size())
        hc = (C.c_uint64 * n_owners)(); hi = (C.c_uint64 * n_owners)(); off = (C.c_uint64 * (n_owners + 1))()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            st = _lib.current_stream_ptr()
            _lib.check(L.sfgpu_eqvec_owner_sizes(_lib.ptr(vec.rowptr), _lib.ptr(vec.hashes), n, n_owners, hc, hi, st))
            sizes = [(int(hc[d]), int(hi[d])) for d in range(n_owners)]
            total = sum(block_bytes(c, l) for c, l in sizes)
            buf = torch.empty(max(total, 8), dtype=torch.uint8, device=self.device)
            _lib.check(L.sfgpu_eqvec_pack_by_owner(_lib.ptr(vec.rowptr), _lib.ptr(vec.ids), _lib.ptr(vec.counts), _lib.ptr(vec.hashes), n,
                                                   n_owners, hc, hi, _lib.ptr(buf), off, st))
        return buf[:total], sizes

    def fold_block(self, builder, block, n_classes, n_ids):
        """upsert the classes of one [counts | lens | ids] block into a started builder"""
        from . import _lib
        if n_classes == 0:
            return
        assert block.data_ptr() % 8 == 0
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            _lib.check(_lib.lib().sfgpu_eq_add_block_device(builder._h, _lib.ptr(block), int(n_classes), int(n_ids), _lib.current_stream_ptr()))

    def export_block(self, vec):
        """the table as one [counts | hashes | lens | ids] block (uint8 device tensor)"""
        from . import _lib
        n, l = int(vec.size()), int(vec.ids.numel())
        buf = torch.empty(max(20 * n + 4 * l, 8), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            st = _lib.current_stream_ptr()
            _lib.check(_lib.lib().sfgpu_eqvec_export_block(_lib.ptr(vec.rowptr), _lib.ptr(vec.ids), _lib.ptr(vec.counts), _lib.ptr(vec.hashes),
                                                           n, l, _lib.ptr(buf), st))
            torch.cuda.current_stream().synchronize()
        return buf[:20 * n + 4 * l]

    def merge_disjoint(self, blocks, sizes):
        """disjoint partitions in export_block format -> the merged table in canonical order, or None if two labels share
        first id and XXH64 (the caller then folds the partitions through a builder)"""
        import ctypes as C
        from . import _lib
        from .eqclass import EqVec
        w = len(blocks)
        n = sum(c for c, _ in sizes); l = sum(x for _, x in sizes)
        if l >= 2 ** 32 or n >= 2 ** 32 or w > 64:        # (sfgpu_eqvec_merge_disjoint takes <= 64 partitions: the caller folds instead)
            return None
        dev = self.device
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev); ids = torch.empty(l, dtype=torch.int32, device=dev)
        counts = torch.empty(n, dtype=torch.int64, device=dev); hashes = torch.empty(n, dtype=torch.int64, device=dev)
        ptrs = (C.c_void_p * w)(*[b.data_ptr() if b.numel() else 0 for b in blocks])
        nc = (C.c_uint64 * w)(*[c for c, _ in sizes]); ni = (C.c_uint64 * w)(*[x for _, x in sizes])
        tie = C.c_int(0)
        with torch.cuda.device(dev):
            torch.cuda.current_stream().synchronize()
            _lib.check(_lib.lib().sfgpu_eqvec_merge_disjoint(ptrs, nc, ni, w, _lib.ptr(rowptr), _lib.ptr(ids), _lib.ptr(counts), _lib.ptr(hashes),
                                                             C.byref(tie), _lib.current_stream_ptr()))
        if tie.value:
            return None
        return EqVec(rowptr, ids, counts, hashes, int(counts.sum().item()) if n else 0)

    def gibbs_sample(self, length, mass, rowptr, ids, counts, num_mapped, n, n_chains=0, seed=1):
        from .gibbs import gibbs_sample
        return gibbs_sample(length, mass, rowptr, ids, counts, num_mapped, n, n_chains=n_chains, seed=seed)

    def tpm(self, exp, sopt):
        return _writer.tpm(exp, sopt)[0]

    def sync(self):
        torch.cuda.synchronize(self.device)


def block_bytes(c, l):
    """SFGPU_BLOCK_BYTES: [counts u64[c] | lens u32[c] | ids u32[l]] padded to 8 bytes"""
    return (12 * c + 4 * l + 7) & ~7


def _all_gather_var(t, group, world):
    """all-gather of 1-D tensors of different lengths (sizes first, then padded payloads)."""
    import torch.distributed as dist
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return [o[:s] for o, s in zip(outs, sizes)]


def nnz_balanced_slices(rowptr_cpu, world):
    """cut [0, C) into `world` contiguous class ranges of ~equal nnz"""
    C = len(rowptr_cpu) - 1
    L = int(rowptr_cpu[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(rowptr_cpu, L * r // world, side="left")))
    cuts.append(C)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


class DistributedQuant:
    def __init__(self, exp: ReadExperiment, sopt: SailfishOpts, group=None, em_mode="auto", engine=None,
                 tol=0.01, max_iter=10000, poll_every=16, merge_mode="auto", min_iter=50):
        self.exp, self.sopt, self.group = exp, sopt, group
        self.world = 1
        self.rank = 0
        if group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(group); self.rank = dist.get_rank(group)
        self.engine = engine or HipEngine(exp.transcripts().device)
        if self.world > 1 and isinstance(self.engine, HipEngine):
            # several ranks on ONE device (dry runs, the tests on a one-GPU box): no kernel of any rank has the chip to itself, and
            # the persistent EM loop (csrc/em_persist.h) needs all its blocks resident -- such ranks keep one kernel per iteration
            import socket
            import torch
            import torch.distributed as dist
            dv = torch.device(self.engine.device)
            me = (socket.gethostname(), str(torch.cuda.get_device_properties(dv).uuid) if hasattr(torch.cuda.get_device_properties(dv), "uuid") else dv.index)
            everyone = [None] * self.world
            dist.all_gather_object(everyone, me, group=group)
            if len(set(everyone)) < self.world:
                from . import _lib
                _lib.lib().sfgpu_em_allow_persistent(0)      # (a switch of the library, not of the process environment)
        self.em_mode = em_mode
        self.tol, self.max_iter, self.poll_every = tol, max_iter, poll_every
        self.min_iter = min_iter            # optimize()'s minIter is 50 (src/CollapsedEMOptimizer.cpp:716); tests shorten the loop
        self.local = self.engine.new_builder()
        self.merged = self.engine.new_builder() if self.world > 1 else None
        # "owner": classes are first reduced at the rank that owns their hash (one all-to-all), then the disjoint
        # partitions are all-gathered; "allgather": every rank receives and folds every rank's whole table
        self.merge_mode = merge_mode if merge_mode != "auto" else ("owner" if self.world > 2 else "allgather")
        self.part = self.engine.new_builder() if (self.world > 1 and self.merge_mode == "owner") else None
        self.problem = None

    # ---- one pass of the hot path ------------------------------------------------------------
    def run(self, ids, off, fl_counts=None, remaining_fl_ops=1):
        eng, exp, sopt = self.engine, self.exp, self.sopt
        info = {}
        # Phase times: the host waits only where the path itself does (finish() and optimize() return results to the host); in
        # between, the phases are told apart by the host clock after those waits -- a device-wide synchronisation per phase
        # boundary cost the step five idle gaps.
        eng.sync(); t0 = time.perf_counter()
        b = self.local
        b.start(); b.add_batch(ids, off); b.finish()
        vec = b.eqVec()
        if self.world > 1:
            eng.sync()
        t1 = time.perf_counter()            # (finish() waited for the table; the export is queued on the builder's stream, ~0.2 ms of device time that the
                                            #  next phase's clock takes)
        bst = b.stats()
        info["t_insert_ms"] = bst["insert_ms"]; info["insert_launches"] = bst.get("insert_launches", 1)
        # a "launch" of the class build is one sub-batch: pass 1 (route) + pass 2 (insert) of the radix-partitioned path
        info["insert_kernels"] = "k_part_route+k_part_insert (per sub-batch)"
        info["t_build_ms"] = (t1 - t0) * 1e3
        if self.world > 1:
            vec = self._merge(vec)
            eng.sync(); t2 = time.perf_counter()
            info["t_merge_ms"] = (t2 - t1) * 1e3
            t1 = t2
        self.last_vec = vec                               # the (merged) class table the EM ran on
        exp.setNumMappedFragments(vec.total_reads)        # every read with a non-empty hit list is mapped
        eng.set_effective_lengths(exp, sopt, fl_counts, remaining_fl_ops)
        t2 = time.perf_counter()
        ok, st, mode = self._em(vec)                      # (optimize() returns when alpha is final)
        t3 = time.perf_counter()
        tpm = eng.tpm(exp, sopt)
        eng.sync(); t4 = time.perf_counter()
        info.update(ok=ok, em_stats=st, em_mode=mode, n_classes=vec.size(), nnz=vec.nnz, tpm=tpm,
                    t_efflen_ms=(t2 - t1) * 1e3, t_em_ms=(t3 - t2) * 1e3, t_tpm_ms=(t4 - t3) * 1e3)
        return info

    # ---- class-table exchange ----------------------------------------------------------------
    def _merge(self, vec):
        if self.merge_mode == "owner":
            part = self._reduce_at_owner(vec)
            merged = self._concat_disjoint(part)
            return merged if merged is not None else self._merge_allgather(part)
        return self._merge_allgather(vec)

    def _concat_disjoint(self, vec):
        """All-gather of the ranks' DISJOINT partitions and assembly of the merged table without hashing anything
        again: the union of disjoint class sets only has to be put into the canonical order (first id, XXH64,
        length, label) -- sfgpu_eqvec_merge_disjoint.  Two different labels with the same first id and the same
        64-bit hash would need the last two keys: then None is returned and the caller folds the partitions
        through a builder instead."""
        import torch.distributed as dist
        w = self.world
        block = self.engine.export_block(vec)                             # [counts i64 | hashes i64 | lens i32 | ids i32]
        dev = block.device
        mine = torch.tensor([int(vec.size()), int(vec.ids.numel())], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(mine) for _ in range(w)]
        dist.all_gather(sizes, mine, group=self.group)
        sizes = [tuple(x) for x in torch.stack(sizes).cpu().tolist()]      # one host sync for all sizes
        nbytes = [20 * c + 4 * l for c, l in sizes]
        pad = torch.zeros(max(max(nbytes), 8), dtype=torch.uint8, device=dev)
        pad[:block.numel()] = block
        blocks = [torch.empty_like(pad) for _ in range(w)]
        dist.all_gather(blocks, pad, group=self.group)
        return self.engine.merge_disjoint([b[:nb] for b, nb in zip(blocks, nbytes)], sizes)

    def _reduce_at_owner(self, vec):
        """Pre-reduction for N > 2 (owner(class) = hash mod G, one all-to-all of (label, count) partials):
        every class goes to the rank that owns its XXH64, which adds up the counts of the ranks' copies.  The result
        is this rank's partition of the merged table -- disjoint from the others' -- so the all-gather that follows
        moves and folds every merged class once instead of every rank's copy of it (N x less merge work, ~N/2 x
        less traffic).  Integer work: the merged table is bit-identical to the all-gather-only merge."""
        import torch.distributed as dist
        w, me = self.world, self.rank
        send, mine_sizes = self.engine.pack_by_owner(vec, w)               # w blocks [counts | lens | ids | pad], owner order
        dev = send.device
        mine = torch.tensor([x for cl in mine_sizes for x in cl], dtype=torch.int64, device=dev)     # [c_0, l_0, c_1, l_1, ...]
        sizes = [torch.zeros_like(mine) for _ in range(w)]
        dist.all_gather(sizes, mine, group=self.group)
        S = torch.stack(sizes).cpu().reshape(w, w, 2).tolist()               # S[src][dst] = (classes, ids); one host sync
        send_bytes = [block_bytes(*S[me][d]) for d in range(w)]
        recv_bytes = [block_bytes(*S[src][me]) for src in range(w)]
        recv = torch.empty(max(sum(recv_bytes), 8), dtype=torch.uint8, device=dev)[:sum(recv_bytes)]
        # the route is a function of the backend alone, so every rank takes the same one (a fallback chosen from a
        # caught exception could split the ranks between two different collectives)
        if dist.get_backend(self.group) == "nccl":
            dist.all_to_all_single(recv, send, output_split_sizes=recv_bytes, input_split_sizes=send_bytes, group=self.group)
        else:
            # gloo (CPU tests, single-device dry runs) has no all-to-all: every rank's whole send buffer is gathered
            # and the block meant for this rank is cut out -- same result, more traffic
            full = _all_gather_var(send, self.group, w)
            pos = 0
            for src in range(w):
                start = sum(block_bytes(*S[src][d]) for d in range(me))
                recv[pos:pos + recv_bytes[src]] = full[src][start:start + recv_bytes[src]]
                pos += recv_bytes[src]
        b = self.part
        b.start()
        pos = 0
        for src in range(w):                                                 # upsert what every rank sent, in rank order
            c, l = S[src][me]
            self.engine.fold_block(b, recv[pos:pos + recv_bytes[src]], c, l)
            pos += recv_bytes[src]
        b.finish()
        return b.eqVec()

    def _merge_allgather(self, vec):
        """One exchange: every rank contributes its class table as one byte block [counts i64[C] | lens i32[C] | ids i32[L]]
        (sizes first, then the padded blocks), and upserts the tables of all ranks, in rank order -> the same table on
        every rank."""
        import torch.distributed as dist
        w = self.world
        block, (cl,) = self.engine.pack_by_owner(vec, 1)                   # one owner: the whole table, canonical order
        dev = block.device
        mine = torch.tensor(list(cl), dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(mine) for _ in range(w)]
        dist.all_gather(sizes, mine, group=self.group)
        sizes = torch.stack(sizes).cpu().tolist()                     # one host sync for all sizes
        nbytes = [block_bytes(c, l) for c, l in sizes]
        pad = torch.zeros(max(max(nbytes), 8), dtype=torch.uint8, device=dev)
        pad[:block.numel()] = block
        blocks = [torch.empty_like(pad) for _ in range(w)]
        dist.all_gather(blocks, pad, group=self.group)
        m = self.merged
        m.start()
        for b, (c, l), nb in zip(blocks, sizes, nbytes):
            self.engine.fold_block(m, b[:nb], c, l)
        m.finish()
        return m.eqVec()

    # ---- EM ------------------------------------------------------------------------------------
    def _pick_mode(self, nnz):
        if self.world == 1:
            return "single"
        if self.em_mode != "auto":
            return self.em_mode
        # (decided by _measure_auto on the first run of a problem of this size: the decision is kept per (M, nnz))
        return getattr(self, "_auto_modes", {}).get(getattr(self, "_auto_key", None)) or "replicated"

    def _allreduce(self):
        """what sums alphaOut over the ranks: libsfgpu's RCCL communicator on the nccl backend (made once, collectively),
        a torch.distributed call otherwise (gloo: CPU tests, several ranks sharing one device)"""
        import torch.distributed as dist
        if getattr(self, "_ar", None) is None:
            self._ar = None
            if isinstance(self.engine, HipEngine) and dist.get_backend(self.group) == "nccl":
                from . import comm as _comm
                # availability is agreed on FIRST (librccl may load on some ranks only): every rank then takes the same branch,
                # so the collective creation below is entered by all ranks or by none (a rank without the library
                # went straight to the MIN all-reduce while the others sat in the id broadcast -- mismatched collectives hang)
                have = torch.tensor([int(bool(_comm.available()))], dtype=torch.int32, device=self.engine.device)
                dist.all_reduce(have, op=dist.ReduceOp.MIN, group=self.group)
                if int(have.item()) == 1:
                    # made collectively and checked collectively: a communicator that does not come up, or does not add, on ANY
                    # rank sends every rank to the torch.distributed call (a split decision would hang the first all-reduce).
                    # from_group's own collectives (the id broadcast) run inside the try on every rank; a rank that raises
                    # before ncclCommInitRank still reaches the flag all-reduce below, and the ranks stuck in ncclCommInitRank
                    # are released by RCCL's own bootstrap timeout.
                    comm, ok = None, 1
                    try:
                        comm = _comm.Comm.from_group(self.group, self.engine.device)
                        probe = torch.ones(4, dtype=torch.float64, device=self.engine.device)
                        comm.all_reduce(probe); self.engine.sync()
                        ok = int(bool((probe == float(self.world)).all()))
                    except Exception:
                        ok = 0
                    flag = torch.tensor([ok], dtype=torch.int32, device=self.engine.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                    if int(flag.item()) == 1:
                        self._ar = comm
                    elif comm is not None:
                        comm.close()
            if self._ar is None:
                group = self.group
                self._ar = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return self._ar

    def _measure_auto(self, p_full, M):
        """auto: time the two things the modes differ in, on this node, once; every rank takes the slowest rank's numbers"""
        import torch.distributed as dist
        if not hasattr(self, "_auto_modes"):
            self._auto_modes = {}
        if self._auto_modes.get(self._auto_key) is not None or self.em_mode != "auto" or self.world == 1:
            return
        self._auto_modes[self._auto_key] = "replicated"
        if not hasattr(p_full, "time_sweep"):
            return
        ar = self._allreduce()
        sweep_us = p_full.time_sweep(20, use_vbem=self.sopt.useVBOpt, tol=self.tol, min_iter=self.min_iter, max_iter=self.max_iter) * 1e3
        if hasattr(ar, "time_all_reduce"):
            ar_us = ar.time_all_reduce(M, 30)
        else:
            buf = torch.zeros(M, dtype=torch.float64, device=self.engine.device)
            ar(buf); self.engine.sync(); t0 = time.perf_counter()
            for _ in range(10):
                ar(buf)
            self.engine.sync(); ar_us = (time.perf_counter() - t0) / 10 * 1e6
        t = torch.tensor([sweep_us, ar_us], dtype=torch.float64, device=self.engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        sweep_us, ar_us = float(t[0]), float(t[1])
        local_us = max(kSweepFloorUs, sweep_us / self.world)
        self._auto_modes[self._auto_key] = "sharded" if local_us + ar_us < sweep_us else "replicated"
        self.auto_measurement = dict(sweep_us_whole_problem=sweep_us, allreduce_us=ar_us, sweep_us_local_estimate=local_us,
                                     chosen=self._auto_modes[self._auto_key], problem=dict(M=self._auto_key[0], nnz_log2_x4=self._auto_key[1]))

    def _em(self, vec):
        exp, sopt = self.exp, self.sopt
        txps = exp.transcripts()
        length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
        # `auto` measures once per problem SIZE: a later run over a different transcriptome / class table decides again
        # (sizes are bucketed to a quarter octave so that run-to-run jitter in nnz does not re-measure)
        import math
        self._auto_key = (int(length.numel()), int(round(4 * math.log2(max(int(vec.nnz), 1)))))
        mode = self._pick_mode(vec.nnz)
        if self.problem is not None:       # release the previous run's device state before building the next
            self.problem.close(); self.problem = None
        kw = dict(use_vbem=sopt.useVBOpt, tol=self.tol, min_iter=self.min_iter, max_iter=self.max_iter)
        # doBiasCorrect (src/CollapsedEMOptimizer.cpp:717): lengths are recomputed at iterations 50 / 500 / 1000
        bias = self.engine.bias_model(exp, sopt) if (sopt.biasCorrect or sopt.gcBiasCorrect) else None
        eff = None
        self.recomputes = 0
        if mode == "replicated" and self.em_mode == "auto" and self.world > 1 and getattr(self, "_auto_modes", {}).get(self._auto_key) is None:
            p = self.engine.em_problem(length, vec.rowptr, vec.ids, vec.counts, exp.numMappedFragments())
            self._measure_auto(p, length.numel())
            mode = self._pick_mode(vec.nnz)
            if mode != "replicated":
                p.close(); p = None
        else:
            p = None
        if mode in ("single", "replicated"):
            if p is None:
                p = self.engine.em_problem(length, vec.rowptr, vec.ids, vec.counts, exp.numMappedFragments())
            if bias is not None:
                rc, st, eff, self.recomputes = p.optimize_bias(bias, **kw)
            else:
                rc, st = p.optimize(**kw)
        else:
            import torch.distributed as dist
            rp_cpu = (vec.rowptr.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
            cuts = nnz_balanced_slices(rp_cpu, self.world)
            c0, c1 = cuts[self.rank], cuts[self.rank + 1]
            j0, j1 = int(rp_cpu[c0]), int(rp_cpu[c1])
            rp_loc = ((vec.rowptr[c0:c1 + 1].to(torch.int64) & 0xFFFFFFFF) - j0).to(torch.int32)
            p = self.engine.em_problem(length, rp_loc, vec.ids[j0:j1], vec.counts[c0:c1], exp.numMappedFragments())
            if bias is None:
                # the whole loop in C: begin -> all-reduce (union of the active sets) -> init -> { sweep, all-reduce,
                # update } with the stop latch polled every poll_every iterations -> finish
                # (one sweep kernel per iteration if EVERY rank's plan allows it: the two forms of the loop notice the stop one iteration
                #  apart, so the ranks agree first -- a MIN all-reduce of one flag per problem)
                if hasattr(p, "sharded_fused_ok"):
                    flag = torch.tensor([int(p.sharded_fused_ok())], dtype=torch.int32, device=(self.engine.device if dist.get_backend(self.group) == "nccl" else "cpu"))
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                    p.set_sharded_fused(bool(int(flag.item())))
                rc, st = p.optimize_sharded(self._allreduce(), poll_every=self.poll_every, **kw)
            else:
                # doBiasCorrect: the loop runs in segments that end at the recompute iterations: the stop bounds are lowered
                # to the next hook, and at a hook the reference would reach (its while condition still true, :820) every
                # rank recomputes the lengths from the replicated alpha; rank 0's result is broadcast so that the ranks
                # stay bit-identical (the 4096-bin expectation is summed with atomics).  Piecewise API, driven from here.
                p.begin(**kw)
                ao = p.alpha_out_view()
                dist.all_reduce(ao, op=dist.ReduceOp.SUM, group=self.group)     # union of the active sets
                p.init()
                user_min, user_max = kw["min_iter"], kw["max_iter"]
                it, conv = 0, False
                while not (it >= user_min and (it >= user_max or conv)):
                    if it in RECOMPUTE_ITERS:
                        new_len, _ = bias.update(p.length_view(), p.alpha_view())
                        dist.broadcast(new_len, src=dist.get_global_rank(self.group, 0), group=self.group)
                        p.rebase(new_len)
                        self.recomputes += 1
                    bound = max(user_min, user_max)                            # the loop cannot end before either
                    nxt = min([h for h in RECOMPUTE_ITERS if h > it] + [bound])
                    p.set_bounds(min(user_min, nxt), nxt)
                    done = False
                    while not done:
                        for _ in range(self.poll_every):
                            p.sweep()
                            dist.all_reduce(ao, op=dist.ReduceOp.SUM, group=self.group)
                            p.update()
                        done, seg = p.poll()
                    it, conv = int(seg["iters"]), bool(seg["converged"])
                eff = p.length_view().clone()
                rc, st = p.finish()
        self.problem = p
        self.problem_mode = mode                          # "sharded": self.problem holds this rank's class slice only
        ok = rc == 0
        if ok:
            txps.estCount.copy_(p.alpha); txps.mass.copy_(p.mass)
            if eff is not None:
                txps.EffectiveLength.copy_(eff)                                # :888
                es, eg = bias.expected()
                exp.setExpectedSeqBias(es); exp.setExpectedGCBias(eg)
        return ok, st, mode

    # ---- posterior sampling: draws are independent, so they are split over the ranks ------------
    def _share(self, n):
        base, extra = divmod(int(n), self.world)
        return base + (1 if self.rank < extra else 0)

    def _gather_rows(self, mine, n_total):
        """concatenate every rank's rows (rank order) on all ranks"""
        if self.world == 1:
            return mine
        flat = _all_gather_var(mine.reshape(-1), self.group, self.world)
        M = mine.shape[1]
        return torch.cat([f.reshape(-1, M) for f in flat], 0)[:n_total]

    def bootstrap(self, n, seed=1, tol=0.01, max_iter=10000):
        """gatherBootstraps over all ranks (classes replicated, draws split, no collective
        until the results are gathered).  Rank r draws its share with the stream seed + r * golden.
        Needs a finished run() in replicated / single mode.  -> float64 [n, M] on every rank"""
        p = self._full_problem()
        rc, out, _ = p.bootstrap(self._share(n), seed=(int(seed) + self.rank * 0x9E3779B97F4A7C15) & (2 ** 64 - 1),
                                 use_vbem=self.sopt.useVBOpt, tol=tol, max_iter=max_iter)
        if rc:
            raise RuntimeError(f"bootstrap failed on rank {self.rank}: rc={rc}")
        return self._gather_rows(out, n)

    def gibbs(self, n, seed=1, n_chains=0):
        """CollapsedGibbsSampler::sample over all ranks: every rank runs its own chains for its share of
        the draws.  -> int32 [n, M] on every rank"""
        exp, sopt = self.exp, self.sopt
        txps = exp.transcripts()
        vec = self.last_vec                               # the table the EM ran on (any merge mode)
        length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
        rc, out = self.engine.gibbs_sample(length, txps.mass, vec.rowptr, vec.ids, vec.counts, exp.numMappedFragments(),
                                           self._share(n), n_chains=n_chains,
                                           seed=(int(seed) + self.rank * 0x9E3779B97F4A7C15) & (2 ** 64 - 1))
        if rc:
            raise RuntimeError(f"gibbs failed on rank {self.rank}: rc={rc}")
        return self._gather_rows(out, n)

    def _full_problem(self):
        """The EM problem over ALL merged classes.  After a "sharded" run self.problem holds only this rank's slice
        (resampling or timing it would silently use 1/N of the classes), so a problem over the whole table is built
        once and kept for the samplers."""
        if self.problem is None:
            raise RuntimeError("run() first")
        if getattr(self, "problem_mode", "single") != "sharded":
            return self.problem
        if getattr(self, "_full", None) is None or self._full_vec is not self.last_vec:
            if getattr(self, "_full", None) is not None:
                self._full.close()
            exp, sopt, vec = self.exp, self.sopt, self.last_vec
            txps = exp.transcripts()
            length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
            self._full = self.engine.em_problem(length, vec.rowptr, vec.ids, vec.counts, exp.numMappedFragments())
            self._full_vec = vec
        return self._full

    def time_sweep(self, n=200):
        """average duration of one sweep launch of THIS rank's problem (its class slice in sharded mode)"""
        p = self.problem
        return p.time_sweep(n, use_vbem=self.sopt.useVBOpt, tol=self.tol, min_iter=self.min_iter, max_iter=self.max_iter)
