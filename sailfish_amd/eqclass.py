"""EquivalenceClassBuilder -- host mirror of include/EquivalenceClassBuilder.hpp:53-119 over
the C ABI (sfgpu_eq_*).  Same call sequence as the reference: start() -> addGroup()* (here
batched: add_batch) -> finish() -> eqVec()."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class EqVec:
    """eqVec() as device-resident CSR: class c has label ids[rowptr[c]:rowptr[c+1]] and
    count counts[c]; hashes[c] = TranscriptGroup::hash (XXH64 of the label)."""

    def __init__(self, rowptr, ids, counts, hashes, total_reads):
        self.rowptr, self.ids, self.counts, self.hashes = rowptr, ids, counts, hashes
        self.total_reads = total_reads

    def size(self):
        return self.rowptr.numel() - 1

    @property
    def nnz(self):
        return self.ids.numel()

    def to_numpy(self):
        u = lambda t, dt: t.cpu().numpy().view(dt)
        return (u(self.rowptr, np.uint32), u(self.ids, np.uint32), u(self.counts, np.uint64), u(self.hashes, np.uint64))


def _as_dev_u32(a, device):
    """Accept numpy/torch (any 32-bit int dtype) and return a contiguous device int32 tensor
    holding the same bits."""
    if isinstance(a, torch.Tensor):
        t = a
        if t.dtype not in (torch.int32, torch.uint32):
            t = t.to(torch.int64).to(torch.int32) if t.dtype != torch.int64 else (t & 0xFFFFFFFF).to(torch.int32)
        return t.contiguous().to(device)
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint32:
        a = a.astype(np.uint32)
    return torch.from_numpy(a.view(np.int32).copy()).to(device)


class EquivalenceClassBuilder:
    def __init__(self, logger=None, expected_classes=0, device="cuda"):
        self._L = _lib.lib()
        self.device = torch.device(device)
        if logger is not None:
            _lib.set_logger(logger)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            sp = _lib.current_stream_ptr()
            _lib.check(self._L.sfgpu_eq_create(C.byref(h), int(expected_classes), sp))
        self._stream = sp.value or 0          # the builder works on its creation stream
        self._h = h
        self._pending = []          # single addGroup() calls buffered into one batch
        self._pending_n = 0
        self._vec = None
        self.n_classes = self.nnz = self.total_reads = 0

    # -- reference interface ---------------------------------------------------------------
    def start(self):
        """start() (EquivalenceClassBuilder.hpp:62)."""
        _lib.check(self._L.sfgpu_eq_start(self._h))
        self._pending, self._pending_n, self._vec = [], 0, None

    def addGroup(self, txps, weights=None):
        """addGroup(TranscriptGroup&&, std::vector<double>&) (:90-108) for one read.  The aux
        weights are all 1.0 at every reference call site and are overwritten by optimize(), so
        they are accepted and ignored.  Reads are buffered and sent as <=1000-read batches, the
        reference's parser-job size (src/SailfishQuantify.cpp:73)."""
        self._pending.append(np.asarray(txps, dtype=np.uint32))
        self._pending_n += 1
        if self._pending_n >= 1000:
            self._flush()

    def add_batch(self, ids, offsets):
        """Many reads at once: label r = ids[offsets[r]:offsets[r+1]] (uint32 both).  Device
        tensors are consumed in place; host arrays go through the library's staging copy."""
        self._flush()
        n = int(offsets.shape[0]) - 1
        if n <= 0:
            return
        # device path only when BOTH arrays already live on a device; a mixed pair is taken through the host path
        on_dev = (isinstance(ids, torch.Tensor) and ids.is_cuda) and (isinstance(offsets, torch.Tensor) and offsets.is_cuda)
        if on_dev:
            ids_t = _as_dev_u32(ids, self.device); off_t = _as_dev_u32(offsets, self.device)
            with torch.cuda.device(self.device):
                if (torch.cuda.current_stream().cuda_stream or 0) != self._stream:
                    torch.cuda.current_stream().synchronize()   # the builder works on its creation stream
            _lib.check(self._L.sfgpu_eq_add_batch_device(self._h, _lib.ptr(ids_t), _lib.ptr(off_t), n))
        else:
            def _host_u32(a):          # no copy for 32-bit integer HOST arrays (a pinned 8 GB batch must stay where it is); other
                                       # integer types are narrowed with one copy, device tensors of a mixed pair come to the host
                a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                a = np.ascontiguousarray(a)
                return a.view(np.uint32) if a.dtype in (np.int32, np.uint32) else a.astype(np.uint32)
            ids_h, off_h = _host_u32(ids), _host_u32(offsets)
            if ids_h.size == 0:
                ids_h = np.zeros(1, np.uint32)
            _lib.check(self._L.sfgpu_eq_add_batch_host(self._h, _lib.ptr(ids_h), _lib.ptr(off_h), n))

    def insertGroups(self, ids, offsets, counts):
        """insertGroup(TranscriptGroup, count) (:82-88) for many groups, with upsert semantics:
        group g = ids[offsets[g]:offsets[g+1]] is added with multiplicity counts[g].  Device tensors."""
        self._flush()
        n = int(offsets.shape[0]) - 1
        if n <= 0:
            return
        ids_t = _as_dev_u32(ids, self.device); off_t = _as_dev_u32(offsets, self.device)
        cnt_t = counts.to(torch.int64).contiguous().to(self.device)
        torch.cuda.current_stream().synchronize()
        _lib.check(self._L.sfgpu_eq_add_weighted_device(self._h, _lib.ptr(ids_t), _lib.ptr(off_t), _lib.ptr(cnt_t), n))

    def stats(self):
        st = _lib.EqStats()
        _lib.check(self._L.sfgpu_eq_get_stats(self._h, C.byref(st)))
        return dict(insert_ms=st.insert_ms, insert_launches=st.insert_launches, table_grows=st.table_grows,
                    deferred_reads=st.deferred_reads, table_slots=st.table_slots,
                    hot_reads=st.hot_reads, spilled_reads=st.spilled_reads, pipeline_drains=st.pipeline_drains)

    def finish(self):
        """finish() (:64-80): returns True; n_classes / total_reads are what the reference logs."""
        self._flush()
        nc, nnz, tot = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(self._L.sfgpu_eq_finish(self._h, C.byref(nc), C.byref(nnz), C.byref(tot)))
        self.n_classes, self.nnz, self.total_reads = nc.value, nnz.value, tot.value
        self._vec = None
        return True

    def eqVec(self) -> EqVec:
        """eqVec() (:110-112) -- canonical order, device CSR."""
        if self._vec is None:
            n, nnz = self.n_classes, self.nnz
            dev = self.device
            rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)      # every element is written by the export
            ids = torch.empty(nnz, dtype=torch.int32, device=dev)
            counts = torch.empty(n, dtype=torch.int64, device=dev)
            hashes = torch.empty(n, dtype=torch.int64, device=dev)
            # the export runs on the builder's creation stream: when that is still torch's current stream, the tensors above and
            # whatever consumes them next are ordered by the stream itself (two host waits less per quantification)
            with torch.cuda.device(dev):
                same_stream = (torch.cuda.current_stream().cuda_stream or 0) == self._stream
            if not same_stream:
                torch.cuda.current_stream().synchronize()
            _lib.check(self._L.sfgpu_eq_export_device(self._h, _lib.ptr(rowptr), _lib.ptr(ids), _lib.ptr(counts),
                                                      _lib.ptr(hashes)))
            if not same_stream:
                torch.cuda.synchronize(dev)
            self._vec = EqVec(rowptr, ids, counts, hashes, self.total_reads)
        return self._vec

    # -- helpers ---------------------------------------------------------------------------
    def _flush(self):
        if not self._pending_n:
            return
        lens = np.fromiter((len(p) for p in self._pending), dtype=np.uint32, count=self._pending_n)
        off = np.zeros(self._pending_n + 1, np.uint32)
        np.cumsum(lens, out=off[1:])
        ids = np.concatenate(self._pending) if off[-1] else np.zeros(1, np.uint32)
        n = self._pending_n
        self._pending, self._pending_n = [], 0
        _lib.check(self._L.sfgpu_eq_add_batch_host(self._h, _lib.ptr(ids), _lib.ptr(off), n))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sfgpu_eq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def xxh64_labels(ids, offsets, device="cuda"):
    """TranscriptGroup hash of every label in a packed batch (src/TranscriptGroup.cpp:9-12)."""
    L = _lib.lib()
    dev = torch.device(device)
    ids_t = _as_dev_u32(ids, dev); off_t = _as_dev_u32(offsets, dev)
    n = off_t.numel() - 1
    out = torch.zeros(max(n, 0), dtype=torch.int64, device=dev)
    if n > 0:
        if ids_t.numel() == 0:
            ids_t = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.sfgpu_xxh64_labels(_lib.ptr(ids_t), _lib.ptr(off_t), n, _lib.ptr(out), _lib.current_stream_ptr()))
    return out
