"""Quasi-mapping front end on the device -- where the reference calls RapMap (SACollector inside processReadsQuasi,
src/SailfishQuantify.cpp:141-142, 192-213, 487-488, 526-528): reads in, the hit records of sailfish_amd.hits out.

RapMap is not part of the reference tree (fetched at build time); this is an exact-seed mapper with its own contract
(csrc/mapper.hip), not RapMap's suffix-array search, and parity with RapMap is unpinned.  Host side: sequence packing
(bytes + offsets), batching, and the driver `quantify_reads` = index -> map -> quantify()."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .hits import HIT_DTYPE


def pack_sequences(seqs, device="cpu"):
    """list of str / bytes -> (uint8 tensor of the bases back to back, int64 offsets[n + 1])"""
    raw = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    off = np.zeros(len(raw) + 1, np.int64)
    np.cumsum([len(r) for r in raw], out=off[1:])
    buf = np.frombuffer(b"".join(raw) or b"\0", dtype=np.uint8).copy()
    return torch.from_numpy(buf).to(device), torch.from_numpy(off).to(device)


class QuasiIndex:
    """k-mer index of a transcriptome on the device (sfgpu_index_build)."""

    def __init__(self, sequences, k=31, max_occ=1000, device="cuda", seeds=2, seed_len=None):
        """seed_len: None = the library's default (scan mode, seeds of min(19, k) bases, matches extended to maximal length);
        0 = the end-seed contract (`seeds` exact k-mers per strand); 8 .. k = scan mode with seeds of that length"""
        self.device = torch.device(device)
        self._L = _lib.lib()
        seq, off = pack_sequences(sequences, self.device)
        self.ref_len = (off[1:] - off[:-1]).to(torch.int32).contiguous()
        self.M = len(sequences)
        self._keep = (seq.contiguous(), off[:-1].contiguous())
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            _lib.check(self._L.sfgpu_index_build(C.byref(self._h), _lib.ptr(self._keep[0]), _lib.ptr(self._keep[1]), _lib.ptr(self.ref_len),
                                                 self.M, int(k), int(max_occ), _lib.current_stream_ptr()))
        kk, npos, nk = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _lib.check(self._L.sfgpu_index_info(self._h, C.byref(kk), C.byref(npos), C.byref(nk)))
        self.k, self.n_positions, self.n_kmers = kk.value, npos.value, nk.value
        self.seeds = 2
        self.seed_len = min(19, self.k)
        if seed_len is not None:
            self.set_scan(seed_len)
        if seeds != 2:
            self.set_seeds(seeds)
            if seed_len is None:
                self.set_scan(0)                           # asking for S end seeds selects the end-seed contract

    def set_scan(self, seed_len):
        """sfgpu_index_set_scan: 0 = end seeds; 8 .. k = scan mode (maximal-match extension) with seeds of that length"""
        _lib.check(self._L.sfgpu_index_set_scan(self._h, int(seed_len)))
        self.seed_len = int(seed_len)

    def set_seeds(self, seeds):
        """seeds per strand (sfgpu_index_set_seeds): 2 = offsets 0 and len - k, every hit kept; 3 .. 8 = seeds spread evenly over
        the read, only the (transcript, strand) pairs that the most seeds hit are kept (more sensitive on reads with errors)"""
        _lib.check(self._L.sfgpu_index_set_seeds(self._h, int(seeds)))
        self.seeds = int(seeds)

    def map_reads(self, reads1, reads2=None):
        """reads1 / reads2: lists of str / bytes, or (uint8 tensor, int64 offsets) pairs already packed.
        -> (hits: uint8 device tensor [n_hits * 24] of HIT_DTYPE records, offsets: int32 device tensor [R + 1])"""
        dev = self.device
        s1, o1 = reads1 if isinstance(reads1, tuple) else pack_sequences(reads1)
        s1, o1 = s1.to(dev).contiguous(), o1.to(dev).contiguous()
        n = int(o1.numel()) - 1
        s2 = o2 = None
        if reads2 is not None:
            s2, o2 = reads2 if isinstance(reads2, tuple) else pack_sequences(reads2)
            s2, o2 = s2.to(dev).contiguous(), o2.to(dev).contiguous()
            assert int(o2.numel()) - 1 == n, "both mate files hold the same number of reads"
        off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        n_hits = C.c_uint64(0)
        cap = max(4 * n, 1024)
        with torch.cuda.device(dev):
            torch.cuda.current_stream().synchronize()
            for _ in range(2):                               # second round only if the first capacity guess was too small
                hits = torch.empty(cap * 24, dtype=torch.uint8, device=dev)
                rc = self._L.sfgpu_map_reads(self._h, _lib.ptr(s1), _lib.ptr(o1), _lib.ptr(s2), _lib.ptr(o2), n, _lib.ptr(hits), cap, _lib.ptr(off),
                                             C.byref(n_hits), _lib.current_stream_ptr())
                if rc == _lib.OK or n_hits.value <= cap:
                    break
                cap = n_hits.value
            _lib.check(rc)
        return hits[: n_hits.value * 24], off

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sfgpu_index_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hits_to_numpy(hits, offsets):
    return hits.cpu().numpy().view(HIT_DTYPE), offsets.cpu().numpy().view(np.uint32)


def quantify_reads(names, sequences, reads1, reads2, lib_format, out_dir, sopt=None, *, k=31, batch_reads=1_000_000, device="cuda", **kw):
    """`sailfish quant` from the reads on: index the transcriptome, map the reads in batches (the reference's parser jobs),
    and hand the hit records to quant.quantify (filtering, classes, effective lengths, EM, writers).  -> (rc, experiment)"""
    from . import quant
    idx = QuasiIndex(sequences, k=k, device=device)
    n = len(reads1)

    def batches():
        for a in range(0, n, batch_reads):
            b = min(n, a + batch_reads)
            h, o = idx.map_reads(reads1[a:b], None if reads2 is None else reads2[a:b])
            yield h, o
    seq_kw = {}
    if sopt is not None and (getattr(sopt, "biasCorrect", False) or getattr(sopt, "gcBiasCorrect", False)):
        s, o = pack_sequences([x + "$" if isinstance(x, str) else bytes(x) + b"$" for x in sequences])
        seq_kw = dict(seq=bytes(s.numpy().tobytes()), seq_off=o[:-1].numpy())
    rc, exp = quant.quantify(names, idx.ref_len.cpu().numpy().view(np.uint32), batches(), lib_format, out_dir, sopt, device=device, **seq_kw, **kw)
    idx.close()
    return rc, exp
