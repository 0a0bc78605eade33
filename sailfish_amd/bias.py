"""Bias-aware effective lengths -- host mirror of sailfish::utils::updateEffectiveLengths
(src/SailfishUtils.cpp:611-926) over the C ABI (sfgpu_bias_*)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

NUM_KMER_BINS = 4096      # ReadKmerDist<6> (include/ReadExperiment.hpp:249)
NUM_GC_BINS = 101         # observedGC_ / expectedGC_ (include/ReadExperiment.hpp:46-47)


class BiasModel:
    """What updateEffectiveLengths reads from ReadExperiment / SailfishOpts, resident on the device (sfgpu_bias).

    seq        uint8 tensor: RapMapSAIndex::seq (all transcripts, any separator bytes between them)
    seq_off    int64 tensor [M]: txpOffsets
    ref_len    int32 tensor [M] holding the uint32 lengths (Transcripts.RefLength)
    txp_eff_len float64 tensor [M]: Transcript::EffectiveLength after the FLD correction
    fl_counts  the vector given to ReadExperiment::setFragLengthDist (length maxFragLen)
    read_bias  4096 uint32 (ReadKmerDist<6>::counts), observed_gc 101 uint32 -- pseudo-counts included
    """

    def __init__(self, seq, seq_off, ref_len, txp_eff_len, fl_counts, read_bias=None, observed_gc=None,
                 num_fwd=0, num_rc=0, seq_bias=False, gc_bias=False, gc_speed_samp=1, gc_size_samp=1):
        self._L = _lib.lib()
        self.device = seq.device
        self.M = int(ref_len.numel())
        self._keep = (seq.contiguous(), seq_off.to(torch.int64).contiguous(), ref_len.contiguous(),
                      txp_eff_len.to(torch.float64).contiguous())
        fl = np.ascontiguousarray(fl_counts, dtype=np.uint32)
        rb = None if read_bias is None else np.ascontiguousarray(read_bias, dtype=np.uint32)
        og = None if observed_gc is None else np.ascontiguousarray(observed_gc, dtype=np.uint32)
        assert rb is None or rb.size == NUM_KMER_BINS
        assert og is None or og.size == NUM_GC_BINS
        inp = _lib.BiasInputs(self.M, _lib.ptr(self._keep[0]).value, _lib.ptr(self._keep[1]).value,
                              _lib.ptr(self._keep[2]).value, _lib.ptr(self._keep[3]).value,
                              fl.ctypes.data, fl.size, int(gc_speed_samp),
                              None if rb is None else rb.ctypes.data, None if og is None else og.ctypes.data,
                              int(num_fwd), int(num_rc), int(bool(seq_bias)), int(bool(gc_bias)), int(gc_size_samp), 0)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.sfgpu_bias_create(C.byref(h), C.byref(inp), _lib.current_stream_ptr()))
        self._h = h

    def update(self, eff_in, alphas):
        """effLensOut = updateEffectiveLengths(sopt, readExp, effLensIn, alphas) -> (tensor[M], stats dict)"""
        eff_in = eff_in.to(torch.float64).contiguous(); alphas = alphas.to(torch.float64).contiguous()
        out = torch.empty_like(eff_in)
        st = _lib.BiasStats()
        with torch.cuda.device(self.device):
            _lib.check(self._L.sfgpu_bias_update(self._h, _lib.ptr(eff_in), _lib.ptr(alphas), _lib.ptr(out), C.byref(st),
                                                 _lib.current_stream_ptr()))
        return out, st.as_dict()

    def expected(self):
        """(expectedSeqBias float64[4096], expectedGCBias float64[101]) after the last update"""
        es = np.zeros(NUM_KMER_BINS); eg = np.zeros(NUM_GC_BINS)
        _lib.check(self._L.sfgpu_bias_expected(self._h, es.ctypes.data, eg.ctypes.data))
        return es, eg

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sfgpu_bias_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
