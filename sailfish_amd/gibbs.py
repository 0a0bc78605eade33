"""CollapsedGibbsSampler -- host mirror of include/CollapsedGibbsSampler.hpp:27-31 /
src/CollapsedGibbsSampler.cpp:198-291 over the C ABI (sfgpu_gibbs_sample)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .experiment import ReadExperiment, SailfishOpts


def gibbs_sample(length, mass, rowptr, ids, counts, num_mapped, n_samples, n_chains=0, seed=1, callback=None):
    """-> (rc, samples int32[n_samples, M] device tensor)"""
    L = _lib.lib()
    dev = length.device
    keep = (length.contiguous(), rowptr.contiguous(), ids.contiguous(), counts.contiguous(), mass.contiguous())
    M = keep[0].numel(); Cn = keep[1].numel() - 1
    prob = _lib.Problem(M, _lib.ptr(keep[0]).value, Cn, _lib.ptr(keep[1]).value,
                        (_lib.ptr(keep[2]).value if keep[2].numel() else None),
                        (_lib.ptr(keep[3]).value if keep[3].numel() else None), int(num_mapped))
    out = torch.zeros((n_samples, M), dtype=torch.int32, device=dev)
    if callback is None:
        cb = _lib.GIBBS_CB(0)
    else:
        cb = _lib.GIBBS_CB(lambda p, m, _u: 1 if callback(np.ctypeslib.as_array(p, shape=(m,)).copy()) else 0)
    with torch.cuda.device(dev):
        torch.cuda.current_stream().synchronize()
        rc = L.sfgpu_gibbs_sample(C.byref(prob), _lib.ptr(keep[4]), int(n_samples), int(n_chains), int(seed), _lib.ptr(out), cb,
                                  None, _lib.current_stream_ptr())
    return rc, out


class CollapsedGibbsSampler:
    """sample(readExp, sopt, writeSample, numSamples) as called at src/SailfishQuantify.cpp:1385-1387."""

    def sample(self, readExp: ReadExperiment, sopt: SailfishOpts, writeSample, numSamples: int, seed=None, n_chains=0) -> bool:
        if sopt.jointLog is not None:
            _lib.set_logger(sopt.jointLog)
        txps = readExp.transcripts()
        vec = readExp.equivalenceClassBuilder().eqVec()
        # the aux weights are the ones optimize() left in eqVec: count/effLen normalised per class
        length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        rc, out = gibbs_sample(length, txps.mass, vec.rowptr, vec.ids, vec.counts, readExp.numMappedFragments(),
                               numSamples, n_chains=n_chains, seed=seed, callback=writeSample)
        self.last_samples = out
        _lib.lib().sfgpu_pool_trim()        # the chain state (4 * nnz * chains bytes) is not needed again: sample() ends a quantification
        _lib.check(rc)
        return True
