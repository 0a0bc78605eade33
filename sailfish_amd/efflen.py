"""Fragment-length distribution -> effective lengths: host mirror of the file-static helpers in
src/SailfishQuantify.cpp (:648-673, :675-704, :769-807, :809-838) and the selection logic at
:961-991 (paired end) / :1035-1043 (single end)."""
import numpy as np
import torch

from . import _lib
from .experiment import ReadExperiment, SailfishOpts


def normal_cf(sopt: SailfishOpts):
    """getNormalFragLengthDist (:648-673)."""
    cf = np.zeros(sopt.maxFragLen)
    _lib.check(_lib.lib().sfgpu_cf_gaussian(sopt.maxFragLen, sopt.fragLenDistPriorMean, sopt.fragLenDistPriorSD, _lib.ptr(cf)))
    return cf


def normal_counts(sopt: SailfishOpts):
    """getNormalFragLengthCounts (:675-704): the integer FLD stored in the ReadExperiment."""
    i = np.arange(sopt.maxFragLen, dtype=np.float64)
    inv = 1.0 / float(sopt.fragLenDistPriorSD)
    x = inv * (i - float(sopt.fragLenDistPriorMean))
    d = np.exp(-0.5 * x * x) * inv
    total = 0.0
    for v in d:          # serial sum, as the reference
        total += v
    if total <= 0:
        return np.zeros(sopt.maxFragLen, np.int32)
    # std::round: half away from zero (all values are >= 0 here)
    return np.floor(d * sopt.numFragSamples / total + 0.5).astype(np.int32)


def counts_cf(fl_counts):
    """correctionFactorsFromCounts (:769-807)."""
    fl = np.ascontiguousarray(fl_counts, dtype=np.uint32)
    cf = np.zeros(len(fl))
    _lib.check(_lib.lib().sfgpu_cf_counts(_lib.ptr(fl), len(fl), _lib.ptr(cf)))
    return cf


def set_effective_lengths(readExp: ReadExperiment, sopt: SailfishOpts, fl_counts=None, remaining_fl_ops=1):
    """The post-mapping block of quasiMapReads: writes Transcript::EffectiveLength on the device.
      single end, or paired end with too few unique-pair observations (remainingFLOps > 0):
          Gaussian prior table (:961-975, :1038-1042)
      else: empirical cumulative-mean table (:976-990), or with --unsmoothedFLD the empirical pdf itself
      --noEffectiveLengthCorrection: EffectiveLength = RefLength (:956-958)."""
    txps = readExp.transcripts()
    L = _lib.lib()
    M = len(txps)
    if sopt.noEffectiveLengthCorrection:
        cf = None
    elif fl_counts is None or remaining_fl_ops > 0:
        readExp.setFragLengthDist(normal_counts(sopt))
        cf = normal_cf(sopt)
    else:
        readExp.setFragLengthDist(np.asarray(fl_counts, dtype=np.int32))
        if sopt.useUnsmoothedFLD:        # computeEmpiricalEffectiveLengths (:985-986, :717-767)
            fl = np.ascontiguousarray(fl_counts, dtype=np.uint32)
            with torch.cuda.device(txps.device):
                _lib.check(L.sfgpu_efflen_empirical(_lib.ptr(fl), len(fl), _lib.ptr(txps.RefLength), M,
                                                    _lib.ptr(txps.EffectiveLength), _lib.current_stream_ptr()))
            return None
        cf = counts_cf(fl_counts)
    with torch.cuda.device(txps.device):
        _lib.check(L.sfgpu_efflen_smoothed(_lib.ptr(txps.RefLength), M, _lib.ptr(cf) if cf is not None else None,
                                           sopt.maxFragLen, _lib.ptr(txps.EffectiveLength), _lib.current_stream_ptr()))
    return cf
