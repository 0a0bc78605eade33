"""quant.sf / aux/ output -- host mirror of src/GZipWriter.cpp:51-92 (eq_classes.txt), :94-192 (writeMeta),
:194-248 (quant.sf) and :249-285 (writeBootstrap)."""
import gzip
import json
import os

import numpy as np
import torch

from . import _lib
from .experiment import ReadExperiment, SailfishOpts


def tpm(readExp: ReadExperiment, sopt: SailfishOpts):
    """TPM column (GZipWriter.cpp:216-245), computed on the device; returns a float64 tensor."""
    txps = readExp.transcripts()
    length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
    out = torch.zeros(len(txps), dtype=torch.float64, device=txps.device)
    with torch.cuda.device(txps.device):
        _lib.check(_lib.lib().sfgpu_tpm(_lib.ptr(txps.estCount), _lib.ptr(length.contiguous()), len(txps),
                                        float(readExp.numMappedFragments()), _lib.ptr(out), _lib.current_stream_ptr()))
    return out, length


def fmt_g(x):
    """cppformat `{}` for double == printf %g (6 significant digits), include/spdlog/details/format.h:2898-2914."""
    return "%g" % x


def write_abundances(path, readExp: ReadExperiment, sopt: SailfishOpts):
    """writeAbundances (GZipWriter.cpp:194-248): Name, Length, EffectiveLength, TPM, NumReads."""
    txps = readExp.transcripts()
    t, length = tpm(readExp, sopt)
    t = t.cpu().numpy(); length = length.cpu().numpy()
    cnt = txps.estCount.cpu().numpy()
    ref = txps.RefLength.cpu().numpy().view(np.uint32)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "quant.sf"), "w") as f:
        f.write("Name\tLength\tEffectiveLength\tTPM\tNumReads\n")
        for i, name in enumerate(txps.RefName):
            eff = fmt_g(float(ref[i])) if sopt.noEffectiveLengthCorrection else fmt_g(length[i])
            f.write(f"{name}\t{int(ref[i])}\t{eff}\t{fmt_g(t[i])}\t{fmt_g(cnt[i])}\n")
    return True


def write_equiv_counts(path, readExp: ReadExperiment, sopt: SailfishOpts):
    """writeEquivCounts (GZipWriter.cpp:51-92): aux/eq_classes.txt (canonical class order)."""
    txps = readExp.transcripts()
    rowptr, ids, counts, _ = readExp.equivalenceClassBuilder().eqVec().to_numpy()
    aux = os.path.join(path, sopt.auxDir)
    os.makedirs(aux, exist_ok=True)
    with open(os.path.join(aux, "eq_classes.txt"), "w") as f:
        f.write(f"{len(txps)}\n{len(counts)}\n")
        for name in txps.RefName:
            f.write(name + "\n")
        for c in range(len(counts)):
            lab = ids[rowptr[c]:rowptr[c + 1]]
            f.write(f"{len(lab)}\t" + "".join(f"{t}\t" for t in lab) + f"{counts[c]}\n")
    return True


SAILFISH_VERSION = "0.10.0"        # sailfish::version (include/SailfishConfig.hpp:32), the release mirrored here
NUM_BIAS_BINS = 4 ** 6             # ReadKmerDist<6>::counts (include/ReadExperiment.hpp:249, ReadKmerDist.hpp:16)


def write_meta(path, readExp: ReadExperiment, sopt: SailfishOpts, start_time: str):
    """writeMeta (GZipWriter.cpp:94-192): aux/bootstrap/names.tsv.gz when sampling is requested, aux/fld.gz and
    the bias-model vectors expected_bias.gz (float64[4096]) / observed_bias.gz (int32[4096]) / expected_gc.gz
    (float64[101]) / observed_gc.gz (int32[101]) (:145-162; all ones without bias correction, as in the reference)
    and aux/meta_info.json (cereal JSON: same keys, same order).  One difference, outside the hot path: fld.gz
    holds the stored fragment-length counts themselves (the reference writes a random_device-seeded 10 000-sample
    realisation of them, EmpiricalDistribution.cpp:125-143)."""
    txps = readExp.transcripts()
    aux = os.path.join(path, sopt.auxDir)
    os.makedirs(aux, exist_ok=True)
    n_boot = int(sopt.numBootstraps)
    n_samp = n_boot if n_boot > 0 else int(sopt.numGibbsSamples)
    if n_samp > 0:
        if len(txps) == 0:
            return False
        bs = os.path.join(aux, "bootstrap")
        os.makedirs(bs, exist_ok=True)
        with gzip.open(os.path.join(bs, "names.tsv.gz"), "wb", compresslevel=6) as f:
            f.write(("\t".join(txps.RefName) + "\n").encode())
    fld = readExp.fragLengthDist()
    fld = np.zeros(sopt.maxFragLen, np.int32) if fld is None else np.asarray(fld, np.int32)
    with gzip.open(os.path.join(aux, "fld.gz"), "wb", compresslevel=6) as f:
        f.write(fld.tobytes())                                   # writeVectorToFile: raw little-endian binary
    for name, vec, dt in (("expected_bias.gz", readExp.expectedSeqBias(), np.float64), ("observed_bias.gz", readExp.readBias(), np.int32),
                          ("expected_gc.gz", readExp.expectedGCBias(), np.float64), ("observed_gc.gz", readExp.observedGC(), np.int32)):
        with gzip.open(os.path.join(aux, name), "wb", compresslevel=6) as f:
            f.write(np.ascontiguousarray(vec).astype(dt).tobytes())
    samp_type = "bootstrap" if n_boot > 0 else ("gibbs" if n_samp > 0 else "none")
    n_obs = readExp.numObservedFragments() or readExp.numMappedFragments()
    info = [("sf_version", SAILFISH_VERSION), ("samp_type", samp_type),
            ("frag_dist_length", int(len(fld) - 1)),             # EmpiricalDistribution::maxValue() of vals = 0..n-1
            ("bias_correct", bool(sopt.biasCorrect)), ("num_bias_bins", NUM_BIAS_BINS),
            ("num_targets", len(txps)), ("num_bootstraps", n_boot),
            ("num_processed", int(n_obs)), ("num_mapped", int(readExp.numMappedFragments())),
            ("percent_mapped", (readExp.numMappedFragments() / n_obs * 100.0) if n_obs else 0.0),
            ("call", "quant"), ("start_time", start_time)]
    with open(os.path.join(aux, "meta_info.json"), "w") as f:
        f.write(json.dumps(dict(info), indent=4))
    return True


class BootstrapWriter:
    """writeBootstrap<T> (GZipWriter.cpp:249-285): every sample is appended as raw binary (float64 for
    bootstrap replicates, int32 for Gibbs samples) to aux/bootstrap/bootstraps.gz.  An instance is the callback
    of EMProblem.bootstrap / gibbs_sample (the C ABI calls it one sample at a time, in draw order)."""

    def __init__(self, path, sopt: SailfishOpts, logger=None):
        self._dir = os.path.join(path, sopt.auxDir, "bootstrap")
        self._f = None
        self._log = logger
        self.written = 0

    def __call__(self, abund):
        if self._f is None:
            os.makedirs(self._dir, exist_ok=True)
            self._f = gzip.open(os.path.join(self._dir, "bootstraps.gz"), "wb", compresslevel=6)
        self._f.write(np.ascontiguousarray(abund).tobytes())
        self.written += 1
        if self._log:
            self._log(0, f"wrote {self.written} bootstraps")
        return True

    def close(self):
        if self._f is not None:
            self._f.close(); self._f = None
