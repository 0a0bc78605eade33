"""quant.sf / eq_classes.txt output -- host mirror of src/GZipWriter.cpp:51-92 and :194-248."""
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
