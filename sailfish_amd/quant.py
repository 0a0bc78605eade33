"""`sailfish quant` after the mapper -- host mirror of the driver in src/SailfishQuantify.cpp:1160-1440
(salmonQuantify) and of quasiMapReads' bookkeeping (:840-1046), over the pieces of this package.

The mapper (RapMap) is outside the path: it is replaced by whoever produces the hit records (sfgpu_hit, one per
quasi-mapping).  From there on everything the reference does is done here, in its order, on the device:

    cmd_info.json -> option checks -> per batch: hit filtering (+ bias / GC samples) -> addGroup
    -> finish -> effective lengths -> [dumpEq] -> optimize (with the bias recompute) -> quant.sf, aux/ meta
    -> [Gibbs | bootstrap samples] -> [gene-level estimates]

Return value as the reference's `int salmonQuantify`: 0 on success, 1 where it logs an error and returns 1."""
import json
import os
import time

import numpy as np
import torch

from . import efflen as _efflen
from . import genes as _genes
from . import hits as _hits
from . import writer as _writer
from .experiment import ReadExperiment, SailfishOpts, Transcripts
from .gibbs import CollapsedGibbsSampler
from .optimizer import CollapsedEMOptimizer


def write_cmd_info(out_dir, options):
    """cmd_info.json (:1262-1276): sf_version first, then the options in the order they were given."""
    os.makedirs(out_dir, exist_ok=True)
    info = [("sf_version", _writer.SAILFISH_VERSION)] + [(k, v) for k, v in (options or {}).items()]
    with open(os.path.join(out_dir, "cmd_info.json"), "w") as f:
        f.write(json.dumps(dict(info), indent=4))


def quantify(names, ref_len, hit_batches, lib_format, out_dir, sopt: SailfishOpts = None, *, seq=None, seq_off=None,
             allow_orphans=False, ignore_lib_compat=False, enforce_lib_compat=False, allow_dovetail=False,
             num_bias_samples=1_000_000, gene_map=None, cmd_options=None, seed=None, device="cuda"):
    """names / ref_len: the index's transcripts.  hit_batches: iterable of (hits, hit_offsets) as hits.filter_hits takes
    them.  seq / seq_off: RapMapSAIndex::seq and txpOffsets (needed with biasCorrect / gcBiasCorrect).
    Returns (rc, ReadExperiment)."""
    sopt = sopt or SailfishOpts()
    log = sopt.jointLog or (lambda lvl, msg: None)
    dev = torch.device(device)
    start_time = time.asctime()
    write_cmd_info(out_dir, cmd_options)
    fmt = _hits.LIBRARY_FORMATS[lib_format.upper()] if isinstance(lib_format, str) else tuple(lib_format)
    paired = fmt[0] == 1
    if sopt.numGibbsSamples > 0 and sopt.numBootstraps > 0:                  # :1280-1286
        log(2, "You cannot perform both Gibbs sampling and bootstrapping. Please choose one.")
        return 1, None
    if sopt.biasCorrect and sopt.gcBiasCorrect:                              # :1293-1297
        log(2, "Enabling both sequence-specific and fragment GC bias correction simultaneously is not yet supported. "
               "Please disable one of these options.")
        return 1, None
    if sopt.gcBiasCorrect and not paired:                                    # :1298-1309
        log(1, "Fragment GC bias correction is currently only implemented for paired-end libraries. It is being disabled")
        sopt.gcBiasCorrect = False

    exp = ReadExperiment(Transcripts(list(names), ref_len, device=dev), sopt)
    do_bias = sopt.biasCorrect or sopt.gcBiasCorrect
    gc_table = None
    if do_bias:
        if seq is None:
            raise ValueError("bias correction needs the transcript sequences (seq, seq_off)")
        exp.setSequences(seq, seq_off)
        if sopt.gcBiasCorrect:
            gc_table = _hits.gc_prefix(exp._seq, exp._seq_off, exp.transcripts().RefLength)
    eq = exp.equivalenceClassBuilder()
    eq.start()                                                               # :1319
    fl = torch.zeros(sopt.maxFragLen, dtype=torch.int32, device=dev)
    rem_fl = int(sopt.numFragSamples)                                        # remainingFLOps (:874)
    rem_bias = int(num_bias_samples)
    d_rb = torch.from_numpy(exp.readBias().view(np.int32).copy()).to(dev) if sopt.biasCorrect else None
    d_gc = torch.from_numpy(exp.observedGC().view(np.int32).copy()).to(dev) if sopt.gcBiasCorrect else None
    stats = None
    for hits, off in hit_batches:                                            # the mapping threads' loop bodies
        ids, out_off, rem_fl, stats = _hits.filter_hits(hits, off, fmt, paired_library=paired, allow_orphans=allow_orphans,
                                                        ignore_lib_compat=ignore_lib_compat, enforce_lib_compat=enforce_lib_compat,
                                                        allow_dovetail=allow_dovetail, max_read_occs=sopt.maxReadOccs,
                                                        max_frag_len=sopt.maxFragLen, fl_counts=fl, remaining_fl_ops=rem_fl,
                                                        stats=stats, device=dev)
        if do_bias:
            rem_bias, _, _ = _hits.sample_bias(hits, off, fmt, exp._seq, exp._seq_off, exp.transcripts().RefLength, read_bias=d_rb,
                                               remaining_bias_samples=rem_bias, observed_gc=d_gc, gc_prefix_table=gc_table,
                                               gc_size_samp=sopt.gcSampFactor, paired_library=paired, allow_orphans=allow_orphans,
                                               max_read_occs=sopt.maxReadOccs, max_frag_len=sopt.maxFragLen, device=dev)
        eq.add_batch(ids, out_off)
    eq.finish()                                                              # :1324
    stats = stats or dict(n_observed=0, n_mapped=0, n_fwd=0, n_rc=0)
    exp.setNumObservedFragments(stats["n_observed"]); exp.setNumMappedFragments(stats["n_mapped"])
    exp.addNumFwd(stats["n_fwd"]); exp.addNumRC(stats["n_rc"])
    if d_rb is not None:
        exp.readBias()[:] = d_rb.cpu().numpy().view(np.uint32)
    if d_gc is not None:
        exp.observedGC()[:] = d_gc.cpu().numpy().view(np.uint32)
    # quasiMapReads' tail: the fragment length distribution and the effective lengths (:937-991, :1035-1043)
    if paired:
        _efflen.set_effective_lengths(exp, sopt, fl_counts=fl.cpu().numpy().view(np.uint32), remaining_fl_ops=rem_fl)
    else:
        _efflen.set_effective_lengths(exp, sopt)
    if sopt.dumpEq:                                                          # :1331-1333
        _writer.write_equiv_counts(out_dir, exp, sopt)
    opt = CollapsedEMOptimizer()
    log(0, "Starting optimizer:\n")
    if not opt.optimize(exp, sopt, 0.01, 10000):                             # :1343-1350
        log(2, "Encountered error during optimization.\nThis should not happen.\nPlease file a bug report on GitHub.\n")
        return 1, exp
    log(0, "Finished optimizer")
    _writer.write_abundances(out_dir, exp, sopt)                             # :1375
    _writer.write_meta(out_dir, exp, sopt, start_time)                       # :1377
    if sopt.numGibbsSamples > 0:                                             # :1379-1397
        w = _writer.BootstrapWriter(out_dir, sopt)
        ok = CollapsedGibbsSampler().sample(exp, sopt, w, sopt.numGibbsSamples, seed=seed)
        w.close()
        if not ok:
            return 1, exp
    elif sopt.numBootstraps > 0:                                             # :1398-1413
        w = _writer.BootstrapWriter(out_dir, sopt)
        ok = opt.gatherBootstraps(exp, sopt, w, 0.01, 10000, seed=seed)
        w.close()
        if not ok:
            return 1, exp
    if gene_map is not None:                                                 # :1416-1426
        try:
            _genes.generate_gene_level_estimates(gene_map, out_dir)
        except ValueError as e:
            log(2, f"Error: [{e}] when trying to compute gene-level estimates. The gene-level file(s) may not exist")
    return 0, exp
