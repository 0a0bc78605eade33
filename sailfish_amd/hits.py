"""Per-read hit filtering on the device -- host mirror of the loop bodies of processReadsQuasi
(src/SailfishQuantify.cpp:215-417 paired end, :530-626 single end) and of sailfish::utils::compatibleHit / hitType
(src/SailfishUtils.cpp:157-289): the mapper's hit records go in, the packed transcript-id lists that
EquivalenceClassBuilder.add_batch takes come out (device-resident), plus the fragment-length samples and the
fragment counters of the ReadExperiment."""
import ctypes as C

import numpy as np
import torch

from . import _lib

# one record per hit: sfgpu_hit (include/sfgpu.h), fields of rapmap's QuasiAlignment that the loop reads
HIT_DTYPE = np.dtype([("tid", "<u4"), ("pos", "<i4"), ("mate_pos", "<i4"), ("frag_len", "<u4"), ("read_len", "<u2"),
                      ("mate_len", "<u2"), ("fwd", "u1"), ("mate_fwd", "u1"), ("mate_status", "u1"), ("pad_", "u1")])
assert HIT_DTYPE.itemsize == 24
SINGLE_END, PAIRED_END_LEFT, PAIRED_END_RIGHT, PAIRED_END_PAIRED = 0, 1, 2, 3        # rapmap::utils::MateStatus
SAME, AWAY, TOWARD, NONE = 0, 1, 2, 3                                                # ReadOrientation
SA, AS, S, A, U = 0, 1, 2, 3, 4                                                      # ReadStrandedness

# parseLibraryFormatStringNew's table (src/SailfishUtils.cpp:69-81): name -> (type, orientation, strandedness)
LIBRARY_FORMATS = {"IU": (1, TOWARD, U), "ISF": (1, TOWARD, SA), "ISR": (1, TOWARD, AS), "OU": (1, AWAY, U),
                   "OSF": (1, AWAY, SA), "OSR": (1, AWAY, AS), "MU": (1, SAME, U), "MSF": (1, SAME, S),
                   "MSR": (1, SAME, A), "U": (0, NONE, U), "SF": (0, NONE, S), "SR": (0, NONE, A)}


def format_id(fmt):
    """LibraryFormat::formatID (include/LibraryFormat.hpp:89-98): type | orientation << 1 | strandedness << 3"""
    t, o, s = fmt
    return (int(t) & 0x01) | ((int(o) & 0x3) << 1) | ((int(s) & 0x7) << 3)


def format_from_id(i):
    """LibraryFormat::formatFromID (include/LibraryFormat.hpp:34-85)"""
    return (int(i) & 0x01, (int(i) >> 1) & 0x3, (int(i) >> 3) & 0x7)


MAX_LIB_TYPE_ID = format_id((1, NONE, U))            # LibraryFormat::maxLibTypeID (:25-30)


def format_check(fmt):
    """LibraryFormat::check (src/LibraryFormat.cpp:6-51): is the combination meaningful?"""
    t, o, s = fmt
    if t == 0:                                       # single end: no orientation, no two-strand protocol
        return o == NONE and s not in (SA, AS)
    if o == NONE:
        return False
    if o == SAME:
        return s in (S, A, U)
    return s in (SA, AS, U)                          # AWAY / TOWARD: the mates come from different strands


def format_str(fmt):
    """operator<<(ostream&, LibraryFormat) (src/LibraryFormat.cpp:53-100), the text of the log lines"""
    t, o, s = fmt
    return ("Library format { type:" + ("single end", "paired end")[t] + ", relative orientation:" +
            {TOWARD: "inward", AWAY: "outward", SAME: "matching", NONE: "none"}[o] + ", strandedness:" +
            {SA: "(sense, antisense)", AS: "(antisense, sense)", S: "sense", A: "antisense", U: "unstranded"}[s] + " }")


def filter_hits(hits, hit_offsets, lib_format, sopt=None, *, paired_library=None, allow_orphans=False,
                ignore_lib_compat=False, enforce_lib_compat=False, allow_dovetail=False, max_read_occs=200,
                max_frag_len=1000, fl_counts=None, remaining_fl_ops=0, stats=None, device="cuda"):
    """hits: numpy structured array (HIT_DTYPE) or a uint8 device tensor of the same bytes; hit_offsets: uint32[R+1].
    lib_format: a name of LIBRARY_FORMATS or a (type, orientation, strandedness) triple.
    Returns (ids int32 device tensor, offsets int32 device tensor [R+1], remaining_fl_ops, stats dict); fl_counts
    (int32 device tensor [max_frag_len]) is updated in place when given."""
    dev = torch.device(device)
    if sopt is not None:
        max_read_occs, max_frag_len = sopt.maxReadOccs, sopt.maxFragLen
    fmt = LIBRARY_FORMATS[lib_format.upper()] if isinstance(lib_format, str) else tuple(lib_format)
    if paired_library is None:
        paired_library = fmt[0] == 1
    if isinstance(hits, torch.Tensor):
        d_hits = hits.to(dev).contiguous()
    else:
        h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        d_hits = torch.from_numpy(h.view(np.uint8).reshape(-1).copy()).to(dev)
    off = hit_offsets
    d_off = off.to(dev).contiguous() if isinstance(off, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(off, np.uint32).view(np.int32).copy()).to(dev)
    R = int(d_off.numel()) - 1
    n_hits = d_hits.numel() // 24
    ids = torch.empty(max(n_hits, 1), dtype=torch.int32, device=dev)
    out_off = torch.empty(R + 1, dtype=torch.int32, device=dev)
    o = _lib.FilterOpts(int(max_read_occs), int(max_frag_len), int(bool(paired_library)), int(not allow_orphans),
                        int(bool(ignore_lib_compat)), int(bool(enforce_lib_compat)), int(bool(allow_dovetail)),
                        _lib.LibFmt(int(fmt[0]), int(fmt[1]), int(fmt[2]), 0))
    st = _lib.FilterStats()
    if stats:
        for k, v in stats.items():
            setattr(st, k, int(v))
    rem = C.c_int64(int(remaining_fl_ops))
    with torch.cuda.device(dev):
        torch.cuda.current_stream().synchronize()
        _lib.check(_lib.lib().sfgpu_filter_hits(_lib.ptr(d_hits), _lib.ptr(d_off), R, C.byref(o), _lib.ptr(ids), _lib.ptr(out_off),
                                                _lib.ptr(fl_counts) if fl_counts is not None else None, C.byref(rem),
                                                C.byref(st), _lib.current_stream_ptr()))
    total = int(out_off[-1].item()) & 0xFFFFFFFF if R >= 0 else 0
    return ids[:total], out_off, int(rem.value), {k: int(getattr(st, k)) for k, _ in _lib.FilterStats._fields_}


def _device_hits(hits, hit_offsets, dev):
    if isinstance(hits, torch.Tensor):
        d_hits = hits.to(dev).contiguous()
    else:
        h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        d_hits = torch.from_numpy(h.view(np.uint8).reshape(-1).copy()).to(dev)
    off = hit_offsets
    d_off = off.to(dev).contiguous() if isinstance(off, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(off, np.uint32).view(np.int32).copy()).to(dev)
    return d_hits, d_off


def gc_prefix(seq, seq_off, ref_len):
    """Transcript::GCCount_ of every transcript (include/Transcript.hpp:183-196), laid out like `seq`: int32 device
    tensor of seq.numel() entries (4 bytes per base) -- what sample_bias reads for the fragment-GC samples."""
    out = torch.zeros(seq.numel(), dtype=torch.int32, device=seq.device)
    so = seq_off.to(torch.int64).contiguous()
    with torch.cuda.device(seq.device):
        _lib.check(_lib.lib().sfgpu_gc_prefix(_lib.ptr(seq), _lib.ptr(so), _lib.ptr(ref_len), int(ref_len.numel()), _lib.ptr(out),
                                              _lib.current_stream_ptr()))
        torch.cuda.current_stream().synchronize()
    return out


def sample_bias(hits, hit_offsets, lib_format, seq, seq_off, ref_len, *, read_bias=None, remaining_bias_samples=0,
                observed_gc=None, gc_prefix_table=None, gc_size_samp=1, paired_library=None, allow_orphans=False,
                max_read_occs=200, max_frag_len=1000, device="cuda"):
    """The bias / GC samples the hit loop collects (src/SailfishQuantify.cpp:270-287, 375-389, 559-581) over the reads and
    hits that survive filter_hits' cuts.  read_bias (int32 device tensor [4096]) and observed_gc ([101]) are updated in
    place when given.  Returns (remaining_bias_samples, n_bias_sampled, n_gc_sampled)."""
    dev = torch.device(device)
    fmt = LIBRARY_FORMATS[lib_format.upper()] if isinstance(lib_format, str) else tuple(lib_format)
    if paired_library is None:
        paired_library = fmt[0] == 1
    d_hits, d_off = _device_hits(hits, hit_offsets, dev)
    R = int(d_off.numel()) - 1
    o = _lib.FilterOpts(int(max_read_occs), int(max_frag_len), int(bool(paired_library)), int(not allow_orphans), 0, 0, 0,
                        _lib.LibFmt(int(fmt[0]), int(fmt[1]), int(fmt[2]), 0))
    so = seq_off.to(torch.int64).contiguous()
    rem = C.c_int64(int(remaining_bias_samples))
    sp = _lib.BiasSampler(_lib.ptr(seq).value, _lib.ptr(so).value, _lib.ptr(ref_len).value,
                          None if read_bias is None else _lib.ptr(read_bias).value, C.pointer(rem),
                          None if observed_gc is None else _lib.ptr(observed_gc).value,
                          None if gc_prefix_table is None else _lib.ptr(gc_prefix_table).value, 0, 0, int(gc_size_samp), 0)
    with torch.cuda.device(dev):
        torch.cuda.current_stream().synchronize()
        _lib.check(_lib.lib().sfgpu_sample_bias(_lib.ptr(d_hits), _lib.ptr(d_off), R, C.byref(o), C.byref(sp), _lib.current_stream_ptr()))
    return int(rem.value), int(sp.n_bias_sampled), int(sp.n_gc_sampled)
