"""ctypes/numpy front end of oracle/liboracle.so (the C restatement in sf_oracle.c).

TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Each wrapper names the reference function it restates (file:line under /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sf_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    elif os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(_HERE, "_ref", "libxxhash_ref.so")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so


class EMStats(C.Structure):
    _fields_ = [("iters", C.c_uint32), ("converged", C.c_uint32), ("max_rel_diff", C.c_double),
                ("alpha_sum", C.c_double), ("n_active", C.c_uint64)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.sfo_xxh64.restype = C.c_uint64
        L.sfo_xxh64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.sfo_digamma.restype = C.c_double
        L.sfo_digamma.argtypes = [C.c_double]
        L.sfo_eq_create.restype = C.c_void_p
        L.sfo_eq_destroy.argtypes = [C.c_void_p]
        L.sfo_eq_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.sfo_eq_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sfo_eq_export.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.sfo_xxh64_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.sfo_cf_gaussian.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
        L.sfo_fld_gaussian_counts.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p]
        L.sfo_cf_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.sfo_efflen_smoothed.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        L.sfo_efflen_empirical.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.sfo_em_optimize.restype = C.c_int
        L.sfo_em_optimize.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_int, C.c_double, C.c_uint32, C.c_uint32, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.sfo_tpm.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.sfo_fld_cdf.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.sfo_index_for_kmer.restype = C.c_uint32
        L.sfo_index_for_kmer.argtypes = [C.c_char_p, C.c_uint32, C.c_int]
        L.sfo_next_kmer_index.restype = C.c_uint32
        L.sfo_next_kmer_index.argtypes = [C.c_uint32, C.c_char, C.c_uint32, C.c_int]
        L.sfo_gc_frac.restype = C.c_int32
        L.sfo_gc_frac.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
        L.sfo_update_efflens.restype = C.c_int
        L.sfo_update_efflens.argtypes = [C.c_void_p] * 7
        L.sfo_em_optimize_bias.restype = C.c_int
        L.sfo_em_optimize_bias.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_uint64, C.c_int, C.c_double, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.sfo_bootstrap.restype = C.c_int
        L.sfo_bootstrap.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_double, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.sfo_gibbs.restype = C.c_int
        L.sfo_gibbs.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def xxh64(data: bytes, seed=0) -> int:
    """XXH64 (src/xxhash.c:346-455)."""
    buf = C.create_string_buffer(data, len(data))
    return int(lib().sfo_xxh64(C.cast(buf, C.c_void_p), len(data), seed))


def xxh64_lists(ids, off):
    """TranscriptGroup hash of each packed label (src/TranscriptGroup.cpp:9-12)."""
    ids = _c(ids, np.uint32); off = _c(off, np.uint64)
    out = np.empty(len(off) - 1, np.uint64)
    lib().sfo_xxh64_lists(_p(ids), _p(off), len(off) - 1, _p(out))
    return out


def digamma(x):
    return float(lib().sfo_digamma(float(x)))


class EqBuilder:
    """EquivalenceClassBuilder restated (include/EquivalenceClassBuilder.hpp:62-112)."""

    def __init__(self):
        self._h = lib().sfo_eq_create()

    def add_batch(self, ids, off):
        ids = _c(ids, np.uint32); off = _c(off, np.uint64)
        if len(ids) == 0:
            ids = np.zeros(1, np.uint32)
        lib().sfo_eq_add(self._h, _p(ids), _p(off), len(off) - 1)

    def add_batch_mt(self, ids, off, n_threads):
        """the same batch over n_threads host threads (per-thread tables, then folded); returns the seconds taken
        (bench.py's all-cores cpu_baseline)"""
        ids = _c(ids, np.uint32); off = _c(off, np.uint64)
        if len(ids) == 0:
            ids = np.zeros(1, np.uint32)
        L = lib()
        L.sfo_eq_add_mt.restype = C.c_double
        L.sfo_eq_add_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        return float(L.sfo_eq_add_mt(self._h, _p(ids), _p(off), len(off) - 1, int(n_threads)))

    def finish(self):
        n = C.c_uint64(); nnz = C.c_uint64(); tot = C.c_uint64()
        lib().sfo_eq_finish(self._h, C.byref(n), C.byref(nnz), C.byref(tot))
        self.n_classes, self.nnz, self.total_reads = n.value, nnz.value, tot.value
        rowptr = np.zeros(n.value + 1, np.uint64); ids = np.zeros(max(nnz.value, 1), np.uint32)
        counts = np.zeros(max(n.value, 1), np.uint64); hashes = np.zeros(max(n.value, 1), np.uint64)
        lib().sfo_eq_export(self._h, _p(rowptr), _p(ids), _p(counts), _p(hashes))
        return rowptr, ids[:nnz.value], counts[:n.value], hashes[:n.value]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().sfo_eq_destroy(self._h); self._h = None
        except Exception:
            pass


def cf_gaussian(max_frag_len=1000, mean=200, sd=80):
    """getNormalFragLengthDist (src/SailfishQuantify.cpp:648-673)."""
    cf = np.zeros(max_frag_len); lib().sfo_cf_gaussian(max_frag_len, mean, sd, _p(cf)); return cf


def fld_gaussian_counts(max_frag_len=1000, mean=200, sd=80, num_samples=10000):
    """getNormalFragLengthCounts (src/SailfishQuantify.cpp:675-704)."""
    d = np.zeros(max_frag_len, np.int32)
    lib().sfo_fld_gaussian_counts(max_frag_len, mean, sd, num_samples, _p(d)); return d


def cf_counts(fl_counts):
    """correctionFactorsFromCounts (src/SailfishQuantify.cpp:769-807)."""
    fl = _c(fl_counts, np.uint32); cf = np.zeros(len(fl))
    lib().sfo_cf_counts(_p(fl), len(fl), _p(cf)); return cf


def efflen_smoothed(ref_len, cf):
    """computeSmoothedEffectiveLengths (src/SailfishQuantify.cpp:809-838)."""
    rl = _c(ref_len, np.uint32); cf = _c(cf, np.float64); eff = np.zeros(len(rl))
    lib().sfo_efflen_smoothed(_p(rl), len(rl), _p(cf), len(cf), _p(eff)); return eff


def efflen_empirical(fl_counts, ref_len):
    """--unsmoothedFLD: computeEmpiricalEffectiveLengths (src/SailfishQuantify.cpp:717-767)."""
    fl = _c(fl_counts, np.uint32); rl = _c(ref_len, np.uint32); eff = np.zeros(len(rl))
    lib().sfo_efflen_empirical(_p(fl), len(fl), _p(rl), len(rl), _p(eff)); return eff


def em_optimize(eff_len, rowptr, ids, counts, num_mapped, use_vbem=False, tol=0.01,
                min_iter=50, max_iter=10000, check_mode=0):
    """CollapsedEMOptimizer::optimize (src/CollapsedEMOptimizer.cpp:711-893).
    Returns (rc, alpha, mass, stats-dict)."""
    eff = _c(eff_len, np.float64); rp = _c(rowptr, np.uint64); ii = _c(ids, np.uint32); cc = _c(counts, np.uint64)
    if len(ii) == 0:
        ii = np.zeros(1, np.uint32)
    if len(cc) == 0:
        cc = np.zeros(1, np.uint64)
    M = len(eff); alpha = np.zeros(max(M, 1)); mass = np.zeros(max(M, 1)); st = EMStats()
    rc = lib().sfo_em_optimize(M, _p(eff), len(rp) - 1, _p(rp), _p(ii), _p(cc), int(num_mapped), int(use_vbem),
                               float(tol), int(min_iter), int(max_iter), int(check_mode),
                               _p(alpha), _p(mass), C.byref(st))
    stats = dict(iters=st.iters, converged=bool(st.converged), max_rel_diff=st.max_rel_diff,
                 alpha_sum=st.alpha_sum, n_active=st.n_active)
    return rc, alpha[:M], mass[:M], stats


class BiasModel(C.Structure):
    """sfo_bias_model: what updateEffectiveLengths reads from ReadExperiment / SailfishOpts."""
    _fields_ = [("M", C.c_uint64), ("seq", C.c_void_p), ("seq_off", C.c_void_p), ("ref_len", C.c_void_p),
                ("txp_eff_len", C.c_void_p), ("fl_counts", C.c_void_p), ("n_fl", C.c_uint32),
                ("read_bias", C.c_void_p), ("observed_gc", C.c_void_p), ("num_fwd", C.c_int64),
                ("num_rc", C.c_int64), ("seq_bias", C.c_int32), ("gc_bias", C.c_int32),
                ("gc_speed_samp", C.c_uint32), ("gc_size_samp", C.c_uint32)]


def make_bias_model(seq, seq_off, ref_len, txp_eff_len, fl_counts, read_bias=None, observed_gc=None,
                    num_fwd=1, num_rc=1, seq_bias=False, gc_bias=False, gc_speed_samp=1, gc_size_samp=1):
    """-> (BiasModel, keepalive list).  seq: bytes / uint8 array holding every transcript (RapMapSAIndex::seq)."""
    sq = np.frombuffer(seq, dtype=np.uint8).copy() if isinstance(seq, (bytes, bytearray)) else _c(seq, np.uint8)
    so = _c(seq_off, np.uint64); rl = _c(ref_len, np.uint32); te = _c(txp_eff_len, np.float64)
    fl = _c(fl_counts, np.uint32)
    rb = _c(read_bias if read_bias is not None else np.ones(4096), np.uint32)
    og = _c(observed_gc if observed_gc is not None else np.ones(101), np.uint32)
    assert len(rb) == 4096 and len(og) == 101 and len(so) == len(rl) == len(te)
    bm = BiasModel(len(rl), _p(sq).value, _p(so).value, _p(rl).value, _p(te).value, _p(fl).value, len(fl),
                   _p(rb).value, _p(og).value, int(num_fwd), int(num_rc), int(bool(seq_bias)), int(bool(gc_bias)),
                   int(gc_speed_samp), int(gc_size_samp))
    return bm, [sq, so, rl, te, fl, rb, og]


def fld_cdf(fl_counts):
    """EmpiricalDistribution::cdf(i) for i in [0, n) (src/EmpiricalDistribution.cpp:29-77, :121-124)
    -> (float32[n], table size)."""
    fl = _c(fl_counts, np.uint32); out = np.zeros(len(fl), np.float32); size = C.c_uint32(0)
    lib().sfo_fld_cdf(_p(fl), len(fl), _p(out), C.byref(size)); return out, size.value


def index_for_kmer(s: bytes, K=6, rc=False):
    """indexForKmer (include/UtilityFunctions.hpp:93-148)."""
    return int(lib().sfo_index_for_kmer(s, K, int(rc)))


def next_kmer_index(idx, ch: bytes, K=6, rc=False):
    """nextKmerIndex (include/UtilityFunctions.hpp:40-90)."""
    return int(lib().sfo_next_kmer_index(idx, ch, K, int(rc)))


def gc_frac(seq: bytes, s, e, step=1):
    """Transcript::gcFrac over the closed interval [s,e] (include/Transcript.hpp:85-95)."""
    return int(lib().sfo_gc_frac(seq, len(seq), step, s, e))


def update_efflens(bm, eff_in, alphas):
    """sailfish::utils::updateEffectiveLengths (src/SailfishUtils.cpp:611-926).
    -> (rc, eff_out, expected_seq[4096], expected_gc[101], n_corrected)"""
    model = bm[0] if isinstance(bm, tuple) else bm
    ei = _c(eff_in, np.float64); al = _c(alphas, np.float64); out = np.zeros(len(ei))
    es = np.ones(4096); eg = np.ones(101); nc = C.c_uint64(0)
    rc = lib().sfo_update_efflens(C.byref(model), _p(ei), _p(al), _p(out), _p(es), _p(eg), C.byref(nc))
    return rc, out, es, eg, nc.value


def em_optimize_bias(bm, eff_len, rowptr, ids, counts, num_mapped, use_vbem=False, tol=0.01,
                     min_iter=50, max_iter=10000):
    """optimize() with biasCorrect / gcBiasCorrect (src/CollapsedEMOptimizer.cpp:711-893, hook :814-840).
    -> (rc, alpha, mass, eff_final, expected_seq, expected_gc, n_recomputes, stats-dict)"""
    model = bm[0] if isinstance(bm, tuple) else bm
    eff = _c(eff_len, np.float64); rp = _c(rowptr, np.uint64); ii = _c(ids, np.uint32); cc = _c(counts, np.uint64)
    M = len(eff); alpha = np.zeros(max(M, 1)); mass = np.zeros(max(M, 1)); efin = np.zeros(max(M, 1))
    es = np.ones(4096); eg = np.ones(101); nr = C.c_uint32(0); st = EMStats()
    rc = lib().sfo_em_optimize_bias(M, _p(eff), len(rp) - 1, _p(rp), _p(ii), _p(cc), int(num_mapped), int(use_vbem),
                                    float(tol), int(min_iter), int(max_iter), C.byref(model), _p(alpha), _p(mass),
                                    _p(efin), _p(es), _p(eg), C.byref(nr), C.byref(st))
    stats = dict(iters=st.iters, converged=bool(st.converged), max_rel_diff=st.max_rel_diff,
                 alpha_sum=st.alpha_sum, n_active=st.n_active)
    return rc, alpha[:M], mass[:M], efin[:M], es, eg, nr.value, stats


def em_iterations_mt(eff_len, rowptr, ids, counts, num_mapped, n_iters, n_threads, use_vbem=False):
    """n_iters EM / VBEM updates over n_threads host threads (classes cut into nnz-balanced ranges, private alphaOut
    copies summed in thread order): bench.py's all-cores cpu_baseline.  -> (seconds per iteration, alpha)"""
    eff = _c(eff_len, np.float64); rp = _c(rowptr, np.uint64); ii = _c(ids, np.uint32); cc = _c(counts, np.uint64)
    out = np.zeros(max(len(eff), 1))
    L = lib()
    L.sfo_em_iterations_mt.restype = C.c_double
    L.sfo_em_iterations_mt.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    sec = L.sfo_em_iterations_mt(len(eff), _p(eff), len(rp) - 1, _p(rp), _p(ii), _p(cc), int(num_mapped), int(use_vbem),
                                 int(n_iters), int(n_threads), _p(out))
    return float(sec), out[: len(eff)]


def tpm(est_count, length, num_mapped):
    """TPM column of quant.sf (src/GZipWriter.cpp:216-245)."""
    a = _c(est_count, np.float64); l = _c(length, np.float64); out = np.zeros(len(a))
    lib().sfo_tpm(len(a), _p(a), _p(l), float(num_mapped), _p(out)); return out


def bootstrap(eff_len, rowptr, ids, counts, B, use_vbem=False, tol=0.01, max_iter=10000, seed=1):
    """gatherBootstraps/doBootstrap (src/CollapsedEMOptimizer.cpp:438-525, 557-709). -> (B, M)"""
    eff = _c(eff_len, np.float64); rp = _c(rowptr, np.uint64); ii = _c(ids, np.uint32); cc = _c(counts, np.uint64)
    out = np.zeros((B, len(eff))); iters = np.zeros(B, np.uint32)
    rc = lib().sfo_bootstrap(len(eff), _p(eff), len(rp) - 1, _p(rp), _p(ii), _p(cc), int(use_vbem), float(tol),
                             int(max_iter), int(B), int(seed), _p(out), _p(iters))
    return rc, out, iters


def gibbs(eff_len, mass, rowptr, ids, counts, num_mapped, S, seed=1):
    """CollapsedGibbsSampler::sample, one chain (src/CollapsedGibbsSampler.cpp:198-291). -> (S, M) int32"""
    eff = _c(eff_len, np.float64); ms = _c(mass, np.float64)
    rp = _c(rowptr, np.uint64); ii = _c(ids, np.uint32); cc = _c(counts, np.uint64)
    out = np.zeros((S, len(eff)), np.int32)
    rc = lib().sfo_gibbs(len(eff), _p(eff), _p(ms), len(rp) - 1, _p(rp), _p(ii), _p(cc), int(num_mapped), int(S),
                         int(seed), _p(out))
    return rc, out


# --- the compiled reference (oracle/_ref), when present -------------------------------------
HIT_DTYPE = np.dtype([("tid", "<u4"), ("pos", "<i4"), ("mate_pos", "<i4"), ("frag_len", "<u4"), ("read_len", "<u2"),
                      ("mate_len", "<u2"), ("fwd", "u1"), ("mate_fwd", "u1"), ("mate_status", "u1"), ("pad_", "u1")])


class _LibFmt(C.Structure):
    _fields_ = [("type", C.c_uint8), ("orientation", C.c_uint8), ("strandedness", C.c_uint8), ("pad_", C.c_uint8)]


class _FilterOpts(C.Structure):
    _fields_ = [("max_read_occs", C.c_uint32), ("max_frag_len", C.c_uint32), ("paired_library", C.c_int32),
                ("discard_orphans", C.c_int32), ("ignore_compat", C.c_int32), ("enforce_compat", C.c_int32),
                ("can_dovetail", C.c_int32), ("expected", _LibFmt)]


class _FilterStats(C.Structure):
    _fields_ = [("n_observed", C.c_uint64), ("n_mapped", C.c_uint64), ("total_hits", C.c_uint64),
                ("upper_bound_hits", C.c_uint64), ("n_fwd", C.c_uint64), ("n_rc", C.c_uint64), ("fl_sampled", C.c_uint64)]


def compatible_single(fmt, is_forward, mate_status):
    """compatibleHit(expected, start, isForward, ms) (src/SailfishUtils.cpp:157-207); fmt = (type, orientation, strandedness)"""
    L = lib(); L.sfo_compatible_single.argtypes = [_LibFmt, C.c_int, C.c_int]; L.sfo_compatible_single.restype = C.c_int
    return bool(L.sfo_compatible_single(_LibFmt(*fmt, 0), int(is_forward), int(mate_status)))


def hit_type(end1_start, end1_fwd, len1, end2_start, end2_fwd, len2, can_dovetail):
    """hitType (src/SailfishUtils.cpp:232-281) -> (type, orientation, strandedness)"""
    L = lib(); L.sfo_hit_type.argtypes = [C.c_int32, C.c_int, C.c_uint32, C.c_int32, C.c_int, C.c_uint32, C.c_int]
    L.sfo_hit_type.restype = _LibFmt
    f = L.sfo_hit_type(int(end1_start), int(end1_fwd), int(len1), int(end2_start), int(end2_fwd), int(len2), int(can_dovetail))
    return (f.type, f.orientation, f.strandedness)


def compatible_pair(expected, observed):
    """compatibleHit(expected, observed) (src/SailfishUtils.cpp:210-229)"""
    L = lib(); L.sfo_compatible_pair.argtypes = [_LibFmt, _LibFmt]; L.sfo_compatible_pair.restype = C.c_int
    return bool(L.sfo_compatible_pair(_LibFmt(*expected, 0), _LibFmt(*observed, 0)))


def filter_hits(hits, hit_off, fmt, paired_library, discard_orphans=True, ignore_compat=False, enforce_compat=False,
                can_dovetail=False, max_read_occs=200, max_frag_len=1000, fl_counts=None, remaining_fl_ops=0):
    """the per-read loop bodies of processReadsQuasi (src/SailfishQuantify.cpp:215-417, 530-626), serial.
    Returns (ids, offsets, fl_counts, remaining_fl_ops, stats dict)."""
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE); off = _c(hit_off, np.uint32)
    R = len(off) - 1
    ids = np.zeros(max(len(h), 1), np.uint32); out_off = np.zeros(R + 1, np.uint32)
    fl = np.zeros(max_frag_len, np.uint32) if fl_counts is None else _c(fl_counts, np.uint32).copy()
    o = _FilterOpts(max_read_occs, max_frag_len, int(paired_library), int(discard_orphans), int(ignore_compat),
                    int(enforce_compat), int(can_dovetail), _LibFmt(*fmt, 0))
    st = _FilterStats(); rem = C.c_int64(int(remaining_fl_ops))
    L = lib()
    L.sfo_filter_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(_FilterOpts), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_int64), C.POINTER(_FilterStats)]
    L.sfo_filter_hits.restype = None
    L.sfo_filter_hits(h.ctypes.data, _p(off), R, C.byref(o), _p(ids), _p(out_off), _p(fl), C.byref(rem), C.byref(st))
    return ids[: out_off[-1]], out_off, fl, int(rem.value), {k: int(getattr(st, k)) for k, _ in _FilterStats._fields_}


class _BiasSampler(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("seq_off", C.c_void_p), ("ref_len", C.c_void_p), ("M", C.c_uint64),
                ("read_bias", C.c_void_p), ("remaining_bias_samples", C.POINTER(C.c_int64)), ("observed_gc", C.c_void_p),
                ("n_bias_sampled", C.c_uint64), ("n_gc_sampled", C.c_uint64), ("gc_size_samp", C.c_uint32), ("pad_", C.c_uint32)]


def filter_hits_bias(hits, hit_off, fmt, paired_library, seq, seq_off, ref_len, read_bias=None, remaining_bias_samples=0,
                     observed_gc=None, **kw):
    """filter_hits plus the samples the loop collects for bias correction: the read-start 6-mer of the first hit that
    yields one (src/SailfishQuantify.cpp:270-287 / :559-581, ReadKmerDist.hpp:35-73; budget numBiasSamples) and the
    fragment GC of every proper pair (:375-389).  read_bias (4096) / observed_gc (101) are accumulated when given.
    Returns (filter_hits' tuple, read_bias, remaining_bias_samples, observed_gc, n_bias_sampled, n_gc_sampled)."""
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE); off = _c(hit_off, np.uint32)
    R = len(off) - 1
    max_frag_len = kw.get("max_frag_len", 1000)
    ids = np.zeros(max(len(h), 1), np.uint32); out_off = np.zeros(R + 1, np.uint32)
    fl = np.zeros(max_frag_len, np.uint32) if kw.get("fl_counts") is None else _c(kw["fl_counts"], np.uint32).copy()
    o = _FilterOpts(kw.get("max_read_occs", 200), max_frag_len, int(paired_library), int(kw.get("discard_orphans", True)),
                    int(kw.get("ignore_compat", False)), int(kw.get("enforce_compat", False)), int(kw.get("can_dovetail", False)),
                    _LibFmt(*fmt, 0))
    st = _FilterStats(); rem = C.c_int64(int(kw.get("remaining_fl_ops", 0)))
    sq = np.frombuffer(seq, dtype=np.uint8).copy() if isinstance(seq, (bytes, bytearray)) else _c(seq, np.uint8)
    so = _c(seq_off, np.uint64); rl = _c(ref_len, np.uint32)
    rb = None if read_bias is None else _c(read_bias, np.uint32).copy()
    og = None if observed_gc is None else _c(observed_gc, np.uint32).copy()
    remb = C.c_int64(int(remaining_bias_samples))
    bs = _BiasSampler(_p(sq).value, _p(so).value, _p(rl).value, len(rl), None if rb is None else _p(rb).value,
                      C.pointer(remb), None if og is None else _p(og).value, 0, 0, int(kw.get("gc_size_samp", 1)), 0)
    L = lib()
    L.sfo_filter_hits_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(_FilterOpts), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_int64), C.POINTER(_FilterStats), C.POINTER(_BiasSampler)]
    L.sfo_filter_hits_bias.restype = None
    L.sfo_filter_hits_bias(h.ctypes.data, _p(off), R, C.byref(o), _p(ids), _p(out_off), _p(fl), C.byref(rem), C.byref(st), C.byref(bs))
    base = (ids[: out_off[-1]], out_off, fl, int(rem.value), {k: int(getattr(st, k)) for k, _ in _FilterStats._fields_})
    return base, rb, int(remb.value), og, int(bs.n_bias_sampled), int(bs.n_gc_sampled)


def ref_xxhash():
    """ctypes handle on the reference's own xxhash.c (oracle/_ref/libxxhash_ref.so) or None."""
    so = os.path.join(_HERE, "_ref", "libxxhash_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.XXH64.restype = C.c_uint64
    L.XXH64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    return L


def multinomial(n, probs, seed=1):
    """the restated MultinomialSampler::operator() (sf_oracle.c `multinomial`) -> uint64[k]"""
    p = _c(probs, np.float64); out = np.zeros(len(p), np.uint64)
    L = lib(); L.sfo_multinomial.restype = None
    L.sfo_multinomial.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.sfo_multinomial(int(seed), int(n), len(p), _p(p), _p(out))
    return out


def ref_sailfish():
    """ctypes handle on oracle/_ref/libsailfish_ref.so -- ref_glue.cpp around the reference units that build unmodified
    with the standard library alone (LibraryFormat.cpp, MultinomialSampler.hpp, cuckoohash_map.hh, xxhash.c) -- or None."""
    so = os.path.join(_HERE, "_ref", "libsailfish_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_format_id.restype = C.c_uint8; L.ref_format_id.argtypes = [C.c_int] * 3
    L.ref_format_from_id.restype = None; L.ref_format_from_id.argtypes = [C.c_uint8, C.c_void_p]
    L.ref_format_check.restype = C.c_int; L.ref_format_check.argtypes = [C.c_int] * 3
    L.ref_format_max_id.restype = C.c_int
    L.ref_format_str.restype = C.c_int; L.ref_format_str.argtypes = [C.c_int] * 3 + [C.c_char_p, C.c_int]
    L.ref_multinomial.restype = None; L.ref_multinomial.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.ref_eq_build.restype = C.c_long
    L.ref_eq_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    return L


def ref_multinomial(n, probs):
    """one call of the REFERENCE's MultinomialSampler (random_device-seeded) -> uint64[k]"""
    p = _c(probs, np.float64); out = np.zeros(len(p), np.uint64)
    ref_sailfish().ref_multinomial(int(n), len(p), _p(p), _p(out))
    return out


def ref_eq_build(ids, off, n_threads=1):
    """label -> count through the REFERENCE's cuckoohash_map::upsert + XXH64, as addGroup uses them
    (ref_glue.cpp).  -> dict {label tuple: (count, hash)} (table order is history dependent)"""
    ids = _c(ids, np.uint32); off = _c(off, np.uint64)
    n = len(off) - 1
    if len(ids) == 0:
        ids = np.zeros(1, np.uint32)
    out_ids = np.zeros(max(len(ids), 1), np.uint32); out_len = np.zeros(max(n, 1), np.uint32)
    out_cnt = np.zeros(max(n, 1), np.uint64); out_hash = np.zeros(max(n, 1), np.uint64)
    nc = ref_sailfish().ref_eq_build(_p(ids), _p(off), n, int(n_threads), _p(out_ids), len(out_ids), _p(out_len), _p(out_cnt),
                                     _p(out_hash), len(out_len))
    assert nc >= 0
    res, pos = {}, 0
    for c in range(nc):
        lab = tuple(out_ids[pos:pos + out_len[c]].tolist()); pos += int(out_len[c])
        assert lab not in res
        res[lab] = (int(out_cnt[c]), int(out_hash[c]))
    return res
