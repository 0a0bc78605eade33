"""ctypes binding of libsfgpu.so (the C ABI declared in include/sfgpu.h).

The HIP library is the product path: there is no CPU fallback.  If the shared object is
missing, or a compute entry point fails, this module raises -- loudly.
"""
import ctypes as C
import os

import torch  # noqa: F401  (imported first so libamdhip64.so.7 resolves to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFGPU_LIB_PATH", os.path.join(_HERE, "csrc", "libsfgpu.so"))   # override: kernel-tuning builds

OK, ERR_INVALID, ERR_HIP, ERR_NO_ACTIVE, ERR_ALPHA_SUM, ERR_RANGE, ERR_STATE, ERR_UNSUPPORTED = range(8)


class SfgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsfgpu error {code}: {msg}")
        self.code = code


class Problem(C.Structure):
    _fields_ = [("M", C.c_uint64), ("d_len", C.c_void_p), ("C", C.c_uint64), ("d_rowptr", C.c_void_p),
                ("d_ids", C.c_void_p), ("d_counts", C.c_void_p), ("num_mapped", C.c_uint64)]


class EmOpts(C.Structure):
    _fields_ = [("use_vbem", C.c_int), ("tol", C.c_double), ("min_iter", C.c_uint32), ("max_iter", C.c_uint32),
                ("check_mode", C.c_int), ("iters_per_launch", C.c_uint32)]


class EmStats(C.Structure):
    _fields_ = [("iters", C.c_uint32), ("converged", C.c_uint32), ("max_rel_diff", C.c_double),
                ("alpha_sum", C.c_double), ("n_active", C.c_uint64), ("loop_ms", C.c_double),
                ("fused", C.c_uint32), ("persistent", C.c_uint32)]

    def as_dict(self):
        return dict(iters=self.iters, converged=bool(self.converged), max_rel_diff=self.max_rel_diff,
                    alpha_sum=self.alpha_sum, n_active=self.n_active, loop_ms=self.loop_ms, fused=bool(self.fused), persistent=bool(self.persistent))


class EqStats(C.Structure):
    _fields_ = [("insert_ms", C.c_double), ("insert_launches", C.c_uint64), ("table_grows", C.c_uint64),
                ("deferred_reads", C.c_uint64), ("table_slots", C.c_uint64),
                ("hot_reads", C.c_uint64), ("spilled_reads", C.c_uint64), ("pipeline_drains", C.c_uint64)]


_LOG_CB = C.CFUNCTYPE(None, C.c_int, C.c_char_p)
ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p)      # sfgpu_allreduce_fn
SAMPLE_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_uint64, C.c_void_p)
GIBBS_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int32), C.c_uint64, C.c_void_p)
_lib = None
_log_keepalive = None

_P = C.c_void_p
class LibFmt(C.Structure):
    _fields_ = [("type", C.c_uint8), ("orientation", C.c_uint8), ("strandedness", C.c_uint8), ("pad_", C.c_uint8)]


class FilterOpts(C.Structure):
    _fields_ = [("max_read_occs", C.c_uint32), ("max_frag_len", C.c_uint32), ("paired_library", C.c_int32),
                ("discard_orphans", C.c_int32), ("ignore_compat", C.c_int32), ("enforce_compat", C.c_int32),
                ("can_dovetail", C.c_int32), ("expected", LibFmt)]


class FilterStats(C.Structure):
    _fields_ = [("n_observed", C.c_uint64), ("n_mapped", C.c_uint64), ("total_hits", C.c_uint64),
                ("upper_bound_hits", C.c_uint64), ("n_fwd", C.c_uint64), ("n_rc", C.c_uint64), ("fl_sampled", C.c_uint64)]


class BiasSampler(C.Structure):
    _fields_ = [("d_seq", C.c_void_p), ("d_seq_off", C.c_void_p), ("d_ref_len", C.c_void_p), ("d_read_bias", C.c_void_p),
                ("remaining_bias_samples", C.POINTER(C.c_int64)), ("d_observed_gc", C.c_void_p), ("d_gc_prefix", C.c_void_p),
                ("n_bias_sampled", C.c_uint64), ("n_gc_sampled", C.c_uint64), ("gc_size_samp", C.c_uint32), ("pad_", C.c_uint32)]


class BiasInputs(C.Structure):
    _fields_ = [("M", C.c_uint64), ("d_seq", C.c_void_p), ("d_seq_off", C.c_void_p), ("d_ref_len", C.c_void_p),
                ("d_txp_eff_len", C.c_void_p), ("h_fl_counts", C.c_void_p), ("max_frag_len", C.c_uint32),
                ("gc_speed_samp", C.c_uint32), ("h_read_bias", C.c_void_p), ("h_observed_gc", C.c_void_p),
                ("num_fwd", C.c_int64), ("num_rc", C.c_int64), ("seq_bias", C.c_int32), ("gc_bias", C.c_int32),
                ("gc_size_samp", C.c_uint32), ("pad_", C.c_uint32)]


class BiasStats(C.Structure):
    _fields_ = [("status", C.c_int32), ("fld_low", C.c_int32), ("fld_high", C.c_int32), ("pad_", C.c_int32),
                ("n_corrected", C.c_uint64), ("n_uncorrected", C.c_uint64)]

    def as_dict(self):
        return dict(status=self.status, fld_low=self.fld_low, fld_high=self.fld_high,
                    n_corrected=self.n_corrected, n_uncorrected=self.n_uncorrected)


_SIGS = {
    "sfgpu_version": (C.c_int, []),
    "sfgpu_has_variants": (C.c_int, []),
    "sfgpu_last_error": (C.c_char_p, []),
    "sfgpu_set_logger": (None, [_LOG_CB]),
    "sfgpu_pool_trim": (C.c_int, []),
    "sfgpu_pool_set_large_limit": (C.c_int, [C.c_longlong]),
    "sfgpu_index_set_seeds": (C.c_int, [_P, C.c_uint32]),
    "sfgpu_index_set_scan": (C.c_int, [_P, C.c_uint32]),
    "sfgpu_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "sfgpu_xxh64_labels": (C.c_int, [_P, _P, C.c_uint32, _P, _P]),
    "sfgpu_eq_create": (C.c_int, [C.POINTER(_P), C.c_uint64, _P]),
    "sfgpu_eq_destroy": (C.c_int, [_P]),
    "sfgpu_eq_start": (C.c_int, [_P]),
    "sfgpu_eq_add_batch_host": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "sfgpu_eq_add_batch_device": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "sfgpu_eq_add_weighted_device": (C.c_int, [_P, _P, _P, _P, C.c_uint32]),
    "sfgpu_eq_get_stats": (C.c_int, [_P, C.POINTER(EqStats)]),
    "sfgpu_eq_finish": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sfgpu_eq_export_device": (C.c_int, [_P, _P, _P, _P, _P]),
    "sfgpu_eq_export_host": (C.c_int, [_P, _P, _P, _P, _P]),
    "sfgpu_cf_gaussian": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, _P]),
    "sfgpu_cf_counts": (C.c_int, [_P, C.c_uint32, _P]),
    "sfgpu_efflen_smoothed": (C.c_int, [_P, C.c_uint64, _P, C.c_uint32, _P, _P]),
    "sfgpu_efflen_empirical": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "sfgpu_index_build": (C.c_int, [C.POINTER(_P), _P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P]),
    "sfgpu_index_destroy": (C.c_int, [_P]),
    "sfgpu_index_info": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sfgpu_map_reads": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64), _P]),
    "sfgpu_filter_hits": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(FilterOpts), _P, _P, _P, C.POINTER(C.c_int64),
                                    C.POINTER(FilterStats), _P]),
    "sfgpu_gc_prefix": (C.c_int, [_P, _P, _P, C.c_uint64, _P, _P]),
    "sfgpu_sample_bias": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(FilterOpts), C.POINTER(BiasSampler), _P]),
    "sfgpu_bias_create": (C.c_int, [C.POINTER(_P), C.POINTER(BiasInputs), _P]),
    "sfgpu_bias_destroy": (C.c_int, [_P]),
    "sfgpu_bias_update": (C.c_int, [_P, _P, _P, _P, C.POINTER(BiasStats), _P]),
    "sfgpu_bias_expected": (C.c_int, [_P, _P, _P]),
    "sfgpu_em_optimize_bias": (C.c_int, [_P, C.POINTER(EmOpts), _P, _P, _P, _P, C.POINTER(C.c_uint32), C.POINTER(EmStats)]),
    "sfgpu_em_create": (C.c_int, [C.POINTER(_P), C.POINTER(Problem), _P]),
    "sfgpu_em_destroy": (C.c_int, [_P]),
    "sfgpu_em_optimize": (C.c_int, [_P, C.POINTER(EmOpts), _P, _P, C.POINTER(EmStats)]),
    "sfgpu_em_begin": (C.c_int, [_P, C.POINTER(EmOpts)]),
    "sfgpu_em_init": (C.c_int, [_P]),
    "sfgpu_em_sweep": (C.c_int, [_P]),
    "sfgpu_em_update": (C.c_int, [_P]),
    "sfgpu_em_poll": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(EmStats)]),
    "sfgpu_em_finish": (C.c_int, [_P, _P, _P, C.POINTER(EmStats)]),
    "sfgpu_em_optimize_sharded": (C.c_int, [_P, C.POINTER(EmOpts), ALLREDUCE_CB, _P, C.c_uint32, _P, _P, C.POINTER(EmStats)]),
    "sfgpu_em_sharded_fused_ok": (C.c_int, [_P]),
    "sfgpu_em_set_sharded_fused": (C.c_int, [_P, C.c_int]),
    "sfgpu_em_stream": (_P, [_P]),
    "sfgpu_comm_available": (C.c_int, []),
    "sfgpu_comm_unique_id": (C.c_int, [_P]),
    "sfgpu_comm_create": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int]),
    "sfgpu_comm_count": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "sfgpu_comm_destroy": (C.c_int, [_P]),
    "sfgpu_comm_allreduce_sum_f64": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "sfgpu_comm_allreduce_fn": (_P, []),
    "sfgpu_comm_time_allreduce": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, _P, C.POINTER(C.c_double)]),
    "sfgpu_eqvec_owner_sizes": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, _P, _P, _P]),
    "sfgpu_eqvec_pack_by_owner": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P, _P]),
    "sfgpu_eq_add_block_device": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P]),
    "sfgpu_eqvec_export_block": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "sfgpu_eqvec_merge_disjoint": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P, _P, _P, C.POINTER(C.c_int), _P]),
    "sfgpu_em_alpha_out": (_P, [_P]),
    "sfgpu_em_alpha": (_P, [_P]),
    "sfgpu_em_lengths": (_P, [_P]),
    "sfgpu_em_set_bounds": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "sfgpu_em_allow_persistent": (C.c_int, [C.c_int]),
    "sfgpu_em_rebase": (C.c_int, [_P, _P]),
    "sfgpu_em_time_sweep": (C.c_int, [_P, C.POINTER(EmOpts), C.c_uint32, C.POINTER(C.c_double)]),
    "sfgpu_bootstrap": (C.c_int, [_P, C.POINTER(EmOpts), C.c_uint32, C.c_uint64, _P, SAMPLE_CB, _P, _P]),
    "sfgpu_bootstrap_counts": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    "sfgpu_gibbs_sample": (C.c_int, [C.POINTER(Problem), _P, C.c_uint32, C.c_uint32, C.c_uint64, _P, GIBBS_CB, _P, _P]),
    "sfgpu_tpm": (C.c_int, [_P, _P, C.c_uint64, C.c_double, _P, _P]),
}


def exported_symbols():
    """Names include/sfgpu.h declares (kept in lock-step with _SIGS by tests/test_abi.py)."""
    return sorted(_SIGS)


def lib():
    """Load libsfgpu.so (built in-tree by __graft_entry__.build()). Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise SfgpuError(rc, lib().sfgpu_last_error().decode("utf-8", "replace"))


def set_logger(fn):
    """fn(level:int, msg:str) or None -- forwarded from the library (the jointLog hook)."""
    global _log_keepalive
    if fn is None:
        _log_keepalive = _LOG_CB(0)
    else:
        _log_keepalive = _LOG_CB(lambda lvl, msg: fn(lvl, msg.decode("utf-8", "replace")))
    lib().sfgpu_set_logger(_log_keepalive)


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device (0 = null stream)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host) address of a contiguous tensor / numpy array."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)
