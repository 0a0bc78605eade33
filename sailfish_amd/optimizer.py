"""CollapsedEMOptimizer -- host mirror of include/CollapsedEMOptimizer.hpp:25-35 /
src/CollapsedEMOptimizer.cpp:711-893 over the C ABI (sfgpu_em_*)."""
import ctypes as C

import torch

from . import _lib
from .experiment import ReadExperiment, SailfishOpts


class EMProblem:
    """Device-resident problem handle (sfgpu_em): CSR of classes + per-transcript lengths."""

    def __init__(self, length_f64, rowptr, ids, counts, num_mapped):
        self._L = _lib.lib()
        self.M = int(length_f64.numel())
        self.C = int(rowptr.numel()) - 1
        self.device = length_f64.device
        # keep the tensors alive: the library reads them in place (caller-owned buffers)
        self._keep = (length_f64.contiguous(), rowptr.contiguous(), ids.contiguous(), counts.contiguous())
        prob = _lib.Problem(self.M, _lib.ptr(self._keep[0]).value, self.C, _lib.ptr(self._keep[1]).value,
                            (_lib.ptr(self._keep[2]).value if self._keep[2].numel() else None),
                            (_lib.ptr(self._keep[3]).value if self._keep[3].numel() else None), int(num_mapped))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            # (no host wait: the handle orders its own stream behind everything already queued on the caller's, sfgpu_em_create)
            _lib.check(self._L.sfgpu_em_create(C.byref(h), C.byref(prob), _lib.current_stream_ptr()))
        self._h = h
        self.alpha = torch.zeros(self.M, dtype=torch.float64, device=self.device)
        self.mass = torch.zeros(self.M, dtype=torch.float64, device=self.device)

    @staticmethod
    def opts(use_vbem=False, tol=0.01, min_iter=50, max_iter=10000, check_mode=0, iters_per_launch=0):
        return _lib.EmOpts(int(use_vbem), float(tol), int(min_iter), int(max_iter), int(check_mode), int(iters_per_launch))

    def optimize(self, **kw):
        """Run the whole loop on the device; returns (rc, stats dict).  alpha/mass hold estCount/mass."""
        o = self.opts(**kw)
        st = _lib.EmStats()
        rc = self._L.sfgpu_em_optimize(self._h, C.byref(o), _lib.ptr(self.alpha), _lib.ptr(self.mass), C.byref(st))
        return rc, st.as_dict()

    def optimize_bias(self, bias, **kw):
        """optimize() with doBiasCorrect (src/CollapsedEMOptimizer.cpp:717, :814-840, :888): `bias` is a
        sailfish_amd.bias.BiasModel.  Returns (rc, stats dict, effLens tensor[M], hooks taken)."""
        o = self.opts(**kw)
        st = _lib.EmStats(); nr = C.c_uint32(0)
        eff = torch.zeros(self.M, dtype=torch.float64, device=self.device)
        rc = self._L.sfgpu_em_optimize_bias(self._h, C.byref(o), bias._h, _lib.ptr(self.alpha), _lib.ptr(self.mass),
                                            _lib.ptr(eff), C.byref(nr), C.byref(st))
        return rc, st.as_dict(), eff, nr.value

    def sharded_fused_ok(self):
        """can this rank's slice run the sharded loop with one sweep kernel per iteration? (sfgpu_em_sharded_fused_ok)"""
        return bool(self._L.sfgpu_em_sharded_fused_ok(self._h))

    def set_sharded_fused(self, on):
        """what the ranks agreed on (the minimum of sharded_fused_ok over the ranks): every rank runs the same form of the loop"""
        _lib.check(self._L.sfgpu_em_set_sharded_fused(self._h, int(bool(on))))

    def optimize_sharded(self, allreduce, poll_every=16, **kw):
        """The sharded loop as ONE call (sfgpu_em_optimize_sharded): this handle holds one rank's slice of the classes,
        `allreduce` leaves the sum of alphaOut over the ranks on every rank -- a sailfish_amd.comm.Comm (ncclAllReduce on
        the loop's stream: no host code between two iterations) or a Python callable(tensor) (dry runs over gloo).
        Returns (rc, stats dict) like optimize()."""
        o = self.opts(**kw)
        st = _lib.EmStats()
        if hasattr(allreduce, "callback"):
            fn, user = allreduce.callback()
        else:
            dev, err = self.device, []

            def _cb(p, n, _u, _s):
                try:
                    allreduce(_tensor_from_ptr(p, n, dev))
                    return 0
                except Exception as e:      # an exception must not unwind through the C frame
                    err.append(e)
                    return 1
            fn, user = _lib.ALLREDUCE_CB(_cb), None
        rc = self._L.sfgpu_em_optimize_sharded(self._h, C.byref(o), fn, user, int(poll_every), _lib.ptr(self.alpha),
                                               _lib.ptr(self.mass), C.byref(st))
        if not hasattr(allreduce, "callback") and err:
            raise err[0]
        return rc, st.as_dict()

    # piecewise API (multi-GPU driver, tests)
    def begin(self, **kw):
        self._o = self.opts(**kw)
        _lib.check(self._L.sfgpu_em_begin(self._h, C.byref(self._o)))

    def init(self):
        _lib.check(self._L.sfgpu_em_init(self._h))

    def sweep(self):
        _lib.check(self._L.sfgpu_em_sweep(self._h))

    def update(self):
        _lib.check(self._L.sfgpu_em_update(self._h))

    def poll(self):
        done = C.c_int(); st = _lib.EmStats()
        _lib.check(self._L.sfgpu_em_poll(self._h, C.byref(done), C.byref(st)))
        return bool(done.value), st.as_dict()

    def finish(self):
        st = _lib.EmStats()
        rc = self._L.sfgpu_em_finish(self._h, _lib.ptr(self.alpha), _lib.ptr(self.mass), C.byref(st))
        return rc, st.as_dict()

    def alpha_out_view(self):
        """alphaOut (M float64) as a torch tensor aliasing the library's buffer, for collectives."""
        p = self._L.sfgpu_em_alpha_out(self._h)
        return _tensor_from_ptr(p, self.M, self.device)

    def alpha_view(self):
        """the loop's current alpha (M float64), aliasing the library's buffer"""
        return _tensor_from_ptr(self._L.sfgpu_em_alpha(self._h), self.M, self.device)

    def length_view(self):
        """effLens as the loop holds them (clamped at 1), aliasing the library's buffer"""
        return _tensor_from_ptr(self._L.sfgpu_em_lengths(self._h), self.M, self.device)

    def set_bounds(self, min_iter, max_iter):
        _lib.check(self._L.sfgpu_em_set_bounds(self._h, int(min_iter), int(max_iter)))

    def rebase(self, length):
        """updateEqClassWeights (:527-555) for the piecewise loop: new lengths, x rebuilt from the current alpha"""
        length = length.to(torch.float64).contiguous()
        _lib.check(self._L.sfgpu_em_rebase(self._h, _lib.ptr(length)))
        torch.cuda.current_stream().synchronize()      # `length` may be a temporary

    def time_sweep(self, n=100, **kw):
        o = self.opts(**kw); ms = C.c_double()
        _lib.check(self._L.sfgpu_em_time_sweep(self._h, C.byref(o), int(n), C.byref(ms)))
        return ms.value

    def bootstrap(self, n, seed=1, callback=None, use_vbem=False, tol=0.01, max_iter=10000):
        """n bootstrap replicates (doBootstrap, src/CollapsedEMOptimizer.cpp:438-525).
        Returns (rc, out[n, M] float64 device tensor, iters[n]); callback(alpha: np.ndarray) -> bool
        is the writeBootstrap hook."""
        import numpy as np
        o = self.opts(use_vbem=use_vbem, tol=tol, min_iter=0, max_iter=max_iter, check_mode=1)
        out = torch.zeros((n, self.M), dtype=torch.float64, device=self.device)
        iters = np.zeros(n, np.uint32)
        if callback is None:
            cb = _lib.SAMPLE_CB(0)
        else:
            def _cb(p, m, _u):
                return 1 if callback(np.ctypeslib.as_array(p, shape=(m,)).copy()) else 0
            cb = _lib.SAMPLE_CB(_cb)
        rc = self._L.sfgpu_bootstrap(self._h, C.byref(o), int(n), int(seed), _lib.ptr(out), cb, None, _lib.ptr(iters))
        return rc, out, iters

    def bootstrap_counts(self, seed, draw):
        """one multinomial resample of the class counts (sampCounts, :468) -> int64 tensor[C]"""
        out = torch.zeros(max(self.C, 1), dtype=torch.int32, device=self.device)
        _lib.check(self._L.sfgpu_bootstrap_counts(self._h, int(seed), int(draw), _lib.ptr(out)))
        return out[: self.C].to(torch.int64) & 0xFFFFFFFF

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sfgpu_em_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _tensor_from_ptr(p, n, device):
    """Zero-copy float64 tensor over device memory owned by the library."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(p), False), "version": 2}
    return torch.as_tensor(h, device=device)


class CollapsedEMOptimizer:
    """optimize(readExp, sopt, relDiffTolerance, maxIter) exactly as the reference's driver calls
    it (src/SailfishQuantify.cpp:1341-1343: tolerance 0.01, maxIter 10000)."""

    def __init__(self):
        self.last_stats = None
        self._problem = None

    def optimize(self, readExp: ReadExperiment, sopt: SailfishOpts, relDiffTolerance: float = 0.01,
                 maxIter: int = 1000) -> bool:
        do_bias = sopt.biasCorrect or sopt.gcBiasCorrect                      # :717
        if sopt.jointLog is not None:
            _lib.set_logger(sopt.jointLog)
        txps = readExp.transcripts()
        vec = readExp.equivalenceClassBuilder().eqVec()
        # :736-737  effLens(i) = noEffectiveLengthCorrection ? RefLength : EffectiveLength
        length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
        prob = EMProblem(length, vec.rowptr, vec.ids, vec.counts, readExp.numMappedFragments())
        self._problem = prob
        eff = None
        if do_bias:
            bias = readExp.biasModel(sopt)
            rc, st, eff, self.last_recomputes = prob.optimize_bias(bias, use_vbem=sopt.useVBOpt, tol=relDiffTolerance,
                                                                   min_iter=50, max_iter=maxIter)
            if rc == _lib.OK:
                es, eg = bias.expected()
                readExp.setExpectedSeqBias(es); readExp.setExpectedGCBias(eg)
        else:
            rc, st = prob.optimize(use_vbem=sopt.useVBOpt, tol=relDiffTolerance, min_iter=50, max_iter=maxIter)
        self.last_stats = st
        if rc in (_lib.ERR_NO_ACTIVE, _lib.ERR_ALPHA_SUM):
            return False            # the reference logs and returns false (:794-798, :877-881)
        _lib.check(rc)
        if eff is not None:
            txps.EffectiveLength.copy_(eff)                                   # :888
        txps.estCount.copy_(prob.alpha)    # setEstCount / setMass (:889-890)
        txps.mass.copy_(prob.mass)
        return True

    def gatherBootstraps(self, readExp: ReadExperiment, sopt: SailfishOpts, writeBootstrap,
                         relDiffTolerance: float = 0.01, maxIter: int = 1000, seed=None) -> bool:
        """gatherBootstraps(readExp, sopt, writeBootstrap, 0.01, 10000)
        (src/SailfishQuantify.cpp:1401-1403; src/CollapsedEMOptimizer.cpp:557-709): draws
        sopt.numBootstraps replicates and hands each alpha vector (numpy float64[M]) to
        writeBootstrap.  The reference seeds from std::random_device; `seed` makes a run repeatable."""
        import os
        if sopt.jointLog is not None:
            _lib.set_logger(sopt.jointLog)
        txps = readExp.transcripts()
        vec = readExp.equivalenceClassBuilder().eqVec()
        length = txps.ref_length_f64() if sopt.noEffectiveLengthCorrection else txps.EffectiveLength
        prob = EMProblem(length, vec.rowptr, vec.ids, vec.counts, readExp.numMappedFragments())
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        rc, out, iters = prob.bootstrap(sopt.numBootstraps, seed=seed, callback=writeBootstrap,
                                        use_vbem=sopt.useVBOpt, tol=relDiffTolerance, max_iter=maxIter)
        self.last_bootstraps, self.last_bootstrap_iters = out, iters
        if rc in (_lib.ERR_NO_ACTIVE, _lib.ERR_ALPHA_SUM):
            return False
        _lib.check(rc)
        return True
