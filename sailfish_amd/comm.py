"""RCCL communicator behind the C ABI (sfgpu_comm_*, csrc/comm.hip): the transport of the sharded EM loop.

torch.distributed keeps its ncclComm_t to itself, so the sharded loop gets a communicator of its own: rank 0 draws the
unique id, torch.distributed broadcasts the 128 bytes (any channel would do), every rank calls ncclCommInitRank through
libsfgpu.  The all-reduce of sfgpu_em_optimize_sharded is then ncclAllReduce enqueued on the loop's stream -- no Python
between two EM iterations."""
import ctypes as C

import torch

from . import _lib

ID_BYTES = 128


def available():
    return bool(_lib.lib().sfgpu_comm_available())


class Comm:
    def __init__(self, world, rank, uid_bytes, device):
        self._L = _lib.lib()
        self.world, self.rank, self.device = int(world), int(rank), torch.device(device)
        buf = (C.c_ubyte * ID_BYTES).from_buffer_copy(bytes(uid_bytes))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.sfgpu_comm_create(C.byref(h), buf, self.world, self.rank))
        self._h = h

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * ID_BYTES)()
        _lib.check(_lib.lib().sfgpu_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_group(cls, group, device):
        """one communicator over the ranks of a torch.distributed group (collective: every rank of the group calls it)"""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        on_dev = dist.get_backend(group) == "nccl"
        t = torch.zeros(ID_BYTES, dtype=torch.uint8, device=(device if on_dev else "cpu"))
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, src=dist.get_global_rank(group, 0), group=group)
        return cls(world, rank, bytes(t.cpu().numpy().tobytes()), device)

    def all_reduce(self, t):
        """in-place SUM of a float64 device tensor, on the current stream"""
        assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous()
        _lib.check(self._L.sfgpu_comm_allreduce_sum_f64(self._h, _lib.ptr(t), t.numel(), _lib.current_stream_ptr()))

    def time_all_reduce(self, n, reps=50):
        """average microseconds of one all-reduce of n doubles"""
        buf = torch.zeros(int(n), dtype=torch.float64, device=self.device)
        us = C.c_double()
        with torch.cuda.device(self.device):
            _lib.check(self._L.sfgpu_comm_time_allreduce(self._h, _lib.ptr(buf), int(n), int(reps), _lib.current_stream_ptr(), C.byref(us)))
        return us.value

    def count(self):
        """the ranks the communicator really spans (ncclCommCount)"""
        n = C.c_int(0)
        _lib.check(self._L.sfgpu_comm_count(self._h, C.byref(n)))
        return n.value

    def callback(self):
        """(function pointer, user pointer) for sfgpu_em_optimize_sharded"""
        fn = C.cast(self._L.sfgpu_comm_allreduce_fn(), _lib.ALLREDUCE_CB)
        return fn, self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sfgpu_comm_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
