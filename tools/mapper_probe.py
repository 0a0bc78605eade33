"""dev probe: index build and mapping rate of the quasi-mapping front end on a GENCODE-scale synthetic transcriptome"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth, _lib
import ctypes as C
dev = torch.device("cuda:0")
M = int(os.environ.get("MAP_M", 80000)); R = int(os.environ.get("MAP_R", 10_000_000)); L = 100
ref_len = synth.transcript_lengths(M, device=dev).long()
off = torch.zeros(M + 1, dtype=torch.int64, device=dev); torch.cumsum(ref_len, 0, out=off[1:])
N = int(off[-1])
g = torch.Generator(device=dev); g.manual_seed(1)
seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (N,), generator=g, device=dev)]
# reads: fragments of 250, both mates 100 bp
t = torch.randint(0, M, (R,), generator=g, device=dev)
ok = ref_len[t] >= 250
p = (torch.rand(R, generator=g, device=dev, dtype=torch.float64) * (ref_len[t] - 250).clamp_min(0).double()).long()
start = off[t] + p
idxs = torch.arange(L, device=dev)
m1 = seq[(start[:, None] + idxs[None, :]).clamp_max(N - 1)]
comp = torch.zeros(256, dtype=torch.uint8, device=dev); comp[list(b"ACGT")] = torch.tensor(list(b"TGCA"), dtype=torch.uint8, device=dev)
m2 = comp[seq[(start[:, None] + 249 - idxs[None, :]).clamp_max(N - 1)].long()]
roff = torch.arange(R + 1, device=dev, dtype=torch.int64) * L
Lb = _lib.lib()
torch.cuda.synchronize(); t0 = time.perf_counter()
h = C.c_void_p()
rl32 = ref_len.to(torch.int32).contiguous(); o0 = off[:-1].contiguous()
_lib.check(Lb.sfgpu_index_build(C.byref(h), _lib.ptr(seq), _lib.ptr(o0), _lib.ptr(rl32), M, 31, 1000, None))
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"index: {M} transcripts, {N/1e6:.1f} M bases: {1e3*(t1-t0):.1f} ms")
if os.environ.get("MAP_SEED_LEN") is not None:
    _lib.check(Lb.sfgpu_index_set_scan(h, int(os.environ["MAP_SEED_LEN"])))       # 0 = end seeds; default: scan mode, 19-base seeds
if float(os.environ.get("MAP_ERR", 0)) > 0:                                        # substitutions at this rate per base
    er = float(os.environ["MAP_ERR"])
    for m in (m1, m2):
        hit = torch.rand(m.shape, generator=g, device=dev) < er
        m[hit] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (int(hit.sum()),), generator=g, device=dev)]
hoff = torch.empty(R + 1, dtype=torch.int32, device=dev); nh = C.c_uint64(0)
hits = torch.empty(2 * R * 24, dtype=torch.uint8, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(Lb.sfgpu_map_reads(h, _lib.ptr(m1.reshape(-1)), _lib.ptr(roff), _lib.ptr(m2.reshape(-1)), _lib.ptr(roff), R, _lib.ptr(hits), 2 * R, _lib.ptr(hoff), C.byref(nh), None))
    torch.cuda.synchronize(); t1 = time.perf_counter()
hv = hits[: nh.value * 24].view(-1, 24)
tid = hv[:, :4].contiguous().view(torch.int32).reshape(-1)
first = hoff[:-1].long(); has = (hoff[1:] > hoff[:-1]) & ok
right = (tid[first.clamp_max(max(nh.value - 1, 0))].long() == t) & has
print(f"map: {R} pairs of 2 x {L} bp in {1e3*(t1-t0):.1f} ms = {R/(t1-t0)/1e6:.1f} M pairs/s; {nh.value} records; "
      f"{int(has.sum())} mapped, first hit = source transcript for {int(right.sum())}")
