"""dev probe: the local (non-communication) cost of the owner-partitioned merge on a cfg3-sized class table: bucketing
the table by owner (what every rank does before the all-to-all) and folding one rank's partition + the gathered
partitions (what it does after), for world = 8, all on one GPU"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
M, P, R, w = 200_000, 4_000_000, 100_000_000, 8
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
eq = sf.EquivalenceClassBuilder(device=dev); eq.start(); eq.add_batch(ids, off); eq.finish(); vec = eq.eqVec()
print("classes", eq.n_classes, "nnz", eq.nnz)
def bucket():
    rp = vec.rowptr.to(torch.int64) & 0xFFFFFFFF
    lens = rp[1:] - rp[:-1]
    C = int(lens.numel())
    owner = ((vec.hashes.to(torch.int64) >> 33) & 0x3FFFFFFF) % w
    order = torch.argsort(owner, stable=True)
    per = torch.bincount(owner, minlength=w)
    lens_s = lens[order]
    off_s = torch.zeros(C + 1, dtype=torch.int64, device=dev); torch.cumsum(lens_s, 0, out=off_s[1:])
    L = int(off_s[-1].item())
    shift = torch.repeat_interleave(rp[:-1][order] - off_s[:-1], lens_s)
    ids_s = vec.ids[(torch.arange(L, device=dev) + shift)]
    cnt_s = vec.counts.to(torch.int64)[order]
    return per, lens_s, off_s, ids_s, cnt_s
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); per, lens_s, off_s, ids_s, cnt_s = bucket(); torch.cuda.synchronize()
    print(f"bucket by owner: {(time.perf_counter()-t)*1e3:.2f} ms")
# one partition = 1/8 of the classes, received from 8 ranks: fold 8 copies of the first partition
c1 = int(per[0].item()); l1 = int(off_s[c1].item())
b = sf.EquivalenceClassBuilder(device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    b.start()
    ln = lens_s[:c1].repeat(w); o = torch.zeros(ln.numel() + 1, dtype=torch.int64, device=dev); torch.cumsum(ln, 0, out=o[1:])
    b.insertGroups(ids_s[:l1].repeat(w), o.to(torch.int32), cnt_s[:c1].repeat(w)); b.finish(); pv = b.eqVec()
    torch.cuda.synchronize(); print(f"fold 8 copies of one partition ({c1} classes each): {(time.perf_counter()-t)*1e3:.2f} ms")
m = sf.EquivalenceClassBuilder(device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    m.start(); m.insertGroups(ids_s, off_s.to(torch.int32), cnt_s); m.finish(); mv = m.eqVec()
    torch.cuda.synchronize(); print(f"fold the 8 gathered partitions ({eq.n_classes} classes): {(time.perf_counter()-t)*1e3:.2f} ms")
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    m.start()
    ln = lens_s.repeat(w); o = torch.zeros(ln.numel() + 1, dtype=torch.int64, device=dev); torch.cumsum(ln, 0, out=o[1:])
    m.insertGroups(ids_s.repeat(w), o.to(torch.int32), cnt_s.repeat(w)); m.finish(); mv = m.eqVec()
    torch.cuda.synchronize(); print(f"(all-gather merge) fold 8 whole tables: {(time.perf_counter()-t)*1e3:.2f} ms")

# assembling the merged table from the gathered disjoint partitions by a key sort (what _concat_disjoint does after
# its all-gather), on the table bucketed above
def concat():
    cnt, hsh_ = cnt_s, vec.hashes.to(torch.int64)[torch.argsort(((vec.hashes.to(torch.int64) >> 33) & 0x3FFFFFFF) % w, stable=True)]
    ln = lens_s
    n = int(cnt.numel())
    src = torch.zeros(n + 1, dtype=torch.int64, device=dev); torch.cumsum(ln, 0, out=src[1:])
    first = ids_s[src[:-1]].to(torch.int64) & 0xFFFFFFFF
    o1 = torch.argsort(hsh_ ^ (-0x8000000000000000), stable=True)
    order = o1[torch.argsort(first[o1], stable=True)]
    f_o, h_o = first[order], hsh_[order]
    dup = bool(((f_o[1:] == f_o[:-1]) & (h_o[1:] == h_o[:-1])).any())
    ln_o = ln[order]
    dst = torch.zeros(n + 1, dtype=torch.int64, device=dev); torch.cumsum(ln_o, 0, out=dst[1:])
    shift = torch.repeat_interleave(src[:-1][order] - dst[:-1], ln_o)
    ids_o = ids_s[torch.arange(int(dst[-1].item()), device=dev) + shift]
    return dup, dst, ids_o, cnt[order]
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); dup, dst, ids_o, cnt_o = concat(); torch.cuda.synchronize()
    print(f"assemble the merged table by key sort: {(time.perf_counter()-t)*1e3:.2f} ms (duplicate keys: {dup})")
assert torch.equal(ids_o, vec.ids) and torch.equal(cnt_o, vec.counts.to(torch.int64))
