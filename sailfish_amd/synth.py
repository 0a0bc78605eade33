"""Deterministic synthetic transcriptomes and packed quasi-mapping hit lists.

This is the input generator for tests and bench.py -- the mapper (RapMap) is outside the hot
path, so its output is synthesised:
  transcripts : len_i = clamp(lognormal(mu=7.3, sigma=0.8), 200, 100000)
  label pool  : P labels; size k = min(200, 1 + Geom(0.25)) (mean 4); members
                sorted{(base + 7 j) mod M}, base ~ U[0, M)
  reads       : read -> label index min(U[0,P), U[0,P))  (skewed towards low indices)
All draws come from a seeded torch.Generator on the requested device."""
import math

import torch


def transcript_lengths(M, seed=42, device="cpu"):
    g = torch.Generator(device=device); g.manual_seed(seed)
    z = torch.randn(M, generator=g, device=device, dtype=torch.float64)
    ln = torch.exp(7.3 + 0.8 * z).clamp(200, 100000)
    return ln.to(torch.int64).to(torch.int32)


def label_pool(M, P, seed=42, device="cpu", max_k=200):
    """-> (pool_off int64[P+1], pool_ids int32[L])"""
    g = torch.Generator(device=device); g.manual_seed(seed + 1)
    u = torch.rand(P, generator=g, device=device, dtype=torch.float64).clamp_min(1e-300)
    k = (1 + torch.floor(torch.log(u) / math.log(0.75))).to(torch.int64).clamp(1, min(max_k, M))
    base = torch.randint(0, M, (P,), generator=g, device=device, dtype=torch.int64)
    off = torch.zeros(P + 1, dtype=torch.int64, device=device)
    torch.cumsum(k, 0, out=off[1:])
    L = int(off[-1])
    cls = torch.repeat_interleave(torch.arange(P, device=device), k)
    j = torch.arange(L, device=device) - off[:-1][cls]
    val = (base[cls] + 7 * j) % M
    # sort members inside each label (only wrapped labels are out of order)
    key = cls * M + val
    key, _ = torch.sort(key)
    ids = (key % M).to(torch.int32)
    return off, ids


def reads_from_pool(pool_off, pool_ids, R, seed=7, device=None, chunk=1 << 23):
    """-> (ids int32[H], offsets int32[R+1]) packed hit lists of R reads (uint32 bit patterns)."""
    device = device or pool_ids.device
    P = pool_off.numel() - 1
    g = torch.Generator(device=device); g.manual_seed(seed)
    k = (pool_off[1:] - pool_off[:-1])
    lens_all = torch.empty(R, dtype=torch.int64, device=device)
    picks = torch.empty(R, dtype=torch.int64, device=device)
    for s in range(0, R, chunk):
        n = min(chunk, R - s)
        a = torch.randint(0, P, (n,), generator=g, device=device)
        b = torch.randint(0, P, (n,), generator=g, device=device)
        p = torch.minimum(a, b)
        picks[s:s + n] = p
        lens_all[s:s + n] = k[p]
    off = torch.zeros(R + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens_all, 0, out=off[1:])
    H = int(off[-1])
    assert H < 2 ** 32, "one batch holds < 2^32 ids"
    ids = torch.empty(H, dtype=torch.int32, device=device)
    for s in range(0, R, chunk):
        n = min(chunk, R - s)
        ln = lens_all[s:s + n]
        o = off[s:s + n]
        tot = int(off[s + n] - off[s])
        rr = torch.repeat_interleave(torch.arange(n, device=device), ln, output_size=tot)
        j = torch.arange(tot, device=device) + int(off[s]) - o[rr]
        ids[int(off[s]):int(off[s]) + tot] = pool_ids[pool_off[picks[s:s + n]][rr] + j]
    off32 = (off & 0xFFFFFFFF).to(torch.int64)
    off32 = torch.where(off32 >= 2 ** 31, off32 - 2 ** 32, off32).to(torch.int32)   # uint32 bits in int32
    return ids, off32


kSliceChunk = 1 << 22


def reads_slice(pool_off, pool_ids, lo, hi, seed=7, device=None):
    """Reads [lo, hi) of ONE experiment whose read stream is defined chunk by chunk (chunk c of 2^22 reads draws
    its labels from a generator seeded with (seed, c)), so any slice is reproducible on its own: N ranks holding
    [r R/N, (r+1) R/N) together hold exactly the reads a single process generates as [0, R) -- the strong-scaling
    shards of bench.py and of the multi-rank tests.  -> (ids int32[H], offsets int32[hi-lo+1])"""
    device = device or pool_ids.device
    P = pool_off.numel() - 1
    k = (pool_off[1:] - pool_off[:-1])
    n = hi - lo
    picks = torch.empty(n, dtype=torch.int64, device=device)
    for c in range(lo // kSliceChunk, (max(hi, lo + 1) - 1) // kSliceChunk + 1):
        g = torch.Generator(device=device); g.manual_seed((int(seed) * 1000003 + c) & 0x7FFFFFFFFFFFFFFF)
        a = torch.randint(0, P, (kSliceChunk,), generator=g, device=device)
        b = torch.randint(0, P, (kSliceChunk,), generator=g, device=device)
        p = torch.minimum(a, b)
        c0 = c * kSliceChunk
        s, e = max(lo, c0), min(hi, c0 + kSliceChunk)
        if e > s:
            picks[s - lo:e - lo] = p[s - c0:e - c0]
    lens = k[picks]
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=off[1:])
    H = int(off[-1])
    assert H < 2 ** 32, "one batch holds < 2^32 ids"
    ids = torch.empty(H, dtype=torch.int32, device=device)
    step = 1 << 23
    for s in range(0, n, step):
        m = min(step, n - s)
        ln = lens[s:s + m]
        tot = int(off[s + m] - off[s])
        rr = torch.repeat_interleave(torch.arange(m, device=device), ln, output_size=tot)
        j = torch.arange(tot, device=device) + int(off[s]) - off[s:s + m][rr]
        ids[int(off[s]):int(off[s]) + tot] = pool_ids[pool_off[picks[s:s + m]][rr] + j]
    off32 = (off & 0xFFFFFFFF).to(torch.int64)
    off32 = torch.where(off32 >= 2 ** 31, off32 - 2 ** 32, off32).to(torch.int32)   # uint32 bits in int32
    return ids, off32


def workload(M, P, R, seed=42, device="cpu"):
    """Convenience: (ref_len int32[M], ids, offsets) for one synthetic experiment."""
    ref_len = transcript_lengths(M, seed, device)
    off, pids = label_pool(M, P, seed, device)
    ids, offs = reads_from_pool(off, pids, R, seed=7, device=device)
    return ref_len, ids, offs
