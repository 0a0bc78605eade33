"""CPU restatement of the quasi-mapping front end's CONTRACT (test infrastructure: only tests/ may import it).

PARITY UNPINNED.  The reference maps reads with RapMap (COMBINE-lab/RapMap @ sf-v0.10.1, fetched by
scripts/fetchRapMap.sh:20; call sites src/SailfishQuantify.cpp:141-142, 192-213, 487-488, 526-528), which is absent
from the reference tree and cannot be built here; nothing in the reference pins its output.  What is restated below is
therefore not RapMap but the contract of this repository's mapper (sailfish_amd/csrc/mapper.hip): the simplest
exact-seed scheme that yields the KIND of record the hot path consumes (the QuasiAlignment fields sfgpu_hit carries),
checked against the ground truth the reference's bundled sample_data carries in its read names.

  index   : every k-mer (k = 31) of every transcript that consists of A/C/G/T only (case folded), with its
            (transcript, position); occurrences of a k-mer in (transcript, position) order, at most `max_occ` kept.
  a read  : two seeds, at offsets 0 and len - k (reads shorter than k do not map), looked up on the forward strand
            (fwd = 1) and as reverse complement (fwd = 0), in the order fwd-seed0, fwd-seed1, rc-seed0, rc-seed1; the FIRST
            occurrence seen for a (transcript, strand) fixes the read's position there: p - seed offset (may be negative).
            The read's hits are sorted by (transcript, strand).
  a pair  : every (left hit, right hit) on one transcript with opposite strands is a PAIRED_END_PAIRED record (status 3,
            fragment length = max end - min start); if there is none, the left hits (status 1) then the right hits
            (status 2) are kept as orphans.
  single  : status 0 records."""
import numpy as np

from .oracle import HIT_DTYPE

_CODE = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}     # ACGTacgt


def _codes(seq: bytes):
    return np.array([_CODE.get(b, 4) for b in seq], np.uint8)


def build_index(transcripts, k=31, max_occ=1000):
    """transcripts: list of bytes -> dict kmer tuple-free key (python int, 2 bits per base, first base most significant)"""
    index = {}
    for t, s in enumerate(transcripts):
        c = _codes(s)
        for p in range(len(c) - k + 1):
            w = c[p:p + k]
            if (w > 3).any():
                continue
            key = 0
            for x in w:
                key = (key << 2) | int(x)
            lst = index.setdefault(key, [])
            if len(lst) < max_occ:
                lst.append((t, p))
    return index


def _key(codes):
    if len(codes) == 0 or (codes > 3).any():
        return None
    key = 0
    for x in codes:
        key = (key << 2) | int(x)
    return key


def seed_offsets(n, k, seeds):
    """offsets of the `seeds` seeds of a read of n bases: floor(j (n - k) / (seeds - 1)); a seed whose offset equals the
    previous one's (a read barely longer than k) is dropped"""
    out = []
    for j in range(seeds):
        o = (j * (n - k)) // (seeds - 1)
        if j == 0 or o != (((j - 1) * (n - k)) // (seeds - 1)):
            out.append((j, o))
    return out


def map_read(index, read: bytes, k=31, seeds=2):
    """seeds = 2: the contract above.  seeds > 2: seeds spread evenly between offsets 0 and len - k; a (transcript, strand) is
    positioned by the first seed (lowest index, fwd strand before rc) that hits it, and only the pairs that the MOST seeds hit
    are kept."""
    c = _codes(read)
    n = len(c)
    if n < k:
        return []
    rc = (3 - c[::-1]).astype(np.uint8); rc[c[::-1] > 3] = 4
    found, votes = {}, {}
    for fwd, q in ((1, c), (0, rc)):
        for j, o in seed_offsets(n, k, seeds):
            key = _key(q[o:o + k])
            if key is None:
                continue
            for t, p in index.get(key, ()):
                found.setdefault((t, fwd), p - o)
                votes.setdefault((t, fwd), set()).add(j)
    if seeds > 2 and len(found) > 1:
        best = max(len(v) for v in votes.values())
        found = {key: p for key, p in found.items() if len(votes[key]) == best}
    return sorted((t, fwd, p) for (t, fwd), p in found.items())


def map_reads(index, reads1, reads2=None, k=31, seeds=2):
    """-> (hits HIT_DTYPE[n], offsets uint32[R + 1])"""
    recs, off = [], [0]
    for i, r1 in enumerate(reads1):
        left = map_read(index, r1, k, seeds)
        if reads2 is None:
            recs += [(t, p, 0, 0, len(r1), 0, f, 0, 0, 0) for t, f, p in left]
        else:
            r2 = reads2[i]
            right = map_read(index, r2, k, seeds)
            paired = [(t, f, p, f2, p2) for t, f, p in left for t2, f2, p2 in right if t2 == t and f2 != f]
            if paired:
                for t, f, p, f2, p2 in paired:
                    recs.append((t, p, p2, max(p + len(r1), p2 + len(r2)) - min(p, p2), len(r1), len(r2), f, f2, 3, 0))
            else:
                recs += [(t, p, 0, 0, len(r1), len(r2), f, 0, 1, 0) for t, f, p in left]
                recs += [(t, p, 0, 0, len(r2), len(r1), f, 0, 2, 0) for t, f, p in right]
        off.append(len(recs))
    return np.array(recs, dtype=HIT_DTYPE), np.array(off, np.uint32)


# ---- the SCAN contract (round 3): RapMap-style maximal-match extension instead of two fixed end seeds ----------------
# PARITY WITH RAPMAP STAYS UNPINNED (see the header).  What follows restates csrc/mapper.hip's scan mode, modelled on what the
# quasi-mapping paper describes (walk the read; look the next seed up; extend the match as far as any occurrence allows --
# the maximal mappable prefix --; jump to its end; the transcripts that the most matches agree on are the hits):
#   * the index is the sorted table of k-mers (k = 31) with their (transcript, position); a seed of s <= k bases is looked up
#     as a PREFIX range of that table (so a seed can only be found at positions that start a whole k-mer: the last k - 1
#     bases of a transcript and k-mers holding a non-ACGT base are not seed starts);
#   * the forward strand (fwd = 1) and the reverse complement are walked in LOCKSTEP: one step on the forward strand, one on
#     the reverse complement, and so on; a match that covers the whole read ends both walks;
#   * a step on a strand: windows q[i:i+s] that hold a non-ACGT base are skipped; the next window is looked up; no occurrence or
#     more than max_occ occurrences -> i += 1; else every occurrence is extended base by base against the transcript (A/C/G/T,
#     case folded) up to the end of the read / transcript; L = the longest extension; the occurrences that reach L, in table
#     order, form a GROUP (i, L), groups numbered in the order they are found; i += L - s + 1; at most 8 groups per mate;
#   * a (transcript, strand) is positioned by the first group that holds it (p - i) and gets one vote per group that holds
#     it; with more than one (transcript, strand) only those with the most votes are kept; hits sorted by (transcript, strand).
MAX_GROUPS = 8


def build_scan_index(transcripts, k=31):
    """-> (sorted list of (kmer key, t, p), transcripts' code arrays): the table the device sorts (key, then (t, p))"""
    codes = [_codes(s) for s in transcripts]
    table = []
    for t, c in enumerate(codes):
        for p in range(len(c) - k + 1):
            w = c[p:p + k]
            if (w > 3).any():
                continue
            key = 0
            for x in w:
                key = (key << 2) | int(x)
            table.append((key, t, p))
    table.sort()
    return table, codes, k


def _prefix_range(table, k, prefix, s):
    import bisect
    lo = bisect.bisect_left(table, (prefix << (2 * (k - s)), -1, -1))
    hi = bisect.bisect_left(table, ((prefix + 1) << (2 * (k - s)), -1, -1))
    return lo, hi


def scan_read(sindex, read: bytes, s=19, max_occ=1000):
    table, tcodes, k = sindex
    c = _codes(read)
    n = len(c)
    if n < s:
        return []
    rc = (3 - c[::-1]).astype(np.uint8); rc[c[::-1] > 3] = 4
    strands = [(1, c), (0, rc)]
    pos = [0, 0]; live = [True, True]
    groups, whole = [], False
    while (live[0] or live[1]) and not whole and len(groups) < MAX_GROUPS:
        for w in (0, 1):                                   # lockstep: one lookup on the forward strand, one on the reverse complement
            if not live[w] or whole or len(groups) >= MAX_GROUPS:
                continue
            fwd, q = strands[w]
            i = pos[w]
            while i + s <= n and (q[i:i + s] > 3).any():   # windows that hold a non-ACGT base are skipped (not a step)
                i += 1
            if i + s > n:
                live[w] = False; continue
            key = _key(q[i:i + s])
            lo, hi = _prefix_range(table, k, key, s)
            if hi == lo or hi - lo > max_occ:
                i += 1
            else:
                ext = []
                for _, t, p in table[lo:hi]:
                    tc = tcodes[t]
                    e = s
                    while i + e < n and p + e < len(tc) and q[i + e] < 4 and tc[p + e] == q[i + e]:
                        e += 1
                    ext.append(e)
                L = max(ext)
                groups.append((fwd, i, L, [(t, p) for (_, t, p), e in zip(table[lo:hi], ext) if e == L]))
                if L == n:
                    whole = True                           # the whole read matched: both walks end
                i += L - s + 1
            if i + s > n:
                live[w] = False
            pos[w] = i
    found, votes = {}, {}
    for g, (fwd, i, L, occ) in enumerate(groups):
        for t, p in occ:
            found.setdefault((t, fwd), p - i)
            votes.setdefault((t, fwd), set()).add(g)
    if len(found) > 1:
        best = max(len(v) for v in votes.values())
        found = {key: p for key, p in found.items() if len(votes[key]) == best}
    return sorted((t, fwd, p) for (t, fwd), p in found.items())


def scan_reads(sindex, reads1, reads2=None, s=19, max_occ=1000):
    """map_reads with the scan contract -> (hits HIT_DTYPE[n], offsets uint32[R + 1])"""
    recs, off = [], [0]
    for i, r1 in enumerate(reads1):
        left = scan_read(sindex, r1, s, max_occ)
        if reads2 is None:
            recs += [(t, p, 0, 0, len(r1), 0, f, 0, 0, 0) for t, f, p in left]
        else:
            r2 = reads2[i]
            right = scan_read(sindex, r2, s, max_occ)
            paired = [(t, f, p, f2, p2) for t, f, p in left for t2, f2, p2 in right if t2 == t and f2 != f]
            if paired:
                for t, f, p, f2, p2 in paired:
                    recs.append((t, p, p2, max(p + len(r1), p2 + len(r2)) - min(p, p2), len(r1), len(r2), f, f2, 3, 0))
            else:
                recs += [(t, p, 0, 0, len(r1), len(r2), f, 0, 1, 0) for t, f, p in left]
                recs += [(t, p, 0, 0, len(r2), len(r1), f, 0, 2, 0) for t, f, p in right]
        off.append(len(recs))
    return np.array(recs, dtype=HIT_DTYPE), np.array(off, np.uint32)
