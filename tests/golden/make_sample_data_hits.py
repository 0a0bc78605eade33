"""Builds tests/golden/sample_data_hits.npz from the reference's bundled sample_data.tgz (BASELINE config 1: 15
transcripts, 10 000 read pairs of 2 x 50 bp) with a stand-in for the mapper.

RapMap (the reference's quasi-mapper, fetched at build time) is not available, so the hit records come from the
simplest exact-seed mapper that yields the same KIND of input the hot path consumes: a read maps to a transcript and
strand if one of its two 31-mer seeds (offsets 0 and 19) occurs there exactly; both mates on one transcript with
opposite strands give a PAIRED_END_PAIRED record (position, mate position, fragment length), otherwise the mates'
hits are kept as orphans (left run, then right run, each ascending in transcript id).  The fixture holds inputs only:
transcript names and lengths, the hit records, and the simulator's truth parsed from the read names
(`@id:transcript:pos:fraglen`).  Run here (needs /root/reference); the GPU box uses the committed .npz."""
import io
import os
import sys
import tarfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import HIT_DTYPE  # noqa: E402

K, READ = 31, 50
COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def main(src="/root/reference/sample_data.tgz", scan=False):
    tf = tarfile.open(src)
    get = lambda n: tf.extractfile("sample_data/" + n).read().decode()
    names, seqs = [], []
    for block in get("transcripts.fasta").split(">")[1:]:
        lines = block.split("\n")
        names.append(lines[0].split()[0]); seqs.append("".join(lines[1:]).upper())
    index = {}
    for t, s in enumerate(seqs):
        for p in range(len(s) - K + 1):
            index.setdefault(s[p:p + K], []).append((t, p))

    def map_read(r):
        found = {}
        for fwd, q in ((1, r), (0, r.encode().translate(COMP)[::-1].decode())):
            for o in (0, READ - K):
                for t, p in index.get(q[o:o + K], ()):
                    found.setdefault((t, fwd), p - o)
        return sorted((t, fwd, p) for (t, fwd), p in found.items())

    r1 = get("reads_1.fastq").split("\n"); r2 = get("reads_2.fastq").split("\n")
    n = len(r1) // 4
    if scan:
        # the SCAN contract (the mapper's default since round 3: maximal-match extension, seeds of 19 bases), restated in
        # oracle/mapper_oracle.py -- same record layout, written to sample_data_hits_scan.npz
        from oracle import mapper_oracle as MO
        sindex = MO.build_scan_index([x.encode() for x in seqs])
        map_read = lambda r: MO.scan_read(sindex, r.encode(), s=19)
    recs, off, truth = [], [0], []
    for i in range(n):
        truth.append(names.index(r1[4 * i].split(":")[1]))
        left, right = map_read(r1[4 * i + 1]), map_read(r2[4 * i + 1])
        paired = [(t, f, p, f2, p2) for t, f, p in left for t2, f2, p2 in right if t2 == t and f2 != f]
        if paired:
            for t, f, p, f2, p2 in paired:
                recs.append((t, p, p2, max(p, p2) + READ - min(p, p2), READ, READ, f, f2, 3, 0))
        else:
            recs += [(t, p, 0, 0, READ, READ, f, 0, 1, 0) for t, f, p in left]
            recs += [(t, p, 0, 0, READ, READ, f, 0, 2, 0) for t, f, p in right]
        off.append(len(recs))
    hits = np.array(recs, dtype=HIT_DTYPE)
    out = os.path.join(HERE, "sample_data_hits_scan.npz" if scan else "sample_data_hits.npz")
    np.savez_compressed(out, names=np.array(names), ref_len=np.array([len(s) for s in seqs], np.uint32),
                        hits=hits.view(np.uint8), offsets=np.array(off, np.uint32), truth=np.array(truth, np.uint32))
    mapped = int((np.diff(off) > 0).sum())
    print(f"{out}: {len(names)} transcripts, {n} read pairs, {len(hits)} hit records, {mapped} reads with a hit, "
          f"{int((hits['mate_status'] == 3).sum())} proper-pair records, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--scan"]
    main(*args, scan="--scan" in sys.argv[1:])
