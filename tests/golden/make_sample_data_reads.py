"""Writes sample_data_reads.npz: the reference's bundled sample_data.tgz (BASELINE config 1: 15 transcripts, 10 000 read
pairs of 2 x 50 bp) as plain arrays -- transcript names and sequences, both mate files' bases, and the simulator's truth
parsed from the read names (`@id:transcript:pos:fraglen`).  Data only: it is the INPUT of the mapping front end
(sailfish_amd/mapper.py); the hit records the mapper must produce from it are sample_data_hits.npz.
Run here (needs /root/reference); the GPU box uses the committed file."""
import os
import tarfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(src="/root/reference/sample_data.tgz"):
    tf = tarfile.open(src)
    get = lambda n: tf.extractfile("sample_data/" + n).read().decode()
    names, seqs = [], []
    for block in get("transcripts.fasta").split(">")[1:]:
        lines = block.split("\n")
        names.append(lines[0].split()[0]); seqs.append("".join(lines[1:]))
    r1 = get("reads_1.fastq").split("\n"); r2 = get("reads_2.fastq").split("\n")
    n = len(r1) // 4
    m1 = np.frombuffer("".join(r1[4 * i + 1] for i in range(n)).encode(), np.uint8).reshape(n, -1)
    m2 = np.frombuffer("".join(r2[4 * i + 1] for i in range(n)).encode(), np.uint8).reshape(n, -1)
    truth = np.array([names.index(r1[4 * i].split(":")[1]) for i in range(n)], np.uint32)
    code = np.zeros(256, np.uint8); code[list(b"ACGT")] = [0, 1, 2, 3]
    assert set(np.unique(np.concatenate([m1.ravel(), m2.ravel()])).tolist()) <= set(b"ACGT")       # the simulator emits no N
    pack = lambda m: np.packbits(np.unpackbits(code[m].reshape(-1, 1), axis=1)[:, 6:].reshape(-1))   # 2 bits per base
    out = os.path.join(HERE, "sample_data_reads.npz")
    np.savez_compressed(out, names=np.array(names), seq=np.frombuffer("".join(seqs).encode(), np.uint8),
                        seq_off=np.cumsum([0] + [len(s) for s in seqs]).astype(np.int64), read_len=np.int64(m1.shape[1]),
                        mate1_2bit=pack(m1), mate2_2bit=pack(m2), truth=truth)
    print(out, os.path.getsize(out), "bytes;", n, "pairs of", m1.shape[1], "bp")


if __name__ == "__main__":
    main()
