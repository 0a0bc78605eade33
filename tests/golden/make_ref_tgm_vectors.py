"""Writes ref_tgm_vectors.json from the REFERENCE's own include/TranscriptGeneMap.hpp, compiled unmodified into
oracle/_ref/libtgm_ref.so (oracle/Makefile `ref`; driver: oracle/ref_glue_tgm.cpp): the gene a transcript name maps to
-- TranscriptGeneMap::geneName, lower_bound without an equality test -- for names in the map, absent names that fall
between two entries, before the first, and past the last.

Run in the container that holds /root/reference:  python tests/golden/make_ref_tgm_vectors.py"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtgm_ref.so"))
L.ref_tgm_gene_names.restype = C.c_size_t
L.ref_tgm_gene_names.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_char_p), C.c_size_t, C.c_char_p, C.c_size_t]


def arr(strs):
    return (C.c_char_p * len(strs))(*[s.encode() for s in strs])


def lookup(tnames, gnames, t2g, queries):
    out = C.create_string_buffer(1 << 20)
    L.ref_tgm_gene_names(arr(tnames), len(tnames), arr(gnames), len(gnames), (C.c_size_t * len(t2g))(*t2g), arr(queries), len(queries), out, len(out))
    return out.value.decode().split("\n")


rng = np.random.default_rng(7)
cases = []
for n_t, n_g in ((4, 3), (40, 11), (300, 60)):
    names = sorted({"ENST%07d.%d" % (rng.integers(0, 10 ** 6), rng.integers(1, 9)) for _ in range(n_t)} | ({"tA", "tB", "tC", "tE"} if n_t == 4 else set()))
    genes = ["G%03d" % i for i in range(n_g)]
    t2g = [int(rng.integers(0, n_g)) for _ in names]
    absent = ["", "A", "ENST", "ENST0500000.1", "ENST9999999.9", "tD", "tZ", "zz", names[0][:-1], names[-1] + "x", names[len(names) // 2] + "0"]
    queries = list(names) + absent
    cases.append({"transcripts": names, "genes": genes, "t2g": t2g, "queries": queries, "gene_of_query": lookup(names, genes, t2g, queries)})

json.dump({"source": "kingsfordgroup/sailfish v0.10.0 include/TranscriptGeneMap.hpp compiled unmodified (oracle/Makefile `ref`, "
                     "-include limits), driven by oracle/ref_glue_tgm.cpp", "cases": cases},
          open(os.path.join(HERE, "ref_tgm_vectors.json"), "w"))
print("wrote ref_tgm_vectors.json:", [len(c["queries"]) for c in cases], "queries")
