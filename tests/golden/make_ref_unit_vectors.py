"""Writes ref_unit_vectors.json from the REFERENCE's own code, run here: oracle/_ref/libsailfish_ref.so is
oracle/ref_glue.cpp (this repo's extern "C" driver) around reference units compiled unmodified from /root/reference
(src/LibraryFormat.cpp, include/MultinomialSampler.hpp, include/cuckoohash_map.hh, src/xxhash.c; recipe: oracle/Makefile).

  library_format : every (type, orientation, strandedness) triple -> formatID(), check(), operator<< text;
                   every id 0..maxLibTypeID() -> formatFromID()
  eq_build       : seeded hit lists -> the label -> (count, XXH64) table that libcuckoo's upsert + the reference's
                   XXH64 produce with addGroup's update rule (4 threads upserting concurrently), sorted by label
  multinomial    : draws of the reference's MultinomialSampler (random_device-seeded: a SAMPLE, for two-sample tests)
                   for a k = 7 problem (linear-scan branch, k <= 100) and a k = 150 problem (binary-search branch)

Run in the container that holds /root/reference:  python tests/golden/make_ref_unit_vectors.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

L = O.ref_sailfish()
assert L is not None, "build oracle/_ref first (make -C oracle ref)"

fmts = []
for t in range(2):
    for o in range(4):
        for s in range(5):
            buf = bytes(200)
            import ctypes as C
            b = C.create_string_buffer(200)
            n = L.ref_format_str(t, o, s, b, 200)
            fmts.append({"format": [t, o, s], "id": int(L.ref_format_id(t, o, s)), "check": bool(L.ref_format_check(t, o, s)),
                         "str": b.value.decode()})
from_id = []
for i in range(L.ref_format_max_id() + 1):
    out = (C.c_int * 3)()
    L.ref_format_from_id(i, out)
    from_id.append({"id": i, "format": [int(out[0]), int(out[1]), int(out[2])]})

rng = np.random.default_rng(20260928)
pool = [tuple(sorted(set(rng.integers(0, 50, rng.integers(1, 9)).tolist()))) for _ in range(150)]
pool = [tuple(range(40)), (7,), (7, 8), (8, 7)] + pool                # a stripe-path label (>= 32 B), and an order-sensitive pair
reads = [pool[min(rng.integers(0, len(pool)), rng.integers(0, len(pool)))] for _ in range(4000)] + [()] * 5
rng.shuffle(reads)
ids = np.array([x for r in reads for x in r], np.uint32)
off = np.zeros(len(reads) + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in reads])
tab = O.ref_eq_build(ids, off, n_threads=4)
assert tab == O.ref_eq_build(ids, off, n_threads=1)
eq = {"ids": ids.tolist(), "off": off.tolist(),
      "table": [{"label": list(k), "count": v[0], "hash": "%016x" % v[1]} for k, v in sorted(tab.items())]}

p7 = np.array([0.05, 0.30, 0.02, 0.25, 0.08, 0.20, 0.10])
p150 = rng.random(150); p150 /= p150.sum()
mn = {"k7": {"n": 20000, "p": p7.tolist(), "draws": [O.ref_multinomial(20000, p7).tolist() for _ in range(48)]},
      "k150": {"n": 60000, "p": p150.tolist(), "draws": [O.ref_multinomial(60000, p150).tolist() for _ in range(24)]}}

json.dump({"source": "kingsfordgroup/sailfish v0.10.0 units compiled unmodified (oracle/Makefile `ref`), driven by oracle/ref_glue.cpp",
           "library_format": {"formats": fmts, "from_id": from_id, "max_id": int(L.ref_format_max_id())},
           "eq_build": eq, "multinomial": mn},
          open(os.path.join(HERE, "ref_unit_vectors.json"), "w"))
print("wrote ref_unit_vectors.json:", len(fmts), "formats,", len(eq["table"]), "classes")
