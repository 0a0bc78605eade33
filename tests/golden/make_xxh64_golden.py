"""Generates tests/golden/xxh64_vectors.json from the REFERENCE's own xxhash.c, compiled
unmodified into oracle/_ref/libxxhash_ref.so by oracle/Makefile (needs /root/reference; run in
the build container only).  The fixture is data: inputs (uint32 lists) and expected XXH64
(seed 0) outputs as hex strings."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

R = O.ref_xxhash()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"


def ref_hash(words):
    b = np.asarray(words, dtype=np.uint32).tobytes()
    buf = C.create_string_buffer(b, len(b))
    return int(R.XXH64(C.cast(buf, C.c_void_p), len(b), 0))


rng = np.random.default_rng(20260927)
vectors = []
# the lists named in SURVEY.md 8c plus every length class of the algorithm
fixed = [[], [5], [2, 9], [1, 2, 3], list(range(9))]
for n in [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 63, 64, 65, 199, 200]:
    fixed.append(rng.integers(0, 2 ** 32, n, dtype=np.uint32).tolist())
for n in [1, 4, 7, 8, 9, 200]:
    fixed.append(sorted(rng.choice(200000, n, replace=False).tolist()))
fixed += [[0], [0xFFFFFFFF], [0] * 8, [0xFFFFFFFF] * 9, [1, 2], [2, 1]]
for w in fixed:
    vectors.append({"ids": [int(x) for x in w], "xxh64": "%016x" % ref_hash(w)})
json.dump({"source": "XXH64(ptr, 4*n, seed=0) of reference src/xxhash.c via oracle/_ref/libxxhash_ref.so",
           "vectors": vectors}, open(os.path.join(os.path.dirname(__file__), "xxh64_vectors.json"), "w"), indent=0)
print(len(vectors), "vectors written")
