"""dev: which reads does the builder lose on the long-label cluster of test_builder_long_labels_that_agree_in_every_sampled_id?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sailfish_amd as sf
from oracle import oracle as O
dev = torch.device("cuda:0")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
with_short = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(5)
n_lab, n = 6000, 40
lab = np.tile(np.arange(0, 3 * n, 3, dtype=np.uint32), (n_lab, 1))
fixed = {0, 1, 2, 3, 4, 5, 6, 7, n - 1, 8 + (n - 8) // 2, 8 + (n - 8) // 4}
free = [k for k in range(8, n - 1) if k not in fixed]
for d, k in enumerate(free[:4]):
    lab[:, k] += ((np.arange(n_lab) // 10 ** d) % 10).astype(np.uint32) * 1000
picks = np.concatenate([np.arange(n_lab), rng.integers(0, n_lab, n_reads - n_lab)])
rng.shuffle(picks)
ids = lab[picks].reshape(-1)
if with_short:
    short = rng.integers(0, 50_000, (n_reads, 2)).astype(np.uint32)
    ids_all = np.concatenate([ids, short.reshape(-1)])
    off = np.concatenate([np.arange(n_reads + 1, dtype=np.uint64) * n, n_reads * n + np.arange(1, n_reads + 1, dtype=np.uint64) * 2]).astype(np.uint32)
else:
    ids_all = ids; off = (np.arange(n_reads + 1, dtype=np.uint64) * n).astype(np.uint32)
ob = O.EqBuilder(); ob.add_batch(ids_all, off.astype(np.uint64)); orp, oids, ocnt, ohash = ob.finish()
eq = sf.EquivalenceClassBuilder(device=dev)
eq.start(); eq.add_batch(torch.from_numpy(ids_all.view(np.int32)).to(dev), torch.from_numpy(off.view(np.int32)).to(dev)); eq.finish()
rp, ii, cc, hh = eq.eqVec().to_numpy()
print("stats", eq.stats(), "classes", eq.n_classes, ob.n_classes, "reads", eq.total_reads, ob.total_reads)
if eq.n_classes == ob.n_classes and np.array_equal(rp, orp.astype(np.uint32)):
    bad = np.nonzero(cc != ocnt)[0]
    print("classes with wrong counts:", len(bad))
    lens = np.diff(rp.astype(np.int64))
    for b in bad[:30]:
        print("  class", b, "len", lens[b], "gpu", cc[b], "oracle", ocnt[b])
    # runs in the input: adjacent identical cluster reads
    adj = np.nonzero(picks[1:] == picks[:-1])[0]
    print("adjacent duplicate picks:", len(adj), "at", adj[:20])
