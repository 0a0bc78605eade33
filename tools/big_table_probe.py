"""dev probe (round 4): class build on a table far beyond 16 M slots -- a pool of tens of millions of distinct labels, so that the
partitioned passes run in groups of 4096 regions (eq_partitioned).  Compare with tools/eq_probe.py EQ_CFG3=1 (1.6 M classes).
  python tools/big_table_probe.py [pool labels, default 60e6] [reads, default 400e6]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sailfish_amd as sf
from sailfish_amd import synth
dev = torch.device("cuda:0")
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
R = int(float(sys.argv[2])) if len(sys.argv) > 2 else 400_000_000
M = 2_000_000
poff, pids = synth.label_pool(M, P, device=dev)
ids, off = synth.reads_from_pool(poff, pids, R, device=dev)
del poff, pids
eq = sf.EquivalenceClassBuilder(device=dev, expected_classes=int(os.environ.get("EXPECTED", "0")))
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    eq.start(); eq.add_batch(ids, off)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eq.finish(); v = eq.eqVec()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = eq.stats()
    print(f"build {it}: add {1e3*(t1-t):.2f} ms = {R/(t1-t)/1e9:.2f} G reads/s, finish+export {1e3*(t2-t1):.2f} ms, classes {eq.n_classes}, "
          f"table {st['table_slots']} slots = {st['table_slots'] // 4096} regions, launches {st['insert_launches']}, grows {st['table_grows']}, "
          f"deferred {st['deferred_reads']}, spilled {st['spilled_reads']}", flush=True)
assert eq.total_reads == R
