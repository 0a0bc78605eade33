# dev: kernel timeline of one bench step (cfg3): per-kernel sums and the largest gaps between consecutive kernels
#   bash tools/step_timeline.sh      (on the GPU box; writes gpurun_out/timeline.txt)
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-host-pinned > /tmp/tl.out 2>&1
tail -1 /tmp/tl.out | cut -c1-300
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/timeline.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")[:48]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last full step = the last k_table_init (builder start) that is followed by >= 4 k_part_route launches before the next one
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_table_init")] + [len(rows)]
full = [starts[j] for j in range(len(starts) - 1) if sum(r[2].startswith("k_part_route") for r in rows[starts[j]:starts[j + 1]]) >= 4]
i0 = full[-1]
ends = [i for i, r in enumerate(rows) if i > i0 and r[2].startswith("k_mass")]
i1 = ends[0] if ends else len(rows) - 1
seg = rows[i0:i1 + 1]
wall = (seg[-1][1] - seg[0][0]) / 1e6
busy = sum(e - s for s, e, _ in seg) / 1e6
print(f"step: {len(seg)} kernels, wall {wall:.3f} ms, kernel time {busy:.3f} ms, gaps {wall - busy:.3f} ms")
acc = collections.OrderedDict()
for s, e, k in seg: acc.setdefault(k, [0, 0.0]); acc[k][0] += 1; acc[k][1] += (e - s) / 1e6
for k, (n, t) in sorted(acc.items(), key=lambda x: -x[1][1])[:25]: print(f"  {k:50s} n={n:5d} {t:8.3f} ms")
gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2], seg[i + 1][2], (seg[i][1] - seg[0][0]) / 1e6) for i in range(len(seg) - 1))[::-1]
print("largest gaps (us): after kernel -> before kernel @ms into step")
for g, a, b, at in gaps[:40]: print(f"  {g:9.1f}  {a} -> {b}  @{at:.2f}")
print("gap histogram: >100us %d, 20-100us %d, 5-20us %d" % (sum(g > 100 for g, *_ in gaps), sum(20 < g <= 100 for g, *_ in gaps), sum(5 < g <= 20 for g, *_ in gaps)))
print("sum of gaps > 5 us: %.3f ms" % (sum(g for g, *_ in gaps if g > 5) / 1e3))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/timeline.txt
# the phase between the last insert and the first sweep, kernel by kernel
python - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/timeline.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")[:60]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_table_init")] + [len(rows)]
full = [starts[j] for j in range(len(starts) - 1) if sum(r[2].startswith("k_part_route") for r in rows[starts[j]:starts[j + 1]]) >= 4]
i0 = full[-1]
last_ins = max(i for i, r in enumerate(rows) if i >= i0 and r[2].startswith("k_part_insert") and i < (starts[starts.index(i0) + 1]))
first_sw = min(i for i, r in enumerate(rows) if i > last_ins and r[2].startswith("k_sweep_lds"))
print("\n== finish -> plan phase (gap before us | kernel us | name)")
for i in range(last_ins, first_sw + 1):
    print(f"  {(rows[i][0] - rows[i - 1][1]) / 1e3:8.1f} {(rows[i][1] - rows[i][0]) / 1e3:8.1f}  {rows[i][2]}")
PY
