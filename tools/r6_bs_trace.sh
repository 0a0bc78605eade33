#!/bin/bash
# round 6: which kernels run while a persistent launch of a bootstrap lane waits for its last blocks?  (kernel trace of two lanes on cfg2)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_bs_trace
cd /tmp && export TMPDIR=/tmp
BSP_SHAPE=cfg2 BSP_N=4 BSP_XBUF=uncached BSP_LANESETS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6_bs_trace -- python $OLDPWD/tools/r6_bs_probe.py > $OLDPWD/gpurun_out/r6_bs_trace/run.log 2>&1
f=$(find /tmp/r6_bs_trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $OLDPWD/gpurun_out/r6_bs_trace/overlap.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def g(r, *names):
    for n in names:
        if n in r: return r[n]
    raise KeyError(names)
ks = [(int(g(r, "Start_Timestamp")), int(g(r, "End_Timestamp")), g(r, "Kernel_Name"), g(r, "Queue_Id", "Queue_ID", "Stream_Id") if any(k in r for k in ("Queue_Id", "Queue_ID", "Stream_Id")) else "?", g(r, "Grid_Size", "Grid_Size_X") if any(k in r for k in ("Grid_Size", "Grid_Size_X")) else "?") for r in rows]
ks.sort()
print("columns:", list(rows[0].keys()))
long = [k for k in ks if "k_em_persist" in k[2] and k[1] - k[0] > 20_000_000]
print(len(long), "persistent launches longer than 20 ms")
for (s, e, n, q, gsz) in long[:3]:
    print(f"--- persist launch queue {q}: start {s} dur {(e - s) / 1e6:.2f} ms")
    for (s2, e2, n2, q2, g2) in ks:
        if e2 > s - 2_000_000 and s2 < e + 200_000 and (s2, e2, n2) != (s, e, n):
            print(f"   {((s2 - s) / 1e3):10.1f} us .. {((e2 - s) / 1e3):10.1f} us  q {q2} grid {g2}  {n2[:90]}")
PY
