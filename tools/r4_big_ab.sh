# dev (round 4): the 18 M-class probe -- groups of the partitioned passes vs the generic kernel, kernel breakdown
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "== generic kernel only (SFGPU_EQ_PARTITION=0)"; EXPECTED=50000000 SFGPU_EQ_PARTITION=0 python $R/tools/big_table_probe.py 2>&1 | tail -1 | cut -c1-220
echo "== groups, kernel trace"
cd /tmp; rm -rf /tmp/bt
EXPECTED=50000000 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -- python $R/tools/big_table_probe.py > /tmp/bt.out 2>&1
tail -1 /tmp/bt.out | cut -c1-200
f=$(find /tmp/bt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not (k.startswith("k_") ): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in sorted(acc.items(), key=lambda x: -sum(x[1]))[:10]:
    n = len(d) // 3 if len(d) >= 3 else len(d)
    print(f"  {k[:36]:36s} n={n:4d} per build sum={sum(d[-n:])/1e3:8.3f} ms  avg {sum(d[-n:])/n:8.1f} us")
PY
