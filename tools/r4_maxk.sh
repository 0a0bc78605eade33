# dev (round 4): how much of the class build is the labels' further granules?  cfg3's reads with labels cut to <= K ids
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for k in 200 9 7 3; do
  cd /tmp; rm -rf /tmp/eqk
  EQ_MAXK=$k EQ_CFG3=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  echo "== max label length $k: $(tail -1 /tmp/eqk.out | cut -c1-90)"
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not k.startswith("k_part"): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3
    print(f"  {k[:30]:30s} n={n:3d} sum={sum(d[-n:])/1e3:8.3f} ms")
PY
done
