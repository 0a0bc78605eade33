# dev: A/B of the class-build kernels under rocprofv3 (cfg3 class build only, tools/eq_probe.py).
#   bash tools/r3_eq_ab.sh "SFGPU_EQ_RING=0" "SFGPU_EQ_RING=1" "SFGPU_EQ_RING=1 SFGPU_EQ_RING_BLOCKS=512" ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  cd /tmp; rm -rf /tmp/eqk
  env $v EQ_CFG3=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  echo "== $v: $(tail -1 /tmp/eqk.out | cut -c1-400)"
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not (k.startswith("k_part") or k in ("k_insert", "k_commit", "k_rehash", "k_table_init", "k_hot_select", "k_sub_batch_begin")) and "scan" not in k.lower(): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3 if len(d) >= 3 else len(d)       # three builds per run: show the last one
    last = d[-n:]
    print(f"  {k[:40]:40s} n={n:3d} sum={sum(last)/1e3:8.3f} ms  calls(us): " + " ".join(f"{x:.0f}" for x in last[:10]))
PY
done
