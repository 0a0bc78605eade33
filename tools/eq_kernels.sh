# dev: per-call durations of the class-build kernels under rocprofv3 for each variant library given
#   bash tools/eq_kernels.sh base cond ...   (names of sailfish_amd/csrc/variants/libsfgpu_<name>.so; "main" = the product library)
export TMPDIR=/tmp
for v in "$@"; do
  cd /tmp; rm -rf /tmp/eqk
  if [ "$v" = main ]; then unset SFGPU_LIB_PATH; else export SFGPU_LIB_PATH=$GRAFT_REPO_ROOT/sailfish_amd/csrc/variants/libsfgpu_$v.so; fi
  EQ_CFG3=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $GRAFT_REPO_ROOT/tools/eq_probe.py > /tmp/eqk.out 2>&1
  echo "== $v: $(tail -1 /tmp/eqk.out | cut -c1-200)"
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not (k.startswith("k_part") or k in ("k_insert", "k_commit", "k_rehash", "k_table_init")) and "scan" not in k.lower(): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3 if len(d) >= 3 else len(d)       # three builds per run: show the last one
    last = d[-n:]
    print(f"  {k[:40]:40s} n={n:3d} sum={sum(last)/1e3:8.3f} ms  calls(us): " + " ".join(f"{x:.0f}" for x in last[:10]))
PY
done
