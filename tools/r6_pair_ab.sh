# dev (round 6): what could a block-local multisplit make of the route pass's stores?  Timing-only builds in which groups of 2 / 4 / 8 lanes
# store their granules as ONE run (wrong bins): route alone, cfg3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in base pair2 pair4 pair8 base; do
  export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_$v.so
  cd /tmp; rm -rf /tmp/eqk
  EQ_CFG3=1 SFGPU_X_ROUTE_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not k.startswith("k_part"): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3 if len(d) >= 3 else len(d)
    print(f"   {sys.argv[2]:12s} {k[:24]:24s} n={n:3d} sum={sum(d[-n:])/1e3:8.3f} ms")
PY
done
