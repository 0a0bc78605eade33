# dev (round 6): where the SHARED form's time goes -- route alone with and without its stores, and what leaves the L2 (WRITE_SIZE, write requests)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {   # $1 = label, $2 = variant, rest = env
  lab=$1; export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_$2.so; shift 2
  cd /tmp; rm -rf /tmp/eqk
  env "$@" EQ_CFG3=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  python - "$f" "$lab" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not k.startswith("k_part"): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3 if len(d) >= 3 else len(d)
    print(f"{sys.argv[2]:24s} {k[:24]:24s} n={n:3d} sum={sum(d[-n:])/1e3:8.3f} ms")
PY
}
pmc() {   # $1 = label, $2 = variant, rest = env
  lab=$1; export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_$2.so; shift 2
  cd /tmp; rm -rf /tmp/eqc
  env "$@" EQ_CFG3=1 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d /tmp/eqc -- python $R/tools/eq_probe.py > /tmp/eqc.out 2>&1 || tail -2 /tmp/eqc.out
  python - "$lab" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/eqc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
        if k.startswith("k_part_route"): acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(sys.argv[1], k, " ".join(f"{c}={x / 1.2e9:.3f}/read" for c, x in sorted(v.items())))
PY
}
run direct base SFGPU_EQ_SHARED=0 SFGPU_X_ROUTE_ONLY=1
run shared base SFGPU_EQ_SHARED=1 SFGPU_X_ROUTE_ONLY=1
run direct_nostore nostore SFGPU_EQ_SHARED=0 SFGPU_X_ROUTE_ONLY=1
run shared_nostore nostore SFGPU_EQ_SHARED=1 SFGPU_X_ROUTE_ONLY=1
pmc direct base SFGPU_EQ_SHARED=0 SFGPU_X_ROUTE_ONLY=1
pmc shared base SFGPU_EQ_SHARED=1 SFGPU_X_ROUTE_ONLY=1
