# dev (round 6): SHARED form, pipelined reservation: kernel times (full pipeline, cfg3 + cfg2), route alone without stores, stress
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {   # $1 = label, $2 = variant, rest = env
  lab=$1; export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_$2.so; shift 2
  cd /tmp; rm -rf /tmp/eqk
  env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  echo "$lab: $(tail -1 /tmp/eqk.out | cut -c1-60)"
  python - "$f" "$lab" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not (k.startswith("k_part") or k.startswith("k_shared")): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3 if len(d) >= 3 else len(d)
    print(f"   {sys.argv[2]:24s} {k[:24]:24s} n={n:3d} sum={sum(d[-n:])/1e3:8.3f} ms")
PY
}
for rep in 1 2; do
run cfg3_direct base EQ_CFG3=1 SFGPU_EQ_SHARED=0
run cfg3_shared base EQ_CFG3=1 SFGPU_EQ_SHARED=1
done
run cfg2_direct base SFGPU_EQ_SHARED=0
run cfg2_shared base SFGPU_EQ_SHARED=1
run direct_nostore nostore EQ_CFG3=1 SFGPU_EQ_SHARED=0 SFGPU_X_ROUTE_ONLY=1
run shared_nostore nostore EQ_CFG3=1 SFGPU_EQ_SHARED=1 SFGPU_X_ROUTE_ONLY=1
cd $R
export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_base.so
SFGPU_EQ_SHARED=1 timeout 300 python tools/builder_stress.py 11 ${STRESS_S:-60} 2>&1 | tail -2 | cut -c1-250
