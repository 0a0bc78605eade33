# dev (round 6): the SHARED form of the route pass (XCD-shared bins, one reservation per step and region) against the direct form:
# per-kernel times under rocprofv3 (cfg3 and cfg2), then the builder stress against the CPU restatement with the form switched on
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-base}
export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_$V.so
run() {   # $1 = label, rest = env
  lab=$1; shift
  cd /tmp; rm -rf /tmp/eqk
  env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  echo "== $lab: $(tail -1 /tmp/eqk.out | cut -c1-220)"
  f=$(find /tmp/eqk -name '*kernel_trace.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not (k.startswith("k_part") or k.startswith("k_shared") or k in ("k_insert", "k_commit")): continue
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    n = len(d) // 3 if len(d) >= 3 else len(d)
    last = d[-n:]
    print(f"  {k[:40]:40s} n={n:3d} sum={sum(last)/1e3:8.3f} ms  calls(us): " + " ".join(f"{x:.0f}" for x in last[:10]))
PY
}
run cfg3_direct EQ_CFG3=1 SFGPU_EQ_SHARED=0
run cfg3_shared EQ_CFG3=1 SFGPU_EQ_SHARED=1
run cfg2_direct SFGPU_EQ_SHARED=0
run cfg2_shared SFGPU_EQ_SHARED=1
cd $R
SFGPU_EQ_SHARED=1 timeout 300 python tools/builder_stress.py 7 ${STRESS_S:-60} 2>&1 | tail -3
