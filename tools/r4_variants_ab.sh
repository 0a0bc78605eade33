# dev (round 4): quick A/B of class-build variants (kernel sums of the third build, cfg3)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  lab=$1; lib=$2; shift 2
  cd /tmp; rm -rf /tmp/eqk
  if [ "$lib" = main ]; then unset SFGPU_LIB_PATH; else export SFGPU_LIB_PATH=$R/sailfish_amd/csrc/variants/libsfgpu_$lib.so; fi
  env "$@" EQ_CFG3=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/eqk -- python $R/tools/eq_probe.py > /tmp/eqk.out 2>&1
  echo "== $lab: $(tail -1 /tmp/eqk.out | cut -c1-70)"
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
}
if [ "$1" != "" ] && [ "$1" != "skip" ]; then "$@"; fi
if [ $# -eq 0 ]; then
run base main SFGPU_EQ_PIPE=0
run sb27 sb27 SFGPU_EQ_PIPE=0
run ring main SFGPU_EQ_RING=1
run b1024 main SFGPU_EQ_BLOCKS=1024
run pipe main SFGPU_EQ_PIPE=1
fi
