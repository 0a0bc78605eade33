# dev: VALU / SALU / LDS instruction counts of the class-build kernels for the env given (one --pmc pass)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  cd /tmp; rm -rf /tmp/eqv
  env $v EQ_CFG3=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/eqv -- python $R/tools/eq_probe.py > /tmp/eqv.out 2>&1
  echo "== $v" | sed 's#/root/repo/sailfish_amd/csrc/variants/##'
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
fs = glob.glob('/tmp/eqv/**/*counter_collection.csv', recursive=True)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if k.startswith("k_part"): acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print("  ", k, " ".join(f"{c.replace('SQ_','')}={x/1.2e9*64:.0f}" for c, x in sorted(v.items()) if 'INSTS' in c), "(per 64 reads)", f"WAIT_ANY/WAVE_CYCLES={v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1):.2f}")
PY
done
