# dev: time and SQ counters of the Gibbs phase kernels over cfg3-scale classes (tools/cfg5_probe.py): one kernel-trace run, one --pmc run
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; rm -rf /tmp/g5 /tmp/g5c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g5 -- python $R/tools/cfg5_probe.py > /tmp/g5.out 2>&1
grep -E "gibbs|classes" /tmp/g5.out | cut -c1-250
f=$(find /tmp/g5 -name '*kernel_stats.csv' | head -1)
grep -i gibbs $f | cut -d, -f1-4 | cut -c1-200
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/g5c -- python $R/tools/cfg5_probe.py > /tmp/g5c.out 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
fs = glob.glob('/tmp/g5c/**/*counter_collection.csv', recursive=True)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if "gibbs" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, v in acc.items():
    print(k, {c.replace("SQ_", ""): f"{x:.3g}" for c, x in v.items()}, f"WAIT_ANY/WAVE_CYCLES={v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1):.2f}", f"VALU per wave {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):.0f}")
PY
