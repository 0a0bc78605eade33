# dev (round 5): SQ counters of k_em_persist per wavefront and step (how many instructions a step costs a thread)
#   tools/r5_persist_counters.sh [shape]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-tiny}
cd /tmp; rm -rf /tmp/pc1
EMP_SHAPES=$SHAPE EMP_MODES=persist rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/pc1 -- python $R/tools/r5_persist_probe.py > /tmp/pc1.out 2>&1
grep -E "==|us/iter" /tmp/pc1.out | cut -c1-120
python - <<'PY'
import csv, glob, collections, re
fs = glob.glob('/tmp/pc1/**/*counter_collection.csv', recursive=True)
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(fs[0])):
    if "k_em_persist" in r["Kernel_Name"]: rows[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])][r["Counter_Name"]] = float(r["Counter_Value"])
iters = [int(x) for x in re.findall(r"iters\s+(\d+)", open('/tmp/pc1.out').read())]
print("dispatches:", len(rows), "iters seen:", iters)
for (d, k), v in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    w = max(v.get("SQ_WAVES", 1), 1)
    print(d, k[-30:], {c.replace("SQ_", ""): round(x / w) for c, x in v.items() if c != "SQ_WAVES"}, "waves", int(w))
PY
