# dev (round 5): the kernels of one EM handle creation + optimize() on cfg3's classes, in launch order with start offsets and durations
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; rm -rf /tmp/pt1
EMP_SHAPES=${1:-cfg3} rocprofv3 --kernel-trace --output-format csv -d /tmp/pt1 -- python $R/tools/r5_plan_probe.py > /tmp/pt1.out 2>&1
grep -E "==|create" /tmp/pt1.out
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pt1/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last k_em_persist launch, and everything after the one before it
idx = [i for i, r in enumerate(rows) if "k_em_persist" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")[-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name}")
    prev_end = e
PY
