# dev: per-call durations of the partition kernels under rocprofv3 (cfg2 class build)
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/abl
rocprofv3 --kernel-trace --output-format csv -d /tmp/abl -- python $GRAFT_REPO_ROOT/tools/eq_probe.py > /tmp/abl.out 2>&1
tail -1 /tmp/abl.out | cut -c1-140
f=$(find /tmp/abl -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for name in ("k_part_hist", "k_part_scatter", "k_part_insert"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    print(name, " ".join(f"{x:.0f}" for x in d[-3:]), "us (last build: sub-batches 1..3)")
PY
