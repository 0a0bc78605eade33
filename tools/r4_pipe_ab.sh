# dev (round 4): the pipelined partition passes against the serial form -- bench step and kernel timeline (cfg3)
#   bash tools/r4_pipe_ab.sh        (on the GPU box; writes gpurun_out/r4_pipe_*.{json,txt})
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for pipe in 1 0; do
  SFGPU_EQ_PIPE=$pipe python $R/bench.py --steps 10 --warmup 3 --no-host-pinned --no-sampling --no-cpu-baseline > $R/gpurun_out/r4_pipe_${pipe}.json 2> $R/gpurun_out/r4_pipe_${pipe}.err
  python - $R/gpurun_out/r4_pipe_${pipe}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pipe", sys.argv[1][-6], "ms/step", round(d["ms_per_step"], 3), "phase", {k: round(v, 3) for k, v in d["phase_ms"].items()}, "launches", d["roofline_class_build"]["launches_per_step"])
PY
done
# kernel timeline of the class build with the pipeline: do route(k + 1) and insert(k) overlap?
cd /tmp; rm -rf /tmp/tl
EQ_CFG3=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/tools/eq_probe.py > /tmp/tl.out 2>&1
tail -1 /tmp/tl.out | cut -c1-500
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python - "$f" > $R/gpurun_out/r4_pipe_timeline.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")[:40]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last build: from the last k_table_init
i0 = max(i for i, r in enumerate(rows) if r[2].startswith("k_table_init"))
seg = [r for r in rows[i0:] if r[2].startswith(("k_part", "k_hot", "k_zero", "k_set", "k_insert", "k_commit", "k_gather", "k_sub"))]
t0 = seg[0][0]
print("kernel                                     start_us    end_us   dur_us")
for s, e, k in seg: print(f"{k:40s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}")
routes = [r for r in seg if r[2].startswith("k_part_route")]; ins = [r for r in seg if r[2].startswith("k_part_insert")]
ov = 0
for rs, re, _ in routes:
    for s, e, _ in ins: ov += max(0, min(re, e) - max(rs, s))
print(f"route total {sum(e - s for s, e, _ in routes) / 1e6:.3f} ms, insert total {sum(e - s for s, e, _ in ins) / 1e6:.3f} ms, overlapped {ov / 1e6:.3f} ms, span {(seg[-1][1] - t0) / 1e6:.3f} ms")
PY
cat $R/gpurun_out/r4_pipe_timeline.txt
