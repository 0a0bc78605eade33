#!/bin/bash
# round 6: kernel timeline of one bench step (the last one): name, start offset, duration, gap to the previous kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_step_trace
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6_st -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-pinned --no-sampling > $R/gpurun_out/r6_step_trace/run.log 2>&1
f=$(find /tmp/r6_st -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/r6_step_trace/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
# the persistent launches of the timed steps: the VBEM ones; take the last-but-extras: find k_part_route sequences
pers = [i for i, k in enumerate(ks) if "k_em_persist<true>" in k[2] or "k_em_persistILb1" in k[2]]
print("persistent launches:", len(pers))
# a step = from the first k_sub_batch_begin / k_part_route after the previous step's k_tpm to its own k_tpm; use the persist index
def step_of(pi):
    j = pi
    while j > 0 and "k_part_route" not in ks[j][2]: j -= 1          # last route before the EM
    # back to the first kernel of this step: walk back over the class build until a gap > 2 ms or a previous persist
    i = j
    while i > 0 and not ("k_em_persist" in ks[i - 1][2]) and ks[i][0] - ks[i - 1][1] < 3_000_000: i -= 1
    e = pi
    while e + 1 < len(ks) and ks[e + 1][0] - ks[e][1] < 300_000 and "k_part_route" not in ks[e + 1][2]: e += 1
    return i, e
for pi in pers[2:5][-1:]:
    i, e = step_of(pi)
    t0 = ks[i][0]
    print(f"step: {e - i + 1} kernels, {(ks[e][1] - t0) / 1e6:.3f} ms")
    prev_end = t0; busy = 0
    for (s, en, n, q) in ks[i:e + 1]:
        busy += en - s
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(en - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q}  {n[:100]}")
        prev_end = max(prev_end, en)
    print(f"busy {busy / 1e6:.3f} ms")
PY
