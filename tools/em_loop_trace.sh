# dev: kernel trace of the EM loop of tools/em_probe.py -- per-kernel means and the idle time between kernels, for the library in SFGPU_LIB_PATH
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/elt
rocprofv3 --kernel-trace --output-format csv -d /tmp/elt -- python $GRAFT_REPO_ROOT/tools/em_probe.py > /tmp/elt.out 2>&1
tail -1 /tmp/elt.out
f=$(find /tmp/elt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")[:40]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last 343 VBEM iterations: take the last 600 kernels that are sweep / update / post / copy
sel = [r for r in rows if r[2].startswith(("k_sweep_lds", "k_update", "k_post_state", "__amd_rocclr_copyBuffer"))][-700:-40]
span = (sel[-1][1] - sel[0][0]) / 1e3
busy = collections.Counter(); cnt = collections.Counter()
for s, e, k in sel: busy[k] += (e - s) / 1e3; cnt[k] += 1
print(f"{len(sel)} kernels over {span:.1f} us; busy {sum(busy.values()):.1f} us")
for k in busy: print(f"  {k:42s} n={cnt[k]:4d} mean {busy[k]/cnt[k]:7.2f} us")
gaps = collections.Counter(); gn = collections.Counter()
for a, b in zip(sel, sel[1:]):
    g = (b[0] - a[1]) / 1e3; key = a[2][:14] + " -> " + b[2][:14]; gaps[key] += g; gn[key] += 1
for k in gaps: print(f"  gap {k:34s} n={gn[k]:4d} mean {gaps[k]/gn[k]:7.2f} us  total {gaps[k]:8.1f}")
PY
