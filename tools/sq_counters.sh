# dev: SQ wave-cycle breakdown of the hot kernels (one rocprofv3 --pmc pass; counts are quad-cycles summed over waves)
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/sq -- python $GRAFT_REPO_ROOT/tools/step_probe.py > /tmp/sq.out 2>&1
tail -1 /tmp/sq.out | cut -c1-160
f=$(find /tmp/sq -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
names = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]
print("%-22s %6s %12s " % ("kernel", "calls", "wave_cyc/call") + " ".join("%9s" % x.replace("SQ_", "")[:9] for x in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = v["SQ_WAVE_CYCLES"]
    if wc <= 0: continue
    print("%-22s %6d %12.0f " % (k[:22], n[k], wc / max(n[k], 1)) + " ".join("%8.1f%%" % (100 * v[x] / wc) for x in names))
PY
