# dev: SQ counters of the class-build kernels for the product library (env from the caller), two --pmc passes
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G4="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY"
G5="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for G in "$G1" "$G4" "$G5"; do
  i=$((i+1)); cd /tmp; rm -rf /tmp/eqc$i
  EQ_CFG3=1 rocprofv3 --pmc $G --output-format csv -d /tmp/eqc$i -- python $R/tools/eq_probe.py > /tmp/eqc$i.out 2>&1 || echo "pass $i failed: $(tail -2 /tmp/eqc$i.out)"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for i in range(1, 4):
    fs = glob.glob(f'/tmp/eqc{i}/**/*counter_collection.csv', recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
        if not k.startswith("k_part"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
READS = 3 * 400e6
for k, v in acc.items():
    print(f"== {k}  (per READ, summed over the launches of 3 builds of 400M reads)")
    for c, x in sorted(v.items()):
        print(f"   {c:40s} {x:16.0f} total   {x / READS:10.4f} per read   ({n[k][c]} launches)")
PY
