# dev: hardware counters of the class-build kernels (cfg3 class build via tools/eq_probe.py), one rocprofv3 --pmc pass per group.
#   bash tools/eq_counters.sh [variant]      (variant = name of sailfish_amd/csrc/variants/libsfgpu_<name>.so, default: the product library)
export TMPDIR=/tmp
if [ -n "$1" ] && [ "$1" != main ]; then export SFGPU_LIB_PATH=$GRAFT_REPO_ROOT/sailfish_amd/csrc/variants/libsfgpu_$1.so; fi
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G4="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY"
G5="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
G6="FETCH_SIZE"
G7="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1)); cd /tmp; rm -rf /tmp/eqc$i
  EQ_CFG3=1 rocprofv3 --pmc $G --output-format csv -d /tmp/eqc$i -- python $GRAFT_REPO_ROOT/tools/eq_probe.py > /tmp/eqc$i.out 2>&1 || echo "pass $i failed: $(tail -2 /tmp/eqc$i.out)"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for i in range(1, 8):
    fs = glob.glob(f'/tmp/eqc{i}/**/*counter_collection.csv', recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
        if not k.startswith("k_part"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
READS = 3 * 400e6            # three builds of 400 M reads per run
for k, v in acc.items():
    print(f"== {k}  (per READ, summed over the launches of 3 builds of 400M reads)")
    for c, x in sorted(v.items()):
        print(f"   {c:40s} {x:16.0f} total   {x / READS:10.4f} per read   ({n[k][c]} launches)")
PY
