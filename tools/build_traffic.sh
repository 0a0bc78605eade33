export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/scw
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/scw -- python $GRAFT_REPO_ROOT/tools/eq_probe.py > /tmp/scw.out 2>&1
tail -1 /tmp/scw.out | cut -c1-110
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/scw/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_part' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE': acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k, v in acc.items(): print(k, 'WRITE KiB per launch (last 2):', [round(x) for x in v[-2:]])
PY
rm -rf /tmp/scf
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/scf -- python $GRAFT_REPO_ROOT/tools/eq_probe.py > /tmp/scf.out 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/scf/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_part' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k, v in acc.items(): print(k, 'FETCH KiB per launch (last 2):', [round(x) for x in v[-2:]])
PY
