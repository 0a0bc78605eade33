# One profile session of the default bench command on the GPU box; everything lands in gpurun_out/<tag>prof and is
# turned into the committed files under profiles/ by tools/make_profiles.py (run here, in the build container).
#   bash tools/profile_round.sh r2
set -x
TAG=${1:-r2}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}prof
rm -rf $O; mkdir -p $O
cd /tmp
# (the sampling legs run concurrent EM lanes and a second of Gibbs kernels: they get a kernel-stats pass of their own below, so that
#  the averages of the step's kernels are those of the timed region)
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-pinned --no-sampling"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.out 2> $O/write.err
# SQ wave-cycle breakdown (one pass, 8 SQ counters) of the same command
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $CMD > $O/sq.out 2> $O/sq.err
rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -- $CMD > $O/ta.out 2> $O/ta.err
python - "$O" <<'PY' > $O/sq_summary.txt
import csv, glob, sys, collections
O = sys.argv[1]
def load(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    fs = glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs: return acc, n
    first = None
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("sfgpu::", "").replace("void ", "")
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        first = first or r["Counter_Name"]
        if r["Counter_Name"] == first: n[k] += 1
    return acc, n
acc, n = load("sq")
names = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]
print("SQ counters per kernel (share of SQ_WAVE_CYCLES, summed over waves; rocprofv3 --pmc, one pass)")
print("%-24s %6s %14s " % ("kernel", "calls", "wave_cyc/call") + " ".join("%10s" % x.replace("SQ_", "")[:10] for x in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = v["SQ_WAVE_CYCLES"]
    if wc <= 0: continue
    print("%-24s %6d %14.0f " % (k[:24], n[k], wc / max(n[k], 1)) + " ".join("%9.1f%%" % (100 * v[x] / wc) for x in names))
acc, n = load("ta")
print()
print("texture-addresser busy cycles and L1->L2 requests per launch (sum over CUs)")
print("%-24s %6s %16s %16s %16s %16s" % ("kernel", "calls", "TA_BUSY/call", "TCC_READ_REQ/call", "TCC_WRITE_REQ/call", "GUI_ACTIVE/call"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["TA_TA_BUSY_sum"]):
    c = max(n[k], 1)
    print("%-24s %6d %16.0f %16.0f %16.0f %16.0f" % (k[:24], n[k], v["TA_TA_BUSY_sum"] / c, v["TCP_TCC_READ_REQ_sum"] / c, v["TCP_TCC_WRITE_REQ_sum"] / c, v["GRBM_GUI_ACTIVE"] / c))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sampling -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-pinned --no-compare-em-modes > $O/sampling.out 2> $O/sampling.err
find $O/sampling -name '*kernel_trace.csv' -delete
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-600
# keep only what is needed (counter csv can be big)
find $O -name '*agent_info.csv' -delete
find $O/sq $O/ta -name '*.csv' -delete
du -sh $O
