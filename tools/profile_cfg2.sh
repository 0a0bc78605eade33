# the three passes of tools/profile_round.sh for --workload cfg2 (kernel stats, FETCH_SIZE, WRITE_SIZE); outputs in gpurun_out/<tag>cfg2prof
set -x
TAG=${1:-r3}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}cfg2prof
rm -rf $O; mkdir -p $O
cd /tmp
CMD="python $R/bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-host-pinned --no-sampling"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.out 2> $O/write.err
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
