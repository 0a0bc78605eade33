set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1prof
rm -rf $O; mkdir -p $O
cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.out 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.out 2> $O/write.err
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-600
find $O -name '*.csv' | head; du -sh $O
# keep only what is needed (counter csv can be big)
find $O -name '*agent_info.csv' -delete
