export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/g5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g5 -- python $GRAFT_REPO_ROOT/tools/cfg5_probe.py > /tmp/g5.out 2>&1
grep -E "gibbs|bootstrap" /tmp/g5.out | cut -c1-200
f=$(find /tmp/g5 -name '*kernel_stats.csv' | head -1)
head -8 $f | cut -c1-160
