# dev: sclk / power of the GPU while the bench step runs back to back (is the step power-limited?)
python bench.py --steps 1200 --warmup 2 --no-cpu-baseline --no-sampling --no-host-pinned --no-compare-em-modes > /tmp/cp.out 2>&1 &
BP=$!
sleep 16
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' ' | cut -c1-300; echo
  sleep 0.5
done
wait $BP
tail -1 /tmp/cp.out | cut -c1-200
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr '\n' ' ' | cut -c1-300; echo
