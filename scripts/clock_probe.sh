#!/bin/bash
# Shader clock and package power while (a) the GPU idles, (b) the ResnetBlock halo GEMM runs back to back, (c) its MFMA-only / no-DMA
# ablations run, (d) the bench step runs: is the MFMA loop power-capped (clock below the 2.4 GHz the 2.5 PFLOP/s peak assumes)?
# Output: gpurun_out/<tag>_clock_probe.txt.   Read-only rocm-smi queries; changes no GPU setting.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/${TAG}_clock_probe.txt
: > $OUT
sampler() {  # label: one line per ~0.7 s until killed
  while true; do
    s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';')
    echo "$1 $(date +%s.%N | cut -c1-14) $s" >> $OUT
    sleep 0.5
  done
}
echo "## idle" >> $OUT
sampler idle & S=$!; sleep 2; kill $S; wait $S 2>/dev/null
for mode in 1 115 101; do
  echo "## bench_conv.py --fast $mode --fwd-only (ResBlock 1024 forward back to back; 115 = MFMA-only loop, 101 = no DMA after the prologue)" >> $OUT
  sampler mode$mode & S=$!
  timeout -k 5 150 python scripts/bench_conv.py --fast $mode --filter "ResBlock 1024" --iters 40000 --fwd-only > gpurun_out/${TAG}_clock_probe_$mode.log 2>&1
  kill $S; wait $S 2>/dev/null
  tail -1 gpurun_out/${TAG}_clock_probe_$mode.log >> $OUT
done
echo "## bench.py --steps 400 (headline step)" >> $OUT
sampler step & S=$!
timeout -k 5 200 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_clock_probe_bench.log 2>&1
kill $S; wait $S 2>/dev/null
tail -1 gpurun_out/${TAG}_clock_probe_bench.log | cut -c1-200 >> $OUT
wc -l $OUT
