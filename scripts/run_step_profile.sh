#!/bin/bash
# bench + rocprofv3 kernel trace of the bench step -> gpurun_out/<tag>_bench.json.log, <tag>_step_breakdown.txt
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench.json.log 2> $R/gpurun_out/${TAG}_bench.err
tail -1 $R/gpurun_out/${TAG}_bench.json.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -f csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_profiled.json.log 2> $R/gpurun_out/${TAG}_bench_profiled.err
cd $R
python scripts/trace_summary.py gpurun_out/prof_${TAG} 70 16 > gpurun_out/${TAG}_step_breakdown.txt
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
find gpurun_out/prof_${TAG} -name "*_kernel_trace.csv" -delete     # large; the summaries stay
head -45 gpurun_out/${TAG}_step_breakdown.txt
