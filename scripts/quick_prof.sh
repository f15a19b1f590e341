#!/bin/bash
# quick: bench (no cpu baseline) + kernel trace breakdown into gpurun_out/$1_*
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -f csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bp.log 2> $R/gpurun_out/${TAG}_bp.err
cd $R
python scripts/trace_summary.py gpurun_out/prof_${TAG} 90 16 > gpurun_out/${TAG}_step_breakdown.txt
find gpurun_out/prof_${TAG} -name "*_kernel_trace.csv" -delete
rm -rf gpurun_out/prof_${TAG}
echo quick done
