#!/bin/bash
# round evidence in one call: default bench line (with cpu_baseline), the same command under rocprofv3 --kernel-trace --stats,
# step breakdown, per-layer profile, HBM traffic (two PMC passes).  Outputs under gpurun_out/<tag>_*.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -f csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_profiled.json.log 2> $R/gpurun_out/${TAG}_bench_profiled.err
cd $R
python scripts/trace_summary.py gpurun_out/prof_${TAG} 70 16 > gpurun_out/${TAG}_step_breakdown.txt
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
find gpurun_out/prof_${TAG} -name "*_kernel_trace.csv" -delete
head -3 gpurun_out/${TAG}_step_breakdown.txt
python scripts/layer_profile.py > gpurun_out/${TAG}_layer_profile.txt 2>&1
sed -n 3,4p gpurun_out/${TAG}_layer_profile.txt
bash scripts/run_pmc_hbm.sh
echo evidence done
