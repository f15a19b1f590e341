#!/bin/bash
# round-4 evidence, part 1: default bench line (with cpu_baseline), the same command under rocprofv3 --kernel-trace --stats, step
# breakdown, per-layer profiles (global bf16, LocalEnhancer, config 2 fp32), HBM traffic (two PMC passes).  Outputs: gpurun_out/<tag>_*
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -f csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_profiled.json.log 2> $R/gpurun_out/${TAG}_bench_profiled.err
cd $R
python scripts/trace_summary.py gpurun_out/prof_${TAG} 80 20 > gpurun_out/${TAG}_step_breakdown.txt
find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/prof_${TAG}
head -3 gpurun_out/${TAG}_step_breakdown.txt
python scripts/layer_profile.py > gpurun_out/${TAG}_layer_profile.txt 2>&1
sed -n 3,4p gpurun_out/${TAG}_layer_profile.txt
python scripts/layer_profile.py --netG local > gpurun_out/${TAG}_layer_profile_local.txt 2>&1
sed -n 3,4p gpurun_out/${TAG}_layer_profile_local.txt
python scripts/layer_profile.py --dtype fp32 --width 512 --height 256 --no-vgg --batch 1 > gpurun_out/${TAG}_layer_profile_fp32.txt 2>&1
sed -n 3,4p gpurun_out/${TAG}_layer_profile_fp32.txt
bash scripts/run_pmc_hbm.sh
python scripts/make_hbm_table.py ${TAG} > /dev/null
mv profiles/${TAG}_hbm_traffic.txt gpurun_out/${TAG}_hbm_traffic.txt
cp profiles/hbm_traffic.json gpurun_out/${TAG}_hbm_traffic.json
rm -rf gpurun_out/pmc_hbm_FETCH_SIZE gpurun_out/pmc_hbm_WRITE_SIZE
echo evidence part 1 done
