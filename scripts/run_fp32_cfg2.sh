#!/bin/bash
# BASELINE config 2 (512x256, GlobalGenerator + 2-scale PatchGAN, fp32, no VGG, batch 1): bench line, per-layer profile and
# kernel-trace breakdown -> gpurun_out/<tag>_cfg2_*
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS="--dtype fp32 --width 512 --height 256 --no-vgg --batch 1"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS > gpurun_out/${TAG}_cfg2_bench.log 2>&1
tail -1 gpurun_out/${TAG}_cfg2_bench.log | cut -c1-300
timeout -k 10 200 python scripts/layer_profile.py $ARGS > gpurun_out/${TAG}_cfg2_layer_profile.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}c2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}c2 -f csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline $ARGS > $R/gpurun_out/${TAG}_cfg2_bp.log 2> $R/gpurun_out/${TAG}_cfg2_bp.err
cd $R
python scripts/trace_summary.py gpurun_out/prof_${TAG}c2 60 40 > gpurun_out/${TAG}_cfg2_step_breakdown.txt
rm -rf gpurun_out/prof_${TAG}c2
echo cfg2 done
