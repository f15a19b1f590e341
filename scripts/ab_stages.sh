#!/bin/bash
# developer-build A/B of the short-K fast configurations on the layers that use them: usage ab_stages.sh "1,44,45,1,44,45"
cd ${GRAFT_REPO_ROOT:-/root/repo}
MODES=${1:-1,44,45,1,44,45}
for f in "layer1" "layer2" "G down 128" "convT 256"; do
JPDSE_HIP_DEV=1 timeout -k 10 200 python scripts/bench_conv.py --fast $MODES --filter "$f" --iters 30 2>&1 | grep -v "amdgpu\|^layer"
done
