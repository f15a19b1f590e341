#!/bin/bash
# developer-build A/B of kernel-selection modes on the short-K layers: usage ab_stages.sh "1,50,1,50"  (round 3 used it with the since retired modes 44 / 45 / 46)
cd ${GRAFT_REPO_ROOT:-/root/repo}
MODES=${1:-1,50,1,50}
for f in "layer1" "layer2" "G down 128" "convT 256"; do
JPDSE_HIP_DEV=1 timeout -k 10 200 python scripts/bench_conv.py --fast $MODES --filter "$f" --iters 30 2>&1 | grep -v "amdgpu\|^layer"
done
