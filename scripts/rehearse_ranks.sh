#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box: 2 and 3 ranks share the card and exchange gradients over gloo
# (bench.py --share-gpu --backend gloo), both overlap schedules and both wire formats.  Not a performance run -- it checks that
# the launcher, the bucketed all-reduce, the timeline marks and the JSON line hold together.  The 8-GPU RCCL run is the driver's.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "2 d_backward bf16" "2 layers fp32" "3 d_backward bf16"; do
  set -- $cfg
  echo "== ranks $1 overlap $2 wire $3"
  timeout -k 5 240 python bench.py --gpus $1 --backend gloo --share-gpu --steps 2 --warmup 1 --batch 1 --grad-reduce $3 \
      --ddp-overlap $2 > gpurun_out/rehearse_$1_$2.log 2> gpurun_out/rehearse_$1_$2.err || { echo "FAILED rc=$?"; exit 1; }
  tail -1 gpurun_out/rehearse_$1_$2.log | cut -c1-200
done
