#!/bin/bash
# step-level sweep of kernel-selection modes against the default (developer build): bench.py --debug-mode m, default interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r04}_mode_sweep.txt
shift
: > $OUT
for m in 1 "$@" 1; do
  v=$(timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --debug-mode $m 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "global mode $m: $v ms" | tee -a $OUT
done
for m in 1 "$@" 1; do
  v=$(timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --netG local --debug-mode $m 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "local  mode $m: $v ms" | tee -a $OUT
done
