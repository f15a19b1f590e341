#!/bin/bash
# the other BASELINE.json configurations on one GPU -> gpurun_out/<tag>_configs.txt (one JSON line each, value / ms_per_step cut out)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_configs.txt
: > $OUT
run() { echo "## bench.py $*" >> $OUT; python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('metric','value','ms_per_step','dtype','config')}))" >> $OUT; }
run --netG local
run --netG local --width 2048 --height 1024 --batch 2
run --width 2048 --height 1024 --batch 2
run --width 2048 --height 1024 --batch 1
run --dtype fp32
run --dtype fp32 --width 512 --height 256 --no-vgg --batch 1
run --dtype fp32 --width 512 --height 256 --no-vgg --batch 16
cat $OUT
