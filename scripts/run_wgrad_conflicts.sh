#!/bin/bash
# weight-gradient kernels after the LDS layout changes: parity tests, per-layer timing, LDS conflict counters
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04w}
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_ops.py tests/test_hip_fullsize_windows.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/${TAG}_tests.log
OUT=$R/gpurun_out/${TAG}_wgrad_conflicts.txt
F='G first,G down,convT,D layer0,D layer1,D layer2,D1 layer1,D1 layer2,G last'
echo "## python scripts/bench_conv.py --fast 58,1,58,1 --iters 20 --filter '$F'" > $OUT
timeout -k 10 300 python scripts/bench_conv.py --fast 58,1,58,1 --iters 20 --filter "$F" 2>/dev/null >> $OUT
cd /tmp && export TMPDIR=/tmp
echo "## rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python3 scripts/bench_conv.py --fast 1 --iters 3 (same filter; per-dispatch averages, wgrad kernels)" >> $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_w -f csv -- python3 $R/scripts/bench_conv.py --fast 1 --iters 3 --filter "$F" > $R/gpurun_out/pmc_w.log 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_w | grep "wgrad_" >> $OUT
rm -rf $R/gpurun_out/pmc_w
cd $R
echo "## python bench.py --steps 20 --warmup 5 --no-cpu-baseline (x2)" >> $OUT
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-175 >> $OUT; done
cat $OUT
