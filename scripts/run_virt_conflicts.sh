#!/bin/bash
# folded-frame halo kernel after the bank-preserving redirects: parity tests, LDS conflict counters, timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_ops.py tests/test_hip_fullsize_windows.py tests/test_hip_step.py -x -q -m gpu > gpurun_out/r04v_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r04v_tests.log
OUT=$R/gpurun_out/r04_virt_conflicts_ab.txt
echo "## python scripts/bench_halo.py 1,1 (shipped build after the change)" > $OUT
timeout -k 10 300 python scripts/bench_halo.py 1,1 2>/dev/null >> $OUT
cd /tmp && export TMPDIR=/tmp
echo "## rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python3 scripts/bench_halo.py 1 (per-dispatch averages)" >> $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_virt -f csv -- python3 $R/scripts/bench_halo.py 1 > $R/gpurun_out/pmc_virt.log 2>&1
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_virt | grep "gemm_halo\|gemm_taps" >> $OUT
rm -rf $R/gpurun_out/pmc_virt
cd $R
echo "## python bench.py --steps 20 --warmup 5 --no-cpu-baseline (x2), --netG local (x1)" >> $OUT
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-175 >> $OUT; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --netG local 2>/dev/null | tail -1 | cut -c1-175 >> $OUT
cat $OUT
