#!/bin/bash
# round-4 evidence, part 2: the other BASELINE.json configurations, SQ counters per kernel, the full GPU test suite with its parity report
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/run_configs.sh $TAG > /dev/null 2>&1
echo configs done; cut -c1-160 gpurun_out/${TAG}_configs.txt
bash scripts/run_pmc_sq.sh && cat gpurun_out/pmc_sq_1.txt gpurun_out/pmc_sq_2.txt gpurun_out/pmc_sq_3.txt > gpurun_out/${TAG}_pmc_kernels.txt
rm -rf gpurun_out/pmc_sq_1 gpurun_out/pmc_sq_2 gpurun_out/pmc_sq_3
echo pmc done
JPDSE_PARITY_REPORT=$R/gpurun_out/${TAG}_parity_report.txt timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/${TAG}_gpu_tests.log
