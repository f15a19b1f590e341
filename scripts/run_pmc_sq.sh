#!/bin/bash
# SQ counters of the bench step (one small group per pass), summarised per kernel by scripts/pmc_summary.py
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_sq_$i -f csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq_$i.log 2>&1
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_sq_$i > $R/gpurun_out/pmc_sq_$i.txt
done
