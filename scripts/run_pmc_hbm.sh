#!/bin/bash
# HBM traffic of the bench step, one counter per pass (FETCH_SIZE + WRITE_SIZE together exceed what one pass can collect)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_hbm_$c -f csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_hbm_$c.log 2>&1
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_hbm_$c > $R/gpurun_out/pmc_hbm_$c.txt
done
