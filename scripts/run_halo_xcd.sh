#!/bin/bash
# halo kernel, XCD-aware tile orders on the one-round ResnetBlock grid: time and memory-side fetch (modes 1 = block b -> tile b, 16 = 2 N x 16 M, 17 = 4 N x 8 M per XCD)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r04_halo_xcd_ab.txt
echo "## python scripts/bench_halo.py 1,16,17,1,17 (resblock rows)" > $OUT
timeout -k 10 300 python scripts/bench_halo.py 1,16,17,1,17 2>/dev/null | grep "resblock\|identical" | head -17 >> $OUT
cd /tmp && export TMPDIR=/tmp
for m in 1 16 17; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_x
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_x -f csv -- python3 $R/scripts/bench_halo.py $m > $R/gpurun_out/pmc_x.log 2>&1
    echo "## mode $m: rocprofv3 --pmc $c -- python3 scripts/bench_halo.py $m (per-dispatch average, KB; FETCH_SIZE to be doubled on gfx950)" >> $OUT
    python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_x gemm_halo | grep "131072" >> $OUT
  done
done
rm -rf $R/gpurun_out/pmc_x
cd $R
for m in 1 17 1 17; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --debug-mode $m 2>/dev/null | tail -1 | cut -c1-175 >> $OUT; done
cat $OUT
