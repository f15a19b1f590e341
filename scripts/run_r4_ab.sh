#!/bin/bash
# round-4 A/B evidence in one call: (a) halo kernel 8 waves x 64x64 wave tile vs 4 waves x 128x64 (developer mode 53), with SQ counters
# for both forms; (b) dgrad2_rows_kernel vs its conflict-free timing-only ablation (mode 54), with the LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/r04_halo_wavetile_ab.txt
echo "## python scripts/bench_halo.py 1,53   (mode 1: shipped 8-wave form; 53: gemm_halo4_kernel, plain forward only -- the data gradient rows of mode 53 run the shipped kernel)" > $OUT
timeout -k 10 300 python scripts/bench_halo.py 1,53 2>/dev/null | grep -v "conv1_2\|conv2_1" >> $OUT
OUT2=gpurun_out/r04_dgrad2_rows_conflicts_ab.txt
echo "## python scripts/bench_conv.py --fast 1,54,1,54 --filter 'G down 64->128,convT 128->64'   (54: conflict-free LDS addresses, TIMING ONLY; the dgrad column of the first layer / fwd column of the second are dgrad2_rows_kernel)" > $OUT2
timeout -k 10 300 python scripts/bench_conv.py --fast 1,54,1,54 --filter "G down 64->128,convT 128->64" --iters 30 2>/dev/null | grep -v amdgpu >> $OUT2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_h4_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_h4_$i -f csv -- python3 $R/scripts/bench_halo.py 1,53 > $R/gpurun_out/pmc_h4_$i.log 2>&1
  echo "## rocprofv3 --pmc $grp -- python3 scripts/bench_halo.py 1,53 (per-dispatch averages)" >> $R/$OUT
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_h4_$i gemm_halo | grep "131072\|65536" >> $R/$OUT
  rm -rf $R/gpurun_out/pmc_h4_$i
done
rm -rf $R/gpurun_out/pmc_d2
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_d2 -f csv -- python3 $R/scripts/bench_conv.py --fast 1,54 --filter "G down 64->128" --iters 5 > $R/gpurun_out/pmc_d2.log 2>&1
echo "## rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -- python3 scripts/bench_conv.py --fast 1,54 --filter 'G down 64->128' (per-dispatch averages)" >> $R/$OUT2
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_d2 dgrad2_rows >> $R/$OUT2
rm -rf $R/gpurun_out/pmc_d2
cat $R/$OUT; cat $R/$OUT2
