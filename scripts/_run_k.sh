set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_fullsize_windows.py tests/test_hip_configs.py -x -q -m gpu -k "g_down_64_128 or convT_128_64 or convT_up or down_s2 or g_up_convT_as_conv or row_streaming or fwd_dgrad_wgrad or fused_moments" 2>&1 | tail -3
timeout -k 10 300 python scripts/bench_conv.py --fast 1,54,1,54 --filter "G down 64->128,convT 128->64" --iters 30 2>/dev/null | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_d2
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_d2 -f csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_conv.py --fast 1,54 --filter "G down 64->128" --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_d2.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_d2 dgrad2_rows
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_d2
