set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_configs.py -x -q -m gpu -k "config2 or 1024x512_train_step" > gpurun_out/r04b_t1.log 2>&1; echo "t1 rc=$?" 
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "fwd_dgrad_wgrad or fused_relu or fused_lrelu" > gpurun_out/r04b_t2.log 2>&1; echo "t2 rc=$?"
timeout -k 10 900 python -m pytest tests/test_hip_fullsize_windows.py -x -q -m gpu --durations=15 > gpurun_out/r04b_t3.log 2>&1; echo "t3 rc=$?"
timeout -k 10 600 python -m pytest tests/test_hip_ddp.py -x -q -m gpu > gpurun_out/r04b_t4.log 2>&1; echo "t4 rc=$?"
bash scripts/run_fp32_cfg2.sh r04b
