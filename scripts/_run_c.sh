set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "nsums or norm_backward_sums or fwd_dgrad_wgrad or instance_norm or pack" > gpurun_out/r04c_t1.log 2>&1; echo "t1 rc=$?"
timeout -k 10 600 python -m pytest tests/test_hip_step.py tests/test_hip_networks.py tests/test_hip_interop.py -x -q -m gpu > gpurun_out/r04c_t2.log 2>&1; echo "t2 rc=$?"
timeout -k 10 600 python -m pytest tests/test_hip_configs.py -x -q -m gpu -k "config2" > gpurun_out/r04c_t3.log 2>&1; echo "t3 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04c_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r04c_bench.log | cut -c1-250
bash scripts/run_fp32_cfg2.sh r04c
