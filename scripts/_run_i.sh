set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --serial-adam 2>/dev/null | tail -1 | cut -c1-180
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-180
done > gpurun_out/r04i_adam_overlap_ab.txt
cat gpurun_out/r04i_adam_overlap_ab.txt
timeout -k 10 600 python -m pytest tests/test_hip_step.py tests/test_hip_interop.py -x -q -m gpu 2>&1 | tail -3
