set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "persistent" > gpurun_out/r04e_t1.log 2>&1; echo "t1 rc=$?"; tail -3 gpurun_out/r04e_t1.log
timeout -k 10 600 python -m pytest tests/test_hip_fullsize_windows.py -x -q -m gpu -k "d_layer1 or d_layer2 or g_down or convT" > gpurun_out/r04e_t2.log 2>&1; echo "t2 rc=$?"; tail -3 gpurun_out/r04e_t2.log
timeout -k 10 600 python scripts/bench_conv.py --fast 50,1,50,1 --filter "D layer1,D layer2,D1 layer1,D1 layer2,G down 128,G down 256,G down 512,convT 512,convT 256" > gpurun_out/r04e_pers_ab.txt 2>&1; echo "ab rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --debug-mode 50 > gpurun_out/r04e_bench_m50.log 2>&1; tail -1 gpurun_out/r04e_bench_m50.log | cut -c1-200
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --debug-mode 1 > gpurun_out/r04e_bench_m1.log 2>&1; tail -1 gpurun_out/r04e_bench_m1.log | cut -c1-200
