cd $GRAFT_REPO_ROOT
export JPDSE_BENCH_VERBOSE=1
timeout -k 5 200 python bench.py --gpus 3 --backend gloo --share-gpu --steps 1 --warmup 1 --batch 1 --grad-reduce bf16 > gpurun_out/r04n3.log 2> gpurun_out/r04n3.err; echo "rc=$?"
grep "bench rank\|Error\|error" gpurun_out/r04n3.err | tail -20
tail -1 gpurun_out/r04n3.log | cut -c1-160
sleep 5; ps aux | grep -c "bench.py" 
