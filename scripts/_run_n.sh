set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 --batch 2 > gpurun_out/r04n_ddp2.log 2> gpurun_out/r04n_ddp2.err; echo "rc=$?"
tail -1 gpurun_out/r04n_ddp2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','n_gpus','ms_per_step','config','multi_gpu')}, indent=1))"
tail -5 gpurun_out/r04n_ddp2.err
timeout -k 10 500 python bench.py --gpus 4 --backend gloo --share-gpu --steps 2 --warmup 1 --batch 1 --grad-reduce fp32 --ddp-overlap layers > gpurun_out/r04n_ddp4.log 2> gpurun_out/r04n_ddp4.err; echo "rc=$?"
tail -1 gpurun_out/r04n_ddp4.log | cut -c1-300
tail -3 gpurun_out/r04n_ddp4.err
