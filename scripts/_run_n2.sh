cd $GRAFT_REPO_ROOT
for cfg in "2 layers fp32" "4 d_backward bf16" "4 layers bf16"; do
  set -- $cfg
  echo "== gpus $1 overlap $2 wire $3" | tee -a gpurun_out/r04n2.log
  timeout -k 5 170 python bench.py --gpus $1 --backend gloo --share-gpu --steps 2 --warmup 1 --batch 1 --grad-reduce $3 --ddp-overlap $2 > gpurun_out/r04n2_$1_$2.log 2> gpurun_out/r04n2_$1_$2.err; echo "rc=$?" | tee -a gpurun_out/r04n2.log
  tail -1 gpurun_out/r04n2_$1_$2.log | cut -c1-160 | tee -a gpurun_out/r04n2.log
done
