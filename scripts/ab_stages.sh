cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in "layer1" "layer2" "G down 128" "G down 256" "G down 512" "convT 1024" "convT 512" "convT 256"; do
JPDSE_HIP_DEV=1 timeout -k 10 200 python scripts/bench_conv.py --fast 1,44,45,1,44,45 --filter "$f" --iters 30 2>&1 | grep -v "amdgpu\|^layer"
done
