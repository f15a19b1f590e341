set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --unfused-norm-sums 2>/dev/null | tail -1 | cut -c1-180
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-180
done > gpurun_out/r04g_nsum_ab.txt
cat gpurun_out/r04g_nsum_ab.txt
bash scripts/quick_prof.sh r04g
head -75 gpurun_out/r04g_step_breakdown.txt
