set -e
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json.log 2> gpurun_out/bench_final.err
tail -1 gpurun_out/bench_final.json.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -f csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/bench_final_profiled.json.log 2> $R/gpurun_out/bench_final_profiled.err
cd $R
python scripts/trace_summary.py gpurun_out/prof_final 60 12 > gpurun_out/step_breakdown_final.txt
tail -1 gpurun_out/bench_final_profiled.json.log | cut -c1-200
