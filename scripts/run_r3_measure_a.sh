#!/bin/bash
# round 3, measurement set A: read-back / timer A/B of the bench step, CU-contention rehearsal, cosine-by-mode diagnostic,
# RCCL world-size-1 kernel trace.  Everything lands in gpurun_out/r3a_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" "--late-readback" "--no-kernel-timers" "--late-readback --no-kernel-timers"; do
  tag=$(echo "base$v" | tr -d ' -')
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('achieved'), json.dumps(d.get('roofline_hbm')))" >> gpurun_out/r3a_bench_ab.txt
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench_full.json.log 2> gpurun_out/r3a_bench_full.err
python scripts/cu_contention.py --ks 0,8,16,32 --steps 10 > gpurun_out/r3a_cu_contention.txt 2> gpurun_out/r3a_cu_contention.err
python scripts/diag_cos_modes.py 1 29 32 > gpurun_out/r3a_cos_modes.txt 2> gpurun_out/r3a_cos_modes.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_rccl
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rccl -f csv -- python3 $R/scripts/diag_rccl_ws1.py > $R/gpurun_out/r3a_rccl_ws1.log 2>&1
cd $R
find gpurun_out/prof_rccl -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3a_rccl_ws1_kernel_stats.csv \;
find gpurun_out/prof_rccl -name "*_kernel_trace.csv" -delete
rm -rf gpurun_out/prof_rccl
echo done
