set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --netG local 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 300 python scripts/layer_profile.py --netG local > gpurun_out/r04h_layer_profile_local.txt 2>&1
head -60 gpurun_out/r04h_layer_profile_local.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_loc
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_loc -f csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --netG local > $GRAFT_REPO_ROOT/gpurun_out/r04h_loc_bp.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/trace_summary.py gpurun_out/prof_loc 60 30 > gpurun_out/r04h_step_breakdown_local.txt
rm -rf gpurun_out/prof_loc
head -95 gpurun_out/r04h_step_breakdown_local.txt
