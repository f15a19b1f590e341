cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_fullsize_windows.py -x -q -m gpu -k "fwd_dgrad_wgrad or local_head" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --netG local --debug-mode 57 2>/dev/null | tail -1 | cut -c1-170
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --netG local --debug-mode 1 2>/dev/null | tail -1 | cut -c1-170
timeout -k 10 300 python scripts/layer_profile.py --netG local 2>/dev/null | grep " 32     3  7"
