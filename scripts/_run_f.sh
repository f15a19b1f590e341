set -o pipefail
cd $GRAFT_REPO_ROOT
JPDSE_PARITY_REPORT=gpurun_out/r04f_parity_report.txt timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r04f_all.log 2>&1; echo "all rc=$?"; tail -22 gpurun_out/r04f_all.log
