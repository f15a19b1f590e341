#!/bin/bash
# same-box A/B of library builds: bench.py with the default library and with each JPDSE_HIP_LIB given, interleaved twice
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
for lib in "" "$@"; do
full=""; [ -n "$lib" ] && full="$PWD/jpd-se_amd/jpdse_hip/$lib"
JPDSE_HIP_LIB=$full timeout -k 10 150 python bench.py --steps 20 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-24s' % ('$lib' or 'default'), d['ms_per_step'], 'adam', d['roofline_hbm']['adam']['ms_per_step'], 'norm fwd', d['roofline_hbm']['forward']['ms_per_step'], 'bwd', d['roofline_hbm']['backward']['ms_per_step'])"
done; done
