cd ${GRAFT_REPO_ROOT:-/root/repo}
true
for i in 1 2; do
for lib in "" "$PWD/jpd-se_amd/jpdse_hip/libjpdse_hip_nont.so"; do
JPDSE_HIP_LIB=$lib timeout -k 10 150 python bench.py --steps 20 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib'[-12:], d['ms_per_step'], d['roofline_hbm']['adam']['ms_per_step'], 'norm fwd', d['roofline_hbm']['forward']['ms_per_step'], 'bwd', d['roofline_hbm']['backward']['ms_per_step'])"
done; done
