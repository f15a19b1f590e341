#!/bin/bash
# VGPR / AGPR / scratch / occupancy of every kernel of the library, from hipcc's kernel-resource-usage remarks (no GPU needed).
# usage: scripts/kernel_resources.sh > profiles/rNN_kernel_resources.txt
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
for f in conv_gemm norm elementwise; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-function -Wno-unused-variable -I$R/include -I$R/jpd-se_amd/csrc \
    --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $R/jpd-se_amd/csrc/$f.hip -o /dev/null > $T/$f.log 2>&1
done
python3 - $T <<'PY'
import re, sys, subprocess, glob
print('%-110s %5s %5s %8s %4s %7s' % ('kernel', 'VGPR', 'AGPR', 'scratch', 'occ', 'LDS'))
for f in sorted(glob.glob(sys.argv[1] + '/*.log')):
  txt = open(f).read()
  for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split()[0]
    try: name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception: pass
    name = name.replace('jpdse::', '').replace('void ', '')
    g = lambda pat: (re.search(pat, b) or [None, '?'])[1]
    print('%-110s %5s %5s %8s %4s %7s' % (name[:110], g(r'VGPRs: (\d+)'), g(r'AGPRs: (\d+)'), g(r'ScratchSize \[bytes/lane\]: (\d+)'),
                                        g(r'Occupancy \[waves/SIMD\]: (\d+)'), g(r'LDS Size \[bytes/block\]: (\d+)')))
PY
rm -rf $T
