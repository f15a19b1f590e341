set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "fwd_dgrad_wgrad" 2>&1 | tail -4
timeout -k 10 600 python -m pytest tests/test_hip_fullsize_windows.py tests/test_hip_configs.py -x -q -m gpu -k "local_trunk or local_enhancer or core_resblock" 2>&1 | tail -4
timeout -k 10 300 python scripts/bench_conv.py --batch 4 --fast 55,1,55,1 --filter "NONE" 2>/dev/null | tail -2
python - <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'jpd-se_amd')
import torch, jpdse_hip
from jpdse_hip import BF16, PAD_REFLECT
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
def timeit(fn, iters=30):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
N, H, W, C = 4, 16, 32, 1024
gfl = 2.0 * N * H * W * C * C * 9 / 1e9
for rep in range(3):
  for m in (55, 1):
    jpdse_hip.set_dev_mode(m)
    layer = HipConv2d(C, C, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=BF16, device=dev)
    x = Act(torch.randn(N, H, W, C, device=dev).bfloat16(), C)
    y, ctx = layer.fwd(x)
    dy = Act(torch.randn_like(y.t.float()).bfloat16(), C)
    t = timeit(lambda: layer.bwd(ctx, dy, False, True))
    print('trunk 1024 @16x32 wgrad  mode %2d: %.4f ms  %5.0f TFLOP/s' % (m, t, gfl / t), flush=True)
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --netG local 2>/dev/null | tail -1 | cut -c1-200
