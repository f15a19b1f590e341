"""Does a world-size-1 `nccl` (RCCL) all-reduce launch a kernel at all?    Run under
  rocprofv3 --kernel-trace --stats -- python3 scripts/diag_rccl_ws1.py
and look for ncclDevKernel* in the kernel statistics: two data-parallel train steps (ngf 16, 64x128) with every gradient
bucket going through dist.all_reduce on a one-rank process group, plus one bare 64 MB all-reduce."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29611')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from ctu.trainers import get_trainer
from ctu.utils.synthetic import synthetic_batch, default_opt
opt = default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16', use_compressed=True, netG='global', ngf=16, ndf=16,
                  batch_size=2)
torch.manual_seed(3)
tr = get_trainer(opt)(opt, 'train')
tr.enable_data_parallel(bucket_bytes=1 << 20)
xd = synthetic_batch(2, 64, 128, seed=5)
xd = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in xd.items()}
for _ in range(2):
  tr.step(xd)
big = torch.ones(16 << 20, device='cuda')
h = dist.all_reduce(big, async_op=True)
h.wait()
torch.cuda.synchronize()
print('buckets G %d, D %d; bare all-reduce ok: %s' % (len(tr.model.grad_buckets['G'].buckets), len(tr.model.grad_buckets['D'].buckets),
                                                        bool((big == 1).all())))
dist.destroy_process_group()
