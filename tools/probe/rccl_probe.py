"""What this RCCL build prints where (one rank on one GPU):  python tools/probe/rccl_probe.py [inproc|shell]
inproc: NCCL_DEBUG* set by os.environ after `import torch`, before init_process_group (what hawkeye_amd.ddp does)."""
import os
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else 'inproc'
import torch
import torch.distributed as dist
log = f'/tmp/hk_rccl_probe_{mode}.log'
if mode == 'inproc':
    os.environ['NCCL_DEBUG'] = 'INFO'
    os.environ['NCCL_DEBUG_SUBSYS'] = 'INIT,GRAPH,TUNING,ENV'
    os.environ['NCCL_DEBUG_FILE'] = log
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.ones(1 << 20, device='cuda')
dist.all_reduce(x)
torch.cuda.synchronize()
print('PY: after first all_reduce', flush=True)
dist.destroy_process_group()
print('PY: after destroy', flush=True)
print('log exists:', os.path.isfile(log), os.path.getsize(log) if os.path.isfile(log) else 0, flush=True)
if os.path.isfile(log):
    print(open(log).read()[:3000])
