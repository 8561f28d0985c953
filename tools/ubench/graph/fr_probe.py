import os, pickle, sys, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29555', RANK='0', WORLD_SIZE='1')
if len(sys.argv) > 1: os.environ['TORCH_NCCL_TRACE_BUFFER_SIZE'] = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from torch._C._distributed_c10d import _dump_nccl_trace
t = torch.ones(1 << 26, device='cuda')
for i in range(3):
    w = dist.all_reduce(t, async_op=True)
def counts():
    a = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=True))
    b = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    return len(a.get('entries', [])), len(b.get('entries', [])), list(b.keys())[:8]
print('env', os.environ.get('TORCH_NCCL_TRACE_BUFFER_SIZE'), 'right after issue:', counts())
torch.cuda.synchronize()
print('after sync:', counts())
for _ in range(50):
    time.sleep(0.01)
    c = counts()
    if c[0] == 0: break
print('after polling:', c)
e = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get('entries', [])
if e: print({k: e[-1][k] for k in e[-1] if k in ('state', 'retired', 'time_discovered_completed_ns', 'profiling_name', 'collective_seq_id')})
dist.destroy_process_group()
