import sys, torch
sys.path.insert(0, '.')
from gedepth_amd.mmrt import bricks
from gedepth_amd import kernels as K
dev = torch.device('cuda')
mode = sys.argv[1]
layers = [bricks.DropPath(0.3).to(dev).train() for _ in range(4)]
x = torch.randn(8, 1000, 96, device=dev, requires_grad=True)
def body():
    h = x
    for l in layers:
        h = l.residual(h, h * 0.5)
    h.float().sum().backward()
    return h
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        x.grad = None; body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print('eager ok')
if mode == 'rand_only':
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        u = torch.rand((4, 8), device=dev)
        r = (u + 0.7).floor() / 0.7
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); print('rand_only ok', r.sum().item())
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        x.grad = None
        out = body()
    print('captured')
    for i in range(3):
        g.replay(); torch.cuda.synchronize(); print('replay', i, out.float().abs().sum().item(), bricks._DROP_PATH_BANK.rows.flatten().tolist()[:8])
