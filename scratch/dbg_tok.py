import torch, sys
sys.path.insert(0, '/root/repo')
from gedepth_amd.kernels import concat_tokens_map, tokens_from_map
dev = torch.device('cuda')
B, C, H, W = 2, 16, 4, 6
fmap = torch.randn(B, C, H, W); pos = torch.randn(1, C, H, W); other = torch.randn(B, 5, H, W); go = torch.randn(B, C + 5, H, W)
fg, og = fmap.to(dev).requires_grad_(True), other.to(dev).requires_grad_(True)
print('leaf', fg.is_leaf, fg.grad_fn)
tok = tokens_from_map(fg, pos.to(dev))
print('tok', tok.grad_fn, 'fg leaf', fg.is_leaf)
out = concat_tokens_map(tok * 0.5, og, identity=fg, tokens_first=True)
print('out', out.grad_fn, 'fg leaf', fg.is_leaf, fg.grad_fn)
out.backward(go.to(dev))
print(fg.grad is None, og.grad is None)
