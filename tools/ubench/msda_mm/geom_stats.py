import sys, torch, math
sys.path.insert(0,'/root/repo')
from gedepth_amd.depth.utils.position_encoding import SinePositionalEncoding
from gedepth_amd.mmrt.bricks import msda_offset_bias
from gedepth_amd import kernels as K
KS=((88,280),(44,140),(22,70),(11,35))
torch.manual_seed(1234)
g=torch.Generator().manual_seed(1)
mode=sys.argv[1] if len(sys.argv)>1 else 'cross'
if mode=='cross':
    nq=176*560
    pe=SinePositionalEncoding(num_feats=256, normalize=False)
    pos=pe.grid(176,560,'cpu')
    lin=torch.nn.Linear(512,2); torch.nn.init.xavier_uniform_(lin.weight); torch.nn.init.constant_(lin.bias,0.)
    ref0=torch.sigmoid(lin(pos.flatten(2)[0].t())).detach()
    order=K.msda_ref_order(ref0,KS[0]).long()
else:
    refs=[]
    for h,w in KS:
        gy,gx=torch.meshgrid((torch.arange(h)+0.5)/h,(torch.arange(w)+0.5)/w,indexing='ij')
        refs.append(torch.stack((gx.reshape(-1),gy.reshape(-1)),-1))
    ref0=torch.cat(refs); nq=ref0.shape[0]
    order=K.msda_tile_order(KS,'cpu').long()
off=(msda_offset_bias(8,4,8)[None].expand(nq,512)+0.05*torch.randn(nq,512,generator=g)).bfloat16().float().view(nq,8,4,8,2)
ref=ref0[order]; off=off[order]
ntiles=(nq+31)//32
pad=ntiles*32-nq
for lvl,(H,W) in enumerate(KS):
    loc=ref[:,None,None,:]+off[:,:,lvl]/torch.tensor([W,H],dtype=torch.float32)
    x=loc[...,0]*W-0.5; y=loc[...,1]*H-0.5
    inm=(x>-1)&(y>-1)&(x<W)&(y<H)
    x0=torch.floor(x).clamp(0,W-1); x1=(torch.floor(x)+1).clamp(0,W-1); y0=torch.floor(y).clamp(0,H-1); y1=(torch.floor(y)+1).clamp(0,H-1)
    big=1e6
    def tilered(t,fn,fill):
        t=torch.where(inm,t,torch.full_like(t,fill))
        t=torch.cat([t,torch.full((pad,8,8),fill)]) if pad else t
        return fn(t.view(ntiles,32,8,8).permute(0,2,1,3).reshape(ntiles,8,-1),-1)
    bx0=tilered(x0,lambda a,d:a.min(d).values,big); bx1=tilered(x1,lambda a,d:a.max(d).values,-big)
    by0=tilered(y0,lambda a,d:a.min(d).values,big); by1=tilered(y1,lambda a,d:a.max(d).values,-big)
    some=bx1>=bx0
    bw=(bx1-bx0+1)[some]; bh=(by1-by0+1)[some]
    print(f'level {lvl} {H}x{W}: tiles x heads with taps {some.float().mean():.3f}  box w mean {bw.mean():.1f} p90 {bw.quantile(0.9):.0f} max {bw.max():.0f}; h mean {bh.mean():.1f} p90 {bh.quantile(0.9):.0f} max {bh.max():.0f}; area mean {(bw*bh).mean():.1f}')
    for (sw,sh) in ((24,16),(32,12),(16,24),(32,24),(48,16),(16,8),(12,8)):
        nx=(torch.floor(bx1/sw)-torch.floor(bx0/sw)+1)[some]; ny=(torch.floor(by1/sh)-torch.floor(by0/sh)+1)[some]
        ov=(nx*ny)
        nblk=math.ceil(W/sw)*math.ceil(H/sh)
        print(f'    superblock {sw}x{sh} ({nblk} per seg): overlaps/tile mean {ov.mean():.2f} p99 {ov.quantile(0.99):.0f} max {ov.max():.0f}; visits per (img) {ov.sum():.0f}')
