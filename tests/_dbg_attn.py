import sys; sys.path[:0]=['/root/repo','/root/repo/obj-gan_amd']
import torch
from objgan_hip import ops
from oracle import torch_ref as tr
dev='cuda:0'
g = torch.Generator().manual_seed(3)
B,ih,iw,L=4,8,8,12
x = torch.randn(B,48,ih,iw,generator=g); src = torch.randn(B,48,L,generator=g)*0.3
lens = torch.randint(1,L+1,(B,),generator=g); lens[0]=L
mask = torch.arange(L).unsqueeze(0) >= lens.unsqueeze(1)
for m in (None, mask):
    wcr, atr = tr.attn_general(x, src, m)
    wcd, atd = ops.attn_general(x.to(dev), src.to(dev), None if m is None else m.to(dev))
    print("mask" if m is not None else "nomask", (wcd.cpu()-wcr).norm()/wcr.norm(), (atd.cpu()-atr).norm()/atr.norm())
    d = (atd.cpu()-atr).abs().reshape(B,L,-1)
    print(d.amax(dim=(1,)).reshape(B,-1)[:, :8])
print(lens)
