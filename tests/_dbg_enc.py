import os, sys, copy
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'obj-gan_amd')]
import torch, encoders
from oracle import ref_harness as rh
dev=torch.device('cuda:0')
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu(); return float((a-b).norm()/b.norm())
g=torch.Generator().manual_seed(1)
x=torch.tanh(torch.randn(2,3,256,256,generator=g))
enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), 9)); rh.seeded_state_(enc.emb_features, 1); rh.seeded_state_(enc.emb_cnn_code, 2); enc.eval()
gr = torch.randn(2,256,17,17,generator=g); gc = torch.randn(2,256,generator=g)
def run(e, xx, gr, gc):
    xx = xx.clone().requires_grad_()
    r,c = e(xx); ((r*gr).sum()+(c*gc).sum()).backward(); return r.detach(), xx.grad
r32, g32 = run(enc, x, gr, gc)
enc64 = copy.deepcopy(enc).double()
r64, g64 = run(enc64, x.double(), gr.double(), gc.double())
encg = copy.deepcopy(enc).to(dev)
rg, gg = run(encg, x.to(dev), gr.to(dev), gc.to(dev))
with torch.backends.cudnn.flags(enabled=False):
    rg2, gg2 = run(encg, x.to(dev), gr.to(dev), gc.to(dev))
print("fwd  cpu32 vs cpu64", rel(r32,r64), " gpu32 vs cpu64", rel(rg,r64), " gpu32(no miopen) vs cpu64", rel(rg2,r64))
print("grad cpu32 vs cpu64", rel(g32,g64), " gpu32 vs cpu64", rel(gg,g64), " gpu32(no miopen) vs cpu64", rel(gg2,g64), " gpu vs cpu32", rel(gg,g32))
