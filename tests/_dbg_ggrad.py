import os, sys, random
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'obj-gan_amd')]
import torch
import model as M, synth_batch, encoders
from miscc.config import cfg
from miscc import losses as L
from miscc.utils import feat_select
from oracle import ref_harness as rh, torch_model as tm
dev=torch.device('cuda:0')
B=2; cfg.TREE.BRANCH_NUM=3; cfg.TRAIN.BATCH_SIZE=B
def rel(a,b):
    a=a.detach().double().cpu(); b=b.detach().double().cpu(); return float((a-b).norm()/b.norm())
b=synth_batch.make_batch(B, seed=5); bg=synth_batch.to_device(b, dev)
g=torch.Generator().manual_seed(1)
fake=[torch.tanh(torch.randn(B,3,s,s,generator=g)) for s in (64,128,256)]
def sd_of(m): return {k:v.detach().clone() for k,v in m.state_dict().items()}
# PatD256
for name, cls, seed in (("pat256", M.PAT_D_NET256, 3), ("shp256", lambda: M.SHP_D_NET256(80), 4), ("objss", lambda: M.OBJ_SS_D_NET(80), 5), ("objls", lambda: M.OBJ_LS_D_NET(80), 6)):
    net = rh.seeded_state_(cls(), seed); sd = sd_of(net); net.to(dev).train()
    xc = fake[2].clone().requires_grad_(); xg = fake[2].to(dev).requires_grad_()
    if name=="pat256":
        f = tm.pat_d(sd, xc); lc = tm._bce(tm.head(f, sd, "UNCOND_DNET."),1) + 0.1*tm._bce(tm.head(f, sd, "COND_DNET.", b["sent_emb"]),1)
        fg = net(xg); lg = L._bce(net.UNCOND_DNET(fg),1) + 0.1*L._bce(net.COND_DNET(fg, bg["sent_emb"]),1)
    elif name=="shp256":
        lc = tm._bce(tm.head(tm.shp_d(sd, xc, b["hmaps"][2]), sd, "UNCOND_DNET."),1)
        lg = L._bce(net.UNCOND_DNET(net(xg, bg["hmaps"][2])),1)
    else:
        large = name=="objls"; nl = 4 if large else 3
        r = b["fm_rois"] if large else b["rois"][0]; rg = bg["fm_rois"] if large else bg["rois"][0]
        raw = torch.randn(B,10,48,generator=g)
        feats, cls_, btc = tm.feat_select(tm.obj_d(sd, xc, b["hmaps"][2], r, nl), raw, r, b["num_rois"], large)
        cond = torch.cat((b["clabels_emb"][cls_], btc),1)
        lc = tm._bce(tm.head(feats, sd, "COND_DNET.", cond),1) + tm._bce(tm.head(feats, sd, "UNCOND_DNET."),1)
        pooled = net(xg, bg["hmaps"][2], rg, bg["num_rois"])
        fs, cl2, bt2 = feat_select(pooled, raw.to(dev), rg, bg["num_rois"], is_large_scale=large)
        condg = torch.cat((bg["clabels_emb"][cl2.to(dev)], bt2),1)
        lg = L._bce(net.COND_DNET(fs, condg),1) + L._bce(net.UNCOND_DNET(fs),1)
        print(name, "K", len(cls_), "pooled rel", rel(pooled, tm.obj_d(sd, fake[2], b["hmaps"][2], r, nl)))
    lc.backward(); lg.backward(); torch.cuda.synchronize()
    print(name, "loss", lc.item(), lg.item(), "dx rel", rel(xg.grad, xc.grad))
# DAMSM
enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), 9)); rh.seeded_state_(enc.emb_features, 1); rh.seeded_state_(enc.emb_cnn_code, 2); enc.eval()
import copy
encg = copy.deepcopy(enc).to(dev).eval()
xc = fake[2].clone().requires_grad_(); xg = fake[2].to(dev).requires_grad_()
labels=torch.arange(B)
rc, cc = enc(xc); rg_, cg = encg(xg)
print("enc fwd rel", rel(rg_, rc), rel(cg, cc))
w0,w1 = tm.words_loss(rc, b["words_embs"], labels, b["cap_lens"], b["class_ids"]); s0,s1 = tm.sent_loss(cc, b["sent_emb"], labels, b["class_ids"])
(w0+w1+s0+s1).backward()
W0,W1,_,_ = L.words_loss(rg_, bg["words_embs"], labels.to(dev), bg["cap_lens"], bg["class_ids"], B); S0,S1,_ = L.sent_loss(cg, bg["sent_emb"], labels.to(dev), bg["class_ids"], B)
(W0+W1+S0+S1).backward(); torch.cuda.synchronize()
print("damsm loss", (w0+w1).item(), (W0+W1).item(), (s0+s1).item(), (S0+S1).item(), "dx rel", rel(xg.grad, xc.grad))
# grads wrt region features only (isolate my words_loss from the encoder)
rc2 = rc.detach().clone().requires_grad_(); rg2 = rc.detach().to(dev).requires_grad_()
w0,w1 = tm.words_loss(rc2, b["words_embs"], labels, b["cap_lens"], b["class_ids"]); (w0+w1).backward()
W0,W1,_,_ = L.words_loss(rg2, bg["words_embs"], labels.to(dev), bg["cap_lens"], bg["class_ids"], B); (W0+W1).backward()
print("words_loss d(regions) rel", rel(rg2.grad, rc2.grad))
