"""Generate tests/golden/step_b2_grads.pt from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_grads.py

Sub-sampled gradient TENSORS of every discriminator parameter (reference miscc/losses.py:163-330 patD_loss /
shpD_loss / objD_loss, each followed by backward()) and of every generator parameter (G_loss + KL_loss with a constant
image encoder, see make_golden.py) for the seeded B = 2 batch and the seeded weights of make_golden.py.  A tensor with
more than 2048 elements is stored as flat[::stride] (stride = ceil(numel / 2048)); the norms of the full tensors are in
step_b2.pt already.  tests/test_modules_gpu.py compares these samples element-wise (rel-L2 <= 1e-3 per network).
"""
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd"), HERE]

from oracle import ref_harness as rh            # noqa: E402
import synth_batch                              # noqa: E402
from oracle import torch_encoders as encoders   # noqa: E402
from make_golden import B, SEEDS                # noqa: E402

CAP = 2048


def grad_samples(module):
    out = {}
    for k, p in module.named_parameters():
        if p.grad is None:
            continue
        flat = p.grad.detach().reshape(-1)
        stride = max(1, -(-flat.numel() // CAP))
        out[k] = flat[::stride].clone()
    return out


def main():
    torch.set_num_threads(8)
    ref = rh.load_reference(branch_num=3, batch_size=B)
    b = synth_batch.make_batch(B, seed=SEEDS["batch"])
    M, Ls, U = ref.model, ref.losses, ref.utils
    G = rh.seeded_state_(M.G_NET(80), SEEDS["G"]).train()
    pats = [rh.seeded_state_(c(), SEEDS["pat"] + i).train() for i, c in enumerate((M.PAT_D_NET64, M.PAT_D_NET128, M.PAT_D_NET256))]
    shps = [rh.seeded_state_(c(80), SEEDS["shp"] + i).train() for i, c in enumerate((M.SHP_D_NET64, M.SHP_D_NET128, M.SHP_D_NET256))]
    objss = rh.seeded_state_(M.OBJ_SS_D_NET(80), SEEDS["objss"]).train()
    objls = rh.seeded_state_(M.OBJ_LS_D_NET(80), SEEDS["objls"]).train()
    enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), SEEDS["inception"]))
    rh.seeded_state_(enc.emb_features, SEEDS["enc_proj"]); rh.seeded_state_(enc.emb_cnn_code, SEEDS["enc_proj"] + 1)
    enc.eval()
    cl = U.form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    glb = int(b["num_rois"].max())
    orig = M.CA_NET.reparametrize
    M.CA_NET.reparametrize = lambda self, mu, logvar: b["ca_eps"] * (logvar * 0.5).exp() + mu
    fake, bt_codes, atts, bt_atts, mu, logvar = G(b["noise"], b["sent_emb"], b["words_embs"],
                                                 b["glove_words_embs"], cl, b["mask"], b["hmaps"], b["rois"],
                                                 b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"], glb)
    M.CA_NET.reparametrize = orig
    bt = [c.detach() for c in bt_codes]
    out = {"seeds": SEEDS, "B": B, "cap": CAP}
    for i in range(3):
        Ls.patD_loss(pats[i], b["imgs"][i], fake[i], b["sent_emb"]).backward()
        out["gradPatD%d" % i] = grad_samples(pats[i])
    for i in range(3):
        random.seed(100 + i)
        Ls.shpD_loss(shps[i], b["imgs"][i], fake[i], b["hmaps"][i], b["rois"][i], b["num_rois"]).backward()
        out["gradShpD%d" % i] = grad_samples(shps[i])
    random.seed(200)
    Ls.objD_loss(objss, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], bt[-1], b["rois"][0], b["num_rois"]).backward()
    out["gradObjSSD"] = grad_samples(objss)
    random.seed(201)
    Ls.objD_loss(objls, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], bt[-1], b["fm_rois"], b["num_rois"],
                 is_large_scale=True).backward()
    out["gradObjLSD"] = grad_samples(objls)
    # generator: constant image encoder (make_golden.py explains why)
    for net in pats + shps + [objss, objls]:
        net.zero_grad()
    labels = torch.arange(B)
    regions_c, code_c = enc(fake[2].detach())
    const_enc = lambda x: (regions_c.detach(), code_c.detach())     # noqa: E731
    total, _ = Ls.G_loss(pats, shps, objss, objls, const_enc, fake, b["hmaps"], b["words_embs"], b["sent_emb"],
                         b["clabels_emb"], bt[-1], labels, b["cap_lens"], b["class_ids"], b["rois"][0],
                         b["fm_rois"], b["num_rois"])
    (total + Ls.KL_loss(mu, logvar)).backward()
    out["gradG_constenc"] = grad_samples(G)
    path = os.path.join(HERE, "step_b2_grads.pt")
    torch.save(out, path)
    print("saved", path, os.path.getsize(path) / 1e6, "MB",
          {k: len(v) for k, v in out.items() if isinstance(v, dict) and k.startswith("grad")})


if __name__ == "__main__":
    main()
