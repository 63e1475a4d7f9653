"""Generate tests/golden/step_b16.pt from the UNMODIFIED reference at the BENCH batch (build container only):

    python tests/golden/make_golden_b16.py

Same recipe as make_golden.py (seeded synthetic batch from obj-gan_amd/synth_batch.py, weights filled by state-dict
key through oracle.ref_harness.seeded_state_), at B = 16: the batch BASELINE.json's configs 2-4 are quoted on and the
one bench.py times.  Row plans, split thresholds and the thin / MFMA switch of the HIP path all depend on the batch, so
the B = 2 fixtures do not pin the bench shapes.  Stored (reference OUTPUTS only, sub-sampled where large):

  * generator forward (model.py:722-795): the three images sub-sampled, mu / logvar, one attention map;
  * patD_loss / shpD_loss / objD_loss (miscc/losses.py:163-330) followed by backward(): the eight losses, the gradient
    norm of every parameter and flat[::stride] samples (<= 512 per tensor) of every gradient tensor;
  * G_loss + KL_loss (miscc/losses.py:333-529) with a CONSTANT image encoder (seeded region / sentence codes; the
    reason is in make_golden.py): total, kl, per-parameter gradient norms and the same samples.

The python `random` module is seeded before every permute_seg user exactly as in make_golden.py.
"""
import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd"), HERE]

from oracle import ref_harness as rh          # noqa: E402
import synth_batch                            # noqa: E402

B = 16
CAP = 512
SEEDS = dict(batch=4321, G=111, pat=121, shp=131, objss=141, objls=142, const_enc=151)


def grad_norms(module):
    return {k: (p.grad.norm().item() if p.grad is not None else 0.0) for k, p in module.named_parameters()}


def grad_samples(module):
    out = {}
    for k, p in module.named_parameters():
        if p.grad is None:
            continue
        flat = p.grad.detach().reshape(-1)
        out[k] = flat[::max(1, -(-flat.numel() // CAP))].clone()
    return out


def main():
    torch.set_num_threads(8)
    t0 = time.time()
    ref = rh.load_reference(branch_num=3, batch_size=B)
    b = synth_batch.make_batch(B, seed=SEEDS["batch"])
    M, Ls, U = ref.model, ref.losses, ref.utils
    G = rh.seeded_state_(M.G_NET(80), SEEDS["G"]).train()
    pats = [rh.seeded_state_(c(), SEEDS["pat"] + i).train()
            for i, c in enumerate((M.PAT_D_NET64, M.PAT_D_NET128, M.PAT_D_NET256))]
    shps = [rh.seeded_state_(c(80), SEEDS["shp"] + i).train()
            for i, c in enumerate((M.SHP_D_NET64, M.SHP_D_NET128, M.SHP_D_NET256))]
    objss = rh.seeded_state_(M.OBJ_SS_D_NET(80), SEEDS["objss"]).train()
    objls = rh.seeded_state_(M.OBJ_LS_D_NET(80), SEEDS["objls"]).train()
    g0 = torch.Generator().manual_seed(SEEDS["const_enc"])
    regions_c, code_c = torch.randn(B, 256, 17, 17, generator=g0), torch.randn(B, 256, generator=g0)

    out = {"seeds": SEEDS, "B": B, "cap": CAP}
    cl = U.form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    glb = int(b["num_rois"].max())
    orig = M.CA_NET.reparametrize
    M.CA_NET.reparametrize = lambda self, mu, logvar: b["ca_eps"] * (logvar * 0.5).exp() + mu
    fake, bt_codes, atts, bt_atts, mu, logvar = G(b["noise"], b["sent_emb"], b["words_embs"],
                                                 b["glove_words_embs"], cl, b["mask"], b["hmaps"], b["rois"],
                                                 b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"], glb)
    M.CA_NET.reparametrize = orig
    print("G forward %.0f s" % (time.time() - t0), flush=True)
    out["fake64_s2"] = fake[0].detach()[:, :, ::2, ::2].clone()
    out["fake128_s4"] = fake[1].detach()[:, :, ::4, ::4].clone()
    out["fake256_s8"] = fake[2].detach()[:, :, ::8, ::8].clone()
    out["fake_sums"] = [(float(f.detach().double().sum()), float(f.detach().double().pow(2).sum())) for f in fake]
    out["bt_c_code_last"] = bt_codes[-1].detach().clone()
    out["att128_s8"] = atts[1].detach()[:, :, ::8, ::8].clone()
    out["mu"], out["logvar"] = mu.detach().clone(), logvar.detach().clone()

    bt = [c.detach() for c in bt_codes]
    for i in range(3):
        e = Ls.patD_loss(pats[i], b["imgs"][i], fake[i], b["sent_emb"])
        e.backward()
        out["errPatD%d" % i] = e.item()
        out["gradPatD%d" % i] = grad_norms(pats[i]); out["gsPatD%d" % i] = grad_samples(pats[i])
    for i in range(3):
        random.seed(100 + i)
        e = Ls.shpD_loss(shps[i], b["imgs"][i], fake[i], b["hmaps"][i], b["rois"][i], b["num_rois"])
        e.backward()
        out["errShpD%d" % i] = e.item()
        out["gradShpD%d" % i] = grad_norms(shps[i]); out["gsShpD%d" % i] = grad_samples(shps[i])
    print("Pat / Shp D %.0f s" % (time.time() - t0), flush=True)
    random.seed(200)
    e = Ls.objD_loss(objss, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], bt[-1], b["rois"][0], b["num_rois"])
    e.backward(); out["errObjSSD"] = e.item()
    out["gradObjSSD"] = grad_norms(objss); out["gsObjSSD"] = grad_samples(objss)
    del e
    random.seed(201)
    e = Ls.objD_loss(objls, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], bt[-1], b["fm_rois"], b["num_rois"],
                     is_large_scale=True)
    e.backward(); out["errObjLSD"] = e.item()
    out["gradObjLSD"] = grad_norms(objls); out["gsObjLSD"] = grad_samples(objls)
    del e
    print("Obj D %.0f s" % (time.time() - t0), flush=True)

    for net in pats + shps + [objss, objls]:
        net.zero_grad()
    labels = torch.arange(B)
    const_enc = lambda x: (regions_c, code_c)     # noqa: E731
    total, logs = Ls.G_loss(pats, shps, objss, objls, const_enc, fake, b["hmaps"], b["words_embs"], b["sent_emb"],
                            b["clabels_emb"], bt[-1], labels, b["cap_lens"], b["class_ids"], b["rois"][0],
                            b["fm_rois"], b["num_rois"])
    kl = Ls.KL_loss(mu, logvar)
    (total + kl).backward()
    out["errG_constenc"] = (total + kl).item(); out["kl"] = kl.item(); out["G_logs"] = logs
    out["gradG_constenc"] = grad_norms(G); out["gsG_constenc"] = grad_samples(G)
    w0, w1, _, _ = Ls.words_loss(regions_c, b["words_embs"], labels, b["cap_lens"], b["class_ids"], B)
    s0, s1, _ = Ls.sent_loss(code_c, b["sent_emb"], labels, b["class_ids"], B)
    out["w_loss"], out["s_loss"] = (w0 + w1).item(), (s0 + s1).item()
    path = os.path.join(HERE, "step_b16.pt")
    torch.save(out, path)
    print({k: v for k, v in out.items() if isinstance(v, float)})
    print("saved %.2f MB after %.0f s" % (os.path.getsize(path) / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
