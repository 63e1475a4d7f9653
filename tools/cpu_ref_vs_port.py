"""Time the UNMODIFIED reference step against the oracle port on the same CPU (build container only: the GPU box
has no /root/reference).  Same seeded batch and weights; the reference's loop body (trainer.py:388-462 order: G
forward, eight discriminator losses + backward + Adam, G loss + backward + Adam, EMA) is driven through
oracle/ref_harness.py exactly as tests/golden/make_golden.py drives it.

    python tools/cpu_ref_vs_port.py [batch] [threads]   ->  profiles/r03_cpu_reference_vs_port.json

bench.py's cpu_baseline times the PORT on the GPU box's host (kind "port"); this file is the evidence that the port
costs what the reference costs.
"""
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd")]

from oracle import ref_harness as rh            # noqa: E402
from oracle import torch_model as tm            # noqa: E402
from oracle import torch_encoders as encoders   # noqa: E402
import synth_batch                              # noqa: E402


def reference_step(ref, nets, opts, ema, enc, b):
    M, Ls, U = ref.model, ref.losses, ref.utils
    G, pats, shps, objss, objls = nets
    B = b["imgs"][0].shape[0]
    cl = U.form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    glb = int(b["num_rois"].max())
    orig = M.CA_NET.reparametrize
    M.CA_NET.reparametrize = lambda self, mu, logvar: b["ca_eps"] * (logvar * 0.5).exp() + mu
    fake, bt_codes, _, _, mu, logvar = G(b["noise"], b["sent_emb"], b["words_embs"], b["glove_words_embs"], cl,
                                         b["mask"], b["hmaps"], b["rois"], b["fm_rois"], b["num_rois"],
                                         b["bt_masks"], b["fm_bt_masks"], glb)
    M.CA_NET.reparametrize = orig
    bt = [c.detach() for c in bt_codes]
    for i in range(3):
        opts["pat"][i].zero_grad()
        Ls.patD_loss(pats[i], b["imgs"][i], fake[i], b["sent_emb"]).backward()
        opts["pat"][i].step()
    for i in range(3):
        opts["shp"][i].zero_grad()
        random.seed(100 + i)
        Ls.shpD_loss(shps[i], b["imgs"][i], fake[i], b["hmaps"][i], b["rois"][i], b["num_rois"]).backward()
        opts["shp"][i].step()
    for net, opt, r, large, seed in ((objss, opts["objss"], b["rois"][0], False, 200),
                                     (objls, opts["objls"], b["fm_rois"], True, 201)):
        opt.zero_grad()
        random.seed(seed)
        e = Ls.objD_loss(net, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], bt[-1], r, b["num_rois"],
                         is_large_scale=large)
        if float(e) > 0:
            e.backward()
            opt.step()
    opts["G"].zero_grad()
    labels = torch.arange(B)
    total, _ = Ls.G_loss(pats, shps, objss, objls, enc, fake, b["hmaps"], b["words_embs"], b["sent_emb"],
                         b["clabels_emb"], bt[-1], labels, b["cap_lens"], b["class_ids"], b["rois"][0],
                         b["fm_rois"], b["num_rois"])
    (total + Ls.KL_loss(mu, logvar)).backward()
    opts["G"].step()
    with torch.no_grad():
        for p, a in zip(G.parameters(), ema):
            a.mul_(0.999).add_(p, alpha=0.001)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
    torch.set_num_threads(threads)
    ref = rh.load_reference(branch_num=3, batch_size=B)
    M = ref.model
    b = synth_batch.make_batch(B, seed=1234)
    G = rh.seeded_state_(M.G_NET(80), 11).train()
    pats = [rh.seeded_state_(c(), 21 + i).train() for i, c in enumerate((M.PAT_D_NET64, M.PAT_D_NET128, M.PAT_D_NET256))]
    shps = [rh.seeded_state_(c(80), 31 + i).train() for i, c in enumerate((M.SHP_D_NET64, M.SHP_D_NET128, M.SHP_D_NET256))]
    objss = rh.seeded_state_(M.OBJ_SS_D_NET(80), 41).train()
    objls = rh.seeded_state_(M.OBJ_LS_D_NET(80), 42).train()
    enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), 51)).eval()
    for p in enc.parameters():
        p.requires_grad_(False)
    adam = lambda ps: torch.optim.Adam(ps, lr=2e-4, betas=(0.5, 0.999))       # noqa: E731
    opts = {"G": adam(G.parameters()), "pat": [adam(n.parameters()) for n in pats],
            "shp": [adam(n.parameters()) for n in shps], "objss": adam(objss.parameters()), "objls": adam(objls.parameters())}
    ema = [p.detach().clone() for p in G.parameters()]
    nets = (G, pats, shps, objss, objls)

    def timed(fn, n=2):
        fn()                                    # warm-up
        t0 = time.time()
        for _ in range(n):
            fn()
        return (time.time() - t0) / n
    t_ref = timed(lambda: reference_step(ref, nets, opts, ema, enc, b))

    # the port: same weights by state-dict key, same batch
    sd_of = lambda m: {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running_" not in k)   # noqa: E731
                       for k, v in m.state_dict().items()}
    sds = {"G": sd_of(G), "pat": [sd_of(n) for n in pats], "shp": [sd_of(n) for n in shps],
           "objss": sd_of(objss), "objls": sd_of(objls)}
    popts = {"G": adam(tm.params_of(sds["G"])), "pat": [adam(tm.params_of(s)) for s in sds["pat"]],
             "shp": [adam(tm.params_of(s)) for s in sds["shp"]], "objss": adam(tm.params_of(sds["objss"])),
             "objls": adam(tm.params_of(sds["objls"]))}
    pema = [p.detach().clone() for p in tm.params_of(sds["G"])]
    penc = encoders.CpuImageEncoder(enc) if hasattr(encoders, "CpuImageEncoder") else enc
    t_port = timed(lambda: tm.train_step(sds, popts, pema, b, image_encoder=penc))
    res = {"batch": B, "threads": threads, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip(),
           "reference_s_per_step": round(t_ref, 3), "port_s_per_step": round(t_port, 3),
           "port_over_reference": round(t_port / t_ref, 4),
           "what": "full G+D step at 256x256 (3 stages, 8 discriminators, DAMSM through the Inception encoder, 9 Adam "
                   "updates, EMA), 1 warm-up + 2 timed steps each; reference = unmodified /root/reference modules "
                   "through oracle/ref_harness.py, port = oracle/torch_model.train_step (what bench.py's cpu_baseline times)"}
    print(json.dumps(res))
    with open(os.path.join(ROOT, "profiles", "r03_cpu_reference_vs_port.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
