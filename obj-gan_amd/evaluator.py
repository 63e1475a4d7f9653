"""Sampling path of the image generator (reference image_generation/evaluator.py:268-340, steps (2)-(4) of
`evaluate`): generated boxes -> SHP_G_NET instance masks -> form_hmaps layout maps -> caption / GloVe
embeddings -> G_NET inference with the EMA weights, plus the Inception predictions of the score.

What is here: the networks, their checkpoints (`build_models`, same files and order as the reference) and
`sampling(...)` for one batch of prepared tensors.  What is not (out of scope): the test
data set / loader, FID statistics, R-precision bookkeeping, TensorFlow Inception, image grids.
"""
import os

import numpy as np
import torch

from miscc.config import cfg
from miscc.utils import mkdir_p, weights_init, form_clabels_feat, form_hmaps
from model import G_NET, SHP_G_NET, RNN_ENCODER
from trainer import category_embeddings


class condGANEvaluator(object):
    def __init__(self, output_dir, data_loader, dataset, device=None):
        self.image_dir = os.path.join(output_dir, 'Image') if output_dir else ''
        self.score_dir = os.path.join(output_dir, 'Score') if output_dir else ''
        for d in (self.image_dir, self.score_dir):
            if d:
                mkdir_p(d)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.batch_size = cfg.TRAIN.BATCH_SIZE
        self.data_loader = data_loader
        for name in ("n_words", "ixtoword", "cats_dict", "cats_index_dict", "cat_labels", "cat_label_lens",
                     "sorted_cat_label_indices"):
            setattr(self, name, getattr(dataset, name, None))
        self.num_classes = len(self.cats_index_dict) if self.cats_index_dict is not None else \
            getattr(dataset, "num_classes", 80)
        self.glove_emb = getattr(dataset, "glove_embed", None)
        self.text_encoder = getattr(dataset, "text_encoder", None)
        self.inception_model = getattr(dataset, "inception_model", None)
        self.netG = self.netShpG = None

    def build_models(self):
        """-> [text_encoder, image_encoder (None: R-precision is out of scope), netG, netShpG]"""
        if self.text_encoder is None:
            enc = RNN_ENCODER(self.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM)
            enc.load_state_dict(torch.load(cfg.TRAIN.NET_E, map_location="cpu"))
            print('Load text encoder from:', cfg.TRAIN.NET_E)
            self.text_encoder = enc
        for p in self.text_encoder.parameters():
            p.requires_grad_(False)
        self.text_encoder.to(self.device).eval()
        netG = G_NET(self.num_classes)
        netG.apply(weights_init)
        netShpG = None
        if cfg.TEST.USE_GT_BOX_SEG > 0:
            netShpG = SHP_G_NET(self.num_classes)
            netShpG.apply(weights_init)
        if cfg.TRAIN.NET_G:
            netG.load_state_dict(torch.load(cfg.TRAIN.NET_G, map_location="cpu"))
            print('Load G from: ', cfg.TRAIN.NET_G)
        if netShpG is not None and cfg.TEST.NET_SHP_G and os.path.exists(cfg.TEST.NET_SHP_G):
            netShpG.load_state_dict(torch.load(cfg.TEST.NET_SHP_G, map_location="cpu"))
            print('Load Shape G from: ', cfg.TEST.NET_SHP_G)
        self.netG = netG.to(self.device).eval()
        self.netShpG = netShpG.to(self.device).eval() if netShpG is not None else None
        if self.glove_emb is not None:
            self.glove_emb.to(self.device).eval()
        if self.inception_model is not None:
            self.inception_model.to(self.device).eval()
        return [self.text_encoder, None, self.netG, self.netShpG]

    def prepare_cat_emb(self):
        return category_embeddings(self.glove_emb.weight, self.cat_labels, self.cat_label_lens,
                                   self.sorted_cat_label_indices, len(self.cats_index_dict)).to(self.device)

    @torch.no_grad()
    def sampling(self, data, clabels_emb, hmap_size, noise_img=None, noise_shp=None):
        """One batch of evaluator.py:290-340.  `data`: dict with rois[3] / fm_rois / num_rois, and either the
        ground-truth layout (hmaps, bt_masks, fm_bt_masks; cfg.TEST.USE_GT_BOX_SEG == 0) or the box maps of
        the shape generator (bbox_maps_fwd, bbox_maps_bwd, bbox_fmaps); captions / cap_lens /
        glove_captions, or precomputed words_embs / sent_emb / glove_words_embs / mask.
        -> dict(fake_imgs, attn_maps, bt_attn_maps, hmaps, raw_masks, is_pred)"""
        d = data
        rois, fm_rois, num_rois = d["rois"], d["fm_rois"], d["num_rois"]
        B = int(num_rois.shape[0])
        max_num_roi = int(torch.max(num_rois))
        if noise_img is None:
            noise_img = torch.randn(B, cfg.GAN.Z_DIM, device=self.device)
        raw_masks = None
        if cfg.TEST.USE_GT_BOX_SEG > 0:
            if noise_shp is None:
                noise_shp = torch.randn(B, cfg.ROI.BOXES_NUM, self.num_classes * 4, device=self.device)
            raw_masks = self.netShpG(noise_shp[:, :max_num_roi], d["bbox_maps_fwd"], d["bbox_maps_bwd"],
                                     d["bbox_fmaps"]).squeeze(2)
            hmaps, bt_masks, fm_bt_masks = form_hmaps(raw_masks, num_rois, rois[0], hmap_size, self.num_classes)
        else:
            hmaps, bt_masks, fm_bt_masks = d["hmaps"], d["bt_masks"], d["fm_bt_masks"]
        if "captions" in d and self.text_encoder is not None:
            captions, cap_lens = d["captions"], d["cap_lens"]
            max_len = int(torch.max(cap_lens))
            words_embs, sent_emb = self.text_encoder(captions, cap_lens, max_len)
            num_words = words_embs.size(2)
            mask = (captions == 0)[:, :num_words]
            gc = d["glove_captions"]
            gw = torch.nn.functional.embedding(gc.reshape(-1), self.glove_emb.weight)
            glove_words_embs = gw.view(gc.size(0), gc.size(1), -1)[:, :num_words].transpose(1, 2)
        else:
            words_embs, sent_emb = d["words_embs"], d["sent_emb"]
            glove_words_embs, mask = d["glove_words_embs"], d["mask"]
        clabels_feat = form_clabels_feat(clabels_emb, rois[0], num_rois)
        fake_imgs, _, attn_maps, bt_attn_maps, _, _ = self.netG(
            noise_img, sent_emb, words_embs, glove_words_embs, clabels_feat, mask, hmaps, rois, fm_rois,
            num_rois, bt_masks, fm_bt_masks, max_num_roi)
        out = {"fake_imgs": fake_imgs, "attn_maps": attn_maps, "bt_attn_maps": bt_attn_maps, "hmaps": hmaps,
               "raw_masks": raw_masks}
        if self.inception_model is not None:
            out["is_pred"] = self.inception_model(fake_imgs[-1])
        return out

    def save_singleimages(self, images, keys, sent_ids):
        """evaluator.py:225-233: [-1, 1] images -> <Image>/<key>_<sent id>.jpg"""
        from PIL import Image
        images = images.detach()
        for i in range(images.size(0)):
            img = images[i].add(1).div(2).mul(255).clamp(0, 255).byte()
            ndarr = img.permute(1, 2, 0).cpu().numpy()
            Image.fromarray(ndarr).save('%s/%s_%d.jpg' % (self.image_dir, keys[i], sent_ids[i]))

    def write_scores(self, predictions):
        from miscc.utils import compute_inception_score, negative_log_posterior_probability
        preds = np.concatenate([p.detach().cpu().numpy() if torch.is_tensor(p) else np.asarray(p)
                                for p in predictions], 0)
        splits = min(10, self.batch_size)
        mean, std = compute_inception_score(preds, splits)
        mean_conf, std_conf = negative_log_posterior_probability(preds, splits)
        if self.score_dir:
            with open('%s/scores.txt' % self.score_dir, 'w') as fp:
                fp.write('mean, std, mean_conf, std_conf \n')
                fp.write('%f, %f, %f, %f' % (mean, std, mean_conf, std_conf))
        return mean, std, mean_conf, std_conf
