"""Generator / discriminator networks of the Obj-GAN image generator on MI355X kernels.

Drop-in for the training-path part of the reference module of the same name (reference
image_generation/model.py:19-81, 452-795, 986-1312): same class names, constructor arguments,
forward signatures, return tuples and state-dict keys (`h_net1_sent.fc.0.weight`,
`h_net2_main.residual.0.block.1.weight`, `img_code.3.running_mean`, `COND_DNET.jointConv.0.weight`
...), so reference checkpoints and the reference trainer/loss code load and run unchanged.

How it differs from the reference: the nn.Conv2d / nn.BatchNorm / nn.Linear objects below are
only PARAMETER HOLDERS (they keep the state-dict layout and the reference initialisation working);
their own forward() is never used.  Each block's forward walks its holders and launches the fused
gfx950 kernels of objgan_hip.ops instead:

  upBlock        nearest-x2 gather fused into the MFMA conv  ->  BatchNorm(batch stats)+GLU pass
  HmapResBlock   reflect-pad gather fused into the conv -> InstanceNorm+GLU -> conv -> InstanceNorm+residual
  G_HMAP / shp_code   reflect conv(+bias) -> InstanceNorm+LeakyReLU -> stride-2 conv with LeakyReLU epilogue
  D encoders     4x4/s2 conv with LeakyReLU epilogue, conv -> BatchNorm+LeakyReLU pass
  attention      GlobalAttention.py (fused kernels), masked max without 5-D temporaries
  object Ds      fused bilinear lift to 512^2, ROIAlign kernel, 4x4 conv with LeakyReLU epilogue

RNN_ENCODER (frozen caption encoder) runs on its own fused LSTM kernel; the frozen image encoders
(CNN_ENCODER, INCEPTION_V3) live in `encoders.py` with their convolutions on the same MFMA kernels.
"""
import numpy as np
import torch
import torch.nn as nn

from miscc.config import cfg
from miscc.utils import pprocess_bt_attns
from GlobalAttention import GlobalAttentionGeneral as ATT_NET
from GlobalAttention import GlobalBUAttentionGeneral as BT_ATT_NET
from models.roi_align.modules.roi_align import RoIAlignAvg
from objgan_hip import ops


# ---------------------------------------------------------------------------------------------
# small helpers shared by the blocks
# ---------------------------------------------------------------------------------------------
# BatchNorm's `num_batches_tracked` counters (pure bookkeeping: momentum is fixed) cost one 4 us launch per
# layer and pass -- 111 per training step.  Inside `deferred_bn_counters()` the increments are collected on
# the host and applied with one multi-tensor add per distinct count when the block exits.
_NBT_PENDING = None


class deferred_bn_counters(object):
    def __enter__(self):
        global _NBT_PENDING
        self.outer = _NBT_PENDING
        if self.outer is None:
            _NBT_PENDING = {}
        return self

    def __exit__(self, *exc):
        global _NBT_PENDING
        if self.outer is None:
            pending, _NBT_PENDING = _NBT_PENDING, None
            by_count = {}
            for t, k in pending.values():
                by_count.setdefault(k, []).append(t)
            for k, ts in by_count.items():
                torch._foreach_add_(ts, k)
        return False


def _bn_act(x, bn, mode):
    """BatchNorm fused with GLU / LeakyReLU: batch statistics + running-stat update in train mode,
    running statistics (forward only) in eval mode -- sampling with the EMA generator."""
    if not bn.training:
        return ops.norm_act_eval(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, mode=mode, eps=bn.eps)
    if bn.num_batches_tracked is not None:
        if _NBT_PENDING is not None:
            ent = _NBT_PENDING.get(id(bn))
            _NBT_PENDING[id(bn)] = (bn.num_batches_tracked, 1 if ent is None else ent[1] + 1)
        else:
            bn.num_batches_tracked += 1
    return ops.norm_act(x, bn.weight, bn.bias, None, bn.running_mean, bn.running_var,
                        per_channel=True, mode=mode, eps=bn.eps, momentum=bn.momentum)


def _in_act(x, inorm, mode, residual=None):
    """InstanceNorm2d(affine=False) fused with GLU / LeakyReLU / residual add."""
    return ops.norm_act(x, None, None, residual, None, None, per_channel=False, mode=mode,
                        eps=inorm.eps)


class GLU(nn.Module):
    """x[:, :C/2] * sigmoid(x[:, C/2:]) -- only reached stand-alone after CA_NET's fc; everywhere
    else it is fused into the normalisation pass."""

    def __init__(self):
        super(GLU, self).__init__()

    def forward(self, x):
        nc = x.size(1)
        assert nc % 2 == 0, 'channels dont divide 2!'
        nc = nc // 2
        return x[:, :nc] * torch.sigmoid(x[:, nc:])


def conv1x1(in_planes, out_planes, bias=False):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=bias)


def conv3x3(in_planes, out_planes):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)


class _UpBlock(nn.Sequential):
    """[Upsample x2 nearest, conv3x3(in, 2*out), norm(2*out), GLU]: spatial size x2."""

    def forward(self, x):
        conv, norm = self[1], self[2]
        y = ops.conv2d(x, conv.weight, None, 1, 1, "zeros", upsample=True)
        if isinstance(norm, nn.InstanceNorm2d):
            return _in_act(y, norm, "glu")
        return _bn_act(y, norm, "glu")


def upBlock(in_planes, out_planes, norm=nn.BatchNorm2d):
    return _UpBlock(nn.Upsample(scale_factor=2, mode='nearest'),
                    conv3x3(in_planes, out_planes * 2),
                    norm(out_planes * 2),
                    GLU())


class _ActSeq(nn.Sequential):
    """[Conv2d, (norm,) LeakyReLU(0.2)] -- conv with the activation in its epilogue, or
    conv -> fused norm+LeakyReLU when a norm layer is present."""

    def forward(self, x):
        conv = self[0]
        pad = conv.padding[0]
        if len(self) == 2:
            return ops.conv2d(x, conv.weight, conv.bias, conv.stride[0], pad, "zeros", act="lrelu")
        y = ops.conv2d(x, conv.weight, conv.bias, conv.stride[0], pad, "zeros")
        norm = self[1]
        if isinstance(norm, nn.InstanceNorm2d):
            return _in_act(y, norm, "lrelu")
        return _bn_act(y, norm, "lrelu")


def downBlock_G(in_planes, out_planes, kernel_size=3, stride=2, padding=1, norm=None):
    layers = [nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                        padding=padding, bias=False)]
    if norm is not None:
        layers.append(norm(out_planes))
    layers.append(nn.LeakyReLU(0.2, inplace=True))
    return _ActSeq(*layers)


class _ReflStem(nn.Sequential):
    """[ReflectionPad2d(1), Conv2d(k3, p0, bias), InstanceNorm2d, LeakyReLU(0.2)]."""

    def forward(self, x):
        conv, inorm = self[1], self[2]
        y = ops.conv2d(x, conv.weight, conv.bias, 1, 1, "reflect")
        return _in_act(y, inorm, "lrelu")


def _shape_stem(ncf, ngf):
    return _ReflStem(nn.ReflectionPad2d(1),
                          nn.Conv2d(ncf, ngf, kernel_size=3, padding=0),
                          nn.InstanceNorm2d(ngf),
                          nn.LeakyReLU(0.2, inplace=True))


class _ResBody(nn.Sequential):
    """[pad, conv c->2c, IN, GLU, pad, conv c->c, IN] of HmapResBlock; forward(x) returns the
    block output INCLUDING the residual (fused into the last normalisation pass)."""

    def forward(self, x):
        y = ops.conv2d(x, self[1].weight, None, 1, 1, "reflect")
        y = _in_act(y, self[2], "glu")
        y = ops.conv2d(y, self[5].weight, None, 1, 1, "reflect")
        return _in_act(y, self[6], None, residual=x)


class HmapResBlock(nn.Module):
    def __init__(self, channel_num):
        super(HmapResBlock, self).__init__()
        self.block = _ResBody(
            nn.ReflectionPad2d(1),
            nn.Conv2d(channel_num, channel_num * 2, kernel_size=3, stride=1, padding=0, bias=False),
            nn.InstanceNorm2d(channel_num * 2),
            GLU(),
            nn.ReflectionPad2d(1),
            nn.Conv2d(channel_num, channel_num, kernel_size=3, stride=1, padding=0, bias=False),
            nn.InstanceNorm2d(channel_num))

    def forward(self, x):
        return self.block(x)


# ---------------------------------------------------------------------------------------------
# frozen text encoder
# ---------------------------------------------------------------------------------------------
class RNN_ENCODER(nn.Module):
    """Bidirectional-LSTM caption encoder (reference model.py:85-179): same constructor arguments,
    state-dict keys (`encoder.weight`, `rnn.weight_ih_l0`, `rnn.weight_hh_l0_reverse`, ...) and
    `forward(captions, cap_lens, max_len) -> (words_emb [B, nhidden, max_len], sent_emb [B, nhidden])`.
    nn.Embedding / nn.LSTM are parameter holders; the forward pass is one fused kernel per step
    (objgan_lstm_bidir_forward).  The encoder is frozen on this path (trainer.py:97-100 of the
    reference): forward only, eval-mode dropout (identity)."""

    def __init__(self, ntoken, ninput=300, drop_prob=0.5, nhidden=128, nlayers=1, bidirectional=True):
        super(RNN_ENCODER, self).__init__()
        self.n_steps = cfg.TEXT.WORDS_NUM
        self.ntoken = ntoken
        self.ninput = ninput
        self.drop_prob = drop_prob
        self.nlayers = nlayers
        self.bidirectional = bidirectional
        self.rnn_type = cfg.RNN_TYPE
        self.num_directions = 2 if bidirectional else 1
        self.nhidden = nhidden // self.num_directions
        if self.rnn_type != 'LSTM' or nlayers != 1 or not bidirectional:
            raise NotImplementedError("the hot path uses the reference default: 1-layer bidirectional LSTM")
        self.encoder = nn.Embedding(self.ntoken, self.ninput)
        self.drop = nn.Dropout(self.drop_prob)
        self.rnn = nn.LSTM(self.ninput, self.nhidden, self.nlayers, batch_first=True,
                           dropout=self.drop_prob, bidirectional=self.bidirectional)
        self.encoder.weight.data.uniform_(-0.1, 0.1)
        self._packed = None

    def init_hidden(self, bsz):
        w = next(self.parameters()).data
        z = w.new_zeros(self.nlayers * self.num_directions, bsz, self.nhidden)
        return (z, z.clone())

    def _weights(self):
        """[2][I][4H] / [2][H][4H] transposed copies (consecutive gate outputs contiguous), rebuilt
        when a weight tensor was replaced or edited."""
        r = self.rnn
        srcs = (r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse,
                r.bias_ih_l0, r.bias_ih_l0_reverse, r.bias_hh_l0, r.bias_hh_l0_reverse)
        key = tuple((t.data_ptr(), t._version) for t in srcs)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                wt_ih = torch.stack((srcs[0].t(), srcs[1].t())).contiguous()
                wt_hh = torch.stack((srcs[2].t(), srcs[3].t())).contiguous()
                b_ih = torch.stack((srcs[4], srcs[5])).contiguous()
                b_hh = torch.stack((srcs[6], srcs[7])).contiguous()
            self._packed = (key, wt_ih, wt_hh, b_ih, b_hh)
        return self._packed[1:]

    def forward(self, captions, cap_lens, max_len, mask=None):
        if self.training and self.drop_prob > 0:
            raise NotImplementedError("RNN_ENCODER is frozen on the training path: call .eval()")
        wt_ih, wt_hh, b_ih, b_hh = self._weights()
        words_emb, sent_emb = ops.lstm_bidir_forward(self.encoder.weight.detach(), captions, cap_lens,
                                                     wt_ih, wt_hh, b_ih, b_hh, max_len)
        return words_emb, sent_emb


# ---------------------------------------------------------------------------------------------
# generator
# ---------------------------------------------------------------------------------------------
class CA_NET(nn.Module):
    """Conditioning augmentation (reference model.py:455-483).  `fixed_eps` (optional attribute)
    injects the reparametrisation noise for parity tests; otherwise it is drawn on the device."""

    def __init__(self):
        super(CA_NET, self).__init__()
        self.t_dim = cfg.TEXT.EMBEDDING_DIM
        self.c_dim = cfg.GAN.CONDITION_DIM
        self.fc = nn.Linear(self.t_dim, self.c_dim * 4, bias=True)
        self.relu = GLU()
        self.fixed_eps = None

    def encode(self, text_embedding):
        x = self.relu(ops.linear(text_embedding, self.fc.weight, self.fc.bias))
        return x[:, :self.c_dim], x[:, self.c_dim:]

    def reparametrize(self, mu, logvar):
        std = (logvar * 0.5).exp()
        eps = self.fixed_eps if self.fixed_eps is not None else torch.randn_like(std)
        return eps * std + mu

    def forward(self, text_embedding):
        mu, logvar = self.encode(text_embedding)
        return self.reparametrize(mu, logvar), mu, logvar


class _FcGlu(nn.Sequential):
    """[Linear(no bias), BatchNorm1d, GLU]."""

    def forward(self, x):
        y = ops.linear(x, self[0].weight, None)
        y = _bn_act(y.reshape(y.shape[0], y.shape[1], 1, 1), self[1], "glu")
        return y.reshape(y.shape[0], y.shape[1])


class INIT_STAGE_G(nn.Module):
    def __init__(self, ngf, ncf):
        super(INIT_STAGE_G, self).__init__()
        self.gf_dim = ngf
        self.in_dim = cfg.GAN.Z_DIM + ncf
        self.define_module()

    def define_module(self):
        nz, ngf = self.in_dim, self.gf_dim
        self.fc = _FcGlu(nn.Linear(nz, ngf * 8 * 8 * 2, bias=False),
                           nn.BatchNorm1d(ngf * 8 * 8 * 2),
                           GLU())
        self.upsample1 = upBlock(ngf, ngf // 2)
        self.upsample2 = upBlock(ngf // 2, ngf // 4)

    def forward(self, z_code, c_code):
        out = self.fc(torch.cat((c_code, z_code), 1))
        out = out.view(-1, self.gf_dim, 8, 8)
        return self.upsample2(self.upsample1(out))


def _max_rois(num_rois):
    from miscc.utils import _host
    nr = _host(num_rois)
    return int(np.amax(nr)) if nr.size else 0


class _BottomUpMixin(object):
    """Object ("bottom-up") branch shared by the stage bodies: label/word attention over the
    box slots, then the box-mask-gated max that paints slot vectors onto the feature grid."""

    def _bottom_up(self, word_embs, glove_word_embs, slabels_feat, mask, bt_mask, max_num_roi,
                   ih, iw, want_att):
        slabels_feat = slabels_feat[:, :, :max_num_roi]
        self.bt_att.applyMask(mask)
        raw_bt_c_code, raw_bt_att = self.bt_att(slabels_feat, glove_word_embs, word_embs)
        bt_mask = bt_mask[:, :max_num_roi]
        bt_c_code = ops.masked_max(raw_bt_c_code, bt_mask, ih, iw)
        bt_slabels_code = ops.masked_max(slabels_feat, bt_mask, ih, iw)
        bt_att = ops.masked_max(raw_bt_att, bt_mask, ih, iw) if want_att else None
        return raw_bt_c_code, bt_c_code, bt_att, bt_slabels_code


class INIT_STAGE_G_MAIN(nn.Module, _BottomUpMixin):
    def __init__(self, ngf, nef, nef2):
        super(INIT_STAGE_G_MAIN, self).__init__()
        self.gf_dim = ngf
        self.ef_dim = nef
        self.ef_dim2 = nef2
        self.define_module()

    def _make_layer(self, block, channel_num):
        return nn.Sequential(*[block(channel_num) for _ in range(cfg.GAN.GLB_R_NUM)])

    def define_module(self):
        ngf, nef, nef2 = self.gf_dim, self.ef_dim, self.ef_dim2
        self.bt_att = BT_ATT_NET(ngf, nef)
        self.residual = self._make_layer(HmapResBlock, ngf * 3 + nef2)
        self.upsample = upBlock(ngf * 3 + nef2, ngf)

    def forward(self, h_code_hmap, h_code1_sent, c_code, word_embs, glove_word_embs, slabels_feat,
                mask, rois, num_rois, bt_mask, glb_max_num_roi, max_num_roi=None):
        ih, iw = h_code_hmap.size(2), h_code_hmap.size(3)
        if max_num_roi is None:
            max_num_roi = _max_rois(num_rois)
        if max_num_roi > 0:
            _, bt_c_code, _, bt_slabels_code = self._bottom_up(
                word_embs, glove_word_embs, slabels_feat, mask, bt_mask, max_num_roi, ih, iw, False)
        else:
            # the reference raises here (`att` undefined, model.py:572); same contract
            raise RuntimeError("INIT_STAGE_G_MAIN needs at least one box in the batch")
        out_code = torch.cat((h_code_hmap, h_code1_sent, bt_c_code, bt_slabels_code), 1)
        return self.upsample(self.residual(out_code))


class G_HMAP(nn.Module):
    """Layout-map encoder (reference model.py:589-617): B x ncf x S x S -> B x 2ngf x S/2 x S/2."""

    def __init__(self, ngf, ncf):
        super(G_HMAP, self).__init__()
        self.gf_dim = ngf
        self.in_dim = ncf
        self.define_module()

    def define_module(self):
        ncf, ngf = self.in_dim, self.gf_dim
        self.conv3x3 = _shape_stem(ncf, ngf)
        self.downsample1 = downBlock_G(ngf, ngf * 2)

    def forward(self, hmap):
        return self.downsample1(self.conv3x3(hmap))


class NEXT_STAGE_G_MAIN(nn.Module, _BottomUpMixin):
    def __init__(self, ngf, nef, nef2):
        super(NEXT_STAGE_G_MAIN, self).__init__()
        self.gf_dim = ngf
        self.ef_dim = nef
        self.ef_dim2 = nef2
        self.define_module()

    def _make_layer(self, block, channel_num):
        return nn.Sequential(*[block(channel_num) for _ in range(cfg.GAN.LOCAL_R_NUM)])

    def define_module(self):
        ngf, nef, nef2 = self.gf_dim, self.ef_dim, self.ef_dim2
        self.att = ATT_NET(ngf, nef)
        self.bt_att = BT_ATT_NET(ngf, nef)
        self.residual = self._make_layer(HmapResBlock, ngf * 3 + nef2)
        self.upsample = upBlock(ngf * 3 + nef2, ngf)

    def forward(self, h_code, h_code_hmap, c_code, word_embs, glove_word_embs, slabels_feat, mask,
                rois, num_rois, bt_mask, glb_max_num_roi, max_num_roi=None):
        idf, ih, iw = h_code.size(1), h_code.size(2), h_code.size(3)
        self.att.applyMask(mask)
        c_code, att = self.att(h_code, word_embs)
        if max_num_roi is None:
            max_num_roi = _max_rois(num_rois)
        B = c_code.size(0)
        raw_bt_c_code = torch.zeros((B, idf, glb_max_num_roi, 1), dtype=c_code.dtype,
                                    device=c_code.device)
        if max_num_roi > 0:
            raw, bt_c_code, bt_att, bt_slabels_code = self._bottom_up(
                word_embs, glove_word_embs, slabels_feat, mask, bt_mask, max_num_roi, ih, iw, True)
            raw_bt_c_code[:, :, :max_num_roi] = raw
        else:
            bt_c_code = torch.zeros_like(c_code)
            bt_att = torch.zeros_like(att)
            bt_slabels_code = torch.zeros((B, self.ef_dim2, ih, iw), dtype=c_code.dtype,
                                          device=c_code.device)
        h_c_code = torch.cat((h_code + h_code_hmap, c_code, bt_c_code, bt_slabels_code), 1)
        out_code = self.upsample(self.residual(h_c_code))
        raw_bt_c_code = raw_bt_c_code.transpose(1, 2).squeeze(-1)
        return out_code, raw_bt_c_code, att, bt_att


class _ToRGB(nn.Sequential):
    def forward(self, x):
        return ops.conv2d(x, self[0].weight, None, 1, 1, "zeros", act="tanh")


class GET_IMAGE_G(nn.Module):
    def __init__(self, ngf):
        super(GET_IMAGE_G, self).__init__()
        self.gf_dim = ngf
        self.img = _ToRGB(conv3x3(ngf, 3), nn.Tanh())

    def forward(self, h_code):
        return self.img(h_code)


class G_NET(nn.Module):
    """Three-stage generator (reference model.py:722-795)."""

    def __init__(self, num_classes):
        super(G_NET, self).__init__()
        ngf = cfg.GAN.GF_DIM
        nef = cfg.TEXT.EMBEDDING_DIM
        nef2 = cfg.TEXT.GLOVE_EMBEDDING_DIM
        ncf = cfg.GAN.CONDITION_DIM
        self.ca_net = CA_NET()
        self.num_classes = num_classes
        if cfg.TREE.BRANCH_NUM > 0:
            self.h_net1_sent = INIT_STAGE_G(ngf * 4, ncf)
            self.h_net1_hmap = G_HMAP(ngf // 2, num_classes)
            self.h_net1_main = INIT_STAGE_G_MAIN(ngf, nef, nef2)
            self.img_net1 = GET_IMAGE_G(ngf)
        if cfg.TREE.BRANCH_NUM > 1:
            self.h_net2_hmap = G_HMAP(ngf // 2, num_classes)
            self.h_net2_main = NEXT_STAGE_G_MAIN(ngf, nef, nef2)
            self.img_net2 = GET_IMAGE_G(ngf)
        if cfg.TREE.BRANCH_NUM > 2:
            self.h_net3_hmap = G_HMAP(ngf // 2, num_classes)
            self.h_net3_main = NEXT_STAGE_G_MAIN(ngf, nef, nef2)
            self.img_net3 = GET_IMAGE_G(ngf)

    def forward(self, z_code, sent_emb, word_embs, glove_word_embs, slabels_feat, mask, hmaps, rois,
                fm_rois, num_rois, bt_masks, fm_bt_masks, glb_max_num_roi):
        fake_imgs, bt_c_codes, att_maps, bt_att_maps = [], [], [], []
        c_code, mu, logvar = self.ca_net(sent_emb)
        # one device->host read of the box counts per forward (the reference does one per stage)
        max_num_roi = _max_rois(num_rois)

        if cfg.TREE.BRANCH_NUM > 0:
            h_code1_hmap = self.h_net1_hmap(hmaps[0])
            h_code1_sent = self.h_net1_sent(z_code, c_code)
            h_code1 = self.h_net1_main(h_code1_hmap, h_code1_sent, c_code, word_embs,
                                       glove_word_embs, slabels_feat, mask, fm_rois, num_rois,
                                       fm_bt_masks, glb_max_num_roi, max_num_roi=max_num_roi)
            fake_imgs.append(self.img_net1(h_code1))
        if cfg.TREE.BRANCH_NUM > 1:
            h_code2_hmap = self.h_net2_hmap(hmaps[1])
            h_code2, bt_c_code2, att1, bt_att1 = self.h_net2_main(
                h_code1, h_code2_hmap, c_code, word_embs, glove_word_embs, slabels_feat, mask,
                rois[0], num_rois, bt_masks[0], glb_max_num_roi, max_num_roi=max_num_roi)
            fake_imgs.append(self.img_net2(h_code2))
            bt_c_codes.append(bt_c_code2)
            if att1 is not None:
                att_maps.append(att1)
            if bt_att1 is not None:
                bt_att_maps.append(bt_att1)
        if cfg.TREE.BRANCH_NUM > 2:
            h_code3_hmap = self.h_net3_hmap(hmaps[2])
            h_code3, bt_c_code3, att2, bt_att2 = self.h_net3_main(
                h_code2, h_code3_hmap, c_code, word_embs, glove_word_embs, slabels_feat, mask,
                rois[1], num_rois, bt_masks[1], glb_max_num_roi, max_num_roi=max_num_roi)
            fake_imgs.append(self.img_net3(h_code3))
            bt_c_codes.append(bt_c_code3)
            if att2 is not None:
                att_maps.append(att2)
            if bt_att2 is not None:
                bt_att_maps.append(bt_att2)
        return fake_imgs, bt_c_codes, att_maps, bt_att_maps, mu, logvar


# ---------------------------------------------------------------------------------------------
# discriminators
# ---------------------------------------------------------------------------------------------
class _JointBlock(nn.Sequential):
    """[conv3x3, BatchNorm2d, LeakyReLU(0.2)]."""

    def forward(self, x):
        conv, bn = self[0], self[1]
        y = ops.conv2d(x, conv.weight, None, 1, 1, "zeros")
        return _bn_act(y, bn, "lrelu")


# ---------------------------------------------------------------------------------------------
# shape generator of the sampling path (reference model.py:800-985; evaluator.py loads it from
# cfg.TEST.NET_SHP_G to turn generated boxes into instance masks).  Forward only.
# ---------------------------------------------------------------------------------------------
def downBlock_3x3(in_planes, out_planes):
    return _ActSeq(nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=2, padding=1, bias=False),
                   nn.InstanceNorm2d(out_planes),
                   nn.LeakyReLU(0.2, inplace=True))


class _GluSeq(nn.Sequential):
    """[conv3x3(in, 2*out), norm(2*out), GLU]: keeps the spatial size."""

    def forward(self, x):
        conv, norm = self[0], self[1]
        y = ops.conv2d(x, conv.weight, None, 1, 1, "zeros")
        if isinstance(norm, nn.InstanceNorm2d):
            return _in_act(y, norm, "glu")
        return _bn_act(y, norm, "glu")


def Block3x3_relu(in_planes, out_planes, norm=nn.BatchNorm2d):
    return _GluSeq(conv3x3(in_planes, out_planes * 2), norm(out_planes * 2), GLU())


class CLSTMCell(nn.Module):
    """Convolutional LSTM cell: one conv over [input, hidden] produces the four gates
    (in, remember, out, cell), reference model.py:819-862."""

    def __init__(self, input_size, hidden_size, kernel_size, padding):
        super(CLSTMCell, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=padding)

    def forward(self, input_, prev_state):
        if prev_state is None:
            size = [input_.size(0), self.hidden_size] + list(input_.shape[2:])
            prev_state = (torch.zeros(size, device=input_.device), torch.zeros(size, device=input_.device))
        prev_hidden, prev_cell = prev_state
        gates = ops.conv2d(torch.cat((input_, prev_hidden), 1), self.Gates.weight, self.Gates.bias,
                           1, self.Gates.padding[0], "zeros")
        in_gate, remember_gate, out_gate, cell_gate = gates.chunk(4, 1)
        cell = torch.sigmoid(remember_gate) * prev_cell + torch.sigmoid(in_gate) * torch.tanh(cell_gate)
        hidden = torch.sigmoid(out_gate) * torch.tanh(cell)
        return hidden, cell


class ResBlock(nn.Module):
    """conv-norm-GLU x2 + conv3x3, plus the input (reference model.py:865-881)."""

    def __init__(self, channel_num, norm=nn.BatchNorm2d):
        super(ResBlock, self).__init__()
        self.block = nn.Sequential(
            conv3x3(channel_num, channel_num * 2), norm(channel_num * 2), GLU(),
            conv3x3(channel_num, channel_num * 2), norm(channel_num * 2), GLU(),
            conv3x3(channel_num, channel_num))

    def forward(self, x):
        out = x
        for i in (0, 3):
            y = ops.conv2d(out, self.block[i].weight, None, 1, 1, "zeros")
            norm = self.block[i + 1]
            out = _in_act(y, norm, "glu") if isinstance(norm, nn.InstanceNorm2d) else _bn_act(y, norm, "glu")
        return ops.conv2d(out, self.block[6].weight, None, 1, 1, "zeros") + x


class GET_SHAPE_G(nn.Module):
    def __init__(self, nbf):
        super(GET_SHAPE_G, self).__init__()
        self.img = nn.Sequential(conv1x1(nbf, 1), nn.Sigmoid())

    def forward(self, h_code):
        return ops.conv2d(h_code, self.img[0].weight, None, 1, 0, "zeros", act="sigmoid")


class SHP_G_NET(nn.Module):
    """Box layouts -> per-box instance masks: bidirectional ConvLSTM over the box sequence at
    16x16, residual blocks, two upBlocks back to 64x64 (reference model.py:898-985)."""

    def __init__(self, num_classes):
        super(SHP_G_NET, self).__init__()
        self.nbf = num_classes
        self.fm_size = cfg.ROI.FM_SIZE
        self.downsample1 = downBlock_3x3(self.nbf, self.nbf * 2)
        self.downsample2 = downBlock_3x3(self.nbf * 2, self.nbf * 4)
        self.fwd_convlstm = CLSTMCell(self.nbf * 4, self.nbf * 2, 3, 1)
        self.bwd_convlstm = CLSTMCell(self.nbf * 4, self.nbf * 2, 3, 1)
        self.jointConv = Block3x3_relu(self.nbf * 8, self.nbf * 4, norm=nn.InstanceNorm2d)
        self.residual = self._make_layer(ResBlock, self.nbf * 4)
        self.upsample1 = upBlock(self.nbf * 4, self.nbf * 2, norm=nn.InstanceNorm2d)
        self.upsample2 = upBlock(self.nbf * 2, self.nbf, norm=nn.InstanceNorm2d)
        self.img_net = GET_SHAPE_G(self.nbf)

    def _make_layer(self, block, channel_num):
        return nn.Sequential(*[block(channel_num, norm=nn.InstanceNorm2d) for _ in range(cfg.GAN.R_NUM)])

    def forward(self, z_code, bbox_maps_fwd, bbox_maps_bwd, bbox_fmaps):
        """z_code [B, R, 4 * nbf] (evaluator.py:278-279); bbox_maps_fwd/bwd [B, R, nbf, S, S]; bbox_fmaps [B, R, 16, 16]
        -> fake_hmaps [B, R, 1, S, S]"""
        B, R = z_code.size(0), z_code.size(1)
        fm = self.fm_size
        z = z_code.unsqueeze(3).repeat(1, 1, 1, fm * fm).view(B, R, -1, fm, fm)
        S = bbox_maps_fwd.size(3)

        def down(maps):
            h = self.downsample2(self.downsample1(maps.reshape(-1, self.nbf, S, S)))
            return h.view(B, R, -1, fm, fm)
        h_fwd, h_bwd = down(bbox_maps_fwd), down(bbox_maps_bwd)
        state_fwd, state_bwd, fwd_lst, bwd_lst = None, None, [], []
        for t in range(R):
            state_fwd = self.fwd_convlstm(h_fwd[:, t], state_fwd)
            fwd_lst.append(state_fwd[0].unsqueeze(1))
            state_bwd = self.bwd_convlstm(h_bwd[:, t], state_bwd)
            bwd_lst.append(state_bwd[0].unsqueeze(1))
        h_code = torch.cat([torch.cat((fwd_lst[t], bwd_lst[R - t - 1], z[:, t:t + 1]), 2) for t in range(R)], 1)
        h_code = self.jointConv(h_code.view(B * R, -1, fm, fm))
        h_code = h_code * bbox_fmaps.reshape(B * R, 1, fm, fm)            # crop to the box
        h_code = self.residual(h_code)
        h_code = self.upsample2(self.upsample1(h_code))
        return self.img_net(h_code).view(B, R, -1, S, S)


def Block3x3_leakRelu(in_planes, out_planes):
    return _JointBlock(conv3x3(in_planes, out_planes),
                        nn.BatchNorm2d(out_planes),
                        nn.LeakyReLU(0.2, inplace=True))


class _Encoder(nn.Sequential):
    """[conv4x4 s2, LReLU, (conv4x4 s2, BN, LReLU) x (n-1)]: spatial size / 2^n."""

    def forward(self, x, x2=None):
        """x2: second part of the input (channels behind x's): conv(cat([x, x2])) with per-part data gradients"""
        mods = list(self)
        i = 0
        while i < len(mods):
            conv = mods[i]
            if isinstance(mods[i + 1], nn.BatchNorm2d):
                if x2 is not None:
                    raise ValueError("_Encoder: a two-part input needs an activation-only first layer")
                y = ops.conv2d(x, conv.weight, None, 2, 1, "zeros")
                x = _bn_act(y, mods[i + 1], "lrelu")
                i += 3
            elif x2 is not None:
                x = ops.conv2d_cat(x, x2, conv.weight, 2, 1, act="lrelu")
                x2 = None
                i += 2
            else:
                x = ops.conv2d(x, conv.weight, None, 2, 1, "zeros", act="lrelu")
                i += 2
        return x


def encode_image_by_ntimes(ngf, ndf, n_layer):
    layers = [nn.Conv2d(3 + ngf, ndf, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True)]
    for n in range(1, n_layer):
        c_prev = ndf * min(2 ** (n - 1), 8)
        c_next = ndf * min(2 ** n, 8)
        layers += [nn.Conv2d(c_prev, c_next, 4, 2, 1, bias=False),
                   nn.BatchNorm2d(c_next),
                   nn.LeakyReLU(0.2, inplace=True)]
    return _Encoder(*layers)


class _ProbHead(nn.Sequential):
    """[Conv2d(c, 1, k4, s2, bias), Sigmoid] -> probabilities."""

    def forward(self, x):
        conv = self[0]
        return ops.conv2d(x, conv.weight, conv.bias, 2, 0, "zeros", act="sigmoid")


class D_GET_LOGITS(nn.Module):
    def __init__(self, ndf, nef, bcondition=False):
        super(D_GET_LOGITS, self).__init__()
        self.df_dim = ndf
        self.ef_dim = nef
        self.layer_num = cfg.GAN.LAYER_D_NUM
        self.bcondition = bcondition
        width = ndf * pow(2, self.layer_num - 1)
        if self.bcondition:
            self.jointConv = Block3x3_leakRelu(width + nef, width)
        self.outlogits = _ProbHead(nn.Conv2d(width, 1, kernel_size=4, stride=2), nn.Sigmoid())

    def forward(self, h_code, c_code=None):
        if self.bcondition and c_code is not None:
            c_code = c_code.view(-1, self.ef_dim, 1, 1).expand(-1, -1, h_code.size(2), h_code.size(3))
            h_c_code = self.jointConv(torch.cat((h_code, c_code), 1))
        else:
            h_c_code = h_code
        return self.outlogits(h_c_code)


class _PatD(nn.Module):
    """Patch discriminator body + (un)conditional heads (reference model.py:1053-1106); the
    heads are invoked by the loss code on the returned features."""

    def __init__(self, b_jcu=True):
        super(_PatD, self).__init__()
        ndf = cfg.GAN.DF_DIM
        nef = cfg.TEXT.EMBEDDING_DIM
        self.img_code = encode_image_by_ntimes(0, ndf, cfg.GAN.LAYER_D_NUM)
        self.UNCOND_DNET = D_GET_LOGITS(ndf, nef, bcondition=False) if b_jcu else None
        self.COND_DNET = D_GET_LOGITS(ndf, nef, bcondition=True)

    def forward(self, x_var):
        return self.img_code(x_var)


class PAT_D_NET64(_PatD):
    pass


class PAT_D_NET128(_PatD):
    pass


class PAT_D_NET256(_PatD):
    pass


class _ShpD(nn.Module):
    """Shape discriminator (reference model.py:1111-1179): image + encoded layout map."""

    def __init__(self, num_classes):
        super(_ShpD, self).__init__()
        ndf = cfg.GAN.DF_DIM
        nef = cfg.TEXT.EMBEDDING_DIM
        ngf = cfg.GAN.GF_DIM // 4
        self.img_code = encode_image_by_ntimes(ngf, ndf, cfg.GAN.LAYER_D_NUM)
        self.shp_code = _shape_stem(num_classes, ngf)
        self.UNCOND_DNET = D_GET_LOGITS(ndf, nef, bcondition=False)

    def encode_seg(self, s_var):
        """shp_code(s_var): a per-sample function of the layout map only, so the real and the fake
        pass of one loss evaluation share it (miscc/losses.py); autograd sums both uses."""
        return self.shp_code(s_var)

    def forward(self, x_var, s_var, s_code=None):
        if s_code is None:
            s_code = self.shp_code(s_var)
        return self.img_code(x_var, s_code)


class SHP_D_NET64(_ShpD):
    pass


class SHP_D_NET128(_ShpD):
    pass


class SHP_D_NET256(_ShpD):
    pass


def _rois_blob(fm_rois, boxes_num):
    """[x, y, w, h] box slots -> ROIAlign rows [batch_idx, x1, y1, x2, y2] (float32), on the
    device.  Mirrors the reference host code (model.py:1213-1214, 1237-1240 and
    miscc/utils.py:365-399): the corner add is done in float64 and rounded to float32 once; the
    batch index of row r is r // BOXES_NUM; ALL slots are pooled, padded ones included."""
    B = fm_rois.shape[0]
    box = fm_rois[:, :, :4].to(torch.float64)
    x1y1 = box[:, :, 0:2]
    x2y2 = x1y1 + box[:, :, 2:4]
    idx = torch.arange(B, device=fm_rois.device, dtype=torch.float64).view(B, 1, 1).expand(B, box.shape[1], 1)
    blob = torch.cat((idx, x1y1, x2y2), dim=2).reshape(B * box.shape[1], 5)
    return blob.to(torch.float32).contiguous()


class _ObjD(nn.Module):
    """ROIAlign-based object discriminator (reference model.py:1184-1312)."""

    n_layer = 3

    def __init__(self, num_classes, b_jcu=True):
        super(_ObjD, self).__init__()
        ndf = cfg.GAN.DF_DIM
        nef = cfg.TEXT.GLOVE_EMBEDDING_DIM + cfg.GAN.GF_DIM
        ngf = cfg.GAN.GF_DIM // 4
        self.roi_size = cfg.ROI.ROI_BASE_SIZE
        self.im_scales = np.array([1])
        n_layer = self.n_layer
        self.img_code = encode_image_by_ntimes(ngf, ndf, n_layer)
        self.shp_code = _shape_stem(num_classes, ngf)
        self.roi_code = _ActSeq(
            nn.Conv2d(ndf * min(2 ** (n_layer - 1), 8), ndf * 4, kernel_size=4, stride=1, padding=1),
            nn.LeakyReLU(0.2, True))
        # NB (reference quirk kept, SURVEY.md trap 2): spatial_scale 1/16 is applied to boxes that
        # are already in feature-map coordinates.
        self.RoIAlignAvg = RoIAlignAvg(self.roi_size, self.roi_size, 1.0 / 16.0)
        self.UNCOND_DNET = D_GET_LOGITS(ndf // 2, nef, bcondition=False) if b_jcu else None
        self.COND_DNET = D_GET_LOGITS(ndf // 2, nef, bcondition=True)

    def encode_seg(self, s_var, img_size=512):
        """shp_code(F.interpolate(s_var, img_size, bilinear, align_corners)) (reference model.py:1217-1226)
        WITHOUT the lifted map: the 80 -> 12 channel contraction runs at the source resolution, the lift /
        reflect-pad / tap shifts are applied to its 9 x 12 planes (objgan_hip.ops.lift_stem_conv).  Depends
        on the layout map only, so the real and fake passes of one loss evaluation share the result."""
        conv, inorm = self.shp_code[1], self.shp_code[2]
        y = ops.lift_stem_conv(s_var, conv.weight, conv.bias, img_size)
        return _in_act(y, inorm, "lrelu")

    def forward(self, x_var, s_var, fm_rois, num_rois, img_size=512, s_code=None):
        x_var = ops.bilinear_resize(x_var, img_size, img_size)
        if s_code is None:
            s_code = self.encode_seg(s_var, img_size)
        x_code = self.img_code(x_var, s_code)
        batch_size = fm_rois.shape[0]
        rois = _rois_blob(fm_rois, cfg.ROI.BOXES_NUM)
        pooled = self.roi_code(self.RoIAlignAvg(x_code, rois))
        return pooled.view(batch_size, cfg.ROI.BOXES_NUM, pooled.size(1), pooled.size(2), pooled.size(3))


class OBJ_SS_D_NET(_ObjD):
    n_layer = 3


class OBJ_LS_D_NET(_ObjD):
    n_layer = 4
