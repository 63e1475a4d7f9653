"""Parity at BASELINE.json's full sizes (per-GPU batch 16, 256x256) through properties that do not
need a CPU evaluation of the full-size operator:

  * convolution: <conv(x, w), g> = <x, dgrad(g)> = <w, wgrad(x, g)> (the three kernels are mutually
    adjoint) and linearity in x, on the largest layers of the step;
  * ROIAlign: the full-size object-discriminator call, BIT-EXACT against the C oracle (2.2 M outputs,
    seconds on the host);
  * normalisation: per-plane mean 0 / variance 1 before the GLU, GLU consistency with the LeakyReLU-free
    path; attention: the maps are distributions over the words; Adam: one fused step over a 77 M-element
    arena against the closed form on a strided sample.

First hardware run: profiles/r02_fullsize_tests_first_run.log (round 2).
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = [pytest.mark.gpu]

B = 16


def _ops():
    from objgan_hip import ops
    return ops


def _dot(a, b):
    return float((a.double() * b.double()).sum())


FULL_CONVS = [
    # Cin, H, W, Cout, k, stride, pad, pad_mode, upsample
    (194, 128, 128, 388, 3, 1, 1, "reflect", False),      # HmapResBlock conv 1 at 128^2
    (194, 128, 128, 194, 3, 1, 1, "reflect", False),      # HmapResBlock conv 2
    (194, 128, 128, 96, 3, 1, 1, "zeros", True),          # upBlock 128^2 -> 256^2 (phased form)
    (96, 256, 256, 192, 4, 2, 1, "zeros", False),         # object-discriminator encoder layer 2
    (80, 256, 256, 24, 3, 1, 1, "reflect", False),        # G_HMAP stem (thin VALU kernels)
    (384, 64, 64, 768, 4, 2, 1, "zeros", False),          # discriminator layer 4
]


@pytest.mark.parametrize("case", FULL_CONVS)
def test_conv_kernels_are_mutually_adjoint_at_full_size(dev, case):
    ops = _ops()
    Cin, H, W, Cout, k, s, p, pm, up = case
    g = torch.Generator(device="cpu").manual_seed(99)
    x = (torch.randn(B, Cin, H, W, generator=g)).to(dev).requires_grad_()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev).requires_grad_()
    y = ops.conv2d(x, w, None, s, p, pm, up, None)
    gy = torch.randn(y.shape, generator=g).to(dev)
    y.backward(gy)
    torch.cuda.synchronize()
    lhs = _dot(y.detach(), gy)
    assert abs(_dot(x.detach(), x.grad) - lhs) <= 2e-4 * abs(lhs) + 1e-3, ("dgrad", lhs, _dot(x.detach(), x.grad))
    assert abs(_dot(w.detach(), w.grad) - lhs) <= 2e-4 * abs(lhs) + 1e-3, ("wgrad", lhs, _dot(w.detach(), w.grad))
    # linearity in x (bias-free, no activation)
    x2 = torch.randn(B, Cin, H, W, generator=g).to(dev)
    with torch.no_grad():
        y2 = ops.conv2d(x2, w, None, s, p, pm, up, None)
        y12 = ops.conv2d(0.5 * x.detach() + x2, w, None, s, p, pm, up, None)
    assert rel_l2(y12, 0.5 * y.detach() + y2) < 1e-5


def test_roi_align_full_size_is_bit_exact(dev):
    """OBJ_SS_D_NET's call: features [16, 384, 64, 64], 160 boxes of the synthetic batch, 6x6 bins."""
    ops = _ops()
    from oracle import roi as oroi
    import model as M
    import synth_batch
    b = synth_batch.make_batch(B, seed=1234)
    rois = M._rois_blob(b["fm_rois"], 10).numpy()
    rng = np.random.RandomState(3)
    feat = rng.randn(B, 384, 64, 64).astype(np.float32)
    want = oroi.forward(feat, rois, 6, 6, 1.0 / 16.0)
    got = ops.roi_align(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev), 6, 6, 1.0 / 16.0)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_instance_norm_glu_statistics_at_full_size(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, 388, 128, 128, generator=g) * 3.0 + 1.5).to(dev)
    y = ops.norm_act(x, mode="glu")                          # InstanceNorm2d + GLU (HmapResBlock)
    z = ops.norm_act(x)                                      # InstanceNorm2d alone
    torch.cuda.synchronize()
    m = z.double().mean(dim=(2, 3))
    v = z.double().var(dim=(2, 3), unbiased=False)
    assert float(m.abs().max()) < 1e-4 and float((v - 1).abs().max()) < 1e-3
    want = z[:, :194] * torch.sigmoid(z[:, 194:])
    assert y.shape == (B, 194, 128, 128) and rel_l2(y, want) < 1e-5


def test_attention_maps_are_distributions_at_full_size(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    L = 12
    x = torch.randn(B, 48, 128, 128, generator=g).to(dev)
    src = torch.randn(B, 48, L, generator=g).to(dev)
    lens = torch.randint(5, L + 1, (B,), generator=g)
    mask = (torch.arange(L).unsqueeze(0) >= lens.unsqueeze(1)).to(dev)
    wc, attn = ops.attn_general(x, src, mask)
    torch.cuda.synchronize()
    assert wc.shape == (B, 48, 128, 128) and attn.shape == (B, L, 128, 128)
    s = attn.double().sum(dim=1)
    assert float((s - 1).abs().max()) < 1e-5 and float(attn.min()) >= 0.0
    # the weighted context is the attention-weighted sum of the source columns
    want = torch.einsum("bcl,blhw->bchw", src.double(), attn.double())
    assert rel_l2(wc, want.float()) < 1e-5


def test_fused_adam_over_a_generator_sized_arena(dev):
    ops = _ops()
    n = 77_400_000 // 4 * 4
    g = torch.Generator(device=dev).manual_seed(7)
    p = torch.randn(n, device=dev, generator=g)
    gr = torch.randn(n, device=dev, generator=g)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    p0 = p.clone()
    lr, b1, b2, eps = 2e-4, 0.5, 0.999, 1e-8
    ops.adam_step_(p, gr, m, v, lr, b1, b2, eps, 1)
    torch.cuda.synchronize()
    idx = torch.arange(0, n, 9973, device=dev)
    gs, ps = gr[idx].double(), p0[idx].double()
    m1, v1 = (1 - b1) * gs, (1 - b2) * gs * gs
    want = ps - lr * (m1 / (1 - b1)) / ((v1 / (1 - b2)).sqrt() + eps)
    assert rel_l2(p[idx], want.float()) < 1e-6
    assert rel_l2(m[idx], m1.float()) < 1e-6 and rel_l2(v[idx], v1.float()) < 1e-6


def test_shape_generator_forward_matches_reference_golden(dev):
    """SHP_G_NET on the gfx950 kernels (ConvLSTM gates, InstanceNorm GLU blocks, phased upBlocks,
    1x1 conv + sigmoid) against the unmodified reference's CPU output (tests/golden/shp_g_ref.pt)."""
    from conftest import ROOT
    import model as M
    import synth_batch
    from oracle import ref_harness as rh
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "shp_g_ref.pt"), weights_only=False)
    net = rh.seeded_state_(M.SHP_G_NET(gold["nbf"]), gold["seed_weights"]).to(dev).eval()
    z, fwd, bwd, fmaps, rois, num = synth_batch.make_shape_inputs(nbf=gold["nbf"])
    with torch.no_grad():
        fake = net(z.to(dev), fwd.to(dev), bwd.to(dev), fmaps.to(dev))
    torch.cuda.synchronize()
    assert fake.shape == gold["fake_hmaps"].shape and rel_l2(fake, gold["fake_hmaps"]) < 1e-3
