"""ORACLE (test infrastructure, CPU only): plain-PyTorch restatement of the frozen encoders around the
hot path -- the torchvision Inception-v3 trunk (torchvision is not installed here; module names and
state-dict keys are torchvision's), the DAMSM image encoder CNN_ENCODER (reference
image_generation/model.py:182-287) and the Inception-score monitor INCEPTION_V3 (model.py:290-315).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(obj-gan_amd/encoders.py) holds the same module tree on the gfx950 kernels and has no CPU path; the two
exchange weights through their state dicts (`cpu_twin`).  It also serves as the `torchvision.models`
stand-in when the UNMODIFIED reference is imported by oracle/ref_harness.py.

Parity unpinned against torchvision itself (absent, no network): the restatement follows the published
torchvision inception.py layer list; what IS pinned is product-vs-oracle on identical seeded weights.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv2d(nn.Module):
    """conv (no bias) -> BatchNorm(eps 1e-3) -> ReLU."""

    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def _chain(cin, spec):
    """spec: list of (cout, kernel, stride, padding) -> nn.ModuleList of BasicConv2d."""
    mods = []
    for cout, k, s, p in spec:
        mods.append(BasicConv2d(cin, cout, kernel_size=k, stride=s, padding=p))
        cin = cout
    return mods


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1, self.branch5x5_2 = _chain(cin, [(48, 1, 1, 0), (64, 5, 1, 2)])
        self.branch3x3dbl_1, self.branch3x3dbl_2, self.branch3x3dbl_3 = _chain(
            cin, [(64, 1, 1, 0), (96, 3, 1, 1), (96, 3, 1, 1)])
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)

    def forward(self, x):
        return torch.cat([
            self.branch1x1(x),
            self.branch5x5_2(self.branch5x5_1(x)),
            self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
            self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1)),
        ], 1)


class InceptionB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1, self.branch3x3dbl_2, self.branch3x3dbl_3 = _chain(
            cin, [(64, 1, 1, 0), (96, 3, 1, 1), (96, 3, 2, 0)])

    def forward(self, x):
        return torch.cat([
            self.branch3x3(x),
            self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
            F.max_pool2d(x, kernel_size=3, stride=2),
        ], 1)


class InceptionC(nn.Module):
    def __init__(self, cin, channels_7x7):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1, self.branch7x7_2, self.branch7x7_3 = _chain(
            cin, [(c7, 1, 1, 0), (c7, (1, 7), 1, (0, 3)), (192, (7, 1), 1, (3, 0))])
        (self.branch7x7dbl_1, self.branch7x7dbl_2, self.branch7x7dbl_3, self.branch7x7dbl_4,
         self.branch7x7dbl_5) = _chain(cin, [(c7, 1, 1, 0), (c7, (7, 1), 1, (3, 0)),
                                             (c7, (1, 7), 1, (0, 3)), (c7, (7, 1), 1, (3, 0)),
                                             (192, (1, 7), 1, (0, 3))])
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = x
        for m in (self.branch7x7dbl_1, self.branch7x7dbl_2, self.branch7x7dbl_3, self.branch7x7dbl_4,
                  self.branch7x7dbl_5):
            bd = m(bd)
        return torch.cat([self.branch1x1(x), b7, bd,
                          self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))], 1)


class InceptionD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1, self.branch3x3_2 = _chain(cin, [(192, 1, 1, 0), (320, 3, 2, 0)])
        (self.branch7x7x3_1, self.branch7x7x3_2, self.branch7x7x3_3, self.branch7x7x3_4) = _chain(
            cin, [(192, 1, 1, 0), (192, (1, 7), 1, (0, 3)), (192, (7, 1), 1, (3, 0)), (192, 3, 2, 0)])

    def forward(self, x):
        b7 = x
        for m in (self.branch7x7x3_1, self.branch7x7x3_2, self.branch7x7x3_3, self.branch7x7x3_4):
            b7 = m(b7)
        return torch.cat([self.branch3x3_2(self.branch3x3_1(x)), b7,
                          F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionE(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        return torch.cat([self.branch1x1(x), b3, bd,
                          self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))], 1)


class InceptionAux(nn.Module):
    """Holder of torchvision's auxiliary head (`AuxLogits.*` checkpoint keys); unused in eval mode."""

    def __init__(self, cin, num_classes):
        super().__init__()
        self.conv0 = BasicConv2d(cin, 128, kernel_size=1)
        self.conv1 = BasicConv2d(128, 768, kernel_size=5)
        self.fc = nn.Linear(768, num_classes)


class Inception3(nn.Module):
    """Inception-v3 with torchvision's module names and state-dict keys."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.AuxLogits = InceptionAux(768, num_classes)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.fc = nn.Linear(2048, num_classes)

    def trunk(self, x, want_regions=False):
        return inception_trunk(self, x, want_regions)

    def forward(self, x):
        return self.fc(self.trunk(x))


TRUNK_MODULES = ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3", "Conv2d_3b_1x1", "Conv2d_4a_3x3",
                 "Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d",
                 "Mixed_6e", "Mixed_7a", "Mixed_7b", "Mixed_7c")


def inception_trunk(m, x, want_regions=False):
    """299x299 image -> (17x17x768 region features, 2048-d pooled code); `m` is any module that
    owns the trunk blocks under their torchvision names."""
    x = m.Conv2d_2b_3x3(m.Conv2d_2a_3x3(m.Conv2d_1a_3x3(x)))
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    x = m.Conv2d_4a_3x3(m.Conv2d_3b_1x1(x))
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    x = m.Mixed_5d(m.Mixed_5c(m.Mixed_5b(x)))
    x = m.Mixed_6e(m.Mixed_6d(m.Mixed_6c(m.Mixed_6b(m.Mixed_6a(x)))))
    regions = x                                     # 768 x 17 x 17
    x = m.Mixed_7c(m.Mixed_7b(m.Mixed_7a(x)))
    x = F.avg_pool2d(x, kernel_size=8).flatten(1)   # 2048
    return (regions, x) if want_regions else x


def inception_v3(**kw):
    return Inception3(**kw)


def seeded_init_(module, seed):
    """Deterministic random weights (He-style scale so activations stay O(1)); identical on every
    machine, so the oracle and the product path share the same frozen encoder without shipping
    a checkpoint."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if name.startswith("AuxLogits.") or ".AuxLogits." in name:
                p.zero_()                   # never-executed holder: kept out of the random stream
                continue
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            elif name.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    return module


class CNN_ENCODER(nn.Module):
    """Image encoder of the DAMSM loss (reference model.py:182-287): frozen Inception-v3 trunk +
    two projections (`emb_features` 1x1 conv 768 -> nef, `emb_cnn_code` Linear 2048 -> nef)."""

    def __init__(self, nef, trunk=None):
        super().__init__()
        self.nef = nef
        net = trunk if trunk is not None else inception_v3()
        for p in net.parameters():
            p.requires_grad = False
        for name in TRUNK_MODULES:          # same attribute / state-dict names as the reference
            setattr(self, name, getattr(net, name))
        self.emb_features = nn.Conv2d(768, nef, kernel_size=1, stride=1, padding=0, bias=False)
        self.emb_cnn_code = nn.Linear(2048, nef)
        self.emb_features.weight.data.uniform_(-0.1, 0.1)
        self.emb_cnn_code.weight.data.uniform_(-0.1, 0.1)

    def forward(self, x):
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=True)
        regions, code = inception_trunk(self, x, want_regions=True)
        return self.emb_features(regions), self.emb_cnn_code(code)


class INCEPTION_V3(nn.Module):
    """Per-step Inception-score monitor (reference model.py:290-315)."""

    def __init__(self, net=None):
        super().__init__()
        self.model = net if net is not None else inception_v3()
        for p in self.model.parameters():
            p.requires_grad = False
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)

    def forward(self, input):
        x = (input * 0.5 + 0.5 - self.mean) / self.std
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=True)
        return F.softmax(self.model(x), dim=-1)


def cpu_twin(module):
    """The plain-PyTorch twin of a product encoder (obj-gan_amd/encoders.py CNN_ENCODER / INCEPTION_V3 /
    Inception3): same tree, weights copied through the state dict."""
    name = type(module).__name__
    if name == "CNN_ENCODER":
        twin = CNN_ENCODER(module.nef)
    elif name == "INCEPTION_V3":
        twin = INCEPTION_V3()
    elif name == "Inception3":
        twin = Inception3()
    else:
        raise TypeError("no CPU twin for %s" % name)
    twin.load_state_dict({k: v.detach().cpu() for k, v in module.state_dict().items()})
    for p in twin.parameters():
        p.requires_grad_(False)
    return twin.eval()


class CpuImageEncoder(object):
    """callable(image) -> (regions, code) on the CPU from a product or oracle CNN_ENCODER."""

    def __init__(self, enc):
        self.enc = enc if type(enc).__module__ == __name__ else cpu_twin(enc)

    def __call__(self, x):
        return self.enc(x)
