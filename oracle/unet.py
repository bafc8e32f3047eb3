"""nnUNet 3d_fullres PlainConvUNet, CPU restatement (oracle; test infrastructure).

The network class is third-party: dynamic-network-architectures==0.2 (poetry.lock of the
reference; pulled in by nnunetv2==2.2.1) and is NOT under /root/reference.  It is built at
/root/reference/dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53 (12 input channels) from
the topology in dg_tta/__resources__/dummy_results/*/plans.json:279-401 and the 105 labels
of dataset.json.  Its arithmetic is nothing but torch.nn.{Conv3d, InstanceNorm3d, LeakyReLU,
ConvTranspose3d} + torch.cat, restated here with the published module/key layout
(encoder.stages.S.0.convs.I.{conv,norm,all_modules.{0,1}}, decoder.{encoder,stages,
transpconvs,seg_layers}) so that a real `checkpoint_final.pth` state-dict loads unchanged.
parity unpinned by the reference (it has no tests); pinned by torch CPU semantics.
"""
import torch
from torch import nn

PLANS_3D_FULLRES = dict(features=(32, 64, 128, 256, 320), strides=(1, 2, 2, 2, 2),
                        n_conv_enc=(2, 2, 2, 2, 2), n_conv_dec=(2, 2, 2, 2),
                        in_channels=12, num_classes=105)


class ConvNormAct(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 3, stride, 1, bias=True)
        self.norm = nn.InstanceNorm3d(cout, eps=1e-5, affine=True)
        self.nonlin = nn.LeakyReLU(negative_slope=1e-2, inplace=True)
        self.all_modules = nn.Sequential(self.conv, self.norm, self.nonlin)

    def forward(self, x):
        return self.all_modules(x)


class StackedConvs(nn.Module):
    def __init__(self, n, cin, cout, first_stride):
        super().__init__()
        self.convs = nn.Sequential(*[ConvNormAct(cin if i == 0 else cout, cout,
                                                 first_stride if i == 0 else 1) for i in range(n)])

    def forward(self, x):
        return self.convs(x)


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        stages, cin = [], cfg["in_channels"]
        for f, s, n in zip(cfg["features"], cfg["strides"], cfg["n_conv_enc"]):
            stages.append(nn.Sequential(StackedConvs(n, cin, f, s)))
            cin = f
        self.stages = nn.Sequential(*stages)

    def forward(self, x):
        skips = []
        for st in self.stages:
            x = st(x)
            skips.append(x)
        return skips


class Decoder(nn.Module):
    def __init__(self, encoder, cfg):
        super().__init__()
        self.encoder = encoder          # registered again, as in the published class (duplicate keys)
        f, st = cfg["features"], cfg["strides"]
        stages, ups, segs = [], [], []
        for s in range(1, len(f)):
            below, skip = f[-s], f[-(s + 1)]
            ups.append(nn.ConvTranspose3d(below, skip, st[-s], st[-s], bias=True))
            stages.append(StackedConvs(cfg["n_conv_dec"][s - 1], 2 * skip, skip, 1))
            segs.append(nn.Conv3d(skip, cfg["num_classes"], 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(ups)
        self.seg_layers = nn.ModuleList(segs)

    def forward(self, skips):
        x = skips[-1]
        for s in range(len(self.stages)):
            x = self.transpconvs[s](x)
            x = torch.cat((x, skips[-(s + 2)]), 1)
            x = self.stages[s](x)
        return self.seg_layers[-1](x)   # deep supervision off (nnUNetPredictor builds it that way)


class PlainConvUNetOracle(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = dict(PLANS_3D_FULLRES if cfg is None else cfg)
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(self.encoder, cfg)

    def forward(self, x):
        return self.decoder(self.encoder(x))


def init_he(model, seed):
    """nnUNet's InitWeights_He(1e-2) [3P]: kaiming_normal_(a=1e-2) on conv/convT weights, zero bias;
    InstanceNorm affine stays (1, 0).  Seeded stand-in for the TS104 checkpoints (not downloadable)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
            fan_in = m.weight.shape[1] * m.weight[0, 0].numel()
            gain = (2.0 / (1 + 1e-2 ** 2)) ** 0.5
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (gain / fan_in ** 0.5))
                if m.bias is not None:
                    m.bias.zero_()
    return model


def perturb_affine(model, seed, scale=0.1):
    """Makes norm gamma/beta and conv biases non-trivial so that parity tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.InstanceNorm3d):
                m.weight.add_(scale * torch.randn(m.weight.shape, generator=g))
                m.bias.add_(scale * torch.randn(m.bias.shape, generator=g))
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                m.bias.add_(scale * torch.randn(m.bias.shape, generator=g))
    return model
