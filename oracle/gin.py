"""GIN random-convolution intensity augmentation, CPU restatement (oracle; test infrastructure).

Follows /root/reference/dg_tta/gin.py:59-122 (GradlessGCReplayNonlinBlock.forward),
:168-230 (GINGroupConv.forward) and :233-241 (gin_aug: IN_CHANNELS=1, N_LAYER=4,
INTERM_CHANNELS=2).  Random draws are made explicit so that a HIP kernel can be fed the
same numbers; `draw_gin_params` reproduces the reference's draw ORDER:
rand(nb) [x.device] -> per layer: randint(2) -> randn(ker) -> randn(shift) [CPU].
"""
import torch
import torch.nn.functional as F

N_LAYER = 4
INTERM = 2
SCALE_POOL = (1, 3)


def layer_channels(in_ch=1):
    """(c_in, c_out) per layer: 1->2->2->2->1."""
    chans = [in_ch] + [INTERM] * (N_LAYER - 1) + [in_ch]
    return list(zip(chans[:-1], chans[1:]))


def draw_gin_params(nb, in_ch=1, device="cpu"):
    """Draws (alpha[nb], ksizes[4], kernels[4], shifts[4]) in the reference's order."""
    alpha = torch.rand(nb, device=device)                       # gin.py:193-195 (after layer setup, before convs)
    ks, kers, shifts = [], [], []
    for cin, cout in layer_channels(in_ch):
        k = SCALE_POOL[int(torch.randint(high=len(SCALE_POOL), size=(1,))[0])]   # gin.py:65-66
        kers.append(torch.randn([cout * nb, cin, k, k, k]))     # gin.py:94-97
        shifts.append(torch.randn([cout * nb, 1, 1, 1]) * 1.0)  # gin.py:98-103
        ks.append(k)
    return alpha, ks, kers, shifts


def gin_chain(x_in, alpha, ks, kers, shifts):
    """GINGroupConv.forward with explicit draws. x_in [nb,nc,D,H,W] fp32."""
    nb, nc = x_in.shape[:2]
    sp = x_in.shape[2:]
    x = x_in
    for li, (k, ker, sh) in enumerate(zip(ks, kers, shifts)):
        c_out = ker.shape[0] // nb
        y = F.conv3d(x.reshape(1, -1, *sp), ker, stride=1, padding=k // 2, dilation=1, groups=nb)
        y = y + sh
        if li < len(ks) - 1:
            y = F.leaky_relu(y)                                  # slope 0.01 (gin.py:112-113)
        x = y.reshape(nb, c_out, *sp)
    a = alpha.view(nb, 1, 1, 1, 1).repeat(1, nc, 1, 1, 1)
    mixed = a * x + (1.0 - a) * x_in                             # gin.py:203
    in_frob = torch.norm(x_in.reshape(nb, nc, -1), dim=(-1, -2), p="fro")
    self_frob = torch.norm(mixed.reshape(nb, nc, -1), dim=(-1, -2), p="fro")
    in_frob = in_frob.view(nb, 1, 1, 1, 1)
    self_frob = self_frob.view(nb, 1, 1, 1, 1)
    return mixed * (1.0 / (self_frob + 1e-5)) * in_frob          # gin.py:228


def gin_aug(x_in):
    alpha, ks, kers, shifts = draw_gin_params(x_in.shape[0], x_in.shape[1], x_in.device)
    return gin_chain(x_in, alpha, ks, kers, shifts)
