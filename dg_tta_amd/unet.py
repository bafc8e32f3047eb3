"""nnUNet 3d_fullres PlainConvUNet executed by the HIP kernels (forward AND backward), as one autograd node.

Drop-in for the network object the reference obtains from nnUNetPredictor (dg_tta/tta/nnunet_utils.py:88-113; class
`dynamic_network_architectures.architectures.unet.PlainConvUNet`, built at
dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53): same state-dict keys (incl. the duplicated `all_modules.*` and
`decoder.encoder.*` entries), `.encoder` attribute (tta.py:210), norm modules whose class name contains
"instancenorm" (torch_utils.py:130-137), forward (pre-)hooks honoured (model_utils.py:22-33).

Data layout in HBM: activations are channels-last [B][D][H][W][C] (fp32, or bf16 storage with fp32 accumulation);
`torch.cat((up, skip), 1)` never happens: the transposed conv writes the first channel half of a pre-allocated
[.., 2C] buffer and the encoder's InstanceNorm+LeakyReLU writes the skip directly into the second half.
"""
import ctypes as C

import os

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, stream_of
from ._state import state_of
from .ops import F32, BF16, F16, _ws, dtype_code, is_cl3d

PLANS_3D_FULLRES = dict(features=(32, 64, 128, 256, 320), strides=(1, 2, 2, 2, 2),
                        n_conv_enc=(2, 2, 2, 2, 2), n_conv_dec=(2, 2, 2, 2),
                        in_channels=12, num_classes=105)
EPS, SLOPE = 1e-5, 1e-2


def _wgrad_on_side_stream():
    import os
    return os.environ.get("DGTTA_WGRAD_STREAM", "1") != "0"


def _pad(c, m):
    return (c + m - 1) // m * m


# ------------------------------------------------------------------------------------------------ parameter holders
class HipConv3d(nn.Module):
    """Parameter holder for a 3x3x3 / 1x1x1 conv (weights in PyTorch layout so checkpoints load unchanged)."""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = cin, cout, k, stride
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout))


class HipConvTranspose3d(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout))


class HipInstanceNorm3d(nn.Module):
    """name contains 'instancenorm' so that release_norms (torch_utils.py:130-137) finds it."""

    def __init__(self, c):
        super().__init__()
        self.num_features, self.eps = c, EPS
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class HipLeakyReLU(nn.Module):
    negative_slope = SLOPE


class ConvNormAct(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = HipConv3d(cin, cout, 3, stride)
        self.norm = HipInstanceNorm3d(cout)
        self.nonlin = HipLeakyReLU()
        self.all_modules = nn.Sequential(self.conv, self.norm, self.nonlin)


class StackedConvs(nn.Module):
    def __init__(self, n, cin, cout, first_stride):
        super().__init__()
        self.convs = nn.Sequential(*[ConvNormAct(cin if i == 0 else cout, cout, first_stride if i == 0 else 1)
                                     for i in range(n)])


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        stages, cin = [], cfg["in_channels"]
        for f, s, n in zip(cfg["features"], cfg["strides"], cfg["n_conv_enc"]):
            stages.append(nn.Sequential(StackedConvs(n, cin, f, s)))
            cin = f
        self.stages = nn.Sequential(*stages)


class Decoder(nn.Module):
    def __init__(self, encoder, cfg):
        super().__init__()
        self.encoder = encoder
        f, st = cfg["features"], cfg["strides"]
        stages, ups, segs = [], [], []
        for s in range(1, len(f)):
            below, skip = f[-s], f[-(s + 1)]
            assert st[-s] == 2, "only stride-2 (kernel-2) transposed convolutions are built"
            ups.append(HipConvTranspose3d(below, skip, 2))
            stages.append(StackedConvs(cfg["n_conv_dec"][s - 1], 2 * skip, skip, 1))
            segs.append(HipConv3d(skip, cfg["num_classes"], 1, 1))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(ups)
        self.seg_layers = nn.ModuleList(segs)


# ------------------------------------------------------------------------------------------------ execution plan
class _Block:
    """One Conv3d+InstanceNorm+LeakyReLU block with its static shape info."""
    __slots__ = ("mod", "cin", "cout", "stride", "cinp", "coutp", "di", "do", "name")


class HipPlainConvUNet(nn.Module):
    """PlainConvUNet whose forward/backward run on hand-written gfx950 kernels.

    act_dtype: torch.float32 (parity mode), torch.bfloat16 or torch.float16 (16-bit storage + MFMA, fp32 accumulation;
    fp16 carries 3 more mantissa bits than bf16 and needs `loss_scale` for its gradients: BASELINE config 5).
    conv_impl: 0 auto (MFMA where covered, else general VALU kernel), 1 force VALU, 2 force MFMA.
    """

    def __init__(self, cfg=None, act_dtype=torch.float32, conv_impl=0):
        super().__init__()
        cfg = dict(PLANS_3D_FULLRES if cfg is None else cfg)
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(self.encoder, cfg)
        self.act_dtype = act_dtype
        self.conv_impl = conv_impl
        # A conv bias in front of InstanceNorm has an identically zero gradient in exact arithmetic (the norm removes
        # the channel mean); autograd in the reference accumulates rounding noise there.  True: report exact zeros and
        # skip the reduction pass; False: compute sum(dy) like autograd does (parity experiments).
        self.exact_zero_bias_grad = False
        # True: backward adds straight into each parameter's .grad (allocated on first use) and returns no gradient
        # tensors to autograd - saves a zero-fill + add per parameter and backward pass.  Tensor hooks on parameters
        # are then not invoked; torch.autograd.grad() w.r.t. parameters is not supported in this mode.
        self.accumulate_grads_in_place = False
        # static loss scale of the fp16 storage path: the TTA loop multiplies the loss gradient by it (tta.tta_epoch) and
        # HipAdamW divides it out; 1 for fp32 / bf16 (their exponent range needs none)
        self.loss_scale = 16384.0 if act_dtype == torch.float16 else 1.0
        self._packed = {}        # id(weight) -> (version, wf, wb)
        self.selected_classes = None   # optional LongTensor: evaluate only these head rows (== map_label 'logits')
        self._fused_warp = None        # (theta on the device, theta on the host) while fuse_output_warp() is active
        self._window_acc = None        # sliding-window target while fuse_window_accumulate() is active

    def __deepcopy__(self, memo):
        # get_model_from_network deep-copies the network per ensemble member: do not drag the packed-weight cache along
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_packed" else copy.deepcopy(v, memo)
        return new

    # -- structure helpers
    def conv_blocks(self):
        enc = [[c for c in st[0].convs] for st in self.encoder.stages]
        dec = [[c for c in st.convs] for st in self.decoder.stages]
        return enc, dec

    def set_selected_classes(self, idx):
        """Fuses map_label(..., 'logits') (torch_utils.py:214-221) into the head: forward returns only these rows."""
        self.selected_classes = None if idx is None else torch.as_tensor(idx, dtype=torch.int32)

    # -- head fused with the inverse warp of its logits (csrc/warp.hip: head_warp_*_kernel)
    def can_fuse_output_warp(self, x_shape, theta_host):
        """True when `forward` can hand back the logits already warped by theta (R_inverse, zeros padding, the TTA grid
        algebra): 16-bit storage, 32 head input channels, 4 / 8 / 12 / 16 selected classes, maps the gather kernel accepts."""
        import os
        if os.environ.get("DGTTA_FUSE_HEAD_WARP", "1") == "0" or self.selected_classes is None:
            return False
        if self.act_dtype not in (torch.float16, torch.bfloat16):
            return False
        b, _, d, h, w = x_shape
        th = theta_host.detach().float().contiguous()
        if tuple(th.shape) != (b, 3, 4) or th.is_cuda:
            return False
        head = self.decoder.seg_layers[-1]
        return bool(_lib.load().dgtta_seghead_warp_supported(th.data_ptr(), b, head.in_channels, int(self.selected_classes.numel()),
                                                             d, h, w, dtype_code(self.act_dtype)))

    def fuse_output_warp(self, theta_dev, theta_host):
        """Context: the next forward returns affine_warp(logits, theta, zeros, tta_grid_algebra) computed by the fused
        head + warp kernels (and its backward runs the fused gather).  Check can_fuse_output_warp first."""
        net = self

        class _Ctx:
            def __enter__(self_):
                net._fused_warp = (theta_dev.float().contiguous(), theta_host.detach().float().contiguous())

            def __exit__(self_, *exc):
                net._fused_warp = None
        return _Ctx()

    # -- head fused with the Gaussian window accumulation of the sliding-window inference (csrc/warp.hip)
    def can_fuse_window_accumulate(self):
        import os
        return (os.environ.get("DGTTA_FUSE_HEAD_ACCUMULATE", "1") != "0" and self.selected_classes is None and
                self.act_dtype in (torch.float16, torch.bfloat16) and self.decoder.seg_layers[-1].in_channels == 32 and
                self.decoder.seg_layers[-1].out_channels <= 112 and not torch.is_grad_enabled())

    def fuse_window_accumulate(self, acc, nsum, gauss, origins):
        """Context (inference, no grad): the next forward adds gauss * logits of window k of the batch into
        acc [X,Y,Z,ncls] / nsum [X,Y,Z] at origins[k] instead of returning the logits (it returns an empty placeholder)."""
        net = self

        if acc.dtype not in (torch.float32, torch.float16):      # the kernel knows these two accumulator storage types
            raise ValueError(f"fuse_window_accumulate: the accumulator is fp32 or fp16, not {acc.dtype}")
        if not (acc.is_contiguous() and nsum.dtype == torch.float32 and gauss.dtype == torch.float32):
            raise ValueError("fuse_window_accumulate: contiguous accumulator, fp32 weight sum and fp32 Gaussian expected")

        class _Ctx:
            def __enter__(self_):
                net._window_acc = (acc, nsum, gauss, list(origins))

            def __exit__(self_, *exc):
                net._window_acc = None
        return _Ctx()

    # -- the same in FEATURE space (round 5, csrc/window_features.hip): the head is linear and last, so the window accumulator
    #    holds the Gaussian-weighted input of the head and the head runs once per voxel at the end
    def can_fuse_window_feature_accumulate(self):
        import os
        return (os.environ.get("DGTTA_FUSE_HEAD_ACCUMULATE", "1") != "0" and self.selected_classes is None and
                self.decoder.seg_layers[-1].in_channels == 32 and not torch.is_grad_enabled())

    def fuse_window_feature_accumulate(self, facc, nsum, gauss, origins):
        """Context (inference, no grad): the next forward adds gauss * z of window k of the batch - z = the 32 feature channels the
        segmentation head reads - into facc [X,Y,Z,32] (fp32) / nsum [X,Y,Z] at origins[k]; the head is NOT evaluated (the forward
        returns an empty placeholder).  Label map: ops.feature_head_argmax with the members' head weights."""
        net = self
        if not (facc.dtype == torch.float32 and facc.is_contiguous() and facc.shape[-1] == 32 and nsum.dtype == torch.float32 and
                gauss.dtype == torch.float32):
            raise ValueError("fuse_window_feature_accumulate: contiguous fp32 accumulator [X,Y,Z,32], fp32 weight sum and Gaussian expected")

        class _Ctx:
            def __enter__(self_):
                net._window_acc = (facc, nsum, gauss, list(origins), "features")

            def __exit__(self_, *exc):
                net._window_acc = None
        return _Ctx()

    def forward(self, x):
        sel = self.selected_classes
        if sel is not None and sel.device != x.device:
            sel = self.selected_classes = sel.to(x.device)
        params = [p for p in self.parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        # Round 6: with the head fused into the inverse warp and 16-bit storage, the network's backward can take the gradient of
        # its output in the storage type (half the bytes for the fused gather).  autograd hands gradients over in the OUTPUT's
        # dtype (fp32), so the offer travels beside the output tensor: a loss that knows it (ops.consistency_loss on the batched
        # pair) leaves its 16-bit gradient in the sink and returns a stride-0 placeholder; anything else is summed as usual.
        sink = None
        if (need_grad and self._fused_warp is not None and self.act_dtype in (torch.bfloat16, torch.float16)
                and os.environ.get("DGTTA_GRAD16", "1") != "0"):
            sink = Grad16Sink(self.act_dtype)
        y = _UNetFn.apply(self, x, sel, need_grad, sink, *params)
        if sink is not None:
            y._dgtta_grad16 = sink
        return y

    # -- packed weights (re-packed only when the parameter changed)
    def packed(self, conv, dt, cinp, coutp, cin_slice=None):
        """Packed weight blob of `conv` in storage format dt (F32 | BF16 | F16), re-packed only when the parameter changed.
        cin_slice = (lo, hi): the blob of the conv restricted to input channels lo..hi-1 (cinp = its padded count) - the data
        gradient of a conv on a concat buffer is evaluated half by half (see _UNetFn.backward)."""
        w = conv.weight
        key = (id(w), dt, cinp, coutp, cin_slice)
        ent = self._packed.get(key)
        if ent is not None and ent[0] == w._version and ent[1].device == w.device:
            return ent[1]
        lib = _lib.load()
        tdt = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[dt]
        nbytes = lib.dgtta_conv3d_packed_bytes(cinp, coutp, dt)
        wpack = torch.empty(nbytes // tdt.itemsize, dtype=tdt, device=w.device)
        wsrc = w.detach() if cin_slice is None else w.detach()[:, cin_slice[0]:cin_slice[1]].contiguous()
        check(lib.dgtta_conv3d_pack_weights(ptr(wsrc), ptr(wpack), wsrc.shape[1], conv.out_channels,
                                            cinp, coutp, dt, stream_of(w.device)), "dgtta_conv3d_pack_weights")
        self._packed[key] = (w._version, wpack)
        return wpack


class Grad16Sink:
    """Side channel for the gradient of the network output in the 16-bit storage type (see HipPlainConvUNet.forward):
    `put` by the loss backward, taken by _UNetFn.backward of the same pass."""

    def __init__(self, dtype):
        self.dtype, self.g16, self.claimed = dtype, None, False

    def claim(self):
        """ONE consumer of the output may use the channel (a second sink-aware loss on the same output takes the fp32 route, so
        that autograd sums its dense gradient with the first one's instead of one `put` overwriting the other)."""
        if self.claimed:
            return False
        self.claimed = True
        return True

    def put(self, g16):
        self.g16 = g16

    def take(self):
        g, self.g16 = self.g16, None
        return g


def set_probe(model, where):
    """bench.py hook: record (start, end) events around the forward conv launch and the weight-gradient launch (sweep + slab
    reduction, on the stream it runs on) of block `where` = (kind, stage, idx) of `model` (None: stop).  Returns the probe dict
    (events are appended to probe["events"] / probe["wgrad_events"] while the model runs)."""
    st = state_of(model)
    st.probe = None if where is None else dict(where=where, events=[])
    return st.probe


def _odim(i, s):
    return (i + 2 - 3) // s + 1


class _UNetFn(torch.autograd.Function):
    """Whole-network autograd node: forward saves raw conv outputs, normalised activations and IN statistics."""

    @staticmethod
    def forward(ctx, net, x, sel, need_grad, sink, *params):
        lib = _lib.load()
        _lib.require_cuda(x)
        dev = x.device
        st = stream_of(dev)
        cfg = net.cfg
        adt = net.act_dtype
        dt = dtype_code(adt)
        impl = net.conv_impl
        B, cin0, D, H, W = x.shape
        assert cin0 == cfg["in_channels"], f"expected {cfg['in_channels']} input channels, got {cin0}"
        nst = len(cfg["features"])
        tot_stride = 1
        for s in cfg["strides"]:
            tot_stride *= s
        assert D % tot_stride == 0 and H % tot_stride == 0 and W % tot_stride == 0, \
            f"patch {D}x{H}x{W} must be divisible by {tot_stride}"
        enc, dec = net.conv_blocks()
        CP = 8 if dt == F32 else 16      # channel padding granule of packed weights / first-layer input

        # ---- input -> NDHWC, padded to CP channels
        cin0p = _pad(cin0, CP)
        if (x.dtype == adt and x.stride(1) == 1 and x.stride(4) == cin0p and x.stride(3) == W * cin0p
                and x.stride(2) == H * W * cin0p and x.stride(0) == D * H * W * cin0p):
            xin = x            # already voxel-major with rows of cin0p (zero padded) channels, e.g. from mind_hook
        else:
            xin = torch.empty((B, D, H, W, cin0p), dtype=adt, device=dev)
            xs = x.contiguous().float()
            check(lib.dgtta_ncdhw_to_ndhwc(ptr(xs), ptr(xin), B, cin0, D * H * W, cin0p, dt, st), "dgtta_ncdhw_to_ndhwc")

        saved = []   # per conv block: dict(u, ldu, y, mr, dims...)
        probe = state_of(net).probe
        ws_cache = {}

        def ws_for(nbytes):
            nb = int(nbytes)
            t = ws_cache.get("ws")
            if t is None or t.numel() < nb:
                t = _ws(nb, dev)
                ws_cache["ws"] = t
            return t

        def run_block(blk_mod, u, ldu, cin, dims_in, z_out=None, ldz=None, where=None, stats_only=False, xbs=0):
            """conv -> IN -> lrelu. u: tensor whose data_ptr()+offset is the input; returns (z, ldz, dims_out, rec).
            stats_only: the InstanceNorm statistics are finalised but not applied (z is not written; rec carries y and mr)."""
            conv, norm = blk_mod.conv, blk_mod.norm
            s = conv.stride
            cout = conv.out_channels
            di, hi, wi = dims_in
            do, ho, wo = _odim(di, s), _odim(hi, s), _odim(wi, s)
            cinp, coutp = _pad(cin, CP), _pad(cout, CP)
            wpack = net.packed(conv, dt, cinp, coutp)
            y = torch.empty((B, do, ho, wo, cout), dtype=adt, device=dev)
            # InstanceNorm statistics ride on the conv epilogue (one reusable buffer: conv -> finalize are stream ordered)
            sbytes = lib.dgtta_conv3d_stats_bytes(B, cout, do, ho, wo)
            stats = ws_cache.get("stats")
            if stats is None or stats.numel() < sbytes:
                stats = ws_cache["stats"] = _ws(sbytes, dev)
            pr = probe if (probe is not None and probe["where"] == where) else None
            if pr is not None:       # bench.py: time this layer's conv launch with events on the launch stream
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            if xbs:      # the input is the level-0 concat buffer as two 32-channel planes (see the encoder loop)
                check(lib.dgtta_conv3d_k3_fwd_blocked(u, xbs, ptr(wpack), ptr(conv.bias), ptr(y), cout, ptr(stats), B, cin, cout, cinp,
                                                      coutp, di, hi, wi, dt, st), "dgtta_conv3d_k3_fwd_blocked")
            else:
                check(lib.dgtta_conv3d_k3_fwd(u, ldu, ptr(wpack), ptr(conv.bias), ptr(y), cout, ptr(stats), B, cin, cout, cinp,
                                              coutp, di, hi, wi, s, dt, impl, st), "dgtta_conv3d_k3_fwd")
            if pr is not None:
                ev1.record()
                pr["events"].append((ev0, ev1, B))
                pr.update(cin=cin, cout=cout, vout=do * ho * wo, batch=B)
            v = do * ho * wo
            mr = torch.empty((B, cout, 2), dtype=torch.float32, device=dev)
            if stats_only:
                zt, zp, ldz_ = None, None, cout
            elif z_out is None:
                zt = torch.empty((B, do, ho, wo, cout), dtype=adt, device=dev)
                zp, ldz_ = zt.data_ptr(), cout
            else:
                zt, zp, ldz_ = z_out[0], z_out[1], ldz
            nb = lib.dgtta_instnorm_ws_bytes(B, cout, v)
            w_ = ws_for(nb)
            check(lib.dgtta_instnorm_lrelu_fwd(ptr(y), cout, ptr(stats), ptr(norm.weight), ptr(norm.bias), ptr(mr), zp, ldz_,
                                               ptr(w_), nb, B, cout, v, EPS, SLOPE, dt, st), "dgtta_instnorm_lrelu_fwd")
            rec = dict(mod=blk_mod, u=u, ldu=ldu, cin=cin, cout=cout, s=s, din=dims_in, dout=(do, ho, wo), y=y, mr=mr,
                       zt=zt, zp=zp, ldz=ldz_, cinp=cinp, coutp=coutp, xbs=xbs)
            return zp, ldz_, (do, ho, wo), rec, zt

        esz = 4 if dt == F32 else 2
        wa = net._window_acc
        feat_fold = (wa is not None and len(wa) > 4 and not need_grad and os.environ.get("DGTTA_FEATURE_FOLD", "1") != "0" and
                     net.decoder.seg_layers[-1].in_channels == cfg["features"][0] == 32)
        # ---- encoder
        dims = (D, H, W)
        u_ptr, ldu, cin = xin.data_ptr(), cin0p, cin0
        keep = [xin]
        cat_bufs = []     # per encoder stage (except last): (tensor [B,d,h,w,2C], C, dims)
        skip_info = []
        for si, blocks in enumerate(enc):
            cstage = cfg["features"][si]
            for bi, blk in enumerate(blocks):
                last = bi == len(blocks) - 1
                z_out, ldz = None, None
                if last and si < nst - 1:
                    s = blk.conv.stride
                    do, ho, wo = _odim(dims[0], s), _odim(dims[1], s), _odim(dims[2], s)
                    # Round 6: where a HALF of the concat buffer is 64 bytes per voxel (32 channels of 16-bit values: level 0), the
                    # two halves are kept as dense PLANES [up | skip] instead of interleaved rows of 2 C channels: the kernels that
                    # read one half (the stride-2 conv of the skip below, its weight gradient, the transposed conv's backward) then
                    # use whole 128-byte lines (the memory side moves whole lines: 4.3x the input fetched before, r05_ab.txt).
                    # The decoder conv that reads BOTH halves takes them as 32-channel blocks (dgtta_conv3d_k3_fwd_blocked) - where
                    # the ring kernels run (asked up front); DGTTA_PLANAR_CAT=0: the interleaved layout everywhere.
                    planar = (cstage * esz == 64 and impl != 1 and os.environ.get("DGTTA_PLANAR_CAT", "1") != "0"
                              and os.environ.get("DGTTA_SPLIT_CAT_GRAD", "1") != "0"
                              and lib.dgtta_conv3d_k3_blocked_supported(B, 2 * cstage, cstage, do, ho, wo, dt) == 1)
                    if planar:
                        cat = torch.empty((2, B, do, ho, wo, cstage), dtype=adt, device=dev)
                        cat_bufs.append((cat, cstage, (do, ho, wo), B * do * ho * wo * cstage))
                        z_out, ldz = (cat, cat[1].data_ptr()), cstage
                    else:
                        cat = torch.empty((B, do, ho, wo, 2 * cstage), dtype=adt, device=dev)
                        cat_bufs.append((cat, cstage, (do, ho, wo), 0))
                        z_out, ldz = (cat, cat.data_ptr() + cstage * esz), 2 * cstage
                u_ptr, ldu, dims, rec, zt = run_block(blk, u_ptr, ldu, cin, dims, z_out, ldz, ("enc", si, bi))
                rec["where"] = ("enc", si, bi)
                saved.append(rec)
                keep.append(zt)
                cin = cstage
        # ---- decoder
        x_low_ptr, x_low_ld, x_low_c, low_dims = u_ptr, ldu, cin, dims
        ups = []
        for k, blocks in enumerate(dec):
            cat, cskip, cdims, cat_xbs = cat_bufs[-(k + 1)]
            up = net.decoder.transpconvs[k]
            nbt = lib.dgtta_convT3d_fwd_ws_bytes(x_low_c, cskip, dt)
            wst = ws_for(nbt)
            cat_ld = cskip if cat_xbs else 2 * cskip           # (planes: the up half is the dense tensor at the buffer's start)
            check(lib.dgtta_convT3d_k2s2_fwd(x_low_ptr, x_low_ld, ptr(up.weight), ptr(up.bias), ptr(cat), cat_ld,
                                             ptr(wst), nbt, B, x_low_c, cskip, low_dims[0], low_dims[1], low_dims[2], dt,
                                             impl, st), "dgtta_convT3d_k2s2_fwd")
            ups.append(dict(mod=up, x=x_low_ptr, ldx=x_low_ld, cin=x_low_c, cout=cskip, din=low_dims, cat=cat))
            u_ptr, ldu, cin, dims = cat.data_ptr(), cat_ld, 2 * cskip, cdims
            for bi, blk in enumerate(blocks):
                # feature-space window accumulation: the block in front of the head hands over its raw conv output and statistics -
                # its InstanceNorm + LeakyReLU apply runs inside the accumulation kernel, z is never written
                fold = feat_fold and k == len(dec) - 1 and bi == len(blocks) - 1
                u_ptr, ldu, dims, rec, zt = run_block(blk, u_ptr, ldu, cin, dims, None, None, ("dec", k, bi), stats_only=fold,
                                                      xbs=cat_xbs if bi == 0 else 0)
                rec["where"] = ("dec", k, bi)
                saved.append(rec)
                keep.append(zt)
                cin = cskip
            x_low_ptr, x_low_ld, x_low_c, low_dims = u_ptr, ldu, cin, dims
        # ---- head (restricted to the selected rows when requested)
        head = net.decoder.seg_layers[-1]
        ncls = head.out_channels
        nsel = ncls if sel is None else int(sel.numel())
        V = D * H * W
        if wa is not None:
            assert not need_grad and sel is None and ldu == head.in_channels and len(wa[3]) == B, "fuse_window_accumulate: misuse"
            acc, nsum, gauss, origins = wa[:4]
            X, Y, Z = acc.shape[:3]
            if len(wa) > 4:      # feature space: no head here
                last = saved[-1]
                nrm = last["mod"].norm
                src0 = last["y"].data_ptr() if feat_fold else u_ptr
                lds = 32 if feat_fold else ldu
                assert lds == 32

                def one_window(k, sx, sy, sz):
                    if feat_fold:
                        check(lib.dgtta_feature_window_accumulate_norm(src0 + k * V * 32 * esz, last["mr"].data_ptr() + k * 32 * 2 * 4,
                                                                       ptr(nrm.weight), ptr(nrm.bias), SLOPE, ptr(gauss), ptr(acc), ptr(nsum),
                                                                       32, D, H, W, X, Y, Z, sx, sy, sz, dt, st),
                              "dgtta_feature_window_accumulate_norm")
                    else:
                        check(lib.dgtta_feature_window_accumulate(src0 + k * V * 32 * esz, ptr(gauss), ptr(acc), ptr(nsum), 32, D, H, W, X,
                                                                  Y, Z, sx, sy, sz, dt, st), "dgtta_feature_window_accumulate")

                # consecutive windows of a sliding-window row overlap along the last axis: one launch per SEGMENT of that axis adds
                # every covering window's contribution in registers (same order, same bits) and touches the accumulator once
                k = 0
                while k < len(origins):
                    j = k + 1
                    while (os.environ.get("DGTTA_FEATURE_SEGMENTS", "1") != "0" and j < len(origins) and origins[j][:2] == origins[k][:2]
                           and origins[j - 1][2] < origins[j][2] < origins[j - 1][2] + W):
                        j += 1
                    if j - k == 1:
                        one_window(k, *origins[k])
                        k = j
                        continue
                    sx, sy = origins[k][:2]
                    zs = [origins[i][2] for i in range(k, j)]
                    cuts = sorted(set(zs + [z + W for z in zs]))
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        cover = [i for i in range(k, j) if origins[i][2] <= a and b <= origins[i][2] + W]
                        for c0 in range(0, len(cover), 4):          # (more than four windows on a voxel: step sizes below a quarter patch)
                            part = cover[c0:c0 + 4]
                            n = len(part)
                            srcs = (C.c_void_p * n)(*[src0 + i * V * 32 * esz for i in part])
                            mrs = (C.c_void_p * n)(*[last["mr"].data_ptr() + i * 32 * 2 * 4 for i in part]) if feat_fold else None
                            zoffs = (C.c_int * n)(*[a - origins[i][2] for i in part])
                            check(lib.dgtta_feature_window_accumulate_multi(srcs, mrs, zoffs, n, ptr(nrm.weight), ptr(nrm.bias), SLOPE,
                                                                            ptr(gauss), ptr(acc), ptr(nsum), 32, D, H, W, b - a, X, Y, Z, sx,
                                                                            sy, a, dt, st), "dgtta_feature_window_accumulate_multi")
                    k = j
                return torch.empty((B, 0, D, H, W), dtype=torch.float32, device=dev)
            for k, (sx, sy, sz) in enumerate(origins):      # overlapping windows: accumulated one after the other
                check(lib.dgtta_seghead_window_accumulate_t(u_ptr + k * V * ldu * esz, ptr(head.weight), ptr(head.bias),
                                                            ptr(gauss), ptr(acc), ptr(nsum), head.in_channels, ncls, D, H, W, X,
                                                            Y, Z, sx, sy, sz, dt, F32 if acc.dtype == torch.float32 else F16, st),
                      "dgtta_seghead_window_accumulate_t")
            return torch.empty((B, 0, D, H, W), dtype=torch.float32, device=dev)
        out = torch.empty((B, D, H, W, nsel), dtype=torch.float32, device=dev)
        fw = net._fused_warp
        if fw is not None:
            assert ldu == head.in_channels and tuple(fw[0].shape) == (B, 3, 4), "fuse_output_warp: shape mismatch"
            check(lib.dgtta_seghead_warp_fwd(u_ptr, ptr(head.weight), ptr(head.bias), ptr(sel), nsel, ptr(fw[0]), ptr(out), B,
                                             head.in_channels, D, H, W, 1, dt, st), "dgtta_seghead_warp_fwd")
        else:
            check(lib.dgtta_seghead_fwd(u_ptr, ldu, ptr(head.weight), ptr(head.bias), ptr(sel), nsel, ptr(out), 1, nsel, B,
                                        head.in_channels, V, dt, st), "dgtta_seghead_fwd")
        if need_grad:
            ctx.net, ctx.sel, ctx.saved, ctx.ups, ctx.keep, ctx.cat_bufs = net, sel, saved, ups, keep, cat_bufs
            ctx.meta = (B, D, H, W, dt, impl, nsel, u_ptr, ldu)
            ctx.params = params
            ctx.fused_warp = fw
            ctx.sink = sink if fw is not None else None
        return out.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        net, sel, saved, ups, cat_bufs = ctx.net, ctx.sel, ctx.saved, ctx.ups, ctx.cat_bufs
        B, D, H, W, dt, impl, nsel, zlast_ptr, zlast_ld = ctx.meta
        params = ctx.params
        dev = gout.device
        st = stream_of(dev)
        adt = net.act_dtype
        esz = 4 if dt == F32 else 2
        CP_ = 8 if dt == F32 else 16      # channel padding granule of packed weights
        grads = {}          # id(param) -> grad tensor

        def want(p):
            return p.requires_grad

        inplace = bool(net.accumulate_grads_in_place)
        ACC = 1 if inplace else 0      # kernels add to the gradient buffers (which then are the parameters' .grad)

        def gbuf(p):
            if inplace:
                if p.grad is None:
                    p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
                return p.grad
            g = grads.get(id(p))
            if g is None:
                g = torch.zeros_like(p, memory_format=torch.contiguous_format)
                grads[id(p)] = g
            return g

        ws_cache = {}

        def ws_for(nbytes, key="ws"):
            nb = int(nbytes)
            t = ws_cache.get(key)
            if t is None or t.numel() < nb:
                t = _ws(nb, dev)
                ws_cache[key] = t
            return t

        # The weight gradients of the conv blocks are leaves of the backward chain (IN-bwd(L) -> dgrad(L) -> IN-bwd(L-1) ...
        # only passes dy on): they run on a SIDE STREAM, so that the MFMA-bound weight-gradient kernels overlap the
        # HBM-bound InstanceNorm passes of the main chain instead of queueing between them (DGTTA_WGRAD_STREAM=0: one stream)
        main_stream = torch.cuda.current_stream(dev)
        side = state_of(net).stream("side_stream", dev) if _wgrad_on_side_stream() else None
        if side is not None:
            side.wait_stream(main_stream)

        def scratch_like(p):
            """Throw-away gradient buffer (only the bias of this layer wants a gradient): it is written by the kernel on
            the side stream, so it comes from that stream's pool - a main-stream block could be handed out again while
            the side kernel still writes to it."""
            if side is None:
                return torch.empty_like(p)
            with torch.cuda.stream(side):
                return torch.empty_like(p)

        V = D * H * W
        g16 = ctx.sink.take() if ctx.sink is not None else None      # the loss left its gradient in the storage type (round 6)
        if g16 is not None and any(gout.stride()):
            # the placeholder has stride 0 everywhere; a dense gout means another consumer of the output contributed: sum in fp32
            gout = gout + g16.permute(0, 4, 1, 2, 3).float()
            g16 = None
        g = None if g16 is not None else gout.contiguous(memory_format=torch.channels_last_3d).float()      # [B,nsel,D,H,W] stored NDHWC
        head = net.decoder.seg_layers[-1]
        cin_h = head.in_channels
        # ---- head backward
        gz = torch.empty((B, D, H, W, cin_h), dtype=adt, device=dev)
        need_hw = want(head.weight) or want(head.bias)
        dws = torch.empty((nsel, cin_h), dtype=torch.float32, device=dev) if need_hw else None
        dbs = torch.empty((nsel,), dtype=torch.float32, device=dev) if need_hw else None
        fw = ctx.fused_warp
        if fw is not None:      # g is the gradient of the WARPED logits: fused gather + W^T (+ weight / bias gradient)
            nb = lib.dgtta_seghead_warp_bwd_ws_bytes(B, cin_h, nsel, D, H, W)
            w_ = ws_for(nb)
            if g16 is not None:
                assert tuple(g16.shape) == (B, D, H, W, nsel) and g16.dtype == adt and g16.is_contiguous()
                check(lib.dgtta_seghead_warp_bwd_g16(zlast_ptr, ptr(g16), ptr(fw[0]), ptr(fw[1]), ptr(head.weight), ptr(sel), nsel,
                                                     ptr(gz), ptr(dws), ptr(dbs), ptr(w_), nb, B, cin_h, D, H, W, 1, 0, dt, st),
                      "dgtta_seghead_warp_bwd_g16")
            else:
                check(lib.dgtta_seghead_warp_bwd(zlast_ptr, ptr(g), ptr(fw[0]), ptr(fw[1]), ptr(head.weight), ptr(sel), nsel,
                                                 ptr(gz), ptr(dws), ptr(dbs), ptr(w_), nb, B, cin_h, D, H, W, 1, 0, dt, st),
                      "dgtta_seghead_warp_bwd")
        else:
            nb = lib.dgtta_seghead_bwd_ws_bytes(B, cin_h, nsel, V)
            w_ = ws_for(nb)
            check(lib.dgtta_seghead_bwd(zlast_ptr, zlast_ld, ptr(g), nsel, ptr(head.weight), ptr(sel), nsel, ptr(gz), cin_h,
                                        ptr(dws), ptr(dbs), ptr(w_), nb, B, cin_h, V, 0, dt, st), "dgtta_seghead_bwd")
        if need_hw:
            gw, gb = gbuf(head.weight), gbuf(head.bias)
            if sel is None:
                gw.view(-1, cin_h).add_(dws)       # buffers start at zero (or hold earlier accumulation steps)
                gb.add_(dbs)
            else:
                gw.view(-1, cin_h).index_add_(0, sel.long(), dws)
                gb.index_add_(0, sel.long(), dbs)

        # gradient buffers for the concat tensors (zero-free: fully written by the consuming conv's dgrad)
        gcat = {}

        gz_ptr, gz_ld = gz.data_ptr(), cin_h
        keep_alive = [gz]
        gstats = None       # InstanceNorm backward sums left by the data gradient that produced the current gz
        first_rec = saved[0]
        # walk blocks in reverse order
        idx = len(saved) - 1
        n_dec_stages = len(ups)
        while idx >= 0:
            rec = saved[idx]
            blk = rec["mod"]
            conv, norm = blk.conv, blk.norm
            cin, cout, s = rec["cin"], rec["cout"], rec["s"]
            di, hi, wi = rec["din"]
            do, ho, wo = rec["dout"]
            v = do * ho * wo
            # -- InstanceNorm + LeakyReLU backward (dy overwrites a fresh dense buffer)
            dy = torch.empty((B, do, ho, wo, cout), dtype=adt, device=dev)
            nb = lib.dgtta_instnorm_ws_bytes(B, cout, v)
            w_ = ws_for(nb)
            dgam = gbuf(norm.weight) if want(norm.weight) else torch.empty_like(norm.weight)
            dbet = gbuf(norm.bias) if want(norm.bias) else torch.empty_like(norm.bias)
            if gstats is not None:      # the data gradient that produced gz also left the reduction's sums
                check(lib.dgtta_instnorm_lrelu_bwd_gstats(gz_ptr, gz_ld, ptr(rec["y"]), cout, ptr(norm.weight),
                                                          ptr(norm.bias), ptr(rec["mr"]), ptr(dy), cout, ptr(dgam), ptr(dbet),
                                                          ptr(gstats), ptr(w_), nb, B, cout, v, SLOPE, ACC, dt, st),
                      "dgtta_instnorm_lrelu_bwd_gstats")
                gstats = None
            else:
                check(lib.dgtta_instnorm_lrelu_bwd(gz_ptr, gz_ld, ptr(rec["y"]), cout, ptr(norm.weight), ptr(norm.bias),
                                                   ptr(rec["mr"]), ptr(dy), cout, ptr(dgam), ptr(dbet), ptr(w_), nb, B, cout,
                                                   v, SLOPE, ACC, dt, st), "dgtta_instnorm_lrelu_bwd")
            # -- weight / bias gradient
            if want(conv.weight) or want(conv.bias):
                # fp32 storage: offer the split workspace - the weight gradient then runs as six launches of the
                # 16-bit matrix-core kernels on exact three-term bf16 splits of x and dy (csrc/conv_wgrad.hip, round 5)
                if dt == F32 and impl != 1 and (s == 1 or not ((di | hi | wi) & 1)):
                    nb = lib.dgtta_conv3d_wgrad_split_ws_bytes(B, cin, cout, do, ho, wo, s)
                else:
                    nb = lib.dgtta_conv3d_wgrad_ws_bytes(B, cin, cout, do, ho, wo)
                dw = gbuf(conv.weight) if want(conv.weight) else scratch_like(conv.weight)
                db = gbuf(conv.bias) if want(conv.bias) else None
                if net.exact_zero_bias_grad:
                    db = None       # gradient buffer stays exactly zero (see HipPlainConvUNet.exact_zero_bias_grad)
                if side is None:
                    w_, st_w = ws_for(nb), st
                else:
                    ev = torch.cuda.Event()
                    ev.record(main_stream)          # dy (and the gradient buffers) are complete on the main stream
                    side.wait_event(ev)
                    dy.record_stream(side)          # the allocator must not hand dy's block out again before the side stream is done
                    with torch.cuda.stream(side):
                        w_ = ws_for(nb, "ws_side")
                    st_w = side.cuda_stream
                pr = state_of(net).probe
                pr = pr if (pr is not None and pr["where"] == rec["where"]) else None
                if pr is not None:       # bench.py: time this layer's weight-gradient launch with events on ITS stream
                    wstream = main_stream if side is None else side
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record(wstream)
                if rec["xbs"]:      # x = the level-0 concat buffer as two 32-channel planes
                    check(lib.dgtta_conv3d_k3_wgrad_blocked(rec["u"], rec["xbs"], ptr(dy), cout, ptr(dw), ptr(db), ptr(w_), nb, B,
                                                            cin, cout, di, hi, wi, ACC, dt, st_w), "dgtta_conv3d_k3_wgrad_blocked")
                else:
                    check(lib.dgtta_conv3d_k3_wgrad(rec["u"], rec["ldu"], ptr(dy), cout, ptr(dw), ptr(db), ptr(w_), nb, B,
                                                    cin, cout, di, hi, wi, s, ACC, dt, impl, st_w), "dgtta_conv3d_k3_wgrad")
                if pr is not None:
                    ev1.record(wstream)
                    pr.setdefault("wgrad_events", []).append((ev0, ev1, B))
            # -- data gradient towards the block input
            where = rec["where"]
            if idx == 0:
                break
            wb = net.packed(conv, dt, rec["cinp"], rec["coutp"])
            kind, sidx, bidx = where
            if kind == "dec" and bidx == 0:
                # input was the concat buffer of decoder stage sidx: gradient for [up | skip]
                cat, cskip, cdims, cat_xbs = cat_bufs[-(sidx + 1)]
                if cskip * esz == 64 and (cat_xbs or os.environ.get("DGTTA_SPLIT_CAT_GRAD", "1") != "0"):
                    # 32 channels of 16-bit values = HALF a 128-byte line: as one [voxel][2 C] tensor every consumer of ONE half of this
                    # gradient (the transposed conv's backward, the stride-2 data gradient's accumulate, the InstanceNorm backward
                    # of the skip block) would fetch whole lines and use 64 bytes of each (profiles/r05_ab.txt, fetch_calib.sh).
                    # The data gradient's two 32-channel output blocks are independent jobs of the kernel anyway: two launches on
                    # the weight halves write two DENSE tensors (same values, the same reads of dy).
                    gc_up = torch.empty((B, *cdims, cskip), dtype=adt, device=dev)
                    gc_skip = torch.empty((B, *cdims, cskip), dtype=adt, device=dev)
                    for half, dst in ((0, gc_up), (1, gc_skip)):
                        wbh = net.packed(conv, dt, _pad(cskip, CP_), rec["coutp"], (half * cskip, (half + 1) * cskip))
                        check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wbh), ptr(dst), cskip, B, cskip, cout, _pad(cskip, CP_),
                                                        rec["coutp"], di, hi, wi, s, 0, dt, impl, st), "dgtta_conv3d_k3_dgrad")
                    gcat[sidx] = (gc_skip, gc_skip.data_ptr(), cskip)
                    gc, gc_ld = gc_up, cskip
                else:
                    gc = torch.empty_like(cat)
                    gcat[sidx] = (gc, gc.data_ptr() + cskip * esz, 2 * cskip)
                    gc_ld = 2 * cskip
                    check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wb), ptr(gc), 2 * cskip, B, cin, cout, rec["cinp"],
                                                    rec["coutp"], di, hi, wi, s, 0, dt, impl, st), "dgtta_conv3d_k3_dgrad")
                # transposed-conv backward: dout = first half of the concat gradient
                up = ups[sidx]
                upm = up["mod"]
                ld0, lh0, lw0 = up["din"]
                glow = torch.empty((B, ld0, lh0, lw0, up["cin"]), dtype=adt, device=dev)
                # fp32 storage: room for the weight gradient as six 16-bit launches on exact bf16 splits (as for the 3x3x3 convs)
                nb = (lib.dgtta_convT3d_bwd_split_ws_bytes if dt == F32 and impl != 1 else
                      lib.dgtta_convT3d_bwd_ws_bytes)(B, up["cin"], up["cout"], ld0, lh0, lw0)
                w_ = ws_for(nb)
                need_w = want(upm.weight) or want(upm.bias)
                dwu = (gbuf(upm.weight) if want(upm.weight) else scratch_like(upm.weight)) if need_w else None
                dbu = gbuf(upm.bias) if want(upm.bias) else None
                if side is None or not need_w:
                    check(lib.dgtta_convT3d_k2s2_bwd(up["x"], up["ldx"], ptr(gc), gc_ld, ptr(upm.weight), ptr(glow),
                                                     up["cin"], ptr(dwu), ptr(dbu), ptr(w_), nb, B, up["cin"], up["cout"],
                                                     ld0, lh0, lw0, ACC, dt, impl, st), "dgtta_convT3d_k2s2_bwd")
                else:
                    # data gradient on the main chain, weight / bias gradient (a leaf) on the side stream
                    ev = torch.cuda.Event()
                    ev.record(main_stream)          # gc is complete
                    check(lib.dgtta_convT3d_k2s2_bwd(up["x"], up["ldx"], ptr(gc), gc_ld, ptr(upm.weight), ptr(glow),
                                                     up["cin"], None, None, ptr(w_), nb, B, up["cin"], up["cout"],
                                                     ld0, lh0, lw0, ACC, dt, impl, st), "dgtta_convT3d_k2s2_bwd")
                    side.wait_event(ev)
                    gc.record_stream(side)
                    with torch.cuda.stream(side):
                        w2 = ws_for(nb, "ws_side")
                    check(lib.dgtta_convT3d_k2s2_bwd(up["x"], up["ldx"], ptr(gc), gc_ld, ptr(upm.weight), None,
                                                     up["cin"], ptr(dwu), ptr(dbu), ptr(w2), nb, B, up["cin"], up["cout"],
                                                     ld0, lh0, lw0, ACC, dt, impl, side.cuda_stream), "dgtta_convT3d_k2s2_bwd")
                gz_ptr, gz_ld = glow.data_ptr(), up["cin"]
                keep_alive = [glow, gc, gcat[sidx][0]]
            elif kind == "enc" and bidx == 0:
                # input was the previous encoder stage's output, which lives in the second half of a concat buffer
                # and already holds the decoder's skip gradient: accumulate into it.
                prev_stage = sidx - 1
                dec_k = n_dec_stages - 1 - prev_stage
                gc, gptr, gld = gcat[dec_k]          # (tensor that owns the skip half, its address, its row pitch)
                check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wb), gptr, gld, B, cin, cout, rec["cinp"],
                                                rec["coutp"], di, hi, wi, s, 1, dt, impl, st), "dgtta_conv3d_k3_dgrad")
                gz_ptr, gz_ld = gptr, gld
                keep_alive = [gc]
            else:
                # input was the previous block's activation z = LeakyReLU(InstanceNorm(y_prev)), consumed by this conv only:
                # the data gradient can leave the sums of that block's InstanceNorm backward (csrc/conv_rows.hip, GST)
                gin = torch.empty((B, di, hi, wi, cin), dtype=adt, device=dev)
                prev = saved[idx - 1]
                if s == 1 and dt != F32 and prev["cout"] == cin and prev["dout"] == (di, hi, wi):
                    pn = prev["mod"].norm
                    gbytes = lib.dgtta_conv3d_stats_bytes(B, cin, di, hi, wi)
                    gbuf_ = ws_for(gbytes, "gstats")
                    produced = C.c_int(0)
                    check(lib.dgtta_conv3d_k3_dgrad_gstats(ptr(dy), cout, ptr(wb), ptr(gin), cin, B, cin, cout, rec["cinp"],
                                                           rec["coutp"], di, hi, wi, ptr(prev["y"]), cin, ptr(prev["mr"]),
                                                           ptr(pn.weight), ptr(pn.bias), SLOPE, ptr(gbuf_), gbytes,
                                                           C.byref(produced), dt, impl, st), "dgtta_conv3d_k3_dgrad_gstats")
                    gstats = gbuf_ if produced.value else None
                else:
                    check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wb), ptr(gin), cin, B, cin, cout, rec["cinp"],
                                                    rec["coutp"], di, hi, wi, s, 0, dt, impl, st), "dgtta_conv3d_k3_dgrad")
                gz_ptr, gz_ld = gin.data_ptr(), cin
                keep_alive = [gin]
            idx -= 1
        del keep_alive, first_rec
        if side is not None:
            main_stream.wait_stream(side)         # gradients complete before anything downstream (optimizer, next pass)
        out = [None, None, None, None, None]
        for p in params:
            out.append(grads.get(id(p)) if p.requires_grad else None)
        return tuple(out)
