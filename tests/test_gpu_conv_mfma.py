"""GPU: MFMA implicit-GEMM conv kernels vs the general VALU kernels (same C ABI, impl=2 vs impl=1) and vs torch CPU
conv3d at a small size.  fp32 must agree to summation-order noise; bf16 is compared against the fp32 kernels run on
the same bf16-rounded operands (bf16 storage, fp32 accumulation => only the output rounding differs)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import reload_kernel_switches

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call_fwd(x, w, bias, stride, dt, impl, cinp, coutp):
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, D, H, W, ld = x.shape
    cout, cin = w.shape[:2]
    tdt = torch.float32 if dt == 0 else torch.bfloat16
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cinp, coutp, dt) // (2 if dt else 4), dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cinp, coutp, dt, stream_of()), "pack")
    do, ho, wo = [(n - 1) // stride + 1 for n in (D, H, W)]
    y = torch.empty((B, do, ho, wo, cout), dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_k3_fwd(ptr(x), ld, ptr(wpack), ptr(bias), ptr(y), cout, None, B, cin, cout, cinp, coutp, D, H,
                                  W, stride, dt, impl, stream_of()), "fwd")
    return y, wpack


def _call_dgrad(dy, wpack, cin, cinp, coutp, dims, stride, dt, impl):
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B = dy.shape[0]
    cout = dy.shape[-1]
    D, H, W = dims
    dx = torch.empty((B, D, H, W, cin), dtype=dy.dtype, device=DEV)
    check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wpack), ptr(dx), cin, B, cin, cout, cinp, coutp, D, H, W, stride,
                                    0, dt, impl, stream_of()), "dgrad")
    return dx


CASES = [  # (B, cin, cout, D, H, W, stride)
    (1, 16, 32, 8, 8, 32, 1), (1, 32, 32, 9, 7, 45, 1), (2, 8, 64, 6, 10, 16, 1), (1, 24, 40, 8, 8, 8, 1),
    (1, 64, 32, 5, 9, 20, 1), (1, 32, 64, 8, 8, 32, 2), (1, 16, 24, 10, 12, 14, 2), (1, 320, 320, 4, 4, 4, 1),
    (1, 32, 64, 6, 10, 72, 2), (2, 64, 128, 4, 6, 64, 2),     # stride 2 with >= 32 output columns: 2 channel blocks per workgroup
]


@pytest.mark.parametrize("case", CASES)
def test_conv_mfma_fp32_matches_valu_and_torch(case):
    B, cin, cout, D, H, W, s = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, D, H, W, cin, device=DEV)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5
    bias = torch.randn(cout, device=DEV)
    cinp, coutp = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    y1, wp = _call_fwd(x, w, bias, s, 0, 1, cinp, coutp)
    y2, _ = _call_fwd(x, w, bias, s, 0, 2, cinp, coutp)
    ref = F.conv3d(x.permute(0, 4, 1, 2, 3).cpu(), w.cpu(), bias.cpu(), stride=s, padding=1).permute(0, 2, 3, 4, 1)
    assert (y1.cpu() - ref).abs().max() < 2e-5 * ref.abs().max() + 1e-5
    assert (y2.cpu() - ref).abs().max() < 2e-5 * ref.abs().max() + 1e-5
    dy = torch.randn_like(y1)
    g1 = _call_dgrad(dy, wp, cin, cinp, coutp, (D, H, W), s, 0, 1)
    g2 = _call_dgrad(dy, wp, cin, cinp, coutp, (D, H, W), s, 0, 2)      # stride 2: 8 parity-class launches
    assert (g1 - g2).abs().max() < 2e-5 * g1.abs().max() + 1e-5


@pytest.mark.parametrize("case", [c for c in CASES if c[1] % 8 == 0])
def test_conv_mfma_bf16(case):
    B, cin, cout, D, H, W, s = case
    torch.manual_seed(sum(case) + 1)
    xb = torch.randn(B, D, H, W, cin, device=DEV).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5).bfloat16().float()
    bias = torch.randn(cout, device=DEV)
    cinp, coutp = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    y_ref, _ = _call_fwd(xb.float(), w, bias, s, 0, 1, (cin + 7) // 8 * 8, (cout + 7) // 8 * 8)    # fp32 math, same operands
    y, wp = _call_fwd(xb, w, bias, s, 1, 2, cinp, coutp)
    err = (y.float() - y_ref).abs().max()
    assert err < 1.0 / 128 * y_ref.abs().max() + 1e-3, f"bf16 conv err {err}"
    y_valu, _ = _call_fwd(xb, w, bias, s, 1, 1, cinp, coutp)
    assert (y.float() - y_valu.float()).abs().max() < 1.0 / 64 * y_ref.abs().max() + 1e-3


ROWS_CASES = [  # (B, cin, cout, D, H, W): ragged tiles, several channel blocks / K-chunks, batch 2, concat-style strides
    (1, 32, 32, 9, 13, 45), (2, 16, 64, 5, 17, 32), (1, 64, 96, 8, 8, 70), (1, 8, 32, 4, 8, 33),
    (2, 32, 64, 32, 32, 64), (1, 16, 128, 16, 64, 32), (1, 32, 32, 36, 40, 32),      # > 256 jobs: persistent loops, block decode
]


@pytest.mark.parametrize("case", ROWS_CASES)
def test_conv_rows_kernel_bf16(case, monkeypatch):
    """The persistent row-reuse kernel (LDS-DMA staging), forced on at small sizes, against the generic MFMA kernel
    and the fp32 VALU kernel on the same bf16 operands: forward with fused InstanceNorm statistics, and data gradient."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 7)
    xb = torch.randn(B, D, H, W, cin, device=DEV).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5).bfloat16().float()
    bias = torch.randn(cout, device=DEV)
    cinp, coutp = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    y_ref, _ = _call_fwd(xb.float(), w, bias, 1, 0, 1, (cin + 7) // 8 * 8, (cout + 7) // 8 * 8)

    def run(rows, order="1"):
        monkeypatch.setenv("DGTTA_CONV_ROWS", rows)
        monkeypatch.setenv("DGTTA_ROWS_ORDER", order)
        reload_kernel_switches()
        wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cinp, coutp, 1) // 2, dtype=torch.bfloat16, device=DEV)
        check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cinp, coutp, 1, stream_of()), "pack")
        y = torch.full((B, D, H, W, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
        nb = lib.dgtta_conv3d_stats_bytes(B, cout, D, H, W)
        st = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_conv3d_k3_fwd(ptr(xb), cin, ptr(wpack), ptr(bias), ptr(y), cout, ptr(st), B, cin, cout, cinp, coutp,
                                      D, H, W, 1, 1, 2, stream_of()), "fwd")
        mr = torch.empty(B, cout, 2, device=DEV)
        z = torch.empty_like(y)
        gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
        nws = lib.dgtta_instnorm_ws_bytes(B, cout, D * H * W)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_instnorm_lrelu_fwd(ptr(y), cout, ptr(st), ptr(gamma), ptr(beta), ptr(mr), ptr(z), cout, ptr(ws), nws,
                                           B, cout, D * H * W, 1e-5, 0.01, 1, stream_of()), "instnorm")
        mean, rstd = mr[..., 0].reshape(-1).clone(), mr[..., 1].reshape(-1).clone()
        dy = torch.randn(B, D, H, W, cout, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).bfloat16()
        dx = _call_dgrad(dy, wpack, cin, cinp, coutp, (D, H, W), 1, 1, 2)
        torch.cuda.synchronize()
        return y, mean, rstd, dx

    y0, m0, r0, dx0 = run("0")
    y1, m1, r1, dx1 = run("1")
    scale = y_ref.abs().max()
    assert torch.isfinite(y1.float()).all()
    assert (y1.float() - y_ref).abs().max() < scale / 128 + 1e-3
    assert (y1.float() - y0.float()).abs().max() < scale / 128 + 1e-3          # both round the same fp32 sums to bf16
    ref_mean = y_ref.reshape(B, -1, cout).mean(1).reshape(-1)
    ref_rstd = (y_ref.reshape(B, -1, cout).var(1, unbiased=False) + 1e-5).rsqrt().reshape(-1)
    assert (m1 - ref_mean).abs().max() < 1e-4 and (m1 - m0).abs().max() < 1e-5
    assert ((r1 - ref_rstd) / ref_rstd).abs().max() < 1e-4 and ((r1 - r0) / r0).abs().max() < 1e-5
    assert (dx1.float() - dx0.float()).abs().max() < dx0.float().abs().max() / 128 + 1e-3
    # job order of the persistent workgroups (round robin in compact blocks vs the contiguous ranges of round 2a): the
    # same tiles are computed, so outputs are identical; only the grouping of the statistics' partial sums differs
    y2, m2, r2, dx2 = run("1", order="0")
    assert torch.equal(y2, y1) and torch.equal(dx2, dx1)
    assert (m2 - m1).abs().max() < 1e-5 and ((r2 - r1) / r1).abs().max() < 1e-5


RING_CASES = [  # (B, cin, cout, D, H, W): 32 / 64 input channels in the forward, 32 / 64 output channels (-> the data gradient's input)
    (1, 32, 32, 9, 13, 45), (2, 32, 64, 6, 17, 32), (1, 32, 96, 8, 8, 70), (1, 64, 32, 7, 9, 33),
    (2, 32, 32, 32, 32, 64), (1, 32, 32, 36, 40, 32), (8, 32, 32, 4, 64, 160),      # the last: 320 columns = two rounds of jobs
    (1, 64, 64, 9, 13, 45), (2, 64, 96, 6, 7, 64), (2, 64, 64, 16, 32, 32), (4, 64, 32, 4, 64, 160),      # 64 channels both ways
]


@pytest.mark.parametrize("dts", ["bf16", "fp16"])
@pytest.mark.parametrize("case", RING_CASES)
def test_conv_ring_kernel(case, dts, monkeypatch):
    """The D-ring kernel (conv_ring.hip: plane ring in LDS, weights in registers, 16x16x32 MFMA), forced on at small sizes,
    against its predecessors (DGTTA_CONV_RING=0: row-reuse / generic MFMA kernel) and the fp32 VALU kernel on the same
    16-bit operands: forward with fused InstanceNorm statistics, data gradient, ragged edges, several channel blocks."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, cin, cout, D, H, W = case
    dt, tdt = (1, torch.bfloat16) if dts == "bf16" else (2, torch.float16)
    torch.manual_seed(sum(case) + 11)
    xb = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5).to(tdt).float()
    bias = torch.randn(cout, device=DEV)
    cinp, coutp = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    y_ref, _ = _call_fwd(xb.float(), w, bias, 1, 0, 1, (cin + 7) // 8 * 8, (cout + 7) // 8 * 8)

    def run(ring):
        monkeypatch.setenv("DGTTA_CONV_RING", ring)
        reload_kernel_switches()
        wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cinp, coutp, dt) // 2, dtype=tdt, device=DEV)
        check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cinp, coutp, dt, stream_of()), "pack")
        y = torch.full((B, D, H, W, cout), float("nan"), dtype=tdt, device=DEV)
        nb = lib.dgtta_conv3d_stats_bytes(B, cout, D, H, W)
        st = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_conv3d_k3_fwd(ptr(xb), cin, ptr(wpack), ptr(bias), ptr(y), cout, ptr(st), B, cin, cout, cinp, coutp,
                                      D, H, W, 1, dt, 2, stream_of()), "fwd")
        mr = torch.empty(B, cout, 2, device=DEV)
        z = torch.empty_like(y)
        gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
        nws = lib.dgtta_instnorm_ws_bytes(B, cout, D * H * W)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_instnorm_lrelu_fwd(ptr(y), cout, ptr(st), ptr(gamma), ptr(beta), ptr(mr), ptr(z), cout, ptr(ws), nws,
                                           B, cout, D * H * W, 1e-5, 0.01, dt, stream_of()), "instnorm")
        mean, rstd = mr[..., 0].reshape(-1).clone(), mr[..., 1].reshape(-1).clone()
        dy = torch.randn(B, D, H, W, cout, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(tdt)
        dx = _call_dgrad(dy, wpack, cin, cinp, coutp, (D, H, W), 1, dt, 2)
        torch.cuda.synchronize()
        return y, mean, rstd, dx

    y0, m0, r0, dx0 = run("0")
    y1, m1, r1, dx1 = run("1")
    scale = y_ref.abs().max()
    tol = scale / (128 if dt == 1 else 1024) + 1e-3
    assert torch.isfinite(y1.float()).all() and torch.isfinite(dx1.float()).all()
    assert (y1.float() - y_ref).abs().max() < tol
    assert (y1.float() - y0.float()).abs().max() < tol
    ref_mean = y_ref.reshape(B, -1, cout).mean(1).reshape(-1)
    ref_rstd = (y_ref.reshape(B, -1, cout).var(1, unbiased=False) + 1e-5).rsqrt().reshape(-1)
    assert (m1 - ref_mean).abs().max() < 1e-4 and (m1 - m0).abs().max() < 1e-5
    assert ((r1 - ref_rstd) / ref_rstd).abs().max() < 1e-4 and ((r1 - r0) / r0).abs().max() < 1e-5
    assert (dx1.float() - dx0.float()).abs().max() < dx0.float().abs().max() / (128 if dt == 1 else 1024) + 1e-3
    # run to run: the same bits (fixed job order, no atomics)
    y2, m2, r2, dx2 = run("1")
    assert torch.equal(y2, y1) and torch.equal(dx2, dx1) and torch.equal(m2, m1) and torch.equal(r2, r1)


def test_conv_mfma_timing_report(capsys):
    """Not a pass/fail perf gate: prints achieved TFLOP/s of the main layer shapes (read in gpurun logs)."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    rows = []
    for dt, name in ((1, "bf16"), (0, "fp32")):
        tdt = torch.bfloat16 if dt else torch.float32
        cp = 16 if dt else 8
        for (cin, cout, n, s) in ((32, 32, 128, 1), (64, 32, 128, 1), (64, 64, 64, 1), (128, 128, 32, 1), (32, 64, 128, 2)):
            if dt == 0 and n == 128 and cin == 64:
                continue
            x = torch.randn(1, n, n, n, cin, device=DEV).to(tdt)
            w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05
            wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // (2 if dt else 4), dtype=tdt, device=DEV)
            check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
            no = (n - 1) // s + 1
            y = torch.empty((1, no, no, no, cout), dtype=tdt, device=DEV)

            def run():
                check(lib.dgtta_conv3d_k3_fwd(ptr(x), cin, ptr(wpack), None, ptr(y), cout, None, 1, cin, cout, cin, cout,
                                              n, n, n, s, dt, 2, stream_of()), "fwd")
            # 3 warm-up launches (first use of a kernel loads its code object and the clocks ramp up after the host-side
            # setup: round 1's driver run showed 5.8 ms for the first row), then the median of 10 individually timed ones
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for e0, e1 in evs:
                e0.record()
                run()
                e1.record()
            torch.cuda.synchronize()
            ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
            tf = 2.0 * 27 * cin * cout * no ** 3 / (ms * 1e-3) / 1e12
            rows.append(f"conv3 {name} {cin:>3}->{cout:<3} {n}^3 s{s}: {ms:8.3f} ms  {tf:8.1f} TFLOP/s")
    with capsys.disabled():
        print("\n" + "\n".join(rows))


def _call_wgrad(x, dy, cin, cout, stride, dt, impl, with_bias=True, accumulate=0, dw=None, db=None):
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, D, H, W, ldx = x.shape
    do, ho, wo = dy.shape[1:4]
    dw = torch.empty((cout, cin, 3, 3, 3), device=DEV) if dw is None else dw
    db = (torch.empty((cout,), device=DEV) if db is None else db) if with_bias else None
    nb = lib.dgtta_conv3d_wgrad_ws_bytes(B, cin, cout, do, ho, wo)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib.dgtta_conv3d_k3_wgrad(ptr(x), ldx, ptr(dy), dy.shape[-1], ptr(dw), ptr(db), ptr(ws), nb, B, cin, cout, D, H,
                                    W, stride, accumulate, dt, impl, stream_of()), "wgrad")
    return dw, db


WCASES = [(1, 16, 32, 8, 8, 32), (1, 32, 32, 9, 7, 45), (2, 8, 64, 6, 10, 16), (1, 64, 32, 5, 9, 20),
          (1, 24, 40, 8, 8, 8), (1, 32, 32, 40, 12, 64), (1, 320, 320, 4, 4, 4)]


@pytest.mark.parametrize("case", WCASES)
def test_wgrad_mfma_fp32(case):
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 7)
    x = torch.randn(B, D, H, W, cin, device=DEV)
    dy = torch.randn(B, D, H, W, cout, device=DEV)
    dw1, db1 = _call_wgrad(x, dy, cin, cout, 1, 0, 1)
    dw2, db2 = _call_wgrad(x, dy, cin, cout, 1, 0, 2)
    xc = x.permute(0, 4, 1, 2, 3).cpu().double()
    gc = dy.permute(0, 4, 1, 2, 3).cpu().double()
    ref = torch.nn.grad.conv3d_weight(xc, (cout, cin, 3, 3, 3), gc, stride=1, padding=1).float()
    scale = ref.abs().max()
    assert (dw1.cpu() - ref).abs().max() < 3e-5 * scale
    assert (dw2.cpu() - ref).abs().max() < 3e-5 * scale
    assert torch.allclose(db1, db2)
    # accumulate: dw += second gradient
    dw3, _ = _call_wgrad(x, dy, cin, cout, 1, 0, 2, accumulate=1, dw=dw2.clone(), db=db2.clone())
    assert (dw3.cpu() - 2 * ref).abs().max() < 6e-5 * scale


def _call_wgrad_split(x, dy, cin, cout, accumulate=0, dw=None, stride=1):
    """fp32 weight gradient with the SPLIT workspace: six 16-bit launches on three-term bf16 splits (round 5)."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, D, H, W, ldx = x.shape
    do, ho, wo = dy.shape[1:4]
    dw = torch.empty((cout, cin, 3, 3, 3), device=DEV) if dw is None else dw
    nb = lib.dgtta_conv3d_wgrad_split_ws_bytes(B, cin, cout, do, ho, wo, stride)
    assert nb > lib.dgtta_conv3d_wgrad_ws_bytes(B, cin, cout, do, ho, wo)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib.dgtta_conv3d_k3_wgrad(ptr(x), ldx, ptr(dy), dy.shape[-1], ptr(dw), None, ptr(ws), nb, B, cin, cout, D, H, W, stride,
                                    accumulate, 0, 2, stream_of()), "wgrad split")
    return dw


@pytest.mark.parametrize("case", [(1, 16, 32, 8, 8, 32), (1, 32, 32, 9, 7, 45), (2, 12, 32, 6, 10, 16), (1, 64, 32, 5, 9, 20),
                                  (1, 24, 40, 8, 8, 8), (2, 32, 32, 40, 12, 64), (1, 320, 320, 4, 4, 4), (4, 64, 64, 16, 32, 32)])
def test_wgrad_fp32_as_six_bf16_products(case, monkeypatch):
    """Round 5: the fp32 weight gradient evaluated on the 16-bit matrix-core kernels - x and dy split EXACTLY
    into three bf16 terms each, the six products with i + j <= 2 accumulated in a fixed order - against a float64 torch
    evaluation, at the accuracy of the fp32 MFMA kernel it replaces (DGTTA_WGRAD_F32_SPLIT=0, run side by side): operands
    with a wide dynamic range (a 16-bit two-term split would lose them), ragged channel counts (12 of 16), batch > 1, the
    accumulate form, run-to-run identical."""
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 9)
    ld = (cin + 7) // 8 * 8 if cin % 4 else cin
    x = torch.zeros(B, D, H, W, ld, device=DEV)
    x[..., :cin] = torch.randn(B, D, H, W, cin, device=DEV) * torch.exp(3 * torch.randn(B, D, H, W, 1, device=DEV))
    dy = torch.randn(B, D, H, W, cout, device=DEV) * torch.exp(2 * torch.randn(1, 1, 1, 1, cout, device=DEV)) * 1e-6
    ref = torch.nn.grad.conv3d_weight(x[..., :cin].permute(0, 4, 1, 2, 3).cpu().double(), (cout, cin, 3, 3, 3),
                                      dy.permute(0, 4, 1, 2, 3).cpu().double(), stride=1, padding=1)
    scale = float(ref.abs().max())
    new = _call_wgrad_split(x, dy, cin, cout)
    monkeypatch.setenv("DGTTA_WGRAD_F32_SPLIT", "0")
    reload_kernel_switches()
    old = _call_wgrad_split(x, dy, cin, cout)              # same workspace, the switch selects the fp32 MFMA kernel
    monkeypatch.delenv("DGTTA_WGRAD_F32_SPLIT")
    reload_kernel_switches()
    e_new, e_old = float((new.cpu().double() - ref).abs().max()) / scale, float((old.cpu().double() - ref).abs().max()) / scale
    assert e_old < 3e-5, e_old
    assert e_new < 3e-5 and e_new < 4 * e_old + 2e-7, (e_new, e_old)      # fp32 rounding level, like its predecessor
    assert torch.equal(new, _call_wgrad_split(x, dy, cin, cout))
    acc = _call_wgrad_split(x, dy, cin, cout, accumulate=1, dw=new.clone())
    assert float((acc.cpu().double() - 2 * ref).abs().max()) / scale < 6e-5
    # stride 2 (the encoder transitions): the same operands as the input of a stride-2 conv of even extent
    if D % 2 == 0 and H % 2 == 0 and W % 2 == 0:
        dy2 = dy[:, ::2, ::2, ::2].contiguous()
        ref2 = torch.nn.grad.conv3d_weight(x[..., :cin].permute(0, 4, 1, 2, 3).cpu().double(), (cout, cin, 3, 3, 3),
                                           dy2.permute(0, 4, 1, 2, 3).cpu().double(), stride=2, padding=1)
        got2 = _call_wgrad_split(x, dy2, cin, cout, stride=2)
        assert float((got2.cpu().double() - ref2).abs().max()) / float(ref2.abs().max()) < 3e-5


@pytest.mark.parametrize("case", [c for c in WCASES if c[1] % 8 == 0])
def test_wgrad_mfma_bf16(case):
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 8)
    x = torch.randn(B, D, H, W, cin, device=DEV).bfloat16()
    dy = torch.randn(B, D, H, W, cout, device=DEV).bfloat16()
    dw_ref, _ = _call_wgrad(x.float(), dy.float(), cin, cout, 1, 0, 1)     # fp32 math on the same bf16 operands
    dw, _ = _call_wgrad(x, dy, cin, cout, 1, 1, 2)
    assert (dw - dw_ref).abs().max() < 1e-4 * dw_ref.abs().max() + 1e-4          # fp32 accumulation: only order differs


def test_wgrad_mfma_timing_report(capsys):
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    rows = []
    for dt, name in ((1, "bf16"), (0, "fp32")):
        tdt = torch.bfloat16 if dt else torch.float32
        for (cin, cout, n) in ((32, 32, 128), (64, 32, 128), (64, 64, 64), (128, 128, 32), (256, 256, 16)):
            if dt == 0 and n == 128 and cin == 64:
                continue
            x = torch.randn(1, n, n, n, cin, device=DEV).to(tdt)
            dy = torch.randn(1, n, n, n, cout, device=DEV).to(tdt)
            dw = torch.empty((cout, cin, 3, 3, 3), device=DEV)
            nb = lib.dgtta_conv3d_wgrad_ws_bytes(1, cin, cout, n, n, n)
            ws = torch.empty(nb, dtype=torch.uint8, device=DEV)

            def run():
                check(lib.dgtta_conv3d_k3_wgrad(ptr(x), cin, ptr(dy), cout, ptr(dw), None, ptr(ws), nb, 1, cin, cout, n, n,
                                                n, 1, 0, dt, 2, stream_of()), "wgrad")
            # 3 warm-up launches (first use of a kernel loads its code object and the clocks ramp up after the host-side
            # setup: round 1's driver run showed 5.8 ms for the first row), then the median of 10 individually timed ones
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for e0, e1 in evs:
                e0.record()
                run()
                e1.record()
            torch.cuda.synchronize()
            ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
            tf = 2.0 * 27 * cin * cout * n ** 3 / (ms * 1e-3) / 1e12
            rows.append(f"wgrad {name} {cin:>3}x{cout:<3} {n}^3: {ms:8.3f} ms  {tf:8.1f} TFLOP/s  (ws {nb / 2**20:.0f} MiB)")
    with capsys.disabled():
        print("\n" + "\n".join(rows))


@pytest.mark.parametrize("case", [(1, 32, 64, 8, 8, 32), (1, 16, 24, 10, 12, 14), (2, 64, 128, 4, 8, 16), (1, 32, 192, 4, 8, 8),
                                  (1, 40, 64, 6, 4, 34)])      # (round 4: three pairs of output-channel blocks; ragged Cin and width)
@pytest.mark.parametrize("dt", [0, 1])
def test_wgrad_stride2_mfma(case, dt):
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 9)
    tdt = torch.bfloat16 if dt else torch.float32
    x = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    dy = torch.randn(B, D // 2, H // 2, W // 2, cout, device=DEV).to(tdt)
    dw1, _ = _call_wgrad(x, dy, cin, cout, 2, dt, 1)
    dw2, _ = _call_wgrad(x, dy, cin, cout, 2, dt, 2)
    assert (dw1 - dw2).abs().max() < 1e-4 * dw1.abs().max() + 1e-4
    if dt == 0:
        ref = torch.nn.grad.conv3d_weight(x.permute(0, 4, 1, 2, 3).cpu().double(), (cout, cin, 3, 3, 3),
                                          dy.permute(0, 4, 1, 2, 3).cpu().double(), stride=2, padding=1).float()
        assert (dw2.cpu() - ref).abs().max() < 3e-5 * ref.abs().max()


@pytest.mark.parametrize("case", [(1, 64, 32, 4, 8, 16), (2, 24, 16, 4, 4, 8), (1, 320, 256, 4, 4, 4), (1, 96, 48, 4, 4, 8)])      # (round 4: an odd number of input-channel blocks)
@pytest.mark.parametrize("dt", [0, 1])
def test_convT_mfma(case, dt):
    """ConvTranspose3d k2 s2 forward / data gradient / weight gradient: MFMA composition vs VALU kernels vs torch."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    import torch.nn.functional as F
    lib = _lib.load()
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 10)
    tdt = torch.bfloat16 if dt else torch.float32
    x = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    w = torch.randn(cin, cout, 2, 2, 2, device=DEV) / cin ** 0.5
    if dt:
        w = w.bfloat16().float()
    bias = torch.randn(cout, device=DEV)
    outs, grads = [], []
    for impl in (1, 2):
        out = torch.empty((B, 2 * D, 2 * H, 2 * W, cout), dtype=tdt, device=DEV)
        nb = lib.dgtta_convT3d_fwd_ws_bytes(cin, cout, dt)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_convT3d_k2s2_fwd(ptr(x), cin, ptr(w), ptr(bias), ptr(out), cout, ptr(ws), nb, B, cin, cout, D, H, W,
                                         dt, impl, stream_of()), "convT fwd")
        outs.append(out)
        dout = torch.randn(B, 2 * D, 2 * H, 2 * W, cout, device=DEV, generator=torch.Generator(DEV).manual_seed(1)).to(tdt)
        dx = torch.empty_like(x)
        dw, db = torch.empty_like(w), torch.empty_like(bias)
        nb = lib.dgtta_convT3d_bwd_ws_bytes(B, cin, cout, D, H, W)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_convT3d_k2s2_bwd(ptr(x), cin, ptr(dout), cout, ptr(w), ptr(dx), cin, ptr(dw), ptr(db), ptr(ws), nb,
                                         B, cin, cout, D, H, W, 0, dt, impl, stream_of()), "convT bwd")
        grads.append((dx, dw, db, dout))
    tol = 2e-5 if dt == 0 else 1.0 / 100
    ref = F.conv_transpose3d(x.float().permute(0, 4, 1, 2, 3).cpu(), w.cpu(), bias.cpu(), stride=2).permute(0, 2, 3, 4, 1)
    for o in outs:
        assert (o.float().cpu() - ref).abs().max() < tol * ref.abs().max() + 1e-4
    (dx1, dw1, db1, _), (dx2, dw2, db2, dout) = grads
    assert (dx1.float() - dx2.float()).abs().max() < tol * dx1.float().abs().max() + 1e-4
    assert (dw1 - dw2).abs().max() < 1e-4 * dw1.abs().max() + 1e-4
    assert torch.allclose(db1, db2)
    if dt == 0:
        xr = x.permute(0, 4, 1, 2, 3).cpu().double().requires_grad_(True)
        wr = w.cpu().double().requires_grad_(True)
        F.conv_transpose3d(xr, wr, None, stride=2).backward(dout.permute(0, 4, 1, 2, 3).cpu().double())
        assert (dx2.cpu() - xr.grad.permute(0, 2, 3, 4, 1).float()).abs().max() < 3e-5 * xr.grad.abs().max()
        assert (dw2.cpu() - wr.grad.float()).abs().max() < 3e-5 * wr.grad.abs().max()


@pytest.mark.parametrize("case", [(2, 64, 32, 8, 8, 16), (1, 24, 16, 4, 6, 8), (1, 320, 256, 4, 4, 4)])
def test_convT_wgrad_fp32_as_six_bf16_products(case, monkeypatch):
    """Round 5: the fp32 ConvTranspose3d weight gradient as six launches of the 16-bit kernel on three-term bf16 splits (the caller
    offers dgtta_convT3d_bwd_split_ws_bytes) against the fp32 MFMA kernel (plain workspace / DGTTA_WGRAD_F32_SPLIT=0) and float64:
    the same rounding level; the data gradient and the bias gradient do not change; accumulate mode adds to what is there."""
    from conftest import reload_kernel_switches
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    import torch.nn.functional as F
    lib = _lib.load()
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 3)
    x = torch.randn(B, D, H, W, cin, device=DEV)
    w = torch.randn(cin, cout, 2, 2, 2, device=DEV) / cin ** 0.5
    dout = torch.randn(B, 2 * D, 2 * H, 2 * W, cout, device=DEV)

    def run(split, accumulate=0, dw0=None):
        nb = (lib.dgtta_convT3d_bwd_split_ws_bytes if split else lib.dgtta_convT3d_bwd_ws_bytes)(B, cin, cout, D, H, W)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w) if dw0 is None else dw0.clone()
        db = torch.zeros(cout, device=DEV) if accumulate else torch.empty(cout, device=DEV)
        check(lib.dgtta_convT3d_k2s2_bwd(ptr(x), cin, ptr(dout), cout, ptr(w), ptr(dx), cin, ptr(dw), ptr(db), ptr(ws), nb, B, cin, cout,
                                         D, H, W, accumulate, 0, 2, stream_of()), "convT bwd")
        torch.cuda.synchronize()
        return dx, dw, db

    assert lib.dgtta_convT3d_bwd_split_ws_bytes(B, cin, cout, D, H, W) > lib.dgtta_convT3d_bwd_ws_bytes(B, cin, cout, D, H, W)
    dx_o, dw_o, db_o = run(False)
    dx_n, dw_n, db_n = run(True)
    wr = w.cpu().double().requires_grad_(True)
    F.conv_transpose3d(x.permute(0, 4, 1, 2, 3).cpu().double(), wr, None, stride=2).backward(dout.permute(0, 4, 1, 2, 3).cpu().double())
    scale = float(wr.grad.abs().max())
    e_old = float((dw_o.cpu().double() - wr.grad).abs().max()) / scale
    e_new = float((dw_n.cpu().double() - wr.grad).abs().max()) / scale
    assert e_old < 3e-5 and e_new < 3e-5 and e_new < 4 * e_old + 2e-7, (e_new, e_old)
    assert not torch.equal(dw_n, dw_o)                                   # (another kernel did run)
    assert torch.equal(dx_n, dx_o) and torch.equal(db_n, db_o)
    assert torch.equal(run(True)[1], dw_n)                               # run to run: the same bits
    _, acc, _ = run(True, accumulate=1, dw0=dw_n)
    assert float((acc.cpu().double() - 2 * wr.grad).abs().max()) / scale < 6e-5
    monkeypatch.setenv("DGTTA_WGRAD_F32_SPLIT", "0")
    reload_kernel_switches()
    assert torch.equal(run(True)[1], dw_o)                               # switched off: the fp32 kernel, whatever the workspace
    monkeypatch.delenv("DGTTA_WGRAD_F32_SPLIT")
    reload_kernel_switches()


@pytest.mark.parametrize("nsel,V", [(16, 8 * 8 * 32), (5, 7 * 9 * 11), (105, 4 * 8 * 16), (105, 64 * 37 + 5), (40, 1000), (128, 130)])
@pytest.mark.parametrize("dt", [0, 1, 2])
def test_head_fast_paths(nsel, V, dt):
    """1x1x1 head (Cin=32) forward / dgrad / wgrad fast paths vs a dense torch reference; round 5: more than 32 evaluated
    classes (the full 105-class head: pre-training, or TTA behind a user-defined output modifier) run the LDS-tiled wide kernels,
    ragged last tiles and several splits included."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    torch.manual_seed(nsel + V)
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    B, cin, ncls = 1, 32, max(105, nsel)
    x = torch.randn(B, V, cin, device=DEV).to(tdt)
    w = torch.randn(ncls, cin, device=DEV)
    bias = torch.randn(ncls, device=DEV)
    sel = None if nsel == ncls else torch.randperm(ncls, device=DEV)[:nsel].int()
    out = torch.empty((B, V, nsel), device=DEV)
    check(lib.dgtta_seghead_fwd(ptr(x), cin, ptr(w), ptr(bias), ptr(sel), nsel, ptr(out), 1, nsel, B, cin, V, dt, stream_of()),
          "head fwd")
    wsel = w if sel is None else w[sel.long()]
    bsel = bias if sel is None else bias[sel.long()]
    ref = x.float() @ wsel.t() + bsel
    assert (out - ref).abs().max() < 2e-5 * ref.abs().max() + 1e-5
    dout = torch.randn(B, V, nsel, device=DEV)
    dx = torch.empty((B, V, cin), dtype=tdt, device=DEV)
    dw, db = torch.empty((nsel, cin), device=DEV), torch.empty((nsel,), device=DEV)
    nb = lib.dgtta_seghead_bwd_ws_bytes(B, cin, nsel, V)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib.dgtta_seghead_bwd(ptr(x), cin, ptr(dout), nsel, ptr(w), ptr(sel), nsel, ptr(dx), cin, ptr(dw), ptr(db), ptr(ws),
                                nb, B, cin, V, 0, dt, stream_of()), "head bwd")
    tol = 2e-5 if dt == 0 else 1e-2
    dx_ref = dout @ wsel
    assert (dx.float() - dx_ref).abs().max() < tol * dx_ref.abs().max() + 1e-5
    dw_ref = dout[0].t() @ x[0].float()
    wtol = 3e-5 if dt == 0 else 2e-2        # bf16 MFMA path rounds dout to bf16
    assert (dw - dw_ref).abs().max() < wtol * dw_ref.abs().max() + 1e-4
    assert torch.allclose(db, dout[0].sum(0), rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_128_conv_two_kernels_agree_and_are_linear(monkeypatch):
    """BASELINE size (128^3, 32 -> 32 channels, bf16): the row-reuse kernel and the generic kernel are independent
    implementations and must agree to output rounding; the fused statistics must match the tensor; and the map is linear
    (conv(x1 + 2 x2) = conv(x1) + 2 conv(x2) up to bf16 rounding of the three outputs)."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    n, c = 128, 32
    g = torch.Generator(DEV).manual_seed(1234)
    x1 = torch.randn(1, n, n, n, c, device=DEV, generator=g).bfloat16()
    x2 = torch.randn(1, n, n, n, c, device=DEV, generator=g).bfloat16()
    w = torch.randn(c, c, 3, 3, 3, device=DEV, generator=g) / (27 * c) ** 0.5
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(c, c, 1) // 2, dtype=torch.bfloat16, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), c, c, c, c, 1, stream_of()), "pack")
    nb = lib.dgtta_conv3d_stats_bytes(1, c, n, n, n)

    def conv(x, rows):
        monkeypatch.setenv("DGTTA_CONV_ROWS", rows)
        reload_kernel_switches()
        y = torch.empty((1, n, n, n, c), dtype=torch.bfloat16, device=DEV)
        st = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_conv3d_k3_fwd(ptr(x), c, ptr(wpack), None, ptr(y), c, ptr(st), 1, c, c, c, c, n, n, n, 1, 1, 2,
                                      stream_of()), "fwd")
        mr = torch.empty(1, c, 2, device=DEV)
        z = torch.empty_like(y)
        one, zero = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        nws = lib.dgtta_instnorm_ws_bytes(1, c, n ** 3)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_instnorm_lrelu_fwd(ptr(y), c, ptr(st), ptr(one), ptr(zero), ptr(mr), ptr(z), c, ptr(ws), nws, 1, c,
                                           n ** 3, 1e-5, 0.01, 1, stream_of()), "instnorm")
        return y.float(), mr[0, :, 0].clone(), mr[0, :, 1].clone()

    y_rows, m_rows, r_rows = conv(x1, "1")
    y_gen, m_gen, r_gen = conv(x1, "0")
    scale = float(y_gen.abs().max())
    assert float((y_rows - y_gen).abs().max()) < scale / 128 + 1e-3
    assert float((m_rows - m_gen).abs().max()) < 1e-5 and float(((r_rows - r_gen) / r_gen).abs().max()) < 1e-5
    flat = y_rows.reshape(-1, c)
    assert float((m_rows - flat.mean(0)).abs().max()) < 2e-4            # statistics of the unrounded values vs the bf16 tensor
    assert float((r_rows - (flat.var(0, unbiased=False) + 1e-5).rsqrt()).abs().max() / r_rows.mean()) < 2e-3
    x3 = (x1.float() + 2.0 * x2.float())
    x3b = x3.bfloat16()
    assert float((x3b.float() - x3).abs().max()) < 0.05                   # x3 itself is rounded to bf16 (|x| < ~8)
    y2, _, _ = conv(x2, "1")
    y3, _, _ = conv(x3b, "1")
    # tolerance: rounding of x3 (2^-9 relative per element, averaged by the 864-term sum) + three output roundings
    assert float((y3 - (y_rows + 2.0 * y2)).abs().max()) < 3.0 * scale / 64 + 1e-2


def test_full_size_128_wgrad_two_kernels_agree(monkeypatch):
    """BASELINE size (128^3, 32 x 32 channels, bf16): the transposed-read weight-gradient kernel against the
    register-transpose kernel (independent staging, MFMA shape and reduction order; fp32 accumulation in both)."""
    n, c = 128, 32
    g = torch.Generator(DEV).manual_seed(4321)
    x = torch.randn(1, n, n, n, c, device=DEV, generator=g).bfloat16()
    dy = torch.randn(1, n, n, n, c, device=DEV, generator=g).bfloat16()
    monkeypatch.setenv("DGTTA_WGRAD_TR", "1")
    reload_kernel_switches()
    dw1, db1 = _call_wgrad(x, dy, c, c, 1, 1, 2)
    monkeypatch.setenv("DGTTA_WGRAD_TR", "0")
    reload_kernel_switches()
    dw0, db0 = _call_wgrad(x, dy, c, c, 1, 1, 2)
    assert float((dw1 - dw0).abs().max()) < 2e-4 * float(dw0.abs().max()) + 1e-3
    assert torch.equal(db1, db0)
    # checksum property: summing dW over taps and input channels = (sum over a 3^3 box of x) . dy, checked for tap 13
    centre = torch.einsum("vc,vk->kc", x.reshape(-1, c).float(), dy.reshape(-1, c).float())
    assert float((dw1[:, :, 1, 1, 1] - centre).abs().max()) < 2e-4 * float(centre.abs().max()) + 1e-2


@pytest.mark.parametrize("case", [(1, 32, 64, 16, 8, 64, 1), (2, 16, 24, 12, 12, 36, 1), (1, 64, 128, 8, 8, 16, 0),
                                  (1, 320, 320, 4, 4, 8, 1), (1, 32, 64, 16, 8, 64, 0)])
@pytest.mark.parametrize("dt", [0, 1])
def test_dgrad_stride2_all_classes(case, dt, monkeypatch):
    """Stride-2 data gradient: the single-pass kernel (8 parity classes accumulated per wave) against the 8-class
    launch and the VALU kernel, with and without accumulation into dx (skip-connection sum); (B, cin, cout, Di, Hi, Wi, acc)."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, cin, cout, D, H, W, acc = case
    tdt = torch.float32 if dt == 0 else torch.bfloat16
    m = 16 if dt else 8
    cinp, coutp = (cin + m - 1) // m * m, (cout + m - 1) // m * m
    torch.manual_seed(sum(case) + dt)
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cout) ** 0.5).to(tdt).float()
    dy = torch.randn(B, D // 2, H // 2, W // 2, cout, device=DEV).to(tdt)
    dx0 = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cinp, coutp, dt) // (2 if dt else 4), dtype=tdt, device=DEV)
    check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cinp, coutp, dt, stream_of()), "pack")

    def run(impl, allcls):
        monkeypatch.setenv("DGTTA_DGRAD_S2_ALLCLS", allcls)
        reload_kernel_switches()
        dx = dx0.clone()
        check(lib.dgtta_conv3d_k3_dgrad(ptr(dy), cout, ptr(wpack), ptr(dx), cin, B, cin, cout, cinp, coutp, D, H, W, 2, acc,
                                        dt, impl, stream_of()), "dgrad")
        torch.cuda.synchronize()
        return dx.float()

    ref, cls8, one = run(1, "0"), run(2, "0"), run(2, "1")
    tol = (2e-5 if dt == 0 else 1.0 / 64) * float(ref.abs().max()) + 1e-4
    assert float((cls8 - ref).abs().max()) < tol
    assert float((one - ref).abs().max()) < tol
    assert float((one - cls8).abs().max()) < tol


@pytest.mark.parametrize("case", [(1, 32, 64, 8, 12, 72), (2, 16, 32, 6, 6, 20), (1, 64, 96, 4, 8, 32), (1, 320, 320, 4, 4, 8)])
def test_wgrad_stride2_one_pass_bf16(case, monkeypatch):
    """Stride-2 weight gradient (bf16): the one-pass kernel (x tile at full resolution, transposed reads with a 2-voxel
    row stride) against the 8 parity-class launch and the VALU kernel on the same operands."""
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case) + 19)
    x = torch.randn(B, D, H, W, cin, device=DEV).bfloat16()
    dy = torch.randn(B, D // 2, H // 2, W // 2, cout, device=DEV).bfloat16()
    ref, _ = _call_wgrad(x.float(), dy.float(), cin, cout, 2, 0, 1)
    monkeypatch.setenv("DGTTA_WGRAD_S2_ONEPASS", "0")
    reload_kernel_switches()
    cls8, _ = _call_wgrad(x, dy, cin, cout, 2, 1, 2)
    monkeypatch.setenv("DGTTA_WGRAD_S2_ONEPASS", "1")
    reload_kernel_switches()
    one, _ = _call_wgrad(x, dy, cin, cout, 2, 1, 2)
    tol = 1e-4 * float(ref.abs().max()) + 1e-4
    assert float((cls8 - ref).abs().max()) < tol
    assert float((one - ref).abs().max()) < tol
    # accumulate into existing gradients
    dw = ref.clone()
    _call_wgrad(x, dy, cin, cout, 2, 1, 2, accumulate=1, dw=dw, db=torch.zeros(cout, device=DEV))
    assert float((dw - 2 * ref).abs().max()) < 2 * tol


@pytest.mark.parametrize("case", [(1, 64, 32, 4, 6, 40), (2, 128, 64, 3, 4, 32), (1, 64, 32, 2, 3, 96)])
@pytest.mark.parametrize("dt", [1, 2])
def test_convT_register_operand_kernel(case, dt, monkeypatch):
    """ConvTranspose k2 s2 of the two large decoder stages: the register-operand GEMM kernel (csrc/convt_gemm.hip) against
    its predecessor (DGTTA_CONVT_GEMM=0) and torch, with the decoder's operand strides: the output / its gradient are the
    first Cout channels of a concat buffer of 2 * Cout, ragged last 32-voxel block, batch 2."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, cin, cout, D, H, W = case
    tdt = torch.bfloat16 if dt == 1 else torch.float16
    torch.manual_seed(sum(case) + dt)
    x = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    w = (torch.randn(cin, cout, 2, 2, 2, device=DEV) / cin ** 0.5).to(tdt).float()
    bias = torch.randn(cout, device=DEV)
    dcat = torch.randn(B, 2 * D, 2 * H, 2 * W, 2 * cout, device=DEV).to(tdt)          # gradient of the concat buffer

    def run(gemm):
        monkeypatch.setenv("DGTTA_CONVT_GEMM", gemm)
        reload_kernel_switches()
        cat = torch.full((B, 2 * D, 2 * H, 2 * W, 2 * cout), 7.0, dtype=tdt, device=DEV)
        nb = lib.dgtta_convT3d_fwd_ws_bytes(cin, cout, dt)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_convT3d_k2s2_fwd(ptr(x), cin, ptr(w), ptr(bias), ptr(cat), 2 * cout, ptr(ws), nb, B, cin, cout, D, H, W,
                                         dt, 2, stream_of()), "convT fwd")
        dx = torch.empty_like(x)
        dw, db = torch.empty_like(w), torch.empty_like(bias)
        nb = lib.dgtta_convT3d_bwd_ws_bytes(B, cin, cout, D, H, W)
        ws2 = torch.empty(nb, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_convT3d_k2s2_bwd(ptr(x), cin, ptr(dcat), 2 * cout, ptr(w), ptr(dx), cin, ptr(dw), ptr(db), ptr(ws2), nb,
                                         B, cin, cout, D, H, W, 0, dt, 2, stream_of()), "convT bwd")
        torch.cuda.synchronize()
        return cat.float(), dx.float(), dw

    cat_new, dx_new, dw_new = run("1")
    cat_old, dx_old, dw_old = run("0")
    assert torch.equal(cat_new[..., cout:], torch.full_like(cat_new[..., cout:], 7.0))      # the skip half is not touched
    ref = F.conv_transpose3d(x.float().permute(0, 4, 1, 2, 3).cpu(), w.cpu(), bias.cpu(), stride=2).permute(0, 2, 3, 4, 1)
    eps = 2.0 ** -8 if dt == 1 else 2.0 ** -11
    assert float((cat_new[..., :cout].cpu() - ref).abs().max()) < eps * float(ref.abs().max()) + 1e-4
    assert float((cat_new - cat_old).abs().max()) < 2 * eps * float(ref.abs().max()) + 1e-4
    xr = x.float().permute(0, 4, 1, 2, 3).cpu().double().requires_grad_(True)
    F.conv_transpose3d(xr, w.cpu().double(), None, stride=2).backward(dcat[..., :cout].float().permute(0, 4, 1, 2, 3).cpu().double())
    gref = xr.grad.permute(0, 2, 3, 4, 1).float()
    assert float((dx_new.cpu() - gref).abs().max()) < eps * float(gref.abs().max()) + 1e-4
    assert float((dx_new - dx_old).abs().max()) < 2 * eps * float(gref.abs().max()) + 1e-4
    assert torch.equal(dw_new, dw_old)              # the weight gradient does not go through the new kernel


@pytest.mark.parametrize("case", [(1, 32, 64, 6, 10, 72), (2, 64, 128, 4, 6, 64), (1, 32, 128, 9, 7, 66), (8, 64, 32, 2, 2, 64)])
@pytest.mark.parametrize("dt", [1, 2])
def test_conv_stride2_register_operand_kernel(case, dt, monkeypatch):
    """Stride-2 forward of the large encoder transitions: the register-operand kernel (csrc/conv_s2.hip) against its
    predecessor (DGTTA_CONV_S2=1) and torch on the same 16-bit operands, with the fused InstanceNorm statistics; odd
    extents (zero padding on the high side too), ragged last 32-voxel block, batch 2 and 8, several channel groups."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    B, cin, cout, D, H, W = case
    tdt = torch.bfloat16 if dt == 1 else torch.float16
    torch.manual_seed(sum(case) + dt)
    xb = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    w = (torch.randn(cout, cin, 3, 3, 3, device=DEV) / (27 * cin) ** 0.5).to(tdt).float()
    bias = torch.randn(cout, device=DEV)
    Do, Ho, Wo = [(n - 1) // 2 + 1 for n in (D, H, W)]

    def run(s2):
        monkeypatch.setenv("DGTTA_CONV_S2", s2)
        reload_kernel_switches()
        wpack = torch.empty(lib.dgtta_conv3d_packed_bytes(cin, cout, dt) // 2, dtype=tdt, device=DEV)
        check(lib.dgtta_conv3d_pack_weights(ptr(w), ptr(wpack), cin, cout, cin, cout, dt, stream_of()), "pack")
        y = torch.full((B, Do, Ho, Wo, cout), float("nan"), dtype=tdt, device=DEV)
        st = torch.zeros(lib.dgtta_conv3d_stats_bytes(B, cout, Do, Ho, Wo), dtype=torch.uint8, device=DEV)
        check(lib.dgtta_conv3d_k3_fwd(ptr(xb), cin, ptr(wpack), ptr(bias), ptr(y), cout, ptr(st), B, cin, cout, cin, cout,
                                      D, H, W, 2, dt, 2, stream_of()), "fwd")
        mr = torch.empty(B, cout, 2, device=DEV)
        z = torch.empty_like(y)
        gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
        nws = lib.dgtta_instnorm_ws_bytes(B, cout, Do * Ho * Wo)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        check(lib.dgtta_instnorm_lrelu_fwd(ptr(y), cout, ptr(st), ptr(gamma), ptr(beta), ptr(mr), ptr(z), cout, ptr(ws), nws,
                                           B, cout, Do * Ho * Wo, 1e-5, 0.01, dt, stream_of()), "instnorm")
        torch.cuda.synchronize()
        return y.float(), mr[..., 0].reshape(-1).clone(), mr[..., 1].reshape(-1).clone()

    y1, m1, r1 = run("0")
    y0, m0, r0 = run("1")
    ref = F.conv3d(xb.float().permute(0, 4, 1, 2, 3).cpu(), w.cpu(), bias.cpu(), stride=2, padding=1).permute(0, 2, 3, 4, 1)
    eps = 2.0 ** -8 if dt == 1 else 2.0 ** -11
    scale = float(ref.abs().max())
    assert torch.isfinite(y1).all()
    assert float((y1.cpu() - ref).abs().max()) < eps * scale + 1e-4
    assert float((y1 - y0).abs().max()) < 2 * eps * scale + 1e-4
    ref_mean = ref.reshape(B, -1, cout).mean(1).reshape(-1)
    ref_rstd = (ref.reshape(B, -1, cout).var(1, unbiased=False) + 1e-5).rsqrt().reshape(-1)
    assert float((m1.cpu() - ref_mean).abs().max()) < 1e-4 and float((m1 - m0).abs().max()) < 1e-5
    assert float(((r1.cpu() - ref_rstd) / ref_rstd).abs().max()) < 1e-4 and float(((r1 - r0) / r0).abs().max()) < 1e-5


WRING_CASES = [(1, 32, 32, 9, 13, 45), (2, 12, 32, 6, 17, 32), (1, 16, 64, 9, 11, 37), (2, 12, 32, 40, 16, 64), (1, 64, 96, 7, 8, 70), (2, 32, 32, 32, 32, 64),
               (1, 32, 64, 36, 40, 32), (8, 32, 32, 4, 64, 160), (1, 24, 32, 6, 9, 33), (1, 40, 64, 5, 10, 40), (1, 8, 32, 7, 8, 32)]
# ragged edges, ragged Cin (also inside a 32-channel block and below 16), channel-block pairs, > 256 columns


@pytest.mark.parametrize("dts", ["bf16", "fp16"])
@pytest.mark.parametrize("case", WRING_CASES)
def test_wgrad_ring_kernel(case, dts, monkeypatch):
    """The persistent weight-gradient sweep (conv_wgrad_ring.hip: 8-wave workgroup per CU, x ring of 5 slices / dy ring of 3,
    counted waits), forced on at small sizes, against its predecessors (DGTTA_WGRAD_RING=0) and torch on the same operands."""
    B, cin, cout, D, H, W = case
    dt, tdt = (1, torch.bfloat16) if dts == "bf16" else (2, torch.float16)
    torch.manual_seed(sum(case) + 21)
    ld = (cin + 7) // 8 * 8
    x = torch.zeros(B, D, H, W, ld, device=DEV, dtype=tdt)
    x[..., :cin] = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    dy = torch.randn(B, D, H, W, cout, device=DEV).to(tdt)

    def run(ring):
        monkeypatch.setenv("DGTTA_WGRAD_RING", ring)
        reload_kernel_switches()
        dw, db = _call_wgrad(x, dy, cin, cout, 1, dt, 2)
        torch.cuda.synchronize()
        return dw

    old, new, new16 = run("0"), run("1"), run("4")       # "4": the 16x16x32 MFMA form with the half-swapped LDS image
    # round 5: "1" shares x operands between neighbouring rows (whole tap columns per wave group, k-step outside the rows);
    # "6" is the form before that (taps dealt round robin, every operand read)
    no_reuse = run("6")
    assert torch.isfinite(no_reuse).all()
    assert float((new - no_reuse).abs().max()) < 1e-4 * float(no_reuse.abs().max()) + 1e-3
    if cin <= 16:                                        # round 5: "1" / "6" run two taps per MFMA for the first layer, "5" one tap
        one_tap = run("5")
        assert torch.isfinite(one_tap).all()
        assert float((no_reuse - one_tap).abs().max()) <= 2e-6 * float(one_tap.abs().max()) + 1e-6      # same products, same k order per tap
    ref = torch.nn.grad.conv3d_weight(x[..., :cin].float().permute(0, 4, 1, 2, 3).cpu().double(), (cout, cin, 3, 3, 3),
                                      dy.float().permute(0, 4, 1, 2, 3).cpu().double(), stride=1, padding=1).float()
    scale = float(ref.abs().max())
    assert torch.isfinite(new).all()
    assert float((new.cpu() - ref).abs().max()) < 2e-4 * scale + 1e-3
    assert float((new - old).abs().max()) < 1e-4 * scale + 1e-3       # same products, another fp32 summation order
    assert float((new16 - old).abs().max()) < 1e-4 * scale + 1e-3
    assert torch.equal(run("1"), new)                                  # run to run: the same bits


WFLAT_CASES = [(8, 64, 64, 8, 8, 8), (8, 32, 96, 4, 4, 4), (2, 64, 32, 16, 16, 16), (1, 24, 40, 5, 7, 6), (3, 320, 320, 4, 4, 4),
               (1, 32, 32, 9, 16, 13), (2, 40, 64, 3, 11, 16), (1, 32, 32, 1, 1, 1), (1, 64, 64, 16, 16, 16), (5, 32, 32, 2, 3, 2)]
# the network's small levels, ragged planes and channels, one sample with D segments, units that do not divide by workgroup


@pytest.mark.parametrize("dts", ["bf16", "fp16"])
@pytest.mark.parametrize("case", WFLAT_CASES)
def test_wgrad_flat_plane_kernel(case, dts, monkeypatch):
    """Planes of W <= 16 as one flat k-run with row pitch W + 2 (conv3_wgrad_flat_kernel, round 6) against the row kernels it
    replaces there (DGTTA_WGRAD_FLAT=0) and torch float64 on the same operands; accumulation into dw; same bits run to run."""
    B, cin, cout, D, H, W = case
    dt, tdt = (1, torch.bfloat16) if dts == "bf16" else (2, torch.float16)
    torch.manual_seed(sum(case) + 77)
    ld = (cin + 7) // 8 * 8
    x = torch.zeros(B, D, H, W, ld, device=DEV, dtype=tdt)
    x[..., :cin] = torch.randn(B, D, H, W, cin, device=DEV).to(tdt)
    dy = torch.randn(B, D, H, W, cout, device=DEV).to(tdt)

    def run(flat, **kw):
        monkeypatch.setenv("DGTTA_WGRAD_FLAT", flat)
        reload_kernel_switches()
        dw, db = _call_wgrad(x, dy, cin, cout, 1, dt, 2, **kw)
        torch.cuda.synchronize()
        return dw

    old, new = run("0"), run("1")
    ref = torch.nn.grad.conv3d_weight(x[..., :cin].float().permute(0, 4, 1, 2, 3).cpu().double(), (cout, cin, 3, 3, 3),
                                      dy.float().permute(0, 4, 1, 2, 3).cpu().double(), stride=1, padding=1).float()
    scale = float(ref.abs().max())
    assert torch.isfinite(new).all()
    assert float((new.cpu() - ref).abs().max()) < 2e-4 * scale + 1e-3
    assert float((new - old).abs().max()) < 1e-4 * scale + 1e-3       # same products, another fp32 summation order
    # the operands are exact in 16 bits and their products exact in fp32: what differs from float64 is fp32 summation alone,
    # and the flat run must be as good at it as the kernels it replaces
    err_new, err_old = float((new.cpu() - ref).abs().max()), float((old.cpu() - ref).abs().max())
    assert err_new <= 2.0 * err_old + 2e-6 * scale, (err_new, err_old, scale)
    assert torch.equal(run("1"), new)
    monkeypatch.setenv("DGTTA_WGRAD_REDUCE_TAPS", "0")      # the slab reduction with one output row per 32 lanes: the same bits
    assert torch.equal(run("1"), new)
    monkeypatch.delenv("DGTTA_WGRAD_REDUCE_TAPS")
    base = torch.randn_like(new)
    acc = run("1", accumulate=1, dw=base.clone())
    assert float((acc - (base + new)).abs().max()) < 1e-5 * scale + 1e-5
    monkeypatch.delenv("DGTTA_WGRAD_FLAT")
    reload_kernel_switches()


@pytest.mark.parametrize("case", [(2, 32, 32, 8, 12, 40), (1, 32, 64, 6, 8, 64), (2, 64, 96, 5, 9, 33)])
@pytest.mark.parametrize("upw", ["2", "3"])
def test_wgrad_units_per_workgroup(case, upw, monkeypatch):
    """Weight-gradient kernels sweeping several units (columns of the volume) into one partial slab: forced here at
    small sizes (the product picks it when there are >= 2 units per workgroup slot) against one unit per workgroup and
    torch; 3 does not divide the unit counts, so the last workgroup of a row runs short."""
    B, cin, cout, D, H, W = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, D, H, W, cin, device=DEV).bfloat16()
    dy = torch.randn(B, D, H, W, cout, device=DEV).bfloat16()

    def run(u):
        monkeypatch.setenv("DGTTA_WGRAD_UPW", u)
        reload_kernel_switches()
        dw, db = _call_wgrad(x, dy, cin, cout, 1, 1, 2)
        torch.cuda.synchronize()
        return dw

    one, many = run("1"), run(upw)
    ref = torch.nn.grad.conv3d_weight(x.float().permute(0, 4, 1, 2, 3).cpu().double(), (cout, cin, 3, 3, 3),
                                      dy.float().permute(0, 4, 1, 2, 3).cpu().double(), stride=1, padding=1).float()
    scale = float(ref.abs().max())
    assert float((many.cpu() - ref).abs().max()) < 2e-4 * scale + 1e-3
    assert float((many - one).abs().max()) < 1e-4 * scale + 1e-3       # same products, another fp32 summation order
