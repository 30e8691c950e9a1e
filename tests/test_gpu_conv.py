"""GPU parity of the implicit-GEMM conv / dgrad / wgrad kernels against a plain
PyTorch fp32 reference of the same op on identical (bf16-representable) inputs.

Tolerance: the kernels accumulate in fp32 and round the result once to bf16, so
|err| <= 2^-8 * |ref| (half a bf16 ulp, rtol 4e-3) plus fp32 accumulation-order
noise; we allow rtol=8e-3, atol=2e-2*scale.  wgrad is fp32 output: rtol 2e-3.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(B, H, W, C, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn(B, H, W, C, generator=g) * scale).to(torch.bfloat16)
    return x.to(dev)


def _nchw(x_nhwc):
    return x_nhwc.float().permute(0, 3, 1, 2).contiguous()


def _close(got, ref, rtol, atol):
    got = got.float()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = (err > lim)
    assert not bad.any(), f"max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g}), {bad.sum().item()} / {bad.numel()} out of tol"


CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 16, 16, 64, 32, 1, 1, 0),
    (2, 16, 16, 32, 64, 3, 1, 1),
    (2, 16, 16, 32, 64, 3, 2, 1),
    (1, 20, 20, 128, 255, 1, 1, 0),      # Detect-like, N not a multiple of 8
    (3, 12, 20, 16, 32, 3, 1, 1),        # stem-like (K tile spans two taps), rows not multiple of 128
    (2, 8, 8, 256, 512, 3, 2, 1),        # several N tiles
    (1, 40, 40, 64, 128, 3, 1, 1),
    (2, 24, 24, 80, 160, 3, 1, 1),       # YOLOv5x / v5m widths: channel counts that are not multiples of 32 (ragged last block)
    (2, 24, 24, 48, 96, 3, 2, 1),
    (2, 16, 16, 80, 80, 1, 1, 0),
    (2, 10, 10, 512, 128, 1, 1, 0),      # pointwise layer on the general (K > 384) weight-gradient tiling
    (2, 20, 20, 256, 256, 1, 1, 0),      # C3 1x1 layer: wide tiling by default, the general one on request (tile_k 128)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", CASES)
def test_conv_fwd(dev, B, H, W, Cin, Cout, k, s, p):
    from yoloseries_amd import hipk
    x = _nhwc(B, H, W, Cin, dev, 1)
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(torch.bfloat16).float().to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ldo = ((Cout + 7) // 8) * 8
    out = torch.full((B, Ho, Wo, ldo), 7.0, dtype=torch.bfloat16, device=dev)
    wp = hipk.pack_weight_fwd(w)
    d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, k, s, p, wp, Cout, hipk.full(out), bias=bias)
    nblk = hipk.conv_stat_blocks(d)
    stats = torch.zeros(nblk, 2, wp.shape[0], device=dev)
    d.stats = stats.data_ptr()
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    ref = F.conv2d(_nchw(x), w, bias, stride=s, padding=p).permute(0, 2, 3, 1)
    _close(out[..., :Cout], ref, 8e-3, 2e-2)
    # statistics are taken over the stored (bf16-rounded) values
    o = out[..., :Cout].float().reshape(-1, Cout)
    ssum = stats[:, 0, :Cout].double().sum(0)
    ssq = stats[:, 1, :Cout].double().sum(0)
    assert torch.allclose(ssum, o.double().sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(ssq, (o.double() ** 2).sum(0), rtol=1e-4, atol=1e-2)


def test_conv_fwd_concat_upsample_silu_res(dev):
    """Two-segment input (first one read through nearest-2x upsample), folded BN + SiLU
    epilogue, residual add and split destination."""
    from yoloseries_amd import hipk
    B, H, W = 2, 16, 16
    lo = _nhwc(B, H // 2, W // 2, 64, dev, 3)
    skip_buf = _nhwc(B, H, W, 96, dev, 4)         # use channels [32, 96) of a wider buffer
    g = torch.Generator().manual_seed(5)
    Cin, Cout = 128, 64
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(torch.bfloat16).float().to(dev)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    res = _nhwc(B, H, W, 32, dev, 6)
    out0 = torch.zeros(B, H, W, 32, dtype=torch.bfloat16, device=dev)
    out1 = torch.zeros(B, H, W, 64, dtype=torch.bfloat16, device=dev)   # write into channels [32,64)
    wp = hipk.pack_weight_fwd(w)
    d = hipk.conv_desc([hipk.Slice(lo, 0, 64, ups=1), hipk.Slice(skip_buf, 32, 64)], hipk.YH_CONV_FWD,
                       B, H, W, H, W, 1, 1, 0, wp, Cout, hipk.full(out0), nsplit=32,
                       out1=hipk.Slice(out1, 32, 32), scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    xin = torch.cat([F.interpolate(_nchw(lo), scale_factor=2, mode="nearest"), _nchw(skip_buf[..., 32:96])], 1)
    z = F.conv2d(xin, w) * scale[None, :, None, None] + shift[None, :, None, None]
    a = F.silu(z).permute(0, 2, 3, 1)
    _close(out0, a[..., :32].to(torch.bfloat16).float() + res.float(), 1e-2, 3e-2)
    _close(out1[..., 32:], a[..., 32:], 8e-3, 2e-2)
    assert (out1[..., :32] == 0).all()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [c for c in CASES if c[4] % 8 == 0])
def test_conv_dgrad(dev, B, H, W, Cin, Cout, k, s, p):
    from yoloseries_amd import hipk
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gy = _nhwc(B, Ho, Wo, Cout, dev, 7)
    g = torch.Generator().manual_seed(8)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5).to(torch.bfloat16).float().to(dev)
    wd = hipk.pack_weight_dgrad(w)
    gx = _nhwc(B, H, W, Cin, dev, 9)          # pre-existing gradient, accumulate on top
    gx0 = gx.clone()
    d = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, Ho, Wo, k, s, p, wd, Cin, hipk.full(gx), accumulate=1)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    x = torch.zeros(B, Cin, H, W, device=dev, requires_grad=True)
    y = F.conv2d(x, w, stride=s, padding=p)
    (ref,) = torch.autograd.grad(y, x, _nchw(gy))
    ref = ref.permute(0, 2, 3, 1).to(torch.bfloat16).float() + gx0.float()
    _close(gx, ref, 1e-2, 4e-2)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", CASES)
@pytest.mark.parametrize("splits,tile_k", [(1, 0), (5, 0), (3, 64), (4, 128)])
def test_conv_wgrad(dev, B, H, W, Cin, Cout, k, s, p, splits, tile_k):
    from yoloseries_amd import hipk
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ldg = ((Cout + 7) // 8) * 8
    gyb = torch.zeros(B, Ho, Wo, ldg, dtype=torch.bfloat16, device=dev)
    gyb[..., :Cout] = _nhwc(B, Ho, Wo, Cout, dev, 10)
    x = _nhwc(B, H, W, Cin, dev, 11)
    dw = torch.zeros(Cout, k * k * Cin, device=dev)
    d = hipk.wgrad_desc(hipk.full(gyb), Cout, hipk.full(x), 0, Cin, B, Ho, Wo, H, W, k, s, p, dw, splits)
    d.tile_k = tile_k                    # 64: 64-pixel k-steps where the layer's tiling has that variant; 128: the general 128-column
                                         # tiling on layers with 128..384 im2col columns; ignored elsewhere
    hipk.wgrad_launch(d)
    torch.cuda.synchronize()
    w = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
    y = F.conv2d(_nchw(x), w, stride=s, padding=p)
    (ref,) = torch.autograd.grad(y, w, _nchw(gyb[..., :Cout]))
    ref = ref.permute(0, 2, 3, 1).reshape(Cout, -1)
    scale = ref.abs().max().item()
    _close(dw, ref, 2e-3, 2e-3 * scale)
    # the workspace form: partial tiles by plain stores, summed in split order by a second kernel — same values within fp32
    # summation noise, and bit-identical from run to run (the atomic form is not)
    import ctypes as C
    from yoloseries_amd._lib import lib
    need = lib().yh_conv_wgrad_ws_bytes(C.byref(d))
    assert need >= Cout * k * k * Cin * 4
    ws = torch.full((need // 4 + 64,), float("nan"), device=dev)
    d.partial, d.partial_bytes = ws.data_ptr(), need
    outs = []
    for _ in range(2):
        dw2 = torch.full_like(dw, 0.25)            # the reduce adds onto what dw holds
        d.dw = dw2.data_ptr()
        hipk.wgrad_launch(d)
        torch.cuda.synchronize()
        outs.append(dw2)
    assert torch.equal(outs[0], outs[1])
    assert torch.isnan(ws[need // 4:]).all()       # nothing written past the advertised size
    _close(outs[0] - 0.25, ref, 2e-3, 2e-3 * scale)
    d.partial_bytes = need - 4
    assert lib().yh_conv_wgrad(C.byref(d), None) != 0      # a workspace that is too small is refused


WGP_CASES = [
    # B, H, W, Cin, Cout, stride, coff_k, Ctot
    (2, 40, 48, 16, 32, 1, 0, 16),       # stem class (space-to-depth image, 16 channels: two taps per 32-column tile, half a tile past tap 8)
    (1, 24, 32, 16, 64, 1, 0, 16),       # YOLOv5l stem
    (2, 48, 40, 32, 32, 1, 0, 32),       # stage-1 bottleneck class; W not a multiple of 16, H a multiple of 16
    (1, 33, 21, 32, 32, 1, 0, 32),       # odd sizes: ragged regions on both axes
    (2, 32, 64, 32, 64, 2, 0, 32),       # stage-1 conv class (stride 2, 18 accumulator tiles)
    (1, 20, 20, 64, 32, 1, 0, 64),
    (1, 16, 32, 32, 48, 1, 32, 96),      # a segment of a concat input: columns [32, 64) of every tap; ragged n-tile
    (2, 24, 40, 64, 64, 1, 0, 64),       # 64 -> 64 (36 accumulator tiles: eight waves)
]
WGP_1X1 = [(2, 24, 40, 64, 64, 0, 64), (1, 33, 17, 32, 32, 0, 32), (2, 16, 16, 64, 40, 64, 128), (1, 20, 36, 16, 64, 0, 16)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,coff,Ctot", WGP_1X1)
def test_conv_wgrad_patch_form_1x1(dev, B, H, W, Cin, Cout, coff, Ctot):
    """the patch form on 1x1 layers (the "patch" is the region itself; persistent blocks, one set of atomics per block)"""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    ldg = ((Cout + 7) // 8) * 8
    gyb = torch.zeros(B, H, W, ldg, dtype=torch.bfloat16, device=dev)
    gyb[..., :Cout] = _nhwc(B, H, W, Cout, dev, 97)
    x = _nhwc(B, H, W, Cin, dev, 98)
    ref = gyb[..., :Cout].float().reshape(-1, Cout).t() @ x.float().reshape(-1, Cin)
    dw = torch.full((Cout, Ctot), 0.25, device=dev)
    d = hipk.wgrad_desc(hipk.full(gyb), Cout, hipk.full(x), coff, Ctot, B, H, W, H, W, 1, 1, 0, dw, 1024)
    d.tile_k = 40
    assert lib().yh_conv_wgrad_patch_ok(C.byref(d)) == 1
    hipk.wgrad_launch(d)
    torch.cuda.synchronize()
    _close(dw[:, coff:coff + Cin] - 0.25, ref, 2e-3, 2e-3 * ref.abs().max().item())
    rest = torch.ones(Ctot, dtype=torch.bool, device=dev)
    rest[coff:coff + Cin] = False
    assert (dw[:, rest] == 0.25).all()



@pytest.mark.parametrize("B,H,W,Cin,Cout,s,coff,Ctot", WGP_CASES)
def test_conv_wgrad_patch_form(dev, B, H, W, Cin, Cout, s, coff, Ctot):
    """conv_wgp_kernel (tile_k 40): weight gradient of the 3x3 small-channel layers from the input patch of a pixel region staged
    once in LDS (nine taps = nine address offsets of transposing reads, persistent blocks) against torch; also with few blocks"""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    ldg = ((Cout + 7) // 8) * 8
    gyb = torch.zeros(B, Ho, Wo, ldg, dtype=torch.bfloat16, device=dev)
    gyb[..., :Cout] = _nhwc(B, Ho, Wo, Cout, dev, 95)
    x = _nhwc(B, H, W, Cin, dev, 96)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(_nchw(x), w, stride=s, padding=1), w, _nchw(gyb[..., :Cout]))
    ref = ref.permute(0, 2, 3, 1)                                  # [N][kh][kw][C]
    for blocks in (512, 3):
        dw = torch.full((Cout, 9 * Ctot), 0.5, device=dev)
        d = hipk.wgrad_desc(hipk.full(gyb), Cout, hipk.full(x), coff, Ctot, B, Ho, Wo, H, W, 3, s, 1, dw, blocks)
        d.tile_k = 40
        assert lib().yh_conv_wgrad_patch_ok(C.byref(d)) == 1
        hipk.wgrad_launch(d)
        torch.cuda.synchronize()
        got = dw.reshape(Cout, 9, Ctot)
        _close(got[:, :, coff:coff + Cin] - 0.5, ref.reshape(Cout, 9, Cin), 2e-3, 2e-3 * ref.abs().max().item())
        rest = torch.ones(Ctot, dtype=torch.bool, device=dev)
        rest[coff:coff + Cin] = False
        assert (got[:, :, rest] == 0.5).all()                       # the other segments' columns are untouched


@pytest.mark.parametrize("tile_k", [0, 64])
@pytest.mark.parametrize("Cout", [32, 48, 64, 80])
def test_conv_wgrad_fused_bn_backward(dev, Cout, tile_k):
    """weight gradient of a layer without a data gradient (the stem: 3x3 on the 16-channel space-to-depth image, 32 / 48 / 64 / 80
    output channels for YOLOv5 s / m / l / x) with the BatchNorm+SiLU backward apply fused into its operand loader
    (yh_wgrad_desc.bn_*): BIT-identical to yh_bn_silu_bwd_apply followed by the plain weight gradient (both through the
    deterministic partial-tile workspace), and close to an fp32 torch evaluation of the same chain"""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    B, H, W, Cin, k = 3, 24, 40, 16, 3
    M = B * H * W
    g = torch.Generator().manual_seed(90 + Cout)
    x = _nhwc(B, H, W, Cin, dev, 91)
    ga = _nhwc(B, H, W, Cout, dev, 92)
    z = _nhwc(B, H, W, Cout, dev, 93)
    z[0, :2] = 0.0                                   # pixels where z (and below ga) are exactly zero: gz there is the constant D, not 0
    ga[0, 0] = 0.0
    mean, invstd = torch.randn(Cout, generator=g) * 0.3, torch.rand(Cout, generator=g) + 0.5
    gamma = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    beta = torch.randn(Cout, generator=g) * 0.2
    scale = gamma.cpu() * invstd
    ws = torch.cat([scale, beta - mean * scale, mean, invstd]).to(dev)
    coef = torch.cat([torch.randn(Cout, generator=g) * 0.05, torch.randn(Cout, generator=g) * 0.05]).to(dev)
    # unfused: the apply pass writes gz, the weight gradient reads it
    gz = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    hipk.bn_silu_bwd_apply(hipk.full(ga), hipk.full(z), ws, gamma, coef, M, hipk.full(gz))

    def run(gy_t, fused):
        dw = torch.zeros(Cout, k * k * Cin, device=dev)
        d = hipk.wgrad_desc(hipk.full(gy_t), Cout, hipk.full(x), 0, Cin, B, H, W, H, W, k, 1, 1, dw, 7)
        d.tile_k = tile_k
        need = lib().yh_conv_wgrad_ws_bytes(C.byref(d))
        wsb = torch.zeros(need // 4 + 16, device=dev)
        d.partial, d.partial_bytes = wsb.data_ptr(), need
        if fused:
            d.bn_z, d.bn_ldz = z.data_ptr(), Cout
            d.bn_ws, d.bn_gamma, d.bn_coef = ws.data_ptr(), gamma.data_ptr(), coef.data_ptr()
        hipk.wgrad_launch(d)
        torch.cuda.synchronize()
        return dw
    ref_dw = run(gz, False)
    got = run(ga, True)
    assert torch.equal(got, ref_dw), (got - ref_dw).abs().max().item()
    # fp32 evaluation of the chain
    a = z.float() * ws[:Cout] + ws[Cout:2 * Cout]
    sg = torch.sigmoid(a)
    dz = ga.float() * (sg * (1 + a * (1 - sg)))
    xhat = (z.float() - ws[2 * Cout:3 * Cout]) * ws[3 * Cout:]
    gzf = gamma * ws[3 * Cout:] * (dz - coef[:Cout] - xhat * coef[Cout:])
    w = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
    (rw,) = torch.autograd.grad(F.conv2d(_nchw(x), w, padding=1), w, gzf.permute(0, 3, 1, 2))
    rw = rw.permute(0, 2, 3, 1).reshape(Cout, -1)
    _close(got, rw, 1e-2, 1e-2 * rw.abs().max().item())


@pytest.mark.parametrize("Cout", [32, 48, 64])
def test_conv_wgrad_patch_form_fused_bn_backward(dev, Cout):
    """the patch form of the stem's weight gradient with the BatchNorm+SiLU backward apply inside the staging of its gy tile
    (conv_wgpf_kernel: tile_k 40 + yh_wgrad_desc.bn_*): the staged gz is the apply pass's gz bit for bit, so the result equals the
    plain patch form on yh_bn_silu_bwd_apply's output up to the order of the fp32 atomics; ragged map (pixels past the edge must
    stay zero although gz = D there) and a channel count that is not a multiple of 32"""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    B, H, W, Cin, k = 3, 21, 40, 16, 3
    M = B * H * W
    g = torch.Generator().manual_seed(190 + Cout)
    x = _nhwc(B, H, W, Cin, dev, 191)
    ga = _nhwc(B, H, W, Cout, dev, 192)
    z = _nhwc(B, H, W, Cout, dev, 193)
    z[0, :2] = 0.0
    ga[0, 0] = 0.0
    mean, invstd = torch.randn(Cout, generator=g) * 0.3, torch.rand(Cout, generator=g) + 0.5
    gamma = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    beta = torch.randn(Cout, generator=g) * 0.2
    scale = gamma.cpu() * invstd
    ws = torch.cat([scale, beta - mean * scale, mean, invstd]).to(dev)
    coef = torch.cat([torch.randn(Cout, generator=g) * 0.05, torch.randn(Cout, generator=g) * 0.05]).to(dev)
    gz = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    hipk.bn_silu_bwd_apply(hipk.full(ga), hipk.full(z), ws, gamma, coef, M, hipk.full(gz))

    def run(gy_t, fused):
        dw = torch.zeros(Cout, k * k * Cin, device=dev)
        d = hipk.wgrad_desc(hipk.full(gy_t), Cout, hipk.full(x), 0, Cin, B, H, W, H, W, k, 1, 1, dw, 0)
        d.tile_k = 40
        if fused:
            d.bn_z, d.bn_ldz = z.data_ptr(), Cout
            d.bn_ws, d.bn_gamma, d.bn_coef = ws.data_ptr(), gamma.data_ptr(), coef.data_ptr()
        assert lib().yh_conv_wgrad_patch_ok(C.byref(d)) == 1
        buf = C.create_string_buffer(96)
        lib().yh_conv_wgrad_patch_name(C.byref(d), buf, 96)
        assert buf.value.decode().startswith("conv_wgpf_kernel<5" if fused else "conv_wgp_kernel<5"), buf.value
        hipk.wgrad_launch(d)
        torch.cuda.synchronize()
        return dw
    ref_dw = run(gz, False)
    got = run(ga, True)
    assert (got - ref_dw).abs().max().item() <= 2e-5 * ref_dw.abs().max().item() + 1e-6
    w = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
    (rw,) = torch.autograd.grad(F.conv2d(_nchw(x), w, padding=1), w, _nchw(gz))
    rw = rw.permute(0, 2, 3, 1).reshape(Cout, -1)
    _close(got, rw, 1e-2, 1e-2 * rw.abs().max().item())


@pytest.mark.parametrize("C0,C1,Cout", [(32, 64, 64), (512, 448, 128)])
def test_conv_wgrad_segment_upsampled(dev, C0, C1, Cout):
    """wgrad of one segment of a concat input, read through the 2x upsample (wide tiling; general tiling)."""
    from yoloseries_amd import hipk
    B, H, W = 2, 16, 16
    lo = _nhwc(B, H // 2, W // 2, C0, dev, 12)
    sk = _nhwc(B, H, W, C1, dev, 13)
    gy = _nhwc(B, H, W, Cout, dev, 14)
    dw = torch.zeros(Cout, C0 + C1, device=dev)
    hipk.wgrad_launch(hipk.wgrad_desc(hipk.full(gy), Cout, hipk.Slice(lo, 0, C0, ups=1), 0, C0 + C1, B, H, W, H, W, 1, 1, 0, dw, 3))
    hipk.wgrad_launch(hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(sk), C0, C0 + C1, B, H, W, H, W, 1, 1, 0, dw, 2))
    torch.cuda.synchronize()
    xin = torch.cat([F.interpolate(_nchw(lo), scale_factor=2, mode="nearest"), _nchw(sk)], 1)
    w = torch.zeros(Cout, C0 + C1, 1, 1, device=dev, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xin, w), w, _nchw(gy))
    ref = ref.reshape(Cout, -1)
    _close(dw, ref, 2e-3, 2e-3 * ref.abs().max().item())


WGS_CASES = [
    # B, H, W, Cin (segment), Cout, k, s, coff_k, Ctot, ups
    (2, 20, 20, 128, 128, 3, 1, 0, 128, 0),      # YOLOv5s stage-3 bottleneck class: 9 tiles, 20-wide map (a 16-pixel stage wraps rows)
    (2, 16, 16, 128, 256, 3, 2, 0, 128, 0),      # stride 2, two n-tiles
    (1, 16, 24, 256, 192, 3, 1, 0, 256, 0),      # two column tiles per tap, ragged n-tile (192 = 128 + 64)
    (2, 20, 20, 256, 255, 1, 1, 0, 256, 0),      # pointwise (scalar walk), N not a multiple of 8 (Detect)
    (2, 8, 8, 128, 128, 3, 1, 128, 384, 0),      # a segment of a concat input: columns [128, 256) of every tap; 8 x 8 map (a stage spans two rows)
    (2, 16, 16, 128, 64, 1, 1, 0, 256, 1),       # segment read through the 2x upsample (not the pointwise walk), half-empty n-tile
    (1, 4, 8, 128, 128, 3, 1, 0, 128, 0),        # ONE work unit: three of the four waves have nothing to do
    (2, 24, 24, 64, 64, 3, 1, 0, 64, 0),         # 64 channels: a column tile is TWO taps (per-lane taps), the fifth tile is half empty (tap 9)
    (1, 32, 32, 64, 128, 3, 2, 0, 64, 0),        # YOLOv5s stage-2 conv class (stride 2, 64 -> 128)
    (2, 16, 16, 96, 160, 3, 1, 0, 96, 0),        # YOLOv5m width: tiles straddle taps at 32-column groups (864 columns: 6.75 tiles)
    (2, 16, 16, 32, 64, 3, 2, 0, 32, 0),         # 32 channels: four taps per tile
    (2, 16, 16, 160, 80, 1, 1, 32, 320, 0),      # pointwise on a 160-channel segment of a 320-channel concat (YOLOv5x): second tile a quarter full
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,coff,Ctot,ups", WGS_CASES)
@pytest.mark.parametrize("groups", [1, 7, -1, 256])
def test_conv_wgrad_wave_private_tiles(dev, B, H, W, Cin, Cout, k, s, coff, Ctot, ups, groups):
    """conv_wgs_kernel (tile_k 129): wave-private 128 x 128 tiles fed by per-wave LDS-DMA rings, stream-K over `splits`
    workgroups (1: one workgroup walks every tile; 7: workgroups end one tile and begin the next; -1: the exact tiles x 3 grid
    with the XCD-aware block map; 256), partial tiles combined through LDS and added to what dw holds.  The operands are slices
    of wider NaN-filled buffers: bytes outside the slices never reach an MFMA."""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ldg = ((Cout + 7) // 8) * 8
    gyw = torch.full((B, Ho, Wo, ldg + 16), float("nan"), dtype=torch.bfloat16, device=dev)
    gyw[..., :ldg] = 0
    gyw[..., :Cout] = _nhwc(B, Ho, Wo, Cout, dev, 20)
    Hs, Ws = H >> ups, W >> ups
    xw = torch.full((B, Hs, Ws, Cin + 32), float("nan"), dtype=torch.bfloat16, device=dev)
    xw[..., 16:16 + Cin] = _nhwc(B, Hs, Ws, Cin, dev, 21)
    dw = torch.full((Cout, k * k * Ctot), 0.5, device=dev)
    d = hipk.wgrad_desc(hipk.Slice(gyw, 0, ldg), Cout, hipk.Slice(xw, 16, Cin, ups=ups), coff, Ctot, B, Ho, Wo, H, W, k, s, p, dw, 1)
    d.tile_k = 129
    T = lib().yh_conv_wgrad_wave_tiles(C.byref(d))
    assert T == ((Cout + 127) // 128) * ((k * k * Cin + 127) // 128)
    d.splits = 3 * T if groups < 0 else groups
    assert lib().yh_conv_wgrad_wave_name(C.byref(d)).decode().startswith("conv_wgs_kernel<")
    hipk.wgrad_launch(d)
    torch.cuda.synchronize()
    xin = _nchw(xw[..., 16:16 + Cin])
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    w = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xin, w, stride=s, padding=p), w, _nchw(gyw[..., :Cout]))
    ref = ref.permute(0, 2, 3, 1)                                  # [Cout][kh][kw][Cin]
    got = dw.reshape(Cout, k, k, Ctot)
    assert not torch.isnan(dw).any()
    _close(got[..., coff:coff + Cin] - 0.5, ref, 2e-3, 2e-3 * ref.abs().max().item())
    mask = torch.ones(Ctot, dtype=torch.bool, device=dev)
    mask[coff:coff + Cin] = False
    assert (got[..., mask] == 0.5).all()                           # the other segments' columns are untouched


def test_conv_wgrad_wave_private_tiles_eligibility(dev):
    """layers the form does not cover fall through to the im2col forms (yh_conv_wgrad_wave_tiles == 0, tile_k 129 ignored)"""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    for (B, H, W, Cin, Cout, k) in [(2, 16, 16, 48, 128, 3), (1, 13, 13, 128, 128, 3), (2, 16, 16, 128, 32, 1)]:
        p = k // 2
        gy = _nhwc(B, H, W, Cout, dev, 22)
        x = _nhwc(B, H, W, Cin, dev, 23)
        dw = torch.zeros(Cout, k * k * Cin, device=dev)
        d = hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, H, W, H, W, k, 1, p, dw, 4)
        d.tile_k = 129
        assert lib().yh_conv_wgrad_wave_tiles(C.byref(d)) == 0
        hipk.wgrad_launch(d)
        torch.cuda.synchronize()
        w = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(_nchw(x), w, padding=p), w, _nchw(gy))
        ref = ref.permute(0, 2, 3, 1).reshape(Cout, -1)
        _close(dw, ref, 2e-3, 2e-3 * ref.abs().max().item())


def _wgs_case(dev, case, seed):
    """operands of one WGS_CASES layer (slices of wider NaN-filled buffers), its descriptor and the fp32 reference"""
    from yoloseries_amd import hipk
    B, H, W, Cin, Cout, k, s, coff, Ctot, ups = case
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ldg = ((Cout + 7) // 8) * 8
    gyw = torch.full((B, Ho, Wo, ldg + 16), float("nan"), dtype=torch.bfloat16, device=dev)
    gyw[..., :ldg] = 0
    gyw[..., :Cout] = _nhwc(B, Ho, Wo, Cout, dev, seed)
    Hs, Ws = H >> ups, W >> ups
    xw = torch.full((B, Hs, Ws, Cin + 32), float("nan"), dtype=torch.bfloat16, device=dev)
    xw[..., 16:16 + Cin] = _nhwc(B, Hs, Ws, Cin, dev, seed + 1)
    dw = torch.full((Cout, k * k * Ctot), 0.5, device=dev)
    d = hipk.wgrad_desc(hipk.Slice(gyw, 0, ldg), Cout, hipk.Slice(xw, 16, Cin, ups=ups), coff, Ctot, B, Ho, Wo, H, W, k, s, p, dw, 1)
    d.tile_k = 129
    xin = _nchw(xw[..., 16:16 + Cin])
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    w = torch.zeros(Cout, Cin, k, k, device=dev, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xin, w, stride=s, padding=p), w, _nchw(gyw[..., :Cout]))
    return d, dw, ref.permute(0, 2, 3, 1), (gyw, xw)


# ---- kernel families: the register-staged kernel (algo 1) and the LDS-DMA ring kernel with its three tiles (algo 2..4)
V3_CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 40, 40, 128, 128, 3, 1, 1),
    (3, 20, 20, 256, 256, 3, 1, 1),      # halo kernel: two n-tiles, 20-wide map (ragged 2-D tiles), 4 channel blocks
    (1, 80, 48, 64, 128, 3, 1, 1),
    (1, 24, 24, 64, 256, 3, 2, 1),
    (2, 16, 16, 256, 128, 1, 1, 0),
    (2, 20, 20, 96, 160, 3, 1, 1),       # 32-channel k-steps (96 % 64 != 0), ragged N tile
    (1, 33, 17, 64, 64, 3, 1, 1),        # odd sizes, rows past M in the last tile
    (3, 8, 8, 512, 256, 1, 1, 0),
    (2, 24, 24, 80, 80, 3, 1, 1),        # halo kernel with a 16-channel tail block (YOLOv5x widths)
    (1, 32, 32, 112, 176, 3, 1, 1),      # 48-channel tail (forward and data gradient)
]


def _kname(d):
    import ctypes as C
    from yoloseries_amd._lib import lib
    buf = C.create_string_buffer(96)
    assert lib().yh_conv_kernel_name(C.byref(d), buf, 96) == 0
    return buf.value.decode()


def _expect_family(d, algo):
    """algo 2..4 must land on the LDS-DMA ring kernel; algo 5 on the halo kernel where the shape is eligible (else skip)"""
    kn = _kname(d)
    if algo in (2, 3, 4):
        assert "conv_v3_kernel" in kn, kn
    if algo == 14:
        assert "conv_v3_kernel<256, 256" in kn, kn
    if algo == 5 and "conv_halo_kernel" not in kn:
        pytest.skip("shape not eligible for the halo kernel")
    if algo == 7:
        assert "conv_dg2_kernel" in kn, kn
    if algo == 8:
        assert "conv_p3_kernel" in kn, kn
    if algo == 9:
        assert "conv_h80_kernel" in kn, kn
    if algo == 12:
        assert "conv_c80_kernel" in kn, kn
    if algo == 13:
        assert "conv_pt_kernel" in kn, kn


@pytest.mark.parametrize("algo", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", V3_CASES)
def test_conv_fwd_algos(dev, B, H, W, Cin, Cout, k, s, p, algo):
    """plain store + BatchNorm partial sums (EPI 1) and the generic epilogue (bias, EPI 2) on every kernel family"""
    from yoloseries_amd import hipk
    x = _nhwc(B, H, W, Cin, dev, 21)
    g = torch.Generator().manual_seed(22)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(torch.bfloat16).float().to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(x), w, None, stride=s, padding=p).permute(0, 2, 3, 1)
    xin = hipk.full(x)
    if Cin % 64:
        # ragged channel count: the input is a channel slice of a wider buffer whose other channels are NaN — the kernels'
        # 128-byte row fetches run over them (and over the next tap's weights) but must never feed them to an MFMA
        xbuf = torch.full((B, H, W, Cin + 16), float("nan"), dtype=torch.bfloat16, device=dev)
        xbuf[..., 8:8 + Cin] = x
        xin = hipk.Slice(xbuf, 8, Cin)
    for with_bias in (False, True):
        out = torch.full((B, Ho, Wo, Cout), 7.0, dtype=torch.bfloat16, device=dev)
        d = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, k, s, p, wp, Cout, hipk.full(out),
                           bias=bias if with_bias else None)
        d.algo = algo
        _expect_family(d, algo)
        stats = None
        if not with_bias:
            stats = torch.zeros(hipk.conv_stat_blocks(d), 2, wp.shape[0], device=dev)
            d.stats = stats.data_ptr()
        hipk.conv_launch(d)
        torch.cuda.synchronize()
        _close(out, ref + (bias if with_bias else 0.0), 8e-3, 2e-2)
        if stats is not None:
            # the sums are taken over the fp32 accumulators, the comparison over the stored bf16 values: allow 4 sigma of
            # the accumulated rounding noise (per element <= 2^-9 |o|)
            o = out.float().reshape(-1, Cout).double()
            tol1 = 1e-2 + 4 * 2.0 ** -9 * (o ** 2).sum(0).sqrt()
            tol2 = 1e-2 + 4 * 2.0 ** -8 * (o ** 4).sum(0).sqrt()
            assert ((stats[:, 0, :Cout].double().sum(0) - o.sum(0)).abs() <= tol1).all()
            assert ((stats[:, 1, :Cout].double().sum(0) - (o ** 2).sum(0)).abs() <= tol2).all()


V3W_CASES = [
    # B, H, W, Cin, Cout, k, s, p — the 256 x 256 tile of the LDS-DMA ring kernel (algo 14): Cin % 64 == 0 and, on the side that
    # becomes N (Cout forward, Cin data gradient), a multiple of 256
    (2, 24, 24, 256, 256, 3, 1, 1),
    (1, 40, 24, 256, 512, 3, 2, 1),      # stride 2 (the data gradient runs as four parity classes), two n-tiles
    (3, 20, 20, 512, 256, 1, 1, 0),      # pointwise, rows past M in the last tile
    (1, 17, 33, 256, 256, 3, 1, 1),      # odd sizes
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", V3W_CASES)
def test_conv_v3_wide_tile(dev, B, H, W, Cin, Cout, k, s, p):
    """conv_v3_kernel<256, 256, ...> (algo 14): forward with statistics / generic epilogue and the data gradient (plain, accumulating,
    fused reduction) against torch, with the bars of the other tiles"""
    test_conv_fwd_algos(dev, B, H, W, Cin, Cout, k, s, p, 14)
    _dgrad_check(dev, B, H, W, Cin, Cout, k, s, p, 14)


@pytest.mark.parametrize("algo", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [c for c in V3_CASES if c[3] > 32])
def test_conv_dgrad_algos(dev, B, H, W, Cin, Cout, k, s, p, algo):
    """data gradient (stride-2 layers run as four parity classes), plain and accumulating, plus the fused
    BatchNorm+SiLU backward reduction of the producer layer (EPI 3) against a torch reference"""
    _dgrad_check(dev, B, H, W, Cin, Cout, k, s, p, algo)


DG2_CASES = [
    # B, H, W, Cin, Cout: data gradient of ConvBnAct(Cin, Cout, 3, 2, 1) on an H x W input (gz is H/2 x W/2 x Cout)
    (2, 64, 64, 32, 64),        # YOLOv5s stage-1 shape class: one 32-channel tile, one 64-channel block, weights resident in LDS
    (1, 48, 80, 64, 128),       # two channel tiles (8 waves), two 64-channel blocks
    (2, 40, 24, 128, 256),      # two blocks along the output channels, ragged regions (20 x 12 map)
    (1, 32, 32, 48, 96),        # YOLOv5m widths: 48 of 64 channels used, 32-channel steps (96 % 64 != 0)
    (3, 16, 16, 256, 512),      # 8 x 8 map: a region covers half an image
    (1, 24, 40, 40, 64),        # channel count that is not a multiple of 32 (masked chunk columns)
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", DG2_CASES)
def test_conv_dgrad_stride2_kernel(dev, B, H, W, Cin, Cout):
    """conv_dg2_kernel (algo 7): the four parity classes of a 3x3 / stride-2 data gradient from ONE LDS patch of gz — plain,
    accumulating and with the fused BatchNorm-backward reduction, against torch; also with 32-channel steps where 64 is the default"""
    _dgrad_check(dev, B, H, W, Cin, Cout, 3, 2, 1, 7)
    if Cout % 64 == 0:
        _dgrad_check(dev, B, H, W, Cin, Cout, 3, 2, 1, 7, tile_k=32)


P3_CASES = [
    # B, H, W, Cin, Cout: ConvBnAct(Cin, Cout, 3, 1, 1) on an H x W map
    (2, 48, 40, 32, 32),        # YOLOv5s stage-1 bottleneck class: one channel tile, weights resident in LDS, two pixel tiles per wave
    (1, 33, 17, 64, 64),        # odd sizes: ragged regions, 64-channel step, two channel tiles
    (2, 24, 24, 32, 64),
    (1, 40, 40, 64, 32),
    (2, 20, 20, 96, 96),        # 32-channel steps (96 % 64 != 0), three channel blocks, two blocks along the output channels
    (1, 16, 16, 128, 128),
    (1, 24, 40, 40 + 24, 40),   # output channel count that is not a multiple of 32 (masked chunk columns)
]


@pytest.mark.parametrize("tile_n", [0, 32])
@pytest.mark.parametrize("B,H,W,Cin,Cout", P3_CASES)
def test_conv_patch3_kernel(dev, B, H, W, Cin, Cout, tile_n):
    """conv_p3_kernel (algo 8): 3x3 / stride-1 layers with few channels from ONE LDS patch of the input — forward with the
    BatchNorm partial sums, data gradient plain / accumulating / with the fused BatchNorm-backward reduction — against torch;
    tile_n 32 = one pixel tile per wave (128-pixel regions) instead of two"""
    from yoloseries_amd import hipk
    x = _nhwc(B, H, W, Cin, dev, 81)
    g = torch.Generator().manual_seed(82)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    out = torch.full((B, H, W, Cout), 3.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(out))
    d.algo, d.tile_n = 8, tile_n
    assert "conv_p3_kernel" in _kname(d), _kname(d)
    stats = torch.full((hipk.conv_stat_blocks(d), 2, wp.shape[0]), float("nan"), device=dev)
    d.stats = stats.data_ptr()
    assert "conv_p3_kernel" in _kname(d) and hipk.conv_stat_blocks(d) == stats.shape[0]
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    ref = F.conv2d(_nchw(x), w, padding=1).permute(0, 2, 3, 1)
    _close(out, ref, 8e-3, 2e-2)
    o = out.float().reshape(-1, Cout).double()
    assert not torch.isnan(stats[:, :, :Cout]).any()
    assert ((stats[:, 0, :Cout].double().sum(0) - o.sum(0)).abs() <= 1e-3 + 1e-5 * o.abs().sum(0)).all()      # sums of the STORED values
    assert ((stats[:, 1, :Cout].double().sum(0) - (o ** 2).sum(0)).abs() <= 1e-3 + 1e-5 * (o ** 2).sum(0)).all()
    if Cout % 32 == 0:                 # the data gradient reads gz with Cout channels: whole 32-channel blocks needed
        _dgrad_check(dev, B, H, W, Cin, Cout, 3, 1, 1, 8, tile_n=tile_n)


P3S2_CASES = [
    # B, H, W (input), Cin, Cout: ConvBnAct(Cin, Cout, 3, 2, 1) — the downsampling layers (models/normal/yolov5s.py:18-40)
    (2, 64, 64, 32, 64),        # YOLOv5s backbone_stage1_conv class: weights resident, one 32-channel block, two channel tiles
    (1, 48, 80, 64, 128),       # two 32-channel blocks (weights reloaded per block), two blocks along the output channels
    (2, 33, 17, 32, 32),        # odd input sizes: output 17 x 9, ragged tiles, the patch reaches past the right / lower edge
    (1, 40, 24, 128, 128),      # four channel blocks
    (3, 16, 16, 96, 40),        # 96 channels (three blocks), output channels not a multiple of 32
    (1, 130, 70, 32, 64),       # several tiles per image row and column
]


@pytest.mark.parametrize("tile_n", [0, 32])
@pytest.mark.parametrize("B,H,W,Cin,Cout", P3S2_CASES)
def test_conv_patch3_stride2_forward(dev, B, H, W, Cin, Cout, tile_n):
    """conv_p3_kernel<..., 2> (algo 8 on a 3x3 / stride-2 / pad-1 forward layer): the (2 TH + 1) x (2 TW + 1) patch staged once, its
    columns de-interleaved by parity — output and BatchNorm partial sums against torch; the data gradient of such a layer is not
    this kernel's (conv_dg2_kernel) and the descriptor falls back to the library default"""
    from yoloseries_amd import hipk
    x = _nhwc(B, H, W, Cin, dev, 83)
    g = torch.Generator().manual_seed(84)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ref = F.conv2d(_nchw(x), w, stride=2, padding=1).permute(0, 2, 3, 1)
    for with_stats in (True, False):
        out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, 3, 2, 1, wp, Cout, hipk.full(out))
        d.algo, d.tile_n = 8, tile_n
        assert "conv_p3_kernel" in _kname(d) and _kname(d).endswith(", 2>"), _kname(d)
        stats = None
        if with_stats:
            stats = torch.full((hipk.conv_stat_blocks(d), 2, wp.shape[0]), float("nan"), device=dev)
            d.stats = stats.data_ptr()
            assert hipk.conv_stat_blocks(d) == stats.shape[0]
        hipk.conv_launch(d)
        torch.cuda.synchronize()
        assert not torch.isnan(out.float()).any()
        _close(out, ref, 8e-3, 2e-2)
        if stats is not None:
            o = out.float().reshape(-1, Cout).double()
            assert not torch.isnan(stats[:, :, :Cout]).any()
            assert ((stats[:, 0, :Cout].double().sum(0) - o.sum(0)).abs() <= 1e-3 + 1e-5 * o.abs().sum(0)).all()
            assert ((stats[:, 1, :Cout].double().sum(0) - (o ** 2).sum(0)).abs() <= 1e-3 + 1e-5 * (o ** 2).sum(0)).all()
    # data gradient of the same layer with algo 8: not eligible -> the default family, still correct
    gy = _nhwc(B, Ho, Wo, Cout, dev, 85)
    gx = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
    if Cout % 8 == 0 and H % 2 == 0 and W % 2 == 0:
        dd = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, Ho, Wo, 3, 2, 1, hipk.pack_weight_dgrad(w), Cin, hipk.full(gx))
        dd.algo = 8
        assert "conv_p3_kernel" not in _kname(dd)


H80_CASES = [
    # B, H, W, Cout: ConvBnAct(80, Cout, 3, 1, 1) on an H x W map (YOLOv5x stage-1 bottlenecks: 80 -> 80 at 320 x 320)
    (2, 32, 32, 80),            # whole 16 x 16 tiles
    (1, 40, 24, 80),            # tiles cut by the right / lower edge
    (3, 19, 23, 80),            # odd sizes: ragged tiles, 16-pixel groups that wrap tile rows; several images per block
    (1, 48, 48, 160),           # two blocks along the output channels
    (2, 7, 5, 80),              # map smaller than a tile
]


@pytest.mark.parametrize("B,H,W,Cout", H80_CASES)
def test_conv_halo80_kernel(dev, B, H, W, Cout):
    """conv_h80_kernel (algo 9): 3x3 / stride-1 layers with 80 input channels on 256-pixel x 80-channel tiles, reduction over the
    flattened (tap, channel) index — plain store, folded BatchNorm + SiLU + residual into a channel slice (the C3 concat buffer) with
    a split destination, accumulate, and the data gradient (flipped taps) — against torch"""
    from yoloseries_amd import hipk
    Cin = 80
    x = _nhwc(B, H, W, Cin, dev, 91)
    g = torch.Generator().manual_seed(92)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(x), w, None, stride=1, padding=1).permute(0, 2, 3, 1)
    # the input as a channel slice of a wider buffer (ld > C); NaN around it must never be read into a result
    xbuf = torch.full((B, H, W, Cin + 16), float("nan"), dtype=torch.bfloat16, device=dev)
    xbuf[..., 8:8 + Cin] = x
    xin = hipk.Slice(xbuf, 8, Cin)
    out = torch.full((B, H, W, Cout), 7.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(out))
    d.algo = 9
    assert "conv_h80_kernel<80, 5, 0>" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, ref, 8e-3, 2e-2)
    # folded BatchNorm + SiLU + residual on the first 80 channels written into a slice of a wider buffer, split destination
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    res = _nhwc(B, H, W, 80, dev, 93)
    cat = torch.full((B, H, W, 160), 5.0, dtype=torch.bfloat16, device=dev)
    a = F.silu(ref * scale + shift)
    if Cout > 80:
        obuf = torch.full((B, H, W, Cout - 80 + 16), 3.0, dtype=torch.bfloat16, device=dev)
        d2 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.Slice(cat, 0, 80), nsplit=80,
                            out1=hipk.Slice(obuf, 8, Cout - 80), scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    else:
        d2 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.Slice(cat, 0, 80),
                            scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    d2.algo = 9
    assert "conv_h80_kernel<80, 5, 2>" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(cat[..., :80], a[..., :80].to(torch.bfloat16).float() + res.float(), 1e-2, 3e-2)
    assert (cat[..., 80:] == 5.0).all()
    if Cout > 80:
        _close(obuf[..., 8:8 + Cout - 80], a[..., 80:], 8e-3, 2e-2)
        assert (obuf[..., :8] == 3.0).all() and (obuf[..., 8 + Cout - 80:] == 3.0).all()
    # accumulate on top of an existing tensor
    acc0 = _nhwc(B, H, W, Cout, dev, 94)
    acc = acc0.clone()
    d3 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(acc), accumulate=1)
    d3.algo = 9
    hipk.conv_launch(d3)
    torch.cuda.synchronize()
    _close(acc, ref.to(torch.bfloat16).float() + acc0.float(), 1e-2, 3e-2)
    # data gradient of an 80 -> Cout' layer: gz has 80 channels, the flipped weight image
    if Cout == 80:
        gy = _nhwc(B, H, W, 80, dev, 95)
        w2 = (torch.randn(80, 80, 3, 3, generator=g) / (80 * 9) ** 0.5).to(torch.bfloat16).float().to(dev)
        wd = hipk.pack_weight_dgrad(w2)
        xz = torch.zeros(B, 80, H, W, device=dev, requires_grad=True)
        (gref,) = torch.autograd.grad(F.conv2d(xz, w2, stride=1, padding=1), xz, _nchw(gy))
        gx = torch.zeros(B, H, W, 80, dtype=torch.bfloat16, device=dev)
        d4 = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, H, W, 3, 1, 1, wd, 80, hipk.full(gx))
        d4.algo = 9
        assert "conv_h80_kernel" in _kname(d4), _kname(d4)
        hipk.conv_launch(d4)
        torch.cuda.synchronize()
        _close(gx, gref.permute(0, 2, 3, 1), 1e-2, 4e-2)


C80_CASES = [
    # B, H, W, stride: ConvBnAct(80, 160, 3, stride, 1) on an H x W map (YOLOv5x stage-1 downsampling: 80 -> 160, stride 2, 640 x 640)
    (2, 32, 32, 2),             # 16 x 16 outputs: one whole 256-pixel tile per image
    (1, 40, 24, 2),             # 240 outputs: a ragged tile
    (3, 19, 23, 2),             # odd input sizes (the last tap row / column inside the image, no bottom / right padding)
    (1, 64, 48, 2),             # three tiles per image, tiles that start in the middle of an output row
    (2, 16, 16, 1),             # stride 1
    (1, 20, 28, 1),             # stride 1, three tiles, the last ragged
]


@pytest.mark.parametrize("B,H,W,stride", C80_CASES)
def test_conv_c80_tap_kernel(dev, B, H, W, stride):
    """conv_c80_kernel (algo 12): 3x3 layers with 80 input and 160 output channels, one tap per stage, no padding of either GEMM
    side — plain store, and folded BatchNorm + bias + SiLU into a channel slice of a wider buffer — against torch"""
    from yoloseries_amd import hipk
    Cin, Cout = 80, 160
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = _nhwc(B, H, W, Cin, dev, 191)
    g = torch.Generator().manual_seed(192)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(x), w, None, stride=stride, padding=1).permute(0, 2, 3, 1)
    # the input as a channel slice of a wider buffer (ld > C); NaN around it must never be read into a result
    xbuf = torch.full((B, H, W, Cin + 16), float("nan"), dtype=torch.bfloat16, device=dev)
    xbuf[..., 8:8 + Cin] = x
    xin = hipk.Slice(xbuf, 8, Cin)
    out = torch.full((B, Ho, Wo, Cout), 7.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, 3, stride, 1, wp, Cout, hipk.full(out))
    d.algo = 12
    assert "conv_c80_kernel" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, ref, 8e-3, 2e-2)
    # a grid smaller than the tile count: persistent workgroups walk several tiles (stage buffers alternate across tiles)
    out.fill_(7.0)
    d.grid_cap = 1
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, ref, 8e-3, 2e-2)
    # folded BatchNorm + bias + SiLU, written into a slice of a wider buffer
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    cat = torch.full((B, Ho, Wo, Cout + 32), 5.0, dtype=torch.bfloat16, device=dev)
    d2 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, 3, stride, 1, wp, Cout, hipk.Slice(cat, 16, Cout),
                        bias=bias, scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
    d2.algo = 12
    assert "conv_c80_kernel" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(cat[..., 16:16 + Cout], F.silu((ref + bias) * scale + shift), 1e-2, 3e-2)
    assert (cat[..., :16] == 5.0).all() and (cat[..., 16 + Cout:] == 5.0).all()
    # not eligible: a residual, statistics, other channel counts -> the library default runs instead
    d3 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, Ho, Wo, H, W, 3, stride, 1, wp, Cout, hipk.full(out), accumulate=1)
    d3.algo = 12
    assert "conv_c80_kernel" not in _kname(d3)


PW_CASES = [
    # B, H, W, Cin, Cout: ConvBnAct(Cin, Cout, 1, 1) on an H x W map (YOLOv5x: C3's cba1 | cba2, cba3 and the bottlenecks' conv_bn_act_1)
    (2, 32, 32, 80, 80),        # whole 128-pixel tiles
    (1, 17, 23, 80, 160),       # ragged last tile, two output-channel groups
    (2, 24, 24, 160, 160),      # 64-pixel tiles, five 32-wide reduction steps
    (1, 9, 7, 160, 80),         # fewer pixels than a tile
    (3, 16, 16, 160, 320),
    (2, 20, 20, 320, 320),      # ten reduction steps, one block per CU
    (1, 11, 13, 320, 160),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", PW_CASES)
def test_conv_pointwise_kernel(dev, B, H, W, Cin, Cout):
    """conv_pw_kernel (algo 10): 1x1 layers with 80 / 160 input channels, pixel tile and weight tile whole in LDS — plain store,
    folded BatchNorm + SiLU into channel slices (split destination, the input itself a slice of a wider buffer), residual,
    accumulate — against torch"""
    from yoloseries_amd import hipk
    x = _nhwc(B, H, W, Cin, dev, 101)
    g = torch.Generator().manual_seed(102)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(x), w).permute(0, 2, 3, 1)
    xbuf = torch.full((B, H, W, 2 * Cin), float("nan"), dtype=torch.bfloat16, device=dev)       # lower half of a concat buffer
    xbuf[..., :Cin] = x
    xin = hipk.Slice(xbuf, 0, Cin)
    out = torch.full((B, H, W, Cout), 7.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, Cout, hipk.full(out))
    d.algo = 10
    assert "conv_pw_kernel<%d, 5, %d, 0>" % (Cin, 2 if Cin == 80 else 1) in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, ref, 8e-3, 2e-2)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    a = F.silu(ref * scale + shift)
    res = _nhwc(B, H, W, 80, dev, 103)
    if Cout > 80:               # cba1 | cba2: the two halves into two slices; a residual on the first
        cat = torch.full((B, H, W, Cout + 16), 5.0, dtype=torch.bfloat16, device=dev)
        d2 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, Cout, hipk.Slice(cat, 0, 80), nsplit=80,
                            out1=hipk.Slice(cat, 88, Cout - 80), scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    else:
        cat = torch.full((B, H, W, 96), 5.0, dtype=torch.bfloat16, device=dev)
        d2 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, Cout, hipk.Slice(cat, 0, 80),
                            scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    d2.algo = 10
    assert "conv_pw_kernel" in _kname(d2) and ", 2>" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(cat[..., :80], a[..., :80].to(torch.bfloat16).float() + res.float(), 1e-2, 3e-2)
    assert (cat[..., 80:88] == 5.0).all()
    if Cout > 80:
        _close(cat[..., 88:88 + Cout - 80], a[..., 80:], 8e-3, 2e-2)
        assert (cat[..., 88 + Cout - 80:] == 5.0).all()
    acc0 = _nhwc(B, H, W, Cout, dev, 104)
    acc = acc0.clone()
    d3 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, Cout, hipk.full(acc), accumulate=1)
    d3.algo = 10
    hipk.conv_launch(d3)
    torch.cuda.synchronize()
    _close(acc, ref.to(torch.bfloat16).float() + acc0.float(), 1e-2, 3e-2)


def _dgrad_check(dev, B, H, W, Cin, Cout, k, s, p, algo, tile_k=0, tile_n=0):
    from yoloseries_amd import hipk
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gy = _nhwc(B, Ho, Wo, Cout, dev, 27)
    g = torch.Generator().manual_seed(28)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5).to(torch.bfloat16).float().to(dev)
    wd = hipk.pack_weight_dgrad(w)
    x = torch.zeros(B, Cin, H, W, device=dev, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, stride=s, padding=p), x, _nchw(gy))
    ref = ref.permute(0, 2, 3, 1)
    # accumulate on top of an existing gradient (generic epilogue)
    gx = _nhwc(B, H, W, Cin, dev, 29)
    gx0 = gx.clone()
    d = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, Ho, Wo, k, s, p, wd, Cin, hipk.full(gx), accumulate=1)
    d.algo, d.tile_k, d.tile_n = algo, tile_k, tile_n
    _expect_family(d, algo)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(gx, ref.to(torch.bfloat16).float() + gx0.float(), 1e-2, 4e-2)
    # plain store + fused reduction: dz = g * silu'(z*scale + shift); partial sums of dz and dz*z per channel
    import ctypes as C
    from yoloseries_amd._lib import lib
    z = _nhwc(B, H, W, Cin, dev, 30)
    ws = torch.cat([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)]).to(dev)
    gx2 = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
    d2 = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, Ho, Wo, k, s, p, wd, Cin, hipk.full(gx2))
    d2.algo, d2.tile_k, d2.tile_n = algo, tile_k, tile_n
    rows = lib().yh_conv_bnr_rows(C.byref(d2))
    assert rows > 0
    slab = torch.zeros(rows, 2, Cin, device=dev)
    d2.bnr_z, d2.bnr_ldz, d2.bnr_C, d2.bnr_ws, d2.bnr_part = z.data_ptr(), Cin, Cin, ws.data_ptr(), slab.data_ptr()
    assert lib().yh_conv_bnr_rows(C.byref(d2)) == rows
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(gx2, ref, 1e-2, 4e-2)
    gq = gx2.float().reshape(-1, Cin).double()
    zz = z.float().reshape(-1, Cin).double()
    a = zz * ws[:Cin].double() + ws[Cin:].double()
    sg = torch.sigmoid(a)
    dz = gq * (sg * (1 + a * (1 - sg)))
    got = slab.double().sum(0)
    assert torch.allclose(got[0], dz.sum(0), rtol=2e-3, atol=2e-3 * dz.abs().sum(0).max().item())
    assert torch.allclose(got[1], (dz * zz).sum(0), rtol=2e-3, atol=2e-3 * (dz * zz).abs().sum(0).max().item())
    # the same as the LAST of several writers: earlier contributions already in the buffer are added (bit-identical to the
    # accumulating generic epilogue above) and the sums are taken over the rounded total
    gx3 = gx0.clone()
    d2.out0, d2.accumulate = gx3.data_ptr(), 1
    assert lib().yh_conv_bnr_rows(C.byref(d2)) == rows
    slab.zero_()
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    assert torch.equal(gx3, gx)
    gq = gx3.float().reshape(-1, Cin).double()
    dz = gq * (sg * (1 + a * (1 - sg)))
    got = slab.double().sum(0)
    assert torch.allclose(got[0], dz.sum(0), rtol=2e-3, atol=2e-3 * dz.abs().sum(0).max().item())
    assert torch.allclose(got[1], (dz * zz).sum(0), rtol=2e-3, atol=2e-3 * (dz * zz).abs().sum(0).max().item())


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
def test_conv_concat_upsample_algos(dev, algo):
    """two-segment input, the first read through the nearest-2x upsample, folded BN + SiLU, residual, split destination"""
    from yoloseries_amd import hipk
    B, H, W = 2, 24, 24
    lo = _nhwc(B, H // 2, W // 2, 64, dev, 33)
    skip_buf = _nhwc(B, H, W, 160, dev, 34)       # channels [32, 160) of a wider buffer
    g = torch.Generator().manual_seed(35)
    Cin, Cout = 192, 128
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16).float().to(dev)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    res = _nhwc(B, H, W, 64, dev, 36)
    out0 = torch.zeros(B, H, W, 64, dtype=torch.bfloat16, device=dev)
    out1 = torch.zeros(B, H, W, 96, dtype=torch.bfloat16, device=dev)   # write into channels [32,96)
    wp = hipk.pack_weight_fwd(w)
    d = hipk.conv_desc([hipk.Slice(lo, 0, 64, ups=1), hipk.Slice(skip_buf, 32, 128)], hipk.YH_CONV_FWD,
                       B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(out0), nsplit=64,
                       out1=hipk.Slice(out1, 32, 64), scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    d.algo = algo
    if algo >= 2:
        assert "conv_v3_kernel" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    xin = torch.cat([F.interpolate(_nchw(lo), scale_factor=2, mode="nearest"), _nchw(skip_buf[..., 32:160])], 1)
    zc = F.conv2d(xin, w, padding=1) * scale[None, :, None, None] + shift[None, :, None, None]
    a = F.silu(zc).permute(0, 2, 3, 1)
    _close(out0, a[..., :64].to(torch.bfloat16).float() + res.float(), 1e-2, 3e-2)
    _close(out1[..., 32:], a[..., 64:], 8e-3, 2e-2)
    assert (out1[..., :32] == 0).all()


@pytest.mark.parametrize("C0,C1,Cout,k", [(80, 80, 160, 1), (48, 80, 96, 1), (80, 48, 64, 3), (40, 24, 32, 1)])
def test_conv_two_segments_ragged_first_segment(dev, C0, C1, Cout, k):
    """a concat input whose FIRST segment is not a multiple of the 32-channel k-step (YOLOv5x stage 1 `cba3`: 80 + 80, YOLOv5m:
    48 + ...): conv_v2_kernel walks the channel blocks per segment (ragged block masked, next segment starts its own block)
    instead of falling back to the 64-bit-pointer kernel — folded BN + SiLU (inference) and raw output + statistics (training),
    with NaN-filled neighbour channels around both slices"""
    from yoloseries_amd import hipk
    B, H, W = 2, 20, 12
    p = k // 2
    a_buf = _nhwc(B, H, W, C0 + 16, dev, 71); a_buf[..., C0:] = float("nan")
    b_buf = _nhwc(B, H, W, C1 + 24, dev, 72); b_buf[..., :8] = float("nan"); b_buf[..., 8 + C1:] = float("nan")
    g = torch.Generator().manual_seed(73)
    w = (torch.randn(Cout, C0 + C1, k, k, generator=g) / (k * k * (C0 + C1)) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    segs = [hipk.Slice(a_buf, 0, C0), hipk.Slice(b_buf, 8, C1)]
    xin = torch.cat([_nchw(a_buf[..., :C0]), _nchw(b_buf[..., 8:8 + C1])], 1)
    ref = F.conv2d(xin, w, padding=p).permute(0, 2, 3, 1)
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    out = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, k, 1, p, wp, Cout, hipk.full(out), scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
    d.algo = 1
    assert "conv_v2_kernel" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, F.silu(ref * scale + shift), 8e-3, 2e-2)
    out2 = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    d2 = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, k, 1, p, wp, Cout, hipk.full(out2))
    d2.algo = 1
    stats = torch.zeros(hipk.conv_stat_blocks(d2), 2, wp.shape[0], device=dev)
    d2.stats = stats.data_ptr()
    assert "conv_v2_kernel" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(out2, ref, 8e-3, 2e-2)
    o = out2.float().reshape(-1, Cout).double()
    assert ((stats[:, 0, :Cout].double().sum(0) - o.sum(0)).abs() <= 1e-2 + 4 * 2.0 ** -9 * (o ** 2).sum(0).sqrt()).all()


@pytest.mark.parametrize("B,H,W", [(2, 24, 64),         # 48 image rows: strips dealt to the waves in order
                                   (4, 32, 64),         # 128 rows = whole bands of 8 rows per XCD: the XCD-aware strip order
                                   (16, 64, 128)])      # ... with two strips per wave (4 096 strips on 512 workgroups)
@pytest.mark.parametrize("Cout", [32, 48, 64, 80])
def test_conv_stem_kernel(dev, Cout, B, H, W):
    """the strip kernel of the stem (3x3 on the 16-channel space-to-depth image, 1..3 output tiles of 32 channels: v5s 32, v5m 48,
    v5l 64, v5x 80): plain store + BatchNorm partial sums, and folded BatchNorm + SiLU (inference)"""
    from yoloseries_amd import hipk
    Cin = 16
    x = _nhwc(B, H, W, Cin, dev, 41)
    g = torch.Generator().manual_seed(42)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(x), w, None, stride=1, padding=1).permute(0, 2, 3, 1)
    # training form: raw output + per-block partial sums
    out = torch.full((B, H, W, Cout), 7.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(out))
    stats = torch.zeros(hipk.conv_stat_blocks(d), 2, wp.shape[0], device=dev)
    d.stats = stats.data_ptr()
    if Cout <= 64:
        assert "conv_stem_kernel<1" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, ref, 8e-3, 2e-2)
    o = out.float().reshape(-1, Cout).double()
    assert ((stats[:, 0, :Cout].double().sum(0) - o.sum(0)).abs() <= 1e-2 + 4 * 2.0 ** -9 * (o ** 2).sum(0).sqrt()).all()
    assert ((stats[:, 1, :Cout].double().sum(0) - (o ** 2).sum(0)).abs() <= 1e-2 + 4 * 2.0 ** -8 * (o ** 4).sum(0).sqrt()).all()
    # inference form: folded BatchNorm + SiLU; the output is a channel slice of a wider buffer that must stay untouched outside it
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    obuf = torch.full((B, H, W, Cout + 16), 3.0, dtype=torch.bfloat16, device=dev)
    d2 = hipk.conv_desc([hipk.full(x)], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.Slice(obuf, 8, Cout),
                        scale=scale, shift=shift, act=hipk.YH_ACT_SILU)
    assert "conv_stem_kernel<2" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(obuf[..., 8:8 + Cout], F.silu(ref * scale + shift), 8e-3, 2e-2)
    assert (obuf[..., :8] == 3.0).all() and (obuf[..., 8 + Cout:] == 3.0).all()


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 24, 24, 160, 160), (1, 40, 40, 128, 320), (3, 17, 33, 96, 160), (1, 20, 20, 64, 480),
                                              (2, 80, 80, 320, 320), (1, 160, 160, 160, 160), (2, 32, 48, 32 + 64 * 3, 160), (1, 16, 16, 64, 160)])
def test_conv_halo160_kernel(dev, B, H, W, Cin, Cout):
    """the 160-wide halo kernel (algo 6: 3x3 / s1, N a multiple of 160, 16x16x32 MFMA tiles of 64 pixels x 80 channels; inference
    epilogues): plain store, folded BatchNorm + SiLU + residual with a split destination, and the data gradient — whole and
    ragged channel blocks (C % 64 = 32), ragged pixel tiles, several N tiles"""
    from yoloseries_amd import hipk
    x = _nhwc(B, H, W, Cin, dev, 51)
    g = torch.Generator().manual_seed(52)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(x), w, None, stride=1, padding=1).permute(0, 2, 3, 1)
    xin = hipk.full(x)
    if Cin % 64:
        xbuf = torch.full((B, H, W, Cin + 16), float("nan"), dtype=torch.bfloat16, device=dev)
        xbuf[..., 8:8 + Cin] = x
        xin = hipk.Slice(xbuf, 8, Cin)
    # plain store
    out = torch.full((B, H, W, Cout), 7.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(out))
    d.algo = 6
    assert "conv_halo160_kernel<0" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out, ref, 8e-3, 2e-2)
    # folded BatchNorm + SiLU + residual on the first 80 channels, split destination
    scale = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(Cout, generator=g).to(dev)
    res = _nhwc(B, H, W, 80, dev, 53)
    out0 = torch.zeros(B, H, W, 80, dtype=torch.bfloat16, device=dev)
    obuf = torch.full((B, H, W, Cout - 80 + 16), 3.0, dtype=torch.bfloat16, device=dev)
    d2 = hipk.conv_desc([xin], hipk.YH_CONV_FWD, B, H, W, H, W, 3, 1, 1, wp, Cout, hipk.full(out0), nsplit=80,
                        out1=hipk.Slice(obuf, 8, Cout - 80), scale=scale, shift=shift, act=hipk.YH_ACT_SILU, res=hipk.full(res))
    d2.algo = 6
    assert "conv_halo160_kernel<2" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    a = F.silu(ref * scale + shift)
    _close(out0, a[..., :80].to(torch.bfloat16).float() + res.float(), 1e-2, 3e-2)
    _close(obuf[..., 8:8 + Cout - 80], a[..., 80:], 8e-3, 2e-2)
    assert (obuf[..., :8] == 3.0).all() and (obuf[..., 8 + Cout - 80:] == 3.0).all()
    # data gradient of a Cout -> Cin' conv whose input has N' = 160 k channels: use the transposed roles (gy has Cin channels here)
    if Cin % 160 == 0:
        gy = _nhwc(B, H, W, Cout, dev, 54)
        wd = hipk.pack_weight_dgrad(w)
        xz = torch.zeros(B, Cin, H, W, device=dev, requires_grad=True)
        (gref,) = torch.autograd.grad(F.conv2d(xz, w, stride=1, padding=1), xz, _nchw(gy))
        gx = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
        d3 = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, H, W, 3, 1, 1, wd, Cin, hipk.full(gx))
        d3.algo = 6
        assert "conv_halo160_kernel" in _kname(d3), _kname(d3)
        hipk.conv_launch(d3)
        torch.cuda.synchronize()
        _close(gx, gref.permute(0, 2, 3, 1), 1e-2, 4e-2)
        # ... accumulating onto an existing gradient (generic epilogue)
        gacc = _nhwc(B, H, W, Cin, dev, 55)
        g0 = gacc.clone()
        d4 = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, W, H, W, 3, 1, 1, wd, Cin, hipk.full(gacc), accumulate=1)
        d4.algo = 6
        assert "conv_halo160_kernel<2" in _kname(d4), _kname(d4)
        hipk.conv_launch(d4)
        torch.cuda.synchronize()
        _close(gacc, gref.permute(0, 2, 3, 1).to(torch.bfloat16).float() + g0.float(), 1e-2, 4e-2)


def test_conv_wgrad_operand_of_two_gib_is_split_over_the_batch(dev):
    """yh_conv_wgrad addresses gy / x with 32-bit buffer offsets; an operand of 2 GiB or more is processed as several launches over
    sub-ranges of the batch (same atomically accumulated dw).  gy here: 8 x 1024 x 1024 x 128 bf16 = 2 GiB."""
    from yoloseries_amd import hipk
    B, H, W, Cin, Cout = 8, 1024, 1024, 8, 128
    g = torch.Generator().manual_seed(61)
    x = (torch.randn(B, H, W, Cin, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    gy = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    for b in range(B):
        gy[b] = (torch.randn(H, W, Cout, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    assert gy.numel() * 2 >= 2 ** 31
    dw = torch.zeros(Cout, Cin, device=dev)
    hipk.wgrad_launch(hipk.wgrad_desc(hipk.full(gy), Cout, hipk.full(x), 0, Cin, B, H, W, H, W, 1, 1, 0, dw, 1024))
    torch.cuda.synchronize()
    ref = torch.zeros(Cout, Cin, dtype=torch.float64, device=dev)
    for b in range(B):                      # fp64 reference image by image (bounded memory)
        ref += gy[b].reshape(-1, Cout).double().t() @ x[b].reshape(-1, Cin).double()
    _close(dw.double(), ref, 2e-3, 2e-3 * ref.abs().max().item())


@pytest.mark.parametrize("case", ["fwd3x3s2", "fwd1x1", "dgrad3x3s2", "concat_upsample"])
def test_conv_input_beyond_two_gib_stays_on_the_lds_dma_kernel(dev, case):
    """An input tensor of 2 GiB or more (34 x 512 x 512 x 128 bf16 = 2.28 GB; the boundary falls on image 32) does not fit one
    buffer descriptor: conv_v3_kernel re-bases its descriptors at every tile.  The result must be bit-identical to the same
    layer run on short slices of the batch (same kernel, whole tensor < 2 GiB) at the start, across the boundary and at the end."""
    from yoloseries_amd import hipk
    B, H, W = 34, 512, 512
    g = torch.Generator().manual_seed(61)

    def rnd(*shape):          # cheap on-device noise in bf16 range (the values only have to differ everywhere)
        t = torch.empty(*shape, dtype=torch.bfloat16, device=dev)
        t.uniform_(-1.0, 1.0)
        return t

    torch.manual_seed(62)
    if case == "dgrad3x3s2":
        Cin, Cout, k, s, p = 64, 512, 3, 2, 1
        Ho, Wo = H // 2, W // 2
        big = [rnd(B, Ho, Wo, Cout)]                                   # gy: 34 x 256 x 256 x 512 = 2.28 GB
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5).to(torch.bfloat16).float().to(dev)
        wp = hipk.pack_weight_dgrad(w)

        def run(ins, nb):
            out = torch.zeros(nb, H, W, Cin, dtype=torch.bfloat16, device=dev)
            d = hipk.conv_desc([hipk.full(ins[0])], hipk.YH_CONV_DGRAD, nb, H, W, Ho, Wo, k, s, p, wp, Cin, hipk.full(out))
            d.algo = 3
            return d, out
    else:
        k, s, p = {"fwd3x3s2": (3, 2, 1), "fwd1x1": (1, 1, 0), "concat_upsample": (1, 1, 0)}[case]
        Cout = 128
        big = [rnd(B, H, W, 128)]
        if case == "concat_upsample":
            big = [rnd(B, H // 2, W // 2, 64), big[0]]                 # first segment read through the nearest-2x upsample
        Cin = sum(t.shape[-1] for t in big)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(torch.bfloat16).float().to(dev)
        wp = hipk.pack_weight_fwd(w)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1

        def run(ins, nb):
            out = torch.zeros(nb, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
            segs = [hipk.full(t) for t in ins]
            if case == "concat_upsample":
                segs[0].ups = 1
            d = hipk.conv_desc(segs, hipk.YH_CONV_FWD, nb, Ho, Wo, H, W, k, s, p, wp, Cout, hipk.full(out))
            d.algo = 3
            return d, out
    assert big[-1].numel() * 2 >= 2 ** 31
    d, out = run(big, B)
    assert "conv_v3_kernel" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    assert out.float().abs().sum().item() > 0
    for lo, hi in ((0, 2), (30, 34), (15, 17)):
        ds, outs = run([t[lo:hi].contiguous() for t in big], hi - lo)
        assert _kname(ds) == _kname(d)
        hipk.conv_launch(ds)
        torch.cuda.synchronize()
        assert torch.equal(outs, out[lo:hi]), (case, lo, hi)
    # one image against torch (fp32 reference of the same bf16 operands)
    i = 32
    if case == "dgrad3x3s2":
        x = torch.zeros(1, Cin, H, W, device=dev, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(x, w, stride=s, padding=p), x, _nchw(big[0][i:i + 1].float()))
        ref = ref.permute(0, 2, 3, 1)
    else:
        xs = [t[i:i + 1].float() for t in big]
        if case == "concat_upsample":
            xs[0] = xs[0].repeat_interleave(2, 1).repeat_interleave(2, 2)
        ref = F.conv2d(_nchw(torch.cat(xs, -1)), w, None, stride=s, padding=p).permute(0, 2, 3, 1)
    _close(out[i:i + 1], ref, 1e-2, 4e-2)


@pytest.mark.parametrize("H,C0,N", [(40, 256, 128), (80, 128, 128), (80, 256, 256)])
def test_conv_pointwise_store_hazard_screen(dev, H, C0, N):
    """conv_pt_kernel at the judged batch (64): eight launches of a plain 1x1 data gradient into NaN-filled outputs, other kernels in
    between, every element against an fp32 matmul.  Regression screen of the store hazard of round 5: a VALU write of the first data
    register directly behind a 16-byte buffer store with a scalar offset reached the last lanes of every 16-lane row of that store —
    a box- and timing-dependent handful of wrong chunks per launch (8 of 12 launches on the box that showed it; csrc/conv_pt.hip
    pt_bstore, tools/pt_race.py)."""
    from yoloseries_amd import hipk
    B = 64
    g = torch.Generator(device=dev).manual_seed(900 + H + C0 + N)
    gy = torch.randn(B, H, H, C0, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(C0, N, 1, 1, device=dev, generator=g) / C0 ** 0.5).to(torch.bfloat16).float()
    wp = hipk.pack_weight_dgrad(w)
    ref = (gy.float().reshape(-1, C0) @ w.reshape(C0, N)).reshape(B, H, H, N)
    for r in range(8):
        out = torch.full((B, H, H, N), float("nan"), dtype=torch.bfloat16, device=dev)
        d = hipk.conv_desc([hipk.full(gy)], hipk.YH_CONV_DGRAD, B, H, H, H, H, 1, 1, 0, wp, N, hipk.full(out))
        d.algo = 13
        assert f"conv_pt_kernel<{C0}, 0, 0>" in _kname(d)
        junk = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 256, device=dev)      # noqa: F841  other bytes in LDS and the caches
        hipk.conv_launch(d)
        torch.cuda.synchronize()
        bad = ~((out.float() - ref).abs() <= 4e-2 + 1e-2 * ref.abs())
        assert not bad.any(), f"launch {r}: {int(bad.sum())} elements wrong, first rows {torch.nonzero(bad.reshape(-1, N).any(1)).flatten()[:4].tolist()}"


PT_CASES = [
    # B, H, W, C0, C1, ups0, ups1, N: conv_pt_kernel (algo 13) — 1x1 layers of the training step with 128 / 256 input channels
    (2, 20, 20, 128, 0, 0, 0, 128),       # bottleneck conv_bn_act_1 (40 x 40 class), 800 pixels = 25 tiles
    (2, 16, 16, 256, 0, 0, 0, 256),       # C3 cba1 | cba2 of stage 3: two output-channel groups
    (3, 13, 17, 128, 0, 0, 0, 64),        # odd pixel count (663: a ragged last tile), fewer channels than a group
    (1, 24, 40, 64, 64, 0, 0, 128),       # cba3 of stage 2: concat of two 64-channel halves
    (2, 16, 24, 128, 128, 0, 0, 256),     # cba3 of stage 3
    (2, 16, 16, 128, 128, 1, 0, 128),     # neck join: the first half through the nearest-2x upsample
    (1, 20, 12, 64, 64, 0, 1, 192),       # ... the second half; N = 1.5 groups
    (1, 40, 40, 256, 0, 0, 0, 255),       # Detect: 255 outputs (row pitch 256), bias
    (4, 40, 40, 128, 0, 0, 0, 128),       # 6 400 pixels: more tiles than one round of workgroups (persistent blocks walk several tiles)
    (2, 20, 20, 512, 0, 0, 0, 256),       # 512 channels (stage 4 / YOLOv5l stage 3): the weight slice of a wave is 128 registers
    (1, 20, 24, 256, 256, 1, 0, 512),     # YOLOv5l neck join: 256 upsampled + 256, four output-channel groups
]


@pytest.mark.parametrize("B,H,W,C0,C1,ups0,ups1,N", PT_CASES)
def test_conv_pointwise_training_kernel(dev, B, H, W, C0, C1, ups0, ups1, N):
    """conv_pt_kernel (algo 13): forward plain / with the BatchNorm partial sums / with bias, data-gradient form accumulating and with
    the fused BatchNorm-backward reduction (plain and as the last of several writers), inputs and outputs as channel slices of
    wider NaN-filled buffers — against fp32 torch"""
    import ctypes as C
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import lib
    g = torch.Generator().manual_seed(300 + N + C0)
    M = B * H * W
    nan = float("nan")
    segs, parts = [], []
    for si, (Cs, ups) in enumerate(((C0, ups0), (C1, ups1))):
        if Cs == 0:
            continue
        buf = torch.full((B, H >> ups, W >> ups, Cs + 16), nan, dtype=torch.bfloat16, device=dev)
        buf[..., 8:8 + Cs] = _nhwc(B, H >> ups, W >> ups, Cs, dev, 310 + si)
        segs.append(hipk.Slice(buf, 8, Cs, ups=ups))
        x = _nchw(buf[..., 8:8 + Cs])
        parts.append(F.interpolate(x, scale_factor=2, mode="nearest") if ups else x)
    xin = torch.cat(parts, 1)
    Ct = C0 + C1
    w = (torch.randn(N, Ct, 1, 1, generator=g) / Ct ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(xin, w).permute(0, 2, 3, 1)
    ldo = ((N + 7) // 8) * 8 + 16
    Nr = ((N + 7) // 8) * 8

    def desc(out, **kw):
        d = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, N, hipk.Slice(out, 8, Nr), **kw)
        d.algo = 13
        return d
    # (1) plain store
    out = torch.full((B, H, W, ldo), 5.0, dtype=torch.bfloat16, device=dev)
    d = desc(out)
    assert f"conv_pt_kernel<{C0}, {C1}, 0>" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out[..., 8:8 + N], ref, 8e-3, 2e-2)
    assert (out[..., :8] == 5.0).all() and (out[..., 8 + Nr:] == 5.0).all()
    # (2) BatchNorm partial sums of the stored values (slab rows = pixel slots of the grid; NaN-filled: every row must be written)
    out.fill_(5.0)
    rows = hipk.conv_stat_blocks(d)
    assert rows > 0 and rows % 8 == 0
    stats = torch.full((rows, 2, wp.shape[0]), nan, device=dev)
    d.stats = stats.data_ptr()
    assert f"conv_pt_kernel<{C0}, {C1}, 1>" in _kname(d) and hipk.conv_stat_blocks(d) == rows
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out[..., 8:8 + N], ref, 8e-3, 2e-2)
    o = out[..., 8:8 + N].float().reshape(-1, N).double()
    assert not torch.isnan(stats).any()
    assert ((stats[:, 0, :N].double().sum(0) - o.sum(0)).abs() <= 1e-3 + 1e-5 * o.abs().sum(0)).all()
    assert ((stats[:, 1, :N].double().sum(0) - (o ** 2).sum(0)).abs() <= 1e-3 + 1e-5 * (o ** 2).sum(0)).all()
    assert (stats[:, :, N:] == 0).all()
    # (3) bias (Detect: EPI 2, no operand from memory) ...
    bias = torch.randn(N, generator=g).to(dev)
    outb = torch.full((B, H, W, ldo), 5.0, dtype=torch.bfloat16, device=dev)
    d2 = desc(outb, bias=bias)
    assert f"conv_pt_kernel<{C0}, {C1}, 2>" in _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(outb[..., 8:8 + N], ref + bias, 1e-2, 4e-2)
    assert (outb[..., :8] == 5.0).all() and (outb[..., 8 + Nr:] == 5.0).all()
    # ... and bias + accumulate onto an existing tensor (EPI 4: the earlier contents arrive by LDS-DMA in the wave's operand slots;
    # 512 channels in all leave no LDS for the slots: the kernel declines and yh_conv_igemm takes another family)
    acc0 = torch.full((B, H, W, ldo), 5.0, dtype=torch.bfloat16, device=dev)
    acc0[..., 8:8 + N] = _nhwc(B, H, W, N, dev, 320)
    acc = acc0.clone()
    d3 = desc(acc, bias=bias, accumulate=1)
    if C0 + C1 == 512:
        assert "conv_pt_kernel" not in _kname(d3)
    else:
        assert f"conv_pt_kernel<{C0}, {C1}, 4>" in _kname(d3)
    hipk.conv_launch(d3)
    torch.cuda.synchronize()
    _close(acc[..., 8:8 + N], (ref + bias).to(torch.bfloat16).float() + acc0[..., 8:8 + N].float(), 1e-2, 4e-2)
    assert (acc[..., :8] == 5.0).all() and (acc[..., 8 + Nr:] == 5.0).all()
    # ... and with a residual (EPI 4: the residual in the slots)
    if C0 + C1 != 512:
        rs = _nhwc(B, H, W, Nr, dev, 325)
        outr = torch.full((B, H, W, ldo), 5.0, dtype=torch.bfloat16, device=dev)
        d5 = desc(outr, bias=bias, res=hipk.full(rs))
        assert f"conv_pt_kernel<{C0}, {C1}, 4>" in _kname(d5)
        hipk.conv_launch(d5)
        torch.cuda.synchronize()
        _close(outr[..., 8:8 + N], (ref + bias) + rs[..., :N].float(), 1e-2, 4e-2)
        assert (outr[..., :8] == 5.0).all() and (outr[..., 8 + Nr:] == 5.0).all()
    if N % 8 or C1 or ups0 or ups1 or C0 + C1 == 512:          # (512 channels: no fused-reduction form, csrc/conv_pt.hip pt_plan)
        return
    # (4) the data-gradient form of a 1x1 layer IS this GEMM (gy [M][Ct] x W^T): fused BatchNorm-backward reduction, plain store and
    # as the last writer of a gradient with earlier contributions
    z = _nhwc(B, H, W, N, dev, 330)
    ws = torch.cat([torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)]).to(dev)
    for accumulate in (0, 1):
        gx0 = _nhwc(B, H, W, N, dev, 340)
        gx = gx0.clone()
        d4 = hipk.conv_desc(segs, hipk.YH_CONV_DGRAD, B, H, W, H, W, 1, 1, 0, wp, N, hipk.full(gx), accumulate=accumulate)
        d4.algo = 13
        rows = lib().yh_conv_bnr_rows(C.byref(d4))
        assert rows > 0
        slab = torch.full((rows, 2, N), nan, device=dev)
        d4.bnr_z, d4.bnr_ldz, d4.bnr_C, d4.bnr_ws, d4.bnr_part = z.data_ptr(), N, N, ws.data_ptr(), slab.data_ptr()
        assert lib().yh_conv_bnr_rows(C.byref(d4)) == rows and f"conv_pt_kernel<{C0}, {C1}, 3>" in _kname(d4)
        hipk.conv_launch(d4)
        torch.cuda.synchronize()
        want = ref.to(torch.bfloat16).float() + gx0.float() if accumulate else ref
        _close(gx, want, 1e-2, 4e-2)
        gq = gx.float().reshape(-1, N).double()
        zz = z.float().reshape(-1, N).double()
        a = zz * ws[:N].double() + ws[N:].double()
        sg = torch.sigmoid(a)
        dz = gq * (sg * (1 + a * (1 - sg)))
        assert not torch.isnan(slab).any()
        got = slab.double().sum(0)
        assert torch.allclose(got[0], dz.sum(0), rtol=2e-3, atol=2e-3 * dz.abs().sum(0).max().item())
        assert torch.allclose(got[1], (dz * zz).sum(0), rtol=2e-3, atol=2e-3 * (dz * zz).abs().sum(0).max().item())


@pytest.mark.parametrize("B,H,W,N", [(2, 20, 20, 320), (1, 17, 23, 160), (3, 40, 40, 320)])
def test_conv_pointwise_kernel_320_channels_inference(dev, B, H, W, N):
    """conv_pt_kernel with 320 input channels (YOLOv5x's bottlenecks at 1280 x 1280: rows of 640 bytes, which straddle the 1-KiB
    transfers and put two rows into a bank period) — the inference forms: plain store, folded BatchNorm + SiLU + residual; the
    training forms are not built for it (statistics fall back to the default kernel)"""
    from yoloseries_amd import hipk
    g = torch.Generator().manual_seed(900 + N + H)
    nan = float("nan")
    buf = torch.full((B, H, W, 320 + 16), nan, dtype=torch.bfloat16, device=dev)
    buf[..., 8:328] = _nhwc(B, H, W, 320, dev, 910)
    segs = [hipk.Slice(buf, 8, 320)]
    w = (torch.randn(N, 320, 1, 1, generator=g) / 320 ** 0.5).to(torch.bfloat16).float().to(dev)
    wp = hipk.pack_weight_fwd(w)
    ref = F.conv2d(_nchw(buf[..., 8:328]), w).permute(0, 2, 3, 1)
    out = torch.full((B, H, W, N + 16), 5.0, dtype=torch.bfloat16, device=dev)
    d = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, N, hipk.Slice(out, 8, N))
    d.algo = 13
    assert "conv_pt_kernel<320, 0, 0>" in _kname(d), _kname(d)
    hipk.conv_launch(d)
    torch.cuda.synchronize()
    _close(out[..., 8:8 + N], ref, 8e-3, 2e-2)
    assert (out[..., :8] == 5.0).all() and (out[..., 8 + N:] == 5.0).all()
    scale, shift = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
    res = _nhwc(B, H, W, N, dev, 920)
    out.fill_(5.0)
    d2 = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, N, hipk.Slice(out, 8, N), scale=scale, shift=shift,
                        act=hipk.YH_ACT_SILU)
    d2.algo = 13
    assert "conv_pt_kernel<320, 0, 2>" in _kname(d2), _kname(d2)
    hipk.conv_launch(d2)
    torch.cuda.synchronize()
    _close(out[..., 8:8 + N], F.silu(ref * scale + shift), 1e-2, 4e-2)
    assert (out[..., :8] == 5.0).all() and (out[..., 8 + N:] == 5.0).all()
    # with a residual the 320-channel form declines (no LDS left for the operand slots): another family takes the layer, same result
    out.fill_(5.0)
    d2r = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, N, hipk.Slice(out, 8, N), scale=scale, shift=shift,
                         act=hipk.YH_ACT_SILU, res=hipk.full(res))
    d2r.algo = 13
    assert "conv_pt_kernel" not in _kname(d2r), _kname(d2r)
    hipk.conv_launch(d2r)
    torch.cuda.synchronize()
    _close(out[..., 8:8 + N], F.silu(ref * scale + shift).to(torch.bfloat16).float() + res.float(), 1e-2, 4e-2)
    assert (out[..., :8] == 5.0).all() and (out[..., 8 + N:] == 5.0).all()
    stats = torch.zeros(4096, 2, wp.shape[0], device=dev)
    d3 = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, 1, 1, 0, wp, N, hipk.Slice(out, 8, N))
    d3.algo, d3.stats = 13, stats.data_ptr()
    assert "conv_pt_kernel" not in _kname(d3), _kname(d3)


def test_conv_pointwise_training_kernel_eligibility(dev):
    """shapes the kernel does not take fall back to the library default (algo 13 ignored)"""
    from yoloseries_amd import hipk
    for (C0, C1, k, N) in [(64, 0, 1, 64), (1024, 0, 1, 256), (128, 64, 1, 128), (128, 0, 3, 128), (96, 0, 1, 96)]:
        B, H, W = 1, 8, 8
        segs = [hipk.full(_nhwc(B, H, W, C0, dev, 1))] + ([hipk.full(_nhwc(B, H, W, C1, dev, 2))] if C1 else [])
        w = torch.randn(N, C0 + C1, k, k, device=dev)
        wp = hipk.pack_weight_fwd(w)
        out = torch.zeros(B, H, W, N, dtype=torch.bfloat16, device=dev)
        d = hipk.conv_desc(segs, hipk.YH_CONV_FWD, B, H, W, H, W, k, 1, k // 2, wp, N, hipk.full(out))
        d.algo = 13
        assert "conv_pt_kernel" not in _kname(d), _kname(d)
        hipk.conv_launch(d)
        torch.cuda.synchronize()
        xin = torch.cat([_nchw(sg.buf) for sg in segs], 1)
        _close(out, F.conv2d(xin, w.to(torch.bfloat16).float(), padding=k // 2).permute(0, 2, 3, 1), 1e-2, 4e-2)
