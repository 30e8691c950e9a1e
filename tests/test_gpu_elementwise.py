"""GPU parity of the BN/SiLU, pooling, upsample-grad, input-transform and arena kernels
against plain PyTorch fp32 references on identical bf16-representable inputs.
Tolerances: outputs are rounded once to bf16 (rtol 8e-3); fp32 reductions rtol 1e-4.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand_bf16(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def _close(got, ref, rtol, atol):
    err = (got.float() - ref.float()).abs()
    lim = atol + rtol * ref.float().abs()
    assert not (err > lim).any(), f"max err {err.max().item():.4g}, {(err > lim).sum().item()} of {err.numel()} out of tol"


@pytest.mark.parametrize("C,M", [(32, 1000), (64, 4096 + 17), (256, 777), (1024, 300)])
def test_bn_silu_fwd_bwd(dev, C, M):
    from yoloseries_amd import hipk
    # conv output held in a wider buffer (channel slice) to exercise ld != C
    ybuf = _rand_bf16((M, C + 16), dev, 1, 2.0)
    y = hipk.Slice(ybuf, 8, C)
    yv = ybuf[:, 8:8 + C].float()
    gamma = (torch.rand(C) + 0.5).to(dev)
    beta = torch.randn(C).to(dev)
    rm = torch.zeros(C, device=dev)
    rv = torch.ones(C, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    # per-"block" partial stats as the conv kernel would emit (2 fake blocks)
    half = M // 2
    stats = torch.zeros(2, 2, C, device=dev)
    stats[0, 0] = yv[:half].sum(0); stats[0, 1] = (yv[:half] ** 2).sum(0)
    stats[1, 0] = yv[half:].sum(0); stats[1, 1] = (yv[half:] ** 2).sum(0)
    ws = torch.zeros(4 * C, device=dev)
    eps, mom = 1e-3, 0.03
    hipk.bn_finalize(stats, 2, C, C, M, gamma, beta, rm, rv, nbt, eps, mom, ws)
    res = _rand_bf16((M, C), dev, 2)
    out = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    hipk.bn_silu_apply(y, ws, M, hipk.full(out), hipk.full(res))
    torch.cuda.synchronize()

    x = yv.clone().requires_grad_(True)
    g_ = gamma.clone().requires_grad_(True)
    b_ = beta.clone().requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    z = F.batch_norm(x, rm_ref, rv_ref, g_, b_, True, mom, eps)
    a = F.silu(z)
    _close(out, a.detach().to(torch.bfloat16).float() + res.float(), 8e-3, 2e-2)
    assert torch.allclose(rm, rm_ref, rtol=1e-4, atol=1e-5) and torch.allclose(rv, rv_ref, rtol=1e-4, atol=1e-5)
    assert nbt.item() == 1
    mean, var = yv.mean(0), yv.var(0, unbiased=False)
    assert torch.allclose(ws[2 * C:3 * C], mean, rtol=1e-4, atol=1e-4)
    assert torch.allclose(ws[3 * C:], (var + eps).rsqrt(), rtol=1e-4, atol=1e-4)

    # backward
    ga = _rand_bf16((M, C), dev, 3)
    (gx_ref, gg_ref, gb_ref) = torch.autograd.grad(a, (x, g_, b_), ga.float())
    nblk = hipk.ew_blocks(M)
    part = torch.zeros(nblk, 2, C, device=dev)
    hipk.bn_silu_bwd_reduce(hipk.full(ga), y, ws, M, part)
    dgamma, dbeta, coef = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(2 * C, device=dev)
    hipk.bn_bwd_finalize(part, nblk, C, M, ws, dgamma, dbeta, coef)
    gy = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    gres = _rand_bf16((M, C), dev, 4)
    gres0 = gres.clone()
    hipk.bn_silu_bwd_apply(hipk.full(ga), y, ws, gamma, coef, M, hipk.full(gy), hipk.full(gres), 1)
    torch.cuda.synchronize()
    sc = gx_ref.abs().max().item()
    _close(gy, gx_ref, 1e-2, 1e-2 * sc)
    assert torch.allclose(dgamma, gg_ref, rtol=2e-3, atol=2e-3 * gg_ref.abs().max().item())
    assert torch.allclose(dbeta, gb_ref, rtol=2e-3, atol=2e-3 * gb_ref.abs().max().item())
    _close(gres, gres0.float() + ga.float(), 8e-3, 1e-2)


@pytest.mark.parametrize("Cs,M", [((32, 32), 4096 + 17), ((64, 40, 24), 1000), ((128, 128), 777), ((8, 16, 8, 32), 333)])
def test_bn_silu_passes_of_a_stacked_layer_in_one_launch(dev, Cs, M):
    """yh_bn_silu_apply_parts / yh_bn_silu_bwd_apply_parts (one pass over whole rows of a stacked ConvBnAct output) are
    bit-identical to the per-part passes: same constants, same arithmetic, different traversal."""
    from yoloseries_amd import hipk
    Ct = sum(Cs)
    ybuf = _rand_bf16((M, Ct + 16), dev, 11, 2.0)
    y = hipk.Slice(ybuf, 8, Ct)
    g = torch.Generator().manual_seed(12)
    fwd_parts, bwd_parts, ref_out, ref_gy = [], [], [], torch.zeros(M, Ct, dtype=torch.bfloat16, device=dev)
    fin_args, bfin_args = [], []
    gy = torch.full((M, Ct + 8), 3.0, dtype=torch.bfloat16, device=dev)
    c0 = 0
    for i, C in enumerate(Cs):
        yp = hipk.Slice(ybuf, 8 + c0, C)
        yv = ybuf[:, 8 + c0:8 + c0 + C].float()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
        stats = torch.zeros(1, 2, C, device=dev)
        stats[0, 0], stats[0, 1] = yv.sum(0), (yv ** 2).sum(0)
        ws = torch.zeros(4 * C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        hipk.bn_finalize(stats, 1, C, C, M, gamma, beta, rm, rv, torch.zeros(1, dtype=torch.int64, device=dev), 1e-3, 0.03, ws)
        fin_args.append(dict(C=C, stats=stats, gamma=gamma, beta=beta, ws=ws, rm=rm, rv=rv))
        # forward: every part has its own destination (a slice of a wider tensor)
        obuf = torch.full((M, C + 8 * (i + 1)), 5.0, dtype=torch.bfloat16, device=dev)
        o_ref = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
        hipk.bn_silu_apply(yp, ws, M, hipk.full(o_ref))
        ref_out.append((obuf, o_ref))
        fwd_parts.append(dict(ws=ws, C=C, out=hipk.Slice(obuf, 8, C)))
        # backward: own incoming gradient, shared gz buffer
        ga = _rand_bf16((M, C + 8), dev, 20 + i)
        gas = hipk.Slice(ga, 0, C)
        nblk = hipk.ew_blocks(M)
        part = torch.zeros(nblk, 2, C, device=dev)
        hipk.bn_silu_bwd_reduce(gas, yp, ws, M, part)
        dgamma, dbeta, coef = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(2 * C, device=dev)
        hipk.bn_bwd_finalize(part, nblk, C, M, ws, dgamma, dbeta, coef)
        hipk.bn_silu_bwd_apply(gas, yp, ws, gamma, coef, M, hipk.Slice(ref_gy, c0, C))
        bfin_args.append(dict(part=part, nblk=nblk, dgamma=dgamma, dbeta=dbeta, coef=coef))
        bwd_parts.append(dict(ws=ws, C=C, ga=gas, gamma=gamma, coef=coef))
        c0 += C
    hipk.bn_silu_apply_parts(y, M, fwd_parts)
    hipk.bn_silu_bwd_apply_parts(y, M, bwd_parts, hipk.Slice(gy, 0, Ct))
    # the finalize kernels of all parts in one launch: same statistics, running buffers, step counters, coefficients
    fin, bfin, refs = [], [], []
    for q, qb in zip(fin_args, bfin_args):
        C = q["C"]
        new = dict(ws=torch.zeros(4 * C, device=dev), rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev),
                   nbt=torch.zeros(1, dtype=torch.int64, device=dev), dgamma=torch.zeros(C, device=dev), dbeta=torch.zeros(C, device=dev),
                   coef=torch.zeros(2 * C, device=dev))
        refs.append((q, qb, new))
        fin.append(dict(ws=new["ws"], C=C, slab=q["stats"].data_ptr(), nblk=1, ldslab=C, gamma=q["gamma"], beta=q["beta"],
                        running_mean=new["rm"], running_var=new["rv"], num_batches=new["nbt"], eps=1e-3, momentum=0.03))
        bfin.append(dict(ws=q["ws"], C=C, slab=qb["part"].data_ptr(), nblk=qb["nblk"], dgamma=new["dgamma"], dbeta=new["dbeta"], coef=new["coef"]))
    hipk.bn_finalize_parts(fin, M)
    hipk.bn_bwd_finalize_parts(bfin, M)
    torch.cuda.synchronize()
    for q, qb, new in refs:
        assert torch.equal(new["ws"], q["ws"]) and torch.equal(new["rm"], q["rm"]) and torch.equal(new["rv"], q["rv"]) and new["nbt"].item() == 1
        assert torch.equal(new["dgamma"], qb["dgamma"]) and torch.equal(new["dbeta"], qb["dbeta"]) and torch.equal(new["coef"], qb["coef"])
    for (obuf, o_ref), C in zip(ref_out, Cs):
        assert torch.equal(obuf[:, 8:8 + C], o_ref)
        assert (obuf[:, :8] == 5.0).all() and (obuf[:, 8 + C:] == 5.0).all()          # nothing outside the slice
    assert torch.equal(gy[:, :Ct], ref_gy) and (gy[:, Ct:] == 3.0).all()
    assert ref_gy.float().abs().sum().item() > 0


def test_bn_fold_and_colsum(dev):
    from yoloseries_amd import hipk
    C, M = 264, 5000
    gamma, beta, rm = torch.randn(C, device=dev), torch.randn(C, device=dev), torch.randn(C, device=dev)
    rv = torch.rand(C, device=dev) + 0.1
    scale, shift = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    hipk.bn_fold(gamma, beta, rm, rv, 1e-3, C, scale, shift)
    s = gamma / torch.sqrt(rv + 1e-3)
    assert torch.allclose(scale, s, rtol=1e-5, atol=1e-6) and torch.allclose(shift, beta - rm * s, rtol=1e-5, atol=1e-5)
    g = _rand_bf16((M, C), dev, 5)
    part = torch.zeros(hipk.ew_blocks(M), 2, C, device=dev)
    out = torch.zeros(C, device=dev)
    hipk.colsum(hipk.full(g), M, part, out)
    torch.cuda.synchronize()
    assert torch.allclose(out, g.float().sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H,W,C", [(2, 20, 20, 64), (2, 40, 40, 24), (1, 13, 17, 8), (1, 7, 128, 16), (1, 3, 5, 8), (1, 9, 130, 8)])
def test_maxpool5_fwd_bwd(dev, B, H, W, C):
    """first maximum in window order on ties, NaNs propagate, backward plain and accumulating, ragged / tiny / wide images"""
    from yoloseries_amd import hipk
    x = _rand_bf16((B, H, W, C), dev, 6)
    # coarse values create ties, which must resolve to the first maximum in window order
    x = (x.float() * 2).round().div(2).to(torch.bfloat16)
    out = torch.zeros_like(x)
    idx = torch.zeros(B, H, W, C, dtype=torch.int8, device=dev)
    hipk.maxpool5_fwd(hipk.full(x), B, H, W, hipk.full(out), idx)
    xn = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.max_pool2d(xn, 5, 1, 2)
    torch.cuda.synchronize()
    assert torch.equal(out.float(), ref.permute(0, 2, 3, 1))
    go = _rand_bf16((B, H, W, C), dev, 7)
    (gref,) = torch.autograd.grad(ref, xn, go.float().permute(0, 3, 1, 2))
    gin = torch.zeros_like(x)
    hipk.maxpool5_bwd(hipk.full(go), idx, B, H, W, hipk.full(gin), 0)
    torch.cuda.synchronize()
    _close(gin, gref.permute(0, 2, 3, 1), 8e-3, 2e-2)
    # accumulate on top of an existing gradient
    gin2 = _rand_bf16((B, H, W, C), dev, 8)
    g0 = gin2.clone()
    hipk.maxpool5_bwd(hipk.full(go), idx, B, H, W, hipk.full(gin2), 1)
    torch.cuda.synchronize()
    _close(gin2, gref.permute(0, 2, 3, 1) + g0.float(), 8e-3, 3e-2)
    # NaNs propagate to every window that contains one
    xn2 = x.clone()
    xn2[0, H // 2, W // 2, :] = float("nan")
    xn2[0, 0, 0, 0] = float("nan")
    out2 = torch.zeros_like(x)
    hipk.maxpool5_fwd(hipk.full(xn2), B, H, W, hipk.full(out2), idx)
    torch.cuda.synchronize()
    ref2 = F.max_pool2d(xn2.float().permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1)
    assert torch.equal(torch.isnan(out2.float()), torch.isnan(ref2))
    ok = ~torch.isnan(ref2)
    assert torch.equal(out2.float()[ok], ref2[ok])


@pytest.mark.parametrize("B,H,W,C", [(3, 20, 20, 256), (2, 12, 17, 72), (1, 5, 3, 8), (2, 20, 24, 128), (2, 24, 32, 80), (1, 40, 40, 72), (2, 40, 48, 16)])
def test_sppf_fused_pools_bit_identical(dev, B, H, W, C):
    """FastSPP's three chained 5x5 pools in one launch per direction (csrc/sppf.hip) against three yh_maxpool5_fwd / _bwd launches:
    outputs, arg-max bytes and the input gradient BIT-identical — with ties (coarse values), NaNs and a ragged channel count; maps of
    up to 480 / 960 / 1920 pixels run with 32 / 16 / 8 channels per block (40 x 40: YOLOv5x at 1280 x 1280)"""
    import ctypes as C_
    from yoloseries_amd import hipk
    from yoloseries_amd._lib import check, lib, stream_ptr
    assert lib().yh_sppf_pool3_ok(H, W, C) == 1 and lib().yh_sppf_pool3_ok(40, 40, C) == 1 and lib().yh_sppf_pool3_ok(40, 49, C) == 0
    cat = torch.zeros(B, H, W, 4 * C, dtype=torch.bfloat16, device=dev)
    x = (_rand_bf16((B, H, W, C), dev, 16).float() * 2).round().div(2).to(torch.bfloat16)
    x[0, H // 2, W // 2, :3] = float("nan")
    cat[..., :C] = x
    ref = cat.clone()
    idx_r = [torch.zeros(B, H, W, C, dtype=torch.int8, device=dev) for _ in range(3)]
    for k in range(3):
        hipk.maxpool5_fwd(hipk.Slice(ref, k * C, C), B, H, W, hipk.Slice(ref, (k + 1) * C, C), idx_r[k])
    idx_f = [torch.full((B, H, W, C), 99, dtype=torch.int8, device=dev) for _ in range(3)]
    sl = [hipk.Slice(cat, k * C, C) for k in range(4)]
    check(lib().yh_sppf_pool3_fwd(sl[0].ptr(), sl[0].ld, B, H, W, C, sl[1].ptr(), sl[2].ptr(), sl[3].ptr(), sl[1].ld,
                                  idx_f[0].data_ptr(), idx_f[1].data_ptr(), idx_f[2].data_ptr(), stream_ptr()), "yh_sppf_pool3_fwd")
    torch.cuda.synchronize()
    assert torch.equal(cat.view(torch.int16), ref.view(torch.int16))
    for a, b_ in zip(idx_f, idx_r):
        assert torch.equal(a, b_)
    # backward: gradient of the concat buffer as cba2's data gradient leaves it, then the pools' backward
    g = _rand_bf16((B, H, W, 4 * C), dev, 17)
    gr = g.clone()
    for k in (2, 1, 0):
        hipk.maxpool5_bwd(hipk.Slice(gr, (k + 1) * C, C), idx_r[k], B, H, W, hipk.Slice(gr, k * C, C), 1)
    gf = g.clone()
    gs = [hipk.Slice(gf, k * C, C) for k in range(4)]
    check(lib().yh_sppf_pool3_bwd(gs[1].ptr(), gs[2].ptr(), gs[3].ptr(), gs[1].ld, idx_f[0].data_ptr(), idx_f[1].data_ptr(), idx_f[2].data_ptr(),
                                  B, H, W, C, gs[0].ptr(), gs[0].ld, 1, stream_ptr()), "yh_sppf_pool3_bwd")
    torch.cuda.synchronize()
    assert torch.equal(gf[..., :C].view(torch.int16), gr[..., :C].view(torch.int16))
    assert torch.equal(gf[..., C:].view(torch.int16), g[..., C:].view(torch.int16))          # the intermediate gradients are left as they were


def test_upsample2_bwd_and_s2d(dev):
    from yoloseries_amd import hipk
    B, Hl, Wl, C = 2, 10, 12, 32
    ghi = _rand_bf16((B, 2 * Hl, 2 * Wl, C), dev, 8)
    glo = _rand_bf16((B, Hl, Wl, C), dev, 9)
    glo0 = glo.clone()
    hipk.upsample2_bwd(hipk.full(ghi), B, Hl, Wl, hipk.full(glo), 1)
    ref = ghi.float().reshape(B, Hl, 2, Wl, 2, C).sum((2, 4)) + glo0.float()
    torch.cuda.synchronize()
    _close(glo, ref, 8e-3, 2e-2)
    x = torch.rand(2, 3, 16, 24, device=dev)
    out = torch.ones(2, 8, 12, 16, dtype=torch.bfloat16, device=dev)
    hipk.input_s2d(x, out)
    torch.cuda.synchronize()
    ref = x.reshape(2, 3, 8, 2, 12, 2).permute(0, 2, 4, 3, 5, 1).reshape(2, 8, 12, 12).to(torch.bfloat16)
    assert torch.equal(out[..., :12], ref) and (out[..., 12:] == 0).all()


def test_arena_kernels(dev):
    from yoloseries_amd import hipk
    n = 100003
    src = torch.randn(n, device=dev)
    idx = torch.randint(-1, n, (2 * n,), dtype=torch.int32, device=dev)
    dst = torch.zeros(2 * n, dtype=torch.bfloat16, device=dev)
    hipk.pack_bf16(src, idx, dst)
    ref = torch.where(idx >= 0, src[idx.clamp(min=0).long()], torch.zeros((), device=dev)).to(torch.bfloat16)
    assert torch.equal(dst, ref)
    d2 = torch.zeros(2 * n, device=dev)
    hipk.gather_f32(src, idx, d2)
    assert torch.equal(d2, torch.where(idx >= 0, src[idx.clamp(min=0).long()], torch.zeros((), device=dev)))
    # SGD nesterov vs torch.optim.SGD, two groups with different lr/wd, two steps
    p = torch.randn(n, device=dev)
    p_ref = [p[:n // 2].clone().requires_grad_(True), p[n // 2:].clone().requires_grad_(True)]
    opt = torch.optim.SGD([{"params": [p_ref[0]], "lr": 0.1, "weight_decay": 0.0},
                           {"params": [p_ref[1]], "lr": 0.01, "weight_decay": 5e-4}], lr=0.1, momentum=0.937, nesterov=True)
    group = torch.zeros(n, dtype=torch.uint8, device=dev); group[n // 2:] = 1
    lr = torch.tensor([0.1, 0.01], device=dev); wd = torch.tensor([0.0, 5e-4], device=dev)
    buf = torch.zeros(n, device=dev)
    for step in range(2):
        g = torch.randn(n, device=dev)
        p_ref[0].grad, p_ref[1].grad = g[:n // 2].clone(), g[n // 2:].clone()
        opt.step()
        hipk.sgd_step(p, g, buf, group, lr, wd, 0.937, True, step == 0)
    torch.cuda.synchronize()
    assert torch.allclose(p, torch.cat([p_ref[0], p_ref[1]]).detach(), rtol=1e-5, atol=1e-6)
    part, out = torch.zeros(4096, device=dev), torch.zeros(1, device=dev)
    hipk.sumsq(src, part, out)
    assert torch.allclose(out, (src.double() ** 2).sum().float(), rtol=1e-5)
    e = torch.randn(n, device=dev); e0 = e.clone()
    hipk.ema_update(e, src, 0.99)
    assert torch.allclose(e, 0.99 * e0 + 0.01 * src, rtol=1e-5, atol=1e-6)
