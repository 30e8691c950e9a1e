"""hipGraph replay of the whole train step (utils/graph.py): forward, loss, backward (incl. the side-stream weight-gradient
branch), clip, SGD-nesterov, EMA captured once and replayed must give the same training trajectory as launching every
kernel from Python; the per-step scalars (learning rate, momentum, EMA decay) are read from device memory."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, graph):
    import bench
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.trainer import ExponentialMovingAverageModel
    from yoloseries_amd.utils import FlatSGD, GraphedStep
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    torch.manual_seed(0)
    B, img = 4, 128
    model = models.YOLOV5Small(3, 80).to(dev).train()
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
    opt = FlatSGD(model, lr=0.002, momentum=0.9, weight_decay=1e-4, nesterov=True)
    ema = ExponentialMovingAverageModel(model)
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(1)).to(dev)
    t = torch.from_numpy(synth_targets(B, img, 80, 12, seed=4, min_boxes=6)).to(dev)      # every head stage gets positives

    def step():
        out = lossf(model(x), t)
        out["tot_loss"].backward()
        opt.clip_grad_norm_(10.0)
        opt.step()
        opt.zero_grad()
        ema.update(model)
        return out
    stepper = GraphedStep(step, pre_replay=[opt.graph_pre_replay, ema.graph_pre_replay], warmup=2, enabled=graph)
    return model, opt, ema, stepper


def test_graph_replay_matches_eager(dev):
    runs = {}
    for graph in (False, True):
        model, opt, ema, stepper = _setup(dev, graph)
        losses = []
        for it in range(8):
            if it == 5:                           # a schedule change between replays must reach the device scalars
                for g in opt.param_groups:
                    g["lr"] = 0.001
                    g["momentum"] = 0.8
            out = stepper()
            losses.append(float(out["tot_loss"].item()))
        torch.cuda.synchronize()
        assert stepper.mode == ("hipGraph replay" if graph else "eager"), stepper.failed
        assert np.isfinite(losses).all(), losses
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
        eflat = torch.cat([p.detach().reshape(-1) for p in ema.ema.parameters()]).cpu().numpy()
        # the schedule change reached the device-resident step scalars (lr per group | weight decay per group | momentum | first)
        scal = opt.scal.cpu().numpy()
        np.testing.assert_allclose(scal[:3], 0.001, rtol=1e-6)
        np.testing.assert_allclose(scal[6], 0.8, rtol=1e-6)
        runs[graph] = (np.array(losses), flat, eflat, opt.steps, ema.update_num)
    (l0, p0, e0, s0, u0), (l1, p1, e1, s1, u1) = runs[False], runs[True]
    assert s0 == s1 == 8 and u0 == u1 == 8
    assert np.isfinite(l1).all()
    # identical kernels on identical inputs, but the weight gradients are summed with fp32 atomics whose order varies from run
    # to run, and a train-mode net amplifies those last-bit differences step by step (two EAGER runs drift apart the same
    # way: tools/graph_variation.py measured, over repeated eager / replayed runs on one box, loss differences of 0, 1.1, 1.9 or
    # 3.0 % by step 8 and EMA differences of up to 4.2e-4 of the largest weight — discrete branches, whichever pair is compared):
    # the first steps must agree tightly, the later ones inside three to five times that run-to-run spread
    # (the tight comparison — 1e-3 over all steps, exact device scalars — is test_graph_replay_matches_eager_deterministic below; this
    # one runs the DEFAULT kernels, whose weight gradients sum partial tiles with fp32 atomics in arrival order)
    # (step 0 runs on identical weights through deterministic kernels: equal; step 1 already sees the first update's atomics order —
    # 0.26 % was observed once in ~10 runs of the suite)
    np.testing.assert_allclose(l1[:1], l0[:1], rtol=1e-6)
    np.testing.assert_allclose(l1[:2], l0[:2], rtol=1e-2)
    np.testing.assert_allclose(l1, l0, rtol=2e-1)
    assert np.abs(p1 - p0).max() <= 3e-2 * np.abs(p0).max()
    assert np.abs(e1 - e0).max() <= 3e-3 * np.abs(e0).max() + 1e-7


def test_graph_replay_matches_eager_deterministic(dev, monkeypatch):
    """the same comparison with every run-to-run source of variation switched off — weight gradients through the workspace form
    (partial tiles by plain stores, summed in split order: bit-reproducible; the atomic patch / wave-private forms and the fused
    stem backward are not eligible with a workspace) — so that eager and replay can be held to 1e-3 over ALL steps: a replay that
    drops or mis-orders a kernel (a stale learning-rate scalar, a missed EMA update) cannot hide inside the atomics' drift"""
    from yoloseries_amd import engine
    monkeypatch.setattr(engine.flags, "WG_WS_BYTES", 256 << 20)
    runs = {}
    for graph in (False, True):
        model, opt, ema, stepper = _setup(dev, graph)
        losses = []
        for it in range(8):
            if it == 5:
                for g in opt.param_groups:
                    g["lr"] = 0.001
                    g["momentum"] = 0.8
            losses.append(float(stepper()["tot_loss"].item()))
        torch.cuda.synchronize()
        assert stepper.mode == ("hipGraph replay" if graph else "eager"), stepper.failed
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
        eflat = torch.cat([p.detach().reshape(-1) for p in ema.ema.parameters()]).cpu().numpy()
        runs[graph] = (np.array(losses), flat, eflat, opt.steps, ema.update_num, opt.scal.cpu().numpy())
    (l0, p0, e0, s0, u0, c0), (l1, p1, e1, s1, u1, c1) = runs[False], runs[True]
    assert s0 == s1 == 8 and u0 == u1 == 8
    np.testing.assert_array_equal(c0, c1)                       # device-resident step scalars after the schedule change
    np.testing.assert_allclose(l1, l0, rtol=1e-3)
    assert np.abs(p1 - p0).max() <= 1e-3 * np.abs(p0).max()
    assert np.abs(e1 - e0).max() <= 1e-4 * np.abs(e0).max() + 1e-7


def test_device_scalar_kernels(dev):
    """yh_sgd_step_dev / yh_ema_advance / yh_ema_update_dev against torch.optim.SGD and the reference's EMA formula"""
    from yoloseries_amd import hipk
    n = 4099
    g = torch.Generator().manual_seed(3)
    p = torch.randn(n, generator=g).to(dev)
    grad = torch.randn(n, generator=g).to(dev)
    group = (torch.arange(n) % 3).to(torch.uint8).to(dev)
    lrs, wds, mom = [0.1, 0.05, 0.2], [0.0, 1e-2, 0.0], 0.9
    ref = [torch.nn.Parameter(p[group == i].clone()) for i in range(3)]
    opt = torch.optim.SGD([{"params": [ref[i]], "lr": lrs[i], "weight_decay": wds[i]} for i in range(3)], lr=0.1, momentum=mom, nesterov=True)
    buf = torch.zeros(n, device=dev)
    scal = torch.tensor(lrs + wds + [mom, 1.0], dtype=torch.float32, device=dev)
    pp = p.clone()
    for it in range(3):
        for i in range(3):
            ref[i].grad = grad[group == i].clone()
        opt.step()
        scal[7] = 1.0 if it == 0 else 0.0
        hipk.sgd_step_dev(pp, grad, buf, group, scal, True)
    for i in range(3):
        torch.testing.assert_close(pp[group == i], ref[i].detach(), rtol=1e-5, atol=1e-6)
    cnt = torch.tensor([41], dtype=torch.int64, device=dev)
    dec = torch.zeros(1, device=dev)
    e = torch.randn(n, generator=g).to(dev)
    e0 = e.clone()
    hipk.ema_advance(cnt, dec, 0.9999, 2000.0)
    hipk.ema_update_dev(e, p, dec)
    d = 0.9999 * (1 - math.exp(-42 / 2000))
    assert int(cnt.item()) == 42 and abs(float(dec.item()) - d) <= 1e-7
    torch.testing.assert_close(e, np.float32(d) * e0 + (1 - np.float32(d)) * p, rtol=1e-5, atol=1e-6)
