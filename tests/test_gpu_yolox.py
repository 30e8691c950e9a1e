"""GPU parity of the YOLOX path (model, SimOTA loss + backward, decode, post-processing) against golden
vectors produced by the reference (tests/golden/g8_yolox.npz) — foreground assignment bit-exact, loss scalars
and gradients within 1e-4 relative (fp32), model outputs with bf16 tolerance."""
import os

import numpy as np
import pytest
import torch

from yoloseries_amd.utils.synth import synth_targets, synth_yolox_heads

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _hypx(dev, img, focal=False, itype="ciou", **kw):
    h = dict(device=dev, num_class=80, input_img_size=[img, img], use_focal_loss=focal, focal_loss_gamma=1.5, focal_loss_alpha=0.25,
             iou_loss_scale=5.0, use_l1=True, l1_loss_scale=1.0, cls_loss_scale=1.0, cof_loss_scale=1.0, class_smooth_factor=1.0,
             cls_pos_weight=1.0, cof_pos_weight=1.0, num_anchors=1, iou_type=itype, topk=13, center_radius=3, num_stage=3,
             iou_threshold=0.2, conf_threshold=0.3, cls_threshold=0.3, max_predictions_per_img=300, mutil_label=False, agnostic=True,
             postprocess_bbox=True, wfb=False, use_tta=False, half=False, compute_metric_conf_threshold=0.001,
             compute_metric_iou_threshold=0.65, compute_metric_cls_threshold=0.001)
    h.update(kw)
    return h


@pytest.mark.parametrize("key", ["plain_ciou", "focal_giou", "plain_iou"])
def test_yolox_loss_golden(dev, key):
    from yoloseries_amd.loss import YOLOXLoss
    g = np.load(os.path.join(G, "g8_yolox.npz"))
    img, batch, focal, seed = (int(v) for v in g[f"{key}_args"])
    lf = YOLOXLoss(_hypx(dev, img, bool(focal), str(g[f"{key}_itype"])))
    for call in range(2):
        t = torch.from_numpy(synth_targets(batch, img, 80, 5, seed=seed + call, min_boxes=2)).to(dev)
        heads = synth_yolox_heads(batch, img, 80, seed=seed + 10 + call)
        preds = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in heads.items()}
        out = lf(preds, t)
        vals = g[f"{key}_c{call}_vals"]
        got = np.array([out["tot_loss"].item(), out["iou_loss"], out["l1_loss"], out["cls_loss"], out["cof_loss"], out["fg_nums"], out["tar_nums"]])
        assert got[5] == vals[5] and got[6] == vals[6], (got, vals)
        np.testing.assert_allclose(got[:5], vals[:5], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(lf.balances, g[f"{key}_c{call}_balances"], rtol=1e-5)
        np.testing.assert_array_equal(t.cpu().numpy(), g[f"{key}_c{call}_tars_after"])
        grads = torch.autograd.grad(out["tot_loss"], list(preds.values()))
        for s, gr in enumerate(grads):
            ref = g[f"{key}_c{call}_grad{s}"]
            assert tuple(gr.shape) == ref.shape
            np.testing.assert_allclose(gr.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_yolox_assignment_bit_exact(dev):
    from yoloseries_amd.loss import YOLOXLoss
    g = np.load(os.path.join(G, "g8_yolox.npz"))
    img, batch, seed_t, seed_p = (int(v) for v in g["assign_args"])
    lf = YOLOXLoss(_hypx(dev, img))
    t = torch.from_numpy(synth_targets(batch, img, 80, 5, seed=seed_t, min_boxes=2)).to(dev)
    heads = synth_yolox_heads(batch, img, 80, seed=seed_p)
    lf({k: torch.from_numpy(v).to(dev) for k, v in heads.items()}, t)
    masks = lf.foreground_masks()
    for s, m in enumerate(masks):
        np.testing.assert_array_equal(m, g[f"assign_s{s}_fg"])
        assert m.sum() == g[f"assign_s{s}_n"][0]


def test_yolox_model_and_evaluator(dev):
    from yoloseries_amd import models
    from yoloseries_amd.trainer import YOLOXEvaluator
    g = np.load(os.path.join(G, "g8_yolox.npz"))
    torch.manual_seed(0)
    m = models.YOLOXSmall(1, 3, 80, 0.01).to(dev).eval()
    x = torch.from_numpy(np.random.RandomState(93).rand(2, 3, 64, 64).astype(np.float32)).to(dev)
    with torch.no_grad():
        outs = m(x)
    assert list(outs.keys()) == ["pred_s", "pred_m", "pred_l"]
    for s, o in enumerate(outs.values()):
        ref = g[f"m_eval64_out{s}"]
        assert tuple(o.shape) == ref.shape
        err = np.abs(o.float().cpu().numpy() - ref)
        assert (err <= 3e-2 * np.abs(ref).max() + 3e-2 * np.abs(ref)).all(), f"stage {s}: max err {err.max()}"
    # decode golden (reference evaluator on stub heads)
    b, img, nc, seed, scale = g["dec_args"]
    heads = synth_yolox_heads(int(b), int(img), int(nc), seed=int(seed), scale=float(scale))
    ev = YOLOXEvaluator(None, _hypx(dev, int(img)))
    dec = ev.decode({k: torch.from_numpy(v).to(dev) for k, v in heads.items()}, int(img))
    np.testing.assert_allclose(dec.cpu().numpy(), g["decoded"], rtol=1e-4, atol=1e-4)
    # post-processing golden (YOLOX thresholds: obj*max(cls) >= conf, class confidence >= thr)
    ev2 = YOLOXEvaluator(None, _hypx(dev, 320, num_class=4))
    res = ev2.numba_nms(torch.from_numpy(g["nms_dec"]).to(dev))
    assert [(-1 if o is None else len(o)) for o in res] == list(g["nms_n"])
    for i, o in enumerate(res):
        if o is not None:
            np.testing.assert_array_equal(o, g[f"nms_out{i}"])


def test_yolox_multi_label_golden(dev):
    """hyp['mutil_label'] for the YOLOX evaluator (trainer/eval_yolox.py:218-221; round 4): every (prediction, class) with
    cls*obj >= cls_thr among the predictions with obj * max(cls) >= conf_thr enters NMS as a candidate of its own — rows and pick
    order of the reference evaluator bit for bit (the candidate table has to grow past one row per prediction), the oracle on fresh
    inputs, and the head path (decode + multi-label filter) against the decoded path"""
    from oracle import postproc as opp
    from yoloseries_amd.trainer import YOLOXEvaluator
    g = np.load(os.path.join(G, "g13_round4.npz"))
    dec = g["mlx_dec"]
    ev = YOLOXEvaluator(None, _hypx(dev, 320, num_class=dec.shape[2] - 5, mutil_label=True))
    outs = ev.numba_nms(torch.from_numpy(dec).to(dev))
    assert [(-1 if o is None else len(o)) for o in outs] == list(g["mlx_n"])
    for i, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o, g[f"mlx_out{i}"])
    assert max(ev.last_ncand) > dec.shape[1]
    r = np.random.RandomState(78)
    d2 = r.uniform(0, 1, (2, 300, 5 + 8)).astype(np.float32)
    d2[..., :2] = r.uniform(20, 300, (2, 300, 2)); d2[..., 2:4] = r.uniform(10, 60, (2, 300, 2))
    ev2 = YOLOXEvaluator(None, _hypx(dev, 320, num_class=8, mutil_label=True, conf_threshold=0.2, cls_threshold=0.1))
    got = ev2.numba_nms(torch.from_numpy(d2).to(dev))
    exp = opp.postprocess_yolox(d2, 0.2, 0.1, 0.2, multi_label=True)
    for a, b in zip(got, exp):
        np.testing.assert_array_equal(a, b)
    heads = synth_yolox_heads(2, 64, 4, seed=5, scale=2.0)
    ev3 = YOLOXEvaluator(None, _hypx(dev, 64, num_class=4, mutil_label=True, conf_threshold=0.05, cls_threshold=0.05))
    hp = {k: torch.from_numpy(v).to(dev) for k, v in heads.items()}
    a = ev3._nms_from_heads(hp, 64)
    b = ev3.numba_nms(ev3.decode(hp, 64))
    assert any(x is not None for x in a)
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            np.testing.assert_array_equal(x, y)


def test_yolox_train_step(dev):
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOXLoss
    torch.manual_seed(0)
    m = models.YOLOXSmall(1, 3, 80, 0.01).to(dev).train()
    lf = YOLOXLoss(_hypx(dev, 128))
    opt = torch.optim.SGD(m.parameters(), lr=0.002, momentum=0.9, nesterov=True)
    x = torch.from_numpy(np.random.RandomState(3).rand(4, 3, 128, 128).astype(np.float32)).to(dev)
    t0 = synth_targets(4, 128, 80, 5, seed=4, min_boxes=2)
    losses = []
    for it in range(6):
        out = lf(m(x), torch.from_numpy(t0.copy()).to(dev))
        opt.zero_grad()
        out["tot_loss"].backward()
        if it == 0:
            missing = [n for n, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all() or p.grad.abs().sum() == 0]
            assert not missing, f"parameters without a finite non-zero gradient: {missing[:8]}"
        opt.step()
        losses.append(out["tot_loss"].item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("topk,maxb", [(13, 20), (10, 40), (20, 40)])
def test_yolox_loss_vs_oracle_640_b8(dev, topk, maxb):
    """BASELINE config #3 scale (640x640, ~10^3 candidate cells per image, dynamic k up to 13) on fresh seeds: SimOTA
    foreground masks bit-identical to the oracle (ties resolved to the lowest index, which is what the kernel documents),
    loss scalars / counts / gradients within 1e-4; the number of cells where the tie rule matters is reported.
    Up to 40 boxes per image runs the matcher's rounds of 16 ground truths three times; topk 20 is beyond the
    wave-list path (<= 16) and takes the IoU / cost matrices in the workspace instead."""
    from oracle.yoloxloss import YOLOXLossOracle
    from yoloseries_amd.loss import YOLOXLoss
    img, B = 640, 8
    hyp = _hypx(dev, img, topk=topk)
    seed = 3101 + maxb
    while True:        # the reference's select_grid falls back to torch.randperm when no cell lies in any box: outside the pinned domain
        tnp = synth_targets(B, img, 80, maxb, seed=seed, min_boxes=4)
        heads = synth_yolox_heads(B, img, 80, seed=seed + 1)
        ohyp = dict(hyp); ohyp["device"] = "cpu"
        o_stable, o_plain = YOLOXLossOracle(dict(ohyp), stable_ties=True), YOLOXLossOracle(dict(ohyp))
        try:
            opreds = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in heads.items()}
            oout = o_stable(opreds, torch.from_numpy(tnp.copy()))
            o_plain({k: torch.from_numpy(v.copy()) for k, v in heads.items()}, torch.from_numpy(tnp.copy()))
            break
        except RuntimeError:
            seed += 10
    ties = int(sum((a != b).sum() for a, b in zip(o_stable.last_fg, o_plain.last_fg)))
    print(f"yolox 640 b8: seed {seed}, fg {oout['fg_nums']}, gt {oout['tar_nums']}, cells decided by the cost-tie rule: {ties}")
    lf = YOLOXLoss(hyp)
    preds = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in heads.items()}
    out = lf(preds, torch.from_numpy(tnp.copy()).to(dev))
    for s, (mk, ofg) in enumerate(zip(lf.foreground_masks(), o_stable.last_fg)):
        np.testing.assert_array_equal(mk, ofg.numpy(), err_msg=f"foreground mask of stage {s}")
    assert out["fg_nums"] == oout["fg_nums"] and out["tar_nums"] == oout["tar_nums"]
    got = np.array([out["tot_loss"].item(), out["iou_loss"], out["l1_loss"], out["cls_loss"], out["cof_loss"]])
    ref = np.array([oout["tot_loss"].item(), oout["iou_loss"], oout["l1_loss"], oout["cls_loss"], oout["cof_loss"]])
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(lf.balances, o_stable.balances, rtol=1e-5)
    grads = torch.autograd.grad(out["tot_loss"], list(preds.values()))
    ograds = torch.autograd.grad(oout["tot_loss"], list(opreds.values()))
    for gr, og in zip(grads, ograds):
        r = og.numpy()
        np.testing.assert_allclose(gr.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * np.abs(r).max())


def test_yolox_learns_a_detection_task(dev):
    """YOLOXSmall + SimOTA loss + FlatSGD on the learnable synthetic task (coloured rectangles, colour = class): the loss must fall
    and the evaluator's detections on fresh images must reach a non-trivial mAP — an end-to-end check of the direction of every
    gradient of the anchor-free path (assignment, IoU / L1 / class / objectness terms, decode, NMS, mAP_v2)."""
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOXLoss
    from yoloseries_amd.trainer import YOLOXEvaluator
    from yoloseries_amd.utils import FlatSGD, mAP_v2
    from yoloseries_amd.utils.synth import synth_shapes_batch
    torch.manual_seed(0)
    img, B, nc = 320, 16, 80
    m = models.YOLOXSmall(1, 3, nc, 0.01).to(dev).train()
    hyp = _hypx(dev, img, loss_items_on_device=True)
    lf = YOLOXLoss(hyp)
    opt = FlatSGD(m, lr=0.01, momentum=0.9, weight_decay=5e-4, nesterov=True)
    losses = []
    nsteps = 300
    for it in range(nsteps):
        lr = 0.01 * min(1.0, (it + 1) / 30) * (0.1 + 0.9 * (1 - it / nsteps))
        for g in opt.param_groups:
            g["lr"] = lr
        im, an = synth_shapes_batch(B, img, 4, 4, seed=1000 + it)
        out = lf(m(torch.from_numpy(im).to(dev)), torch.from_numpy(an).to(dev))
        out["tot_loss"].backward()
        opt.clip_grad_norm_(10.0)
        opt.step()
        opt.zero_grad()
        if it % 10 == 0 or it >= nsteps - 10:
            losses.append(float(out["tot_loss"].item()))
    assert np.isfinite(losses).all() and np.mean(losses[-10:]) < 0.6 * np.mean(losses[:3]), losses
    m.eval()
    ev = YOLOXEvaluator(m, hyp, compute_metric=True)
    gts, preds = [], []
    for k in range(2):
        im, an = synth_shapes_batch(B, img, 4, 4, seed=5000 + k)
        outs = ev(torch.from_numpy(im).to(dev))
        for b in range(B):
            gts.append(an[b][an[b][:, 4] >= 0][:, :5])
            preds.append(outs[b].numpy() if outs[b] is not None else np.zeros((0, 6), np.float32))
    assert any(len(p) for p in preds)
    mp, m50, prec, rec = mAP_v2(gts, preds).get_mean_metrics()
    assert m50 > 0.08 and rec > 0.12, (mp, m50, prec, rec)


def test_yolox_tta_matches_oracle_composition(dev):
    """YOLOXEvaluator with use_tta (trainer/eval_yolox.py:94-122, 153-168; the shipped config's default): three passes (scale 1 /
    0.83 + flip-y / 0.67 + flip-x), each decoded with ITS input height, un-scaled and un-flipped, concatenated before NMS.  The stub
    model records what it is fed and returns fixed heads; decode and post-processing against the oracle."""
    import torch.nn.functional as F
    from oracle import postproc as opp
    from yoloseries_amd.trainer import YOLOXEvaluator
    img = 320
    heads = synth_yolox_heads(2, img, 80, seed=61)
    for v in heads.values():                        # a few confident cells so that NMS has something to do
        v[:, :, 4] -= 2.0
        v[:, :, 4, ::5, ::7] += 5.0
        v[:, :, 5:] -= 2.0
        v[:, :, 5 + 3, ::5, ::7] += 5.0
    ht = {k: torch.from_numpy(v).to(dev) for k, v in heads.items()}
    seen = []

    def stub(x):
        seen.append(x.detach().cpu())
        return ht
    ev = YOLOXEvaluator(stub, _hypx(dev, img, use_tta=True))
    x = torch.rand(2, 3, img, img, generator=torch.Generator().manual_seed(4)).to(dev)
    merged, parts = ev.test_time_augmentation(x)
    assert len(seen) == 3 and all(s.shape == (2, 3, img, img) for s in seen) and torch.equal(seen[0], x.cpu())
    for sc, f, k in ((0.83, 2, 1), (0.67, 3, 2)):
        nh = int(sc * img)
        want = F.pad(F.interpolate(x.cpu().flip(dims=(f,)), size=(nh, nh), align_corners=False, mode='bilinear'),
                     [0, img - nh, 0, img - nh], value=0.447)
        assert torch.allclose(seen[k], want, atol=1e-6)
    dec = opp.decode_yolox(list(heads.values()), img)
    want = []
    for sc, f in ((1, None), (0.83, 2), (0.67, 3)):
        d = dec.copy()
        d[..., :4] /= np.float32(sc)
        if f == 2:
            d[..., 1] = img - d[..., 1]
        if f == 3:
            d[..., 0] = img - d[..., 0]
        want.append(d)
    want = np.concatenate(want, axis=1)
    assert merged.shape == want.shape and len(parts) == 3
    np.testing.assert_allclose(merged.cpu().numpy(), want, rtol=2e-5, atol=2e-4)
    seen.clear()
    res = ev(x)
    assert len(seen) == 3
    ref = opp.postprocess_yolox(merged.cpu().numpy(), 0.3, 0.3, 0.2)
    assert any(r is not None and len(r) > 0 for r in ref)
    for o, r in zip(res, ref):
        assert (o is None) == (r is None)
        if r is not None:
            np.testing.assert_array_equal(o.numpy(), r)
