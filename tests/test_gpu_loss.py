"""GPU parity of the YOLOv5 loss path (assign / CIoU / focal-BCE / balances / backward)
through the C ABI against (a) golden vectors produced by the reference and (b) the oracle
on fresh seeded inputs.  Bars: assignment indices bit-exact, tar_box
bit-exact (pure +,-,*,/ fp32 chain), loss scalars and gradients within 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import v5loss as ov5
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_head_outputs, synth_targets

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _hyp(img, focal, dev, nc=80):
    return dict(device=dev, num_class=nc, input_img_size=[img, img], use_focal_loss=focal, focal_loss_gamma=1.5,
                focal_loss_alpha=0.25, iou_loss_scale=0.05, cls_loss_scale=0.5, cof_loss_scale=1.0, anchor_match_thr=4.0,
                class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0)


@pytest.mark.parametrize("key", ["hand", "synth"])
def test_assign_bit_exact(dev, key):
    from yoloseries_amd.loss import YOLOV5Loss
    g = np.load(os.path.join(G, "g2_match.npz"))
    if key == "hand":
        targets = g["targets"]
    else:
        b, img, nc, mb, seed = g["synth_args"]
        targets = synth_targets(int(b), int(img), int(nc), int(mb), seed=int(seed))
    lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), _hyp(640, True, dev))
    outs = lf.assign(torch.from_numpy(targets).to(dev), [(80, 80), (40, 40), (20, 20)])
    for s in range(3):
        tbox, cls, img_i, anc, gy, gx = outs[s]
        for name, val in (("cls", cls), ("img", img_i), ("anc", anc), ("gy", gy), ("gx", gx)):
            np.testing.assert_array_equal(val.cpu().numpy(), g[f"{key}_s{s}_{name}"])
        np.testing.assert_array_equal(tbox.cpu().numpy(), g[f"{key}_s{s}_tbox"])


def test_match_reference_signature(dev):
    """YOLOV5Loss.match with the reference's own argument convention (normalised targets)."""
    from oracle.bbox import xyxy2xywhn
    from yoloseries_amd.loss import YOLOV5Loss
    g = np.load(os.path.join(G, "g2_match.npz"))
    t = g["targets"].copy()
    t[..., :4] = xyxy2xywhn(t[..., :4], [640, 640])
    tt = np.concatenate([np.broadcast_to(t[None], (3,) + t.shape), np.broadcast_to(np.arange(3, dtype=np.float32)[:, None, None, None], (3,) + t.shape[:2] + (1,))], -1)
    lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), _hyp(640, True, dev))
    for s, fm in enumerate((80, 40, 20)):
        ds = np.float32(640) / np.float32(fm)
        anchor_stage = torch.from_numpy(COCO_ANCHORS[s] / ds).to(dev)
        outs = lf.match(torch.from_numpy(np.ascontiguousarray(tt)).to(dev), anchor_stage, (fm, fm))
        for name, val in zip(("tbox", "cls", "img", "anc", "gy", "gx"), outs):
            np.testing.assert_array_equal(val.cpu().numpy(), g[f"hand_s{s}_{name}"])


@pytest.mark.parametrize("key", ["small_focal", "small_plain", "big_focal"])
def test_loss_golden(dev, key):
    from yoloseries_amd.loss import YOLOV5Loss
    g = np.load(os.path.join(G, "g3_loss.npz"))
    img, batch, focal, seed_t, seed_p, ncalls, pscale = g[f"{key}_args"]
    img, batch, ncalls = int(img), int(batch), int(ncalls)
    lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), _hyp(img, bool(focal), dev))
    for call in range(ncalls):
        t = synth_targets(batch, img, 80, 6 if img < 640 else 20, seed=int(seed_t) + call)
        heads = synth_head_outputs(batch, img, 80, 3, seed=int(seed_p) + call, scale=float(pscale))
        preds = [torch.from_numpy(h).to(dev).requires_grad_(True) for h in heads]
        out = lf(preds, torch.from_numpy(t).to(dev))
        vals = g[f"{key}_c{call}_vals"]
        got = np.array([out["tot_loss"].item(), out["iou_loss"], out["cof_loss"], out["cls_loss"], out["tar_nums"]])
        assert got[4] == vals[4]
        np.testing.assert_allclose(got[:4], vals[:4], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(lf.balances, g[f"{key}_c{call}_balances"], rtol=1e-5)
        grads = torch.autograd.grad(out["tot_loss"], preds)
        for s, gr in enumerate(grads):
            gn = gr.cpu().numpy()
            if f"{key}_c{call}_grad{s}" in g:
                ref = g[f"{key}_c{call}_grad{s}"]
                np.testing.assert_allclose(gn, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
            else:
                flat = gn.reshape(-1)
                ref = g[f"{key}_c{call}_gval{s}"]
                np.testing.assert_allclose(flat[g[f"{key}_c{call}_gidx{s}"]], ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
                np.testing.assert_allclose([flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum()],
                                           g[f"{key}_c{call}_gsum{s}"], rtol=1e-4)


@pytest.mark.parametrize("focal", [True, False])
def test_loss_vs_oracle_bf16_layout(dev, focal):
    """bf16 cell-major predictions (the layout the HIP model emits, ld=256) with duplicate
    positives per cell (many boxes, small maps); oracle evaluated on the same bf16 values."""
    from yoloseries_amd.loss import YOLOV5Loss
    img, B = 128, 4
    t = synth_targets(B, img, 80, 24, seed=5, min_boxes=12)
    heads = synth_head_outputs(B, img, 80, 3, seed=6, scale=1.5)
    heads_bf = [torch.from_numpy(h).to(torch.bfloat16) for h in heads]
    hyp = _hyp(img, focal, dev)
    lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), hyp)
    bufs, preds = [], []
    for h in heads_bf:
        Bn, Ct, hh, ww = h.shape
        buf = torch.zeros(Bn, hh, ww, 256, dtype=torch.bfloat16, device=dev)
        buf[..., :Ct] = h.to(dev).permute(0, 2, 3, 1)
        bufs.append(buf)
        preds.append(buf.as_strided((Bn, Ct, hh, ww), (hh * ww * 256, 1, ww * 256, 256)).requires_grad_(True))
    out = lf(preds, torch.from_numpy(t).to(dev))
    grads = torch.autograd.grad(out["tot_loss"], preds)
    ohyp = dict(hyp); ohyp["device"] = "cpu"
    of = ov5.V5LossOracle(COCO_ANCHORS, ohyp)
    opreds = [h.float().requires_grad_(True) for h in heads_bf]
    oout = of(opreds, t)
    ograds = torch.autograd.grad(oout["tot_loss"], opreds)
    assert out["tar_nums"] == oout["tar_nums"]
    np.testing.assert_allclose([out["tot_loss"].item(), out["iou_loss"], out["cof_loss"], out["cls_loss"]],
                               [oout["tot_loss"].item(), oout["iou_loss"], oout["cof_loss"], oout["cls_loss"]], rtol=1e-4)
    np.testing.assert_allclose(lf.balances, of.balances, rtol=1e-5)
    for gr, og in zip(grads, ograds):
        ref = og.numpy()
        # gradient is stored in bf16: half-ulp relative 2^-9 plus absolute floor
        np.testing.assert_allclose(gr.float().cpu().numpy(), ref, rtol=6e-3, atol=1e-3 * np.abs(ref).max())
    # the padding column (channel 255) of the gradient buffers must be zero: the head's data / weight gradient kernels
    # read all ld = 256 columns of every cell
    for gr in grads:
        Bn, Ct, hh, ww = gr.shape
        ld = gr.stride(3)
        assert ld == 256 and Ct == 255
        whole = gr.as_strided((Bn, hh, ww, ld), (hh * ww * ld, ww * ld, ld, 1))
        assert (whole[..., Ct:] == 0).all()


def test_ciou_and_iou_utils(dev):
    from yoloseries_amd import utils as U
    g = np.load(os.path.join(G, "g1_boxes.npz"))
    b1 = torch.from_numpy(g["b1"]).to(dev).requires_grad_(True)
    b2 = torch.from_numpy(g["b2"]).to(dev)
    c = U.gpu_CIoU(b1, b2)
    np.testing.assert_allclose(c.detach().cpu().numpy(), g["ciou"], rtol=0, atol=1e-5)
    (gr,) = torch.autograd.grad(c.sum(), b1)
    np.testing.assert_allclose(gr.cpu().numpy(), g["ciou_grad_b1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(U.gpu_Giou(b1.detach(), b2).cpu().numpy(), g["giou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(U.gpu_DIoU(b1.detach(), b2).cpu().numpy(), g["diou"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(U.gpu_iou(b1.detach()[:96], b2[:80]).cpu().numpy(), g["iou_mat"])
    np.testing.assert_array_equal(U.numba_iou(g["b1"][480:], g["b2"][470:]), g["numba_iou_mat"])
    np.testing.assert_array_equal(U.xyxy2xywh(b2).cpu().numpy(), g["xyxy2xywh"])
    np.testing.assert_array_equal(U.xywh2xyxy(b2).cpu().numpy(), g["xywh2xyxy"])
    np.testing.assert_array_equal(U.xyxy2xywhn(b2, [640, 640]).cpu().numpy(), g["xyxy2xywhn"])
