"""CPU: the oracle (oracle/*.py) against the golden vectors generated from the reference
(tools/gen_golden.py).  Integer/index outputs must be bit-exact; float outputs of pure
+,-,*,/ chains must be bit-exact too; transcendental chains (sigmoid/atan/exp) within 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import bbox, postproc, v5loss
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_head_outputs, synth_targets

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_g1_boxes():
    g = load("g1_boxes.npz")
    b1, b2 = g["b1"], g["b2"]
    np.testing.assert_array_equal(bbox.gpu_iou(b1[:96], b2[:80]), g["iou_mat"])
    np.testing.assert_array_equal(bbox.numba_iou(b1[480:], b2[470:]), g["numba_iou_mat"])   # NaN == NaN positions too
    np.testing.assert_array_equal(bbox.gpu_giou(b1, b2), g["giou"])
    np.testing.assert_array_equal(bbox.gpu_diou(b1, b2), g["diou"])
    np.testing.assert_allclose(bbox.gpu_ciou(b1, b2), g["ciou"], rtol=0, atol=2e-6)     # atan: libm vs torch
    np.testing.assert_array_equal(bbox.xyxy2xywh(b2), g["xyxy2xywh"])
    np.testing.assert_array_equal(bbox.xywh2xyxy(b2), g["xywh2xyxy"])
    np.testing.assert_array_equal(bbox.xyxy2xywhn(b2, [640, 640]), g["xyxy2xywhn"])
    # torch restatement of CIoU and its gradient
    t1 = torch.from_numpy(b1).requires_grad_(True)
    c = v5loss._ciou_t(t1, torch.from_numpy(b2))
    (gr,) = torch.autograd.grad(c.sum(), t1)
    np.testing.assert_allclose(c.detach().numpy(), g["ciou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(gr.numpy(), g["ciou_grad_b1"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("key", ["hand", "synth"])
def test_g2_match(key):
    g = load("g2_match.npz")
    if key == "hand":
        targets = g["targets"]
    else:
        b, img, nc, mb, seed = g["synth_args"]
        targets = synth_targets(int(b), int(img), int(nc), int(mb), seed=int(seed))
    for s, fm in enumerate((80, 40, 20)):
        tbox, cls, img_i, anc, gy, gx = v5loss.match(targets, COCO_ANCHORS[s], fm, fm, [640, 640], 4.0)
        assert len(tbox) == len(g[f"{key}_s{s}_tbox"]) and len(tbox) > 0
        for name, val in (("cls", cls), ("img", img_i), ("anc", anc), ("gy", gy), ("gx", gx)):
            np.testing.assert_array_equal(val, g[f"{key}_s{s}_{name}"])
        np.testing.assert_array_equal(tbox, g[f"{key}_s{s}_tbox"])


def _hyp(img, focal, nc=80):
    return dict(device="cpu", num_class=nc, input_img_size=[img, img], use_focal_loss=focal, focal_loss_gamma=1.5,
                focal_loss_alpha=0.25, iou_loss_scale=0.05, cls_loss_scale=0.5, cof_loss_scale=1.0, anchor_match_thr=4.0,
                class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0)


@pytest.mark.parametrize("key", ["small_focal", "small_plain", "big_focal"])
def test_g3_loss(key):
    g = load("g3_loss.npz")
    img, batch, focal, seed_t, seed_p, ncalls, pscale = g[f"{key}_args"]
    img, batch, ncalls = int(img), int(batch), int(ncalls)
    lf = v5loss.V5LossOracle(COCO_ANCHORS, _hyp(img, bool(focal)))
    for call in range(ncalls):
        t = synth_targets(batch, img, 80, 6 if img < 640 else 20, seed=int(seed_t) + call)
        heads = synth_head_outputs(batch, img, 80, 3, seed=int(seed_p) + call, scale=float(pscale))
        preds = [torch.from_numpy(h).requires_grad_(True) for h in heads]
        out = lf(preds, t)
        vals = g[f"{key}_c{call}_vals"]
        got = np.array([out["tot_loss"].item(), out["iou_loss"], out["cof_loss"], out["cls_loss"], out["tar_nums"]])
        np.testing.assert_allclose(got, vals, rtol=2e-6, atol=1e-7)
        assert got[4] == vals[4]
        np.testing.assert_allclose(lf.balances, g[f"{key}_c{call}_balances"], rtol=1e-7)
        grads = torch.autograd.grad(out["tot_loss"], preds)
        for s, gr in enumerate(grads):
            if f"{key}_c{call}_grad{s}" in g:
                np.testing.assert_allclose(gr.numpy(), g[f"{key}_c{call}_grad{s}"], rtol=1e-4, atol=1e-8)
            else:
                flat = gr.numpy().reshape(-1)
                np.testing.assert_allclose(flat[g[f"{key}_c{call}_gidx{s}"]], g[f"{key}_c{call}_gval{s}"], rtol=1e-4, atol=1e-9)
                np.testing.assert_allclose([flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum()],
                                           g[f"{key}_c{call}_gsum{s}"], rtol=1e-5)


def test_g4_decode():
    g = load("g4_decode.npz")
    b, img, nc, a, seed, scale = g["args"]
    heads = synth_head_outputs(int(b), int(img), int(nc), int(a), seed=int(seed), scale=float(scale))
    dec = postproc.decode_v5(heads, COCO_ANCHORS, (8, 16, 32))
    np.testing.assert_allclose(dec, g["decoded"], rtol=2e-5, atol=1e-5)   # sigmoid: numpy vs torch CPU, contract is 1e-4


@pytest.mark.parametrize("key", ["std", "metric", "nonagn", "nopost", "cap", "tie", "zero", "empty"])
def test_g5_nms(key):
    g = load("g5_nms.npz")
    dec = g[f"{key}_dec"]
    metric, agn, post, maxp = (int(v) for v in g[f"{key}_cfg"])
    conf, cls_t, iou_t = (0.001, 0.001, 0.65) if metric else (0.3, 0.3, 0.2)
    outs = postproc.postprocess_v5(dec, conf, cls_t, iou_t, class_aware=bool(agn), max_keep=maxp, merge_filter=bool(post))
    ns = g[f"{key}_n"]
    assert [(-1 if o is None else len(o)) for o in outs] == list(ns)
    for i, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o, g[f"{key}_out{i}"])     # same rows in the same (pick) order, bit-exact


def test_g5_numba_nms_function():
    g = load("g5_nms.npz")
    keep = postproc.numba_nms(g["fn_boxes"], g["fn_scores"], 0.45)
    np.testing.assert_array_equal(np.array(keep), g["fn_numba_keep_0.45"])


# ------------------------------------------------------------------ YOLOX (g8)
def _hypx(img, focal, itype):
    return dict(device="cpu", num_class=80, input_img_size=[img, img], use_focal_loss=focal, focal_loss_gamma=1.5, focal_loss_alpha=0.25,
                iou_loss_scale=5.0, use_l1=True, l1_loss_scale=1.0, cls_loss_scale=1.0, cof_loss_scale=1.0, class_smooth_factor=1.0,
                cls_pos_weight=1.0, cof_pos_weight=1.0, num_anchors=1, iou_type=itype, topk=13, center_radius=3, num_stage=3)


@pytest.mark.parametrize("key", ["plain_ciou", "focal_giou", "plain_iou"])
def test_g8_yolox_loss(key):
    from oracle.yoloxloss import YOLOXLossOracle
    from yoloseries_amd.utils.synth import synth_yolox_heads
    g = load("g8_yolox.npz")
    img, batch, focal, seed = (int(v) for v in g[f"{key}_args"])
    lf = YOLOXLossOracle(_hypx(img, bool(focal), str(g[f"{key}_itype"])))
    for call in range(2):
        t = torch.from_numpy(synth_targets(batch, img, 80, 5, seed=seed + call, min_boxes=2))
        heads = synth_yolox_heads(batch, img, 80, seed=seed + 10 + call)
        preds = {k: torch.from_numpy(v).requires_grad_(True) for k, v in heads.items()}
        out = lf(preds, t)
        vals = g[f"{key}_c{call}_vals"]
        got = np.array([out["tot_loss"].item(), out["iou_loss"], out["l1_loss"], out["cls_loss"], out["cof_loss"], out["fg_nums"], out["tar_nums"]])
        assert got[5] == vals[5] and got[6] == vals[6]
        np.testing.assert_allclose(got[:5], vals[:5], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(lf.balances, g[f"{key}_c{call}_balances"], rtol=1e-7)
        np.testing.assert_array_equal(t.numpy(), g[f"{key}_c{call}_tars_after"])      # in-place xyxy -> xywh of the caller's tensor
        grads = torch.autograd.grad(out["tot_loss"], list(preds.values()))
        for s, gr in enumerate(grads):
            ref = g[f"{key}_c{call}_grad{s}"]
            np.testing.assert_allclose(gr.numpy(), ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())


def test_g8_yolox_assign():
    from oracle.bbox import xyxy2xywh
    from oracle.yoloxloss import YOLOXLossOracle
    from yoloseries_amd.utils.synth import synth_yolox_heads
    g = load("g8_yolox.npz")
    img, batch, seed_t, seed_p = (int(v) for v in g["assign_args"])
    lf = YOLOXLossOracle(_hypx(img, False, "ciou"))
    t = synth_targets(batch, img, 80, 5, seed=seed_t, min_boxes=2)
    t[..., :4] = xyxy2xywh(t[..., :4])
    heads = synth_yolox_heads(batch, img, 80, seed=seed_p)
    for s, (k, v) in enumerate(heads.items()):
        h, w = v.shape[-2:]
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
        grid = torch.stack((xs, ys), dim=2).float().reshape(-1, 2)
        p = torch.from_numpy(v).permute(0, 1, 3, 4, 2).contiguous().reshape(batch, h * w, -1)
        tb, tcof, tcls, tl1, fg, nfg, ngt = lf.label_assign(torch.from_numpy(t), p, grid, img / h)
        np.testing.assert_array_equal(fg.numpy(), g[f"assign_s{s}_fg"])
        np.testing.assert_array_equal(tb.numpy(), g[f"assign_s{s}_tbox"])
        np.testing.assert_allclose(tcls.numpy(), g[f"assign_s{s}_tcls"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(tl1.numpy(), g[f"assign_s{s}_tl1"], rtol=1e-6, atol=1e-7)
        assert [nfg, ngt] == list(g[f"assign_s{s}_n"])
        assert fg.sum() > 0


def test_g8_yolox_decode_and_nms():
    from yoloseries_amd.utils.synth import synth_yolox_heads
    g = load("g8_yolox.npz")
    b, img, nc, seed, scale = g["dec_args"]
    heads = synth_yolox_heads(int(b), int(img), int(nc), seed=int(seed), scale=float(scale))
    dec = postproc.decode_yolox(list(heads.values()), int(img))
    np.testing.assert_allclose(dec, g["decoded"], rtol=2e-5, atol=1e-5)
    d = g["nms_dec"]
    outs = []
    for i in range(d.shape[0]):
        cand = postproc.candidates_yolox(d[i], 0.3, 0.3)
        rows, _ = postproc.nms_image(cand, 0.2, True, 300, True)
        outs.append(rows)
    assert [(-1 if o is None else len(o)) for o in outs] == list(g["nms_n"])
    for i, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o, g[f"nms_out{i}"])


def test_g9_map_v2():
    """host-side metric mirror (yoloseries_amd/utils/mAP.py) against the reference's mAP_v2 on identical detections"""
    from yoloseries_amd.utils.mAP import mAP_v2
    g = load("g9_map.npz")
    n = int(g["n"])
    gts = [g[f"gt{i}"] for i in range(n)]
    preds = [g[f"pred{i}"] for i in range(n)]
    m = mAP_v2(gts, preds).compute_ap_per_class()
    np.testing.assert_array_equal(m["unique_cls"], g["unique_cls"])
    np.testing.assert_allclose(m["ap"], g["ap"], rtol=1e-12, atol=1e-12)
    for k in ("precision", "recall", "f1"):
        np.testing.assert_allclose(m[k], g[k], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mAP_v2(gts, preds).get_mean_metrics(), g["mean"], rtol=1e-12)
    assert g["mean"][1] > 0.1


# ---------------------------------------------------------------- G10 letterbox + collate format (host side)
def test_letterbox_and_collate_match_reference():
    """dataset/data_collater.py:20-64, utils/data_aug.py:21-70, utils/bbox_tools.py:38-49 on geometries the
    reference runs without OpenCV (scale == 1): resize_info, padded image, boxes and the (B,maxbox,6) tensor."""
    from yoloseries_amd.dataset import SyntheticDetectionDataset, fixed_imgsize_collate_fn, test_dataset_collate_fn
    from yoloseries_amd.utils.letterbox import letter_resize_bbox, letter_resize_img
    g = np.load(os.path.join(G, "g10_collate.npz"))
    length, nc, mb, seed0 = (int(v) for v in g["dataset_args"])
    items = []
    for i, hw in enumerate(g["shapes"]):
        ds = SyntheticDetectionDataset(length, img_hw=tuple(int(v) for v in hw), num_class=nc, max_boxes=mb, seed=seed0 + i)
        img, ann, iid = ds[i]
        if i == 3:
            ann = {'bboxes': np.zeros((0, 4), np.float32), 'classes': []}
        items.append((img, ann, iid))
        out, info = letter_resize_img(img, [640, 640])
        got = np.array([info['scale'], info['pad_top'], info['pad_left'], info['pad_bottom'], info['pad_right'],
                        info['org_shape'][0], info['org_shape'][1]], np.float64)
        assert np.array_equal(got, g[f"lr{i}_info"])
        assert out.dtype == np.uint8 and np.array_equal(out.astype(np.float64).sum(axis=(0, 1)), g[f"lr{i}_sum"])
        assert np.array_equal(out.reshape(-1)[::997], g[f"lr{i}_sample"])
        if len(ann['classes']):
            assert np.array_equal(letter_resize_bbox(np.array(ann['bboxes'], np.float64).copy(), info), g[f"lb{i}"])
    batch = fixed_imgsize_collate_fn(items, [640, 640])
    assert batch['img'].dtype == torch.float32 and batch['ann'].dtype == torch.float32
    assert np.array_equal(batch['ann'].numpy(), g["c_ann"])
    assert np.array_equal(batch['img'].double().sum(dim=(2, 3)).numpy(), g["c_img_sum"])
    assert np.array_equal(batch['img'].reshape(-1)[::9973].numpy(), g["c_img_sample"])
    info = np.array([[r['scale'], r['pad_top'], r['pad_left'], r['pad_bottom'], r['pad_right']] for r in batch['resize_info']], np.float64)
    assert np.array_equal(info, g["c_info"]) and np.array_equal(np.array(batch['img_id']), g["c_ids"])
    tb = test_dataset_collate_fn([(torch.full((3, 64, 96), float(k)), {'scale': 1.0 + k}) for k in range(3)])
    assert np.array_equal(tb['img'].double().sum(dim=(1, 2, 3)).numpy(), g["t_img_sum"]) and tb['resize_info'][2]['scale'] == 3.0


def test_letterbox_resize_paths():
    """scale != 1 (nearest-neighbour index rule of cv2.INTER_NEAREST) and the test-time minimal padding:
    geometry invariants of utils/data_aug.py:21-70 (OpenCV is absent, these paths have no reference vector)."""
    from yoloseries_amd.utils.letterbox import letter_resize_img, resize_nearest
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(375, 500, 3), dtype=np.uint8)
    out, info = letter_resize_img(img, 640)
    assert out.shape == (640, 640, 3) and abs(info['scale'] - 1.28) < 1e-12
    rh, rw = int(375 * 1.28), int(500 * 1.28)
    assert info['pad_top'] == (640 - rh) // 2 and info['pad_left'] == (640 - rw) // 2
    assert info['pad_top'] + info['pad_bottom'] + rh == 640 and info['pad_left'] + info['pad_right'] + rw == 640
    inner = out[info['pad_top']:info['pad_top'] + rh, info['pad_left']:info['pad_left'] + rw]
    ys = np.minimum(np.floor(np.arange(rh) * (375 / rh)).astype(int), 374)
    xs = np.minimum(np.floor(np.arange(rw) * (500 / rw)).astype(int), 499)
    assert np.array_equal(inner, img[ys][:, xs]) and np.array_equal(resize_nearest(img, rw, rh), inner)
    assert (out[:info['pad_top']] == 128).all() and (out[info['pad_top'] + rh:] == 128).all()
    # test time: pad only up to the next multiple of the stride
    out2, info2 = letter_resize_img(img, 640, training=False)
    assert out2.shape[0] % 64 == 0 and out2.shape[1] % 64 == 0 and out2.shape[1] == 640 and out2.shape[0] == 512
    assert info2['pad_top'] + info2['pad_bottom'] == 512 - rh and info2['pad_left'] + info2['pad_right'] == 0
    # only_ds never upsamples
    small = rs.randint(0, 256, size=(100, 200, 3), dtype=np.uint8)
    out3, info3 = letter_resize_img(small, 640, only_ds=True)
    assert info3['scale'] == 1.0 and out3.shape == (640, 640, 3)


def test_g12_multi_label_candidates():
    """hyp['mutil_label'] (trainer/eval_yolov5.py:276-279): one candidate per (prediction, class) — the oracle's rows and pick
    order equal the reference evaluator's, bit for bit"""
    g = load("g12_round3.npz")
    outs = postproc.postprocess_v5(g["ml_dec"], 0.3, 0.3, 0.2, class_aware=True, max_keep=300, merge_filter=True, multi_label=True)
    assert [(-1 if o is None else len(o)) for o in outs] == list(g["ml_n"])
    assert max(g["ml_n"]) > 40
    for i, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o, g[f"ml_out{i}"])


def test_g13_yolox_multi_label_candidates():
    """hyp['mutil_label'] of the YOLOX evaluator (trainer/eval_yolox.py:218-221): pre-filter obj * max(cls) >= conf, then one
    candidate per (prediction, class) — the oracle's rows and pick order equal the reference evaluator's, bit for bit"""
    g = load("g13_round4.npz")
    conf, cls, iou = (float(v) for v in g["mlx_thr"])
    outs = postproc.postprocess_yolox(g["mlx_dec"], conf, cls, iou, class_aware=True, max_keep=300, merge_filter=True, multi_label=True)
    assert [(-1 if o is None else len(o)) for o in outs] == list(g["mlx_n"])
    assert max(g["mlx_n"]) > 40 and min(g["mlx_n"]) == -1
    for i, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o, g[f"mlx_out{i}"])


def test_g14_evaluator_bbox_iou_and_do_nms_status():
    """the evaluators' unclamped bbox_iou (trainer/eval_yolov5.py:237-258, trainer/eval_yolox.py:177-199): the oracle's restatement
    against the reference's outputs, bit for bit (NaN where the reference has NaN); and what the reference's own do_nms does —
    None without candidates, IndexError with any (utils/nms.py:62-63) — which is why do_nms parity is anchored on the oracle's loop"""
    from oracle.bbox import evaluator_bbox_iou
    g = np.load(os.path.join(G, "g14_round5.npz"))
    got = evaluator_bbox_iou(g["iou_b1"], g["iou_b2"])
    for key in ("iou_v5", "iou_yolox"):
        ref = g[key]
        assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
    fin = g["iou_v5"][np.isfinite(g["iou_v5"])]
    assert np.isnan(g["iou_v5"]).any() and (fin < 0).any()          # the fixture exercises 0 / 0 and the negative-union quirk
    # boxes apart on both axes: a positive value where every clamped IoU is 0
    assert (g["iou_v5"][(bbox.numba_iou(g["iou_b1"], g["iou_b2"]) == 0)] > 0).any()
    assert list(g["donms_status"]) == [0, -1, -1]
    outs = postproc.do_nms_v5(g["donms_dec"], 0.3, 0.3, 0.2)
    assert outs[0] is None and len(outs[1]) == 1 and len(outs[2]) >= 1
