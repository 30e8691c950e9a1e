#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only where /root/reference exists (the build container); never on the GPU box.
The reference is imported read-only with stub modules for absent third-party
packages (numba -> identity decorator, loguru/cv2/emoji/seaborn -> no-ops), exactly
as SURVEY.md §8(c) describes.  Only inputs/outputs (data) are written — no reference
source.  Inputs that are large are described by RandomState seeds
(yoloseries_amd/utils/synth.py) instead of being stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def install_shims():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ident = lambda f=None, *a, **k: (f if callable(f) else (lambda g: g))  # noqa: E731
    stub("numba", njit=ident, jit=ident)

    class _L:
        def __getattr__(self, n):
            if n == "catch":
                return ident
            return lambda *a, **k: None
    stub("loguru", logger=_L())

    class _Sink(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Sink(n)

        def __call__(self, *a, **k):
            return None
    sys.modules["cv2"] = _Sink("cv2")
    stub("emoji", emojize=lambda s, *a, **k: s)
    stub("seaborn")


def make_hyp(num_class=80, img=640, focal=True, **kw):
    hyp = dict(device="cpu", num_class=num_class, input_img_size=[img, img],
               use_focal_loss=focal, focal_loss_gamma=1.5, focal_loss_alpha=0.25,
               iou_loss_scale=0.05, cls_loss_scale=0.5, cof_loss_scale=1.0, anchor_match_thr=4.0,
               class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0,
               iou_threshold=0.2, conf_threshold=0.3, cls_threshold=0.3, max_predictions_per_img=300,
               iou_type="iou", mutil_label=False, agnostic=True, postprocess_bbox=True, wfb=False,
               use_tta=False, half=False,
               compute_metric_conf_threshold=0.001, compute_metric_iou_threshold=0.65,
               compute_metric_cls_threshold=0.001)
    hyp.update(kw)
    return hyp


def main():
    if not os.path.isdir(REF):
        sys.exit("tools/gen_golden.py needs /root/reference (build container only)")
    sys.dont_write_bytecode = True
    install_shims()
    sys.path.insert(0, REF)
    sys.path.insert(1, ROOT)
    import torch
    import loss as ref_loss
    import models as ref_models
    import trainer as ref_trainer
    import utils as ref_utils
    if sys.argv[1:] == ["--only", "g10"]:
        gen_g10(ref_utils)
        return
    if sys.argv[1:] == ["--only", "g11"]:
        gen_g11(ref_utils, ref_models)
        return
    if sys.argv[1:] == ["--only", "g12"]:
        gen_g12(ref_models, ref_trainer)
        return
    if sys.argv[1:] == ["--only", "g13"]:
        gen_g13(ref_trainer)
        return
    if sys.argv[1:] == ["--only", "g14"]:
        gen_g14(ref_models, ref_trainer)
        return
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_head_outputs, synth_targets

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    anchors_t = torch.from_numpy(COCO_ANCHORS.copy())

    # ------------------------------------------------------------------ G1 boxes
    rs = np.random.RandomState(101)

    def rand_boxes(n, lo=0, hi=100):
        c = rs.uniform(lo, hi, (n, 2)); wh = np.exp(rs.uniform(np.log(1), np.log(60), (n, 2)))
        return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    b1 = rand_boxes(512); b2 = rand_boxes(512)
    b2[:64] = b1[:64] + rs.uniform(-2, 2, (64, 4)).astype(np.float32)        # strongly overlapping pairs
    b2[:64, 2:] = np.maximum(b2[:64, 2:], b2[:64, :2] + 0.5)
    deg1 = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [5, 5, 5, 5], [0, 0, 1, 1], [0, 0, 4, 8], [3, 3, 9, 9], [0, 0, 10, 10], [2, 2, 4, 4]] * 4, np.float32)
    deg2 = np.array([[0, 0, 10, 10], [20, 20, 30, 30], [5, 5, 5, 5], [1, 1, 2, 2], [0, 0, 8, 4], [3, 3, 9, 9], [10, 0, 20, 10], [0, 0, 10, 10]] * 4, np.float32)
    p1 = np.concatenate([b1, deg1]); p2 = np.concatenate([b2, deg2])
    t1 = torch.from_numpy(p1).requires_grad_(True); t2 = torch.from_numpy(p2)
    ciou = ref_utils.gpu_CIoU(t1, t2)
    (gc,) = torch.autograd.grad(ciou.sum(), t1)
    np.savez_compressed(os.path.join(OUT, "g1_boxes.npz"), b1=p1, b2=p2,
                        ciou=ciou.detach().numpy(), ciou_grad_b1=gc.numpy(),
                        giou=ref_utils.gpu_Giou(t1.detach(), t2).numpy(), diou=ref_utils.gpu_DIoU(t1.detach(), t2).numpy(),
                        iou_mat=ref_utils.gpu_iou(t1.detach()[:96], t2[:80]).numpy(),
                        numba_iou_mat=ref_utils.numba_iou(p1[480:], p2[470:]),
                        xyxy2xywh=ref_utils.xyxy2xywh(t2).numpy(), xywh2xyxy=ref_utils.xywh2xyxy(t2).numpy(),
                        xyxy2xywhn=ref_utils.xyxy2xywhn(t2, [640, 640]).numpy())

    # ------------------------------------------------------------------ G2 match
    hyp = make_hyp()
    lf = ref_loss.YOLOV5Loss(anchors_t, hyp)
    B, MB = 4, 12
    tg = -np.ones((B, MB, 6), np.float32)
    hand = [  # xmin ymin xmax ymax : borders, cell boundaries (multiples of 8/16/32), edges of the image
        [0, 0, 64, 64], [100, 120, 164, 200], [8, 8, 24, 24], [16, 16, 48, 80], [320, 320, 352, 352],
        [600, 600, 640, 640], [0, 300, 30, 420], [255.5, 127.5, 287.5, 159.5], [4, 4, 12, 12],
        [630, 10, 640, 30], [300, 0, 420, 20], [63.9, 63.9, 192.1, 192.1]]
    rs = np.random.RandomState(7)
    for b in range(B):
        n = [12, 7, 1, 0][b]
        for j in range(n):
            box = hand[(j + 3 * b) % len(hand)] if (j % 2 == 0) else None
            if box is None:
                c = rs.uniform(20, 620, 2); wh = np.exp(rs.uniform(np.log(6), np.log(400), 2))
                box = [max(c[0] - wh[0] / 2, 0), max(c[1] - wh[1] / 2, 0), min(c[0] + wh[0] / 2, 640), min(c[1] + wh[1] / 2, 640)]
            tg[b, j, :4] = box; tg[b, j, 4] = rs.randint(80); tg[b, j, 5] = b
    g2 = {"targets": tg}

    def run_match(targets_np, key):
        t = torch.from_numpy(targets_np.copy())
        bsz, mb = t.shape[:2]
        tt = t.clone()
        tt[..., :4] = ref_utils.xyxy2xywhn(tt[..., :4], hyp["input_img_size"])
        tt = tt.repeat(3, 1, 1, 1).contiguous()
        aid = torch.arange(3, dtype=torch.float32).reshape(-1, 1)[:, None, None, :].repeat(1, bsz, mb, 1)
        tt = torch.cat([tt, aid], -1).contiguous()
        for i, fm in enumerate((80, 40, 20)):
            ds = 640 / torch.tensor(fm)
            outs = lf.match(tt, anchors_t[i] / ds, (torch.tensor(fm), torch.tensor(fm)))
            for name, o in zip(("tbox", "cls", "img", "anc", "gy", "gx"), outs):
                g2[f"{key}_s{i}_{name}"] = o.numpy()
    run_match(tg, "hand")
    g2["synth_args"] = np.array([8, 640, 80, 20, 11])      # batch, img, nc, max_boxes, seed
    run_match(synth_targets(8, 640, 80, 20, seed=11), "synth")
    np.savez_compressed(os.path.join(OUT, "g2_match.npz"), **g2)

    # ------------------------------------------------------------------ G3 loss
    g3 = {}

    def run_loss(key, img, batch, focal, seed_t, seed_p, ncalls=1, store_full=True, pscale=1.0):
        hyp3 = make_hyp(img=img, focal=focal)
        lossf = ref_loss.YOLOV5Loss(anchors_t, hyp3)
        g3[f"{key}_args"] = np.array([img, batch, int(focal), seed_t, seed_p, ncalls, pscale], np.float64)
        for call in range(ncalls):
            tnp = synth_targets(batch, img, 80, 6 if img < 640 else 20, seed=seed_t + call)
            heads = synth_head_outputs(batch, img, 80, 3, seed=seed_p + call, scale=pscale)
            preds = [torch.from_numpy(h).requires_grad_(True) for h in heads]
            out = lossf(preds, torch.from_numpy(tnp))
            grads = torch.autograd.grad(out["tot_loss"], preds)
            g3[f"{key}_c{call}_vals"] = np.array([out["tot_loss"].item(), out["iou_loss"], out["cof_loss"], out["cls_loss"], out["tar_nums"]], np.float64)
            g3[f"{key}_c{call}_balances"] = np.array(lossf.balances, np.float64)
            for s, g in enumerate(grads):
                gn = g.numpy()
                if store_full:
                    g3[f"{key}_c{call}_grad{s}"] = gn
                else:
                    flat = gn.reshape(-1)
                    rsx = np.random.RandomState(1000 + s)
                    idx = np.unique(np.concatenate([rsx.randint(0, flat.size, 4096), np.argsort(-np.abs(flat))[:512]]))
                    g3[f"{key}_c{call}_gidx{s}"] = idx.astype(np.int64)
                    g3[f"{key}_c{call}_gval{s}"] = flat[idx]
                    g3[f"{key}_c{call}_gsum{s}"] = np.array([flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum()])
    run_loss("small_focal", 64, 2, True, 21, 31, ncalls=2)
    run_loss("small_plain", 64, 2, False, 22, 32, ncalls=1)
    run_loss("big_focal", 640, 2, True, 23, 33, ncalls=1, store_full=False)
    np.savez_compressed(os.path.join(OUT, "g3_loss.npz"), **g3)

    # ------------------------------------------------------------------ G4 decode
    hyp4 = make_hyp(img=64)
    heads = synth_head_outputs(2, 64, 80, 3, seed=41, scale=2.0)

    class StubYolo:
        def __call__(self, x):
            return [torch.from_numpy(h) for h in heads]
    ev = ref_trainer.YOLOV5Evaluator(StubYolo(), anchors_t, hyp4)
    dec = ev.do_inference(torch.zeros(2, 3, 64, 64))
    np.savez_compressed(os.path.join(OUT, "g4_decode.npz"), args=np.array([2, 64, 80, 3, 41, 2.0]), decoded=dec.numpy())

    # ------------------------------------------------------------------ G5 NMS
    g5 = {}
    nc5 = 4

    def clustered_decoded(seed, nimg, nclust, per, img=320, jitter=6.0, conf_lo=0.05, tie=False, zero_area=False):
        r = np.random.RandomState(seed)
        out = np.zeros((nimg, nclust * per + 40, 5 + nc5), np.float32)
        for b in range(nimg):
            rows = []
            for k in range(nclust):
                c = r.uniform(30, img - 30, 2); wh = r.uniform(20, 90, 2); cl = r.randint(nc5)
                for _ in range(per):
                    cc = c + r.uniform(-jitter, jitter, 2); ww = wh * r.uniform(0.8, 1.25, 2)
                    if tie:
                        cc = np.round(cc / 4) * 4; ww = np.round(ww / 8) * 8 + 8
                    cls = r.uniform(0.0, 0.2, nc5); cls[cl] = r.uniform(0.5, 1.0)
                    rows.append(np.concatenate([cc, ww, [r.uniform(conf_lo, 1.0)], cls]))
            for _ in range(40):   # background rows, mostly below conf thresholds
                rows.append(np.concatenate([r.uniform(0, img, 2), r.uniform(5, 50, 2), [r.uniform(0, 0.02)], r.uniform(0, 0.3, nc5)]))
            arr = np.array(rows, np.float32)
            if tie:   # exact score ties: quantise conf and class scores
                arr[:, 4] = np.round(arr[:, 4] * 8) / 8
                arr[:, 5:] = np.round(arr[:, 5:] * 8) / 8
            if zero_area:
                arr[::7, 2] = 0.0          # zero-width boxes -> 0/0 IoU with themselves
            out[b] = arr[r.permutation(len(arr))]
        return out

    def run_nms(key, dec, compute_metric, agnostic, post, max_pred=300):
        h = make_hyp(num_class=nc5, img=320, agnostic=agnostic, postprocess_bbox=post, max_predictions_per_img=max_pred)
        e = ref_trainer.YOLOV5Evaluator(None, anchors_t, h, compute_metric=compute_metric)
        res = e.numba_nms(torch.from_numpy(dec.copy()))
        g5[f"{key}_dec"] = dec
        g5[f"{key}_cfg"] = np.array([int(compute_metric), int(agnostic), int(post), max_pred])
        g5[f"{key}_n"] = np.array([-1 if r is None else len(r) for r in res])
        for i, r in enumerate(res):
            if r is not None:
                g5[f"{key}_out{i}"] = np.asarray(r, np.float32)
    run_nms("std", clustered_decoded(51, 3, 12, 10), False, True, True)
    run_nms("metric", clustered_decoded(52, 2, 10, 12, conf_lo=0.0005), True, True, True)
    run_nms("nonagn", clustered_decoded(53, 2, 10, 10), False, False, True)
    run_nms("nopost", clustered_decoded(54, 2, 10, 10), False, True, False)
    run_nms("cap", clustered_decoded(55, 1, 120, 4, jitter=1.0), True, True, False, max_pred=50)
    run_nms("tie", clustered_decoded(56, 2, 8, 12, tie=True), False, True, True)
    run_nms("zero", clustered_decoded(57, 2, 8, 8, zero_area=True), False, True, True)
    empty = clustered_decoded(58, 2, 4, 4); empty[0, :, 4] = 0.0
    run_nms("empty", empty, False, True, True)
    # function-level numba_nms / gpu_nms
    r = np.random.RandomState(59)
    fb = rand_boxes(300, 0, 200); fs = r.uniform(0, 1, 300).astype(np.float32); fs[::11] = 0.0
    g5["fn_boxes"], g5["fn_scores"] = fb, fs
    g5["fn_numba_keep_0.45"] = np.array(ref_utils.numba_nms(fb, fs, 0.45))
    # utils.gpu_nms (utils/nms.py:30-65) raises IndexError for M > 1 in the reference (mask of shape
    # (1,M) indexes a 1-D score tensor, :62-63), so it has no golden vector; see DESIGN.md.
    np.savez_compressed(os.path.join(OUT, "g5_nms.npz"), **g5)

    # ------------------------------------------------------------------ G6 blocks + G8 fuse
    g6 = {}

    def fill_state(mod, seed):
        r = np.random.RandomState(seed)
        sd = mod.state_dict()
        for k2, v in sd.items():
            if k2.endswith("num_batches_tracked"):
                continue
            shape = tuple(v.shape)
            if k2.endswith("running_var"):
                a = r.uniform(0.5, 1.5, shape)
            elif k2.endswith("bn.weight"):
                a = r.uniform(0.7, 1.3, shape)
            elif k2.endswith(("running_mean", "bn.bias", ".bias")):
                a = r.randn(*shape) * 0.2
            else:
                fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
                a = r.randn(*shape) / np.sqrt(fan_in)
            sd[k2] = torch.from_numpy(a.astype(np.float32))
        mod.load_state_dict(sd)

    def run_block(key, mod, cin, seed, hw=8, multi=False):
        fill_state(mod, seed)
        r = np.random.RandomState(seed + 1)
        x = (r.randn(2, cin, hw, hw)).astype(np.float32)
        g6[f"{key}_args"] = np.array([seed, cin, hw])
        mod.eval()
        with torch.no_grad():
            g6[f"{key}_eval"] = mod(torch.from_numpy(x.copy())).numpy()
        mod.train()
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        y = mod(xt)
        go = torch.from_numpy(r.randn(*y.shape).astype(np.float32))
        params = [p for p in mod.parameters()]
        grads = torch.autograd.grad(y, [xt] + params, go)
        g6[f"{key}_train"] = y.detach().numpy()
        g6[f"{key}_gout"] = go.numpy()
        g6[f"{key}_gx"] = grads[0].numpy()
        for (n, _), g in zip(mod.named_parameters(), grads[1:]):
            gf = g.double().reshape(-1)     # signature instead of the full tensor keeps the fixture small
            g6[f"{key}_gp_{n}"] = np.concatenate([[gf.sum().item(), gf.abs().sum().item(), gf.norm().item()], gf[:29].numpy()])
        for n, b in mod.named_buffers():
            g6[f"{key}_buf_{n}"] = b.numpy().copy()
    run_block("cba1x1", ref_utils.ConvBnAct(32, 64, 1, 1), 32, 601)
    run_block("cba3x3s2", ref_utils.ConvBnAct(32, 64, 3, 2, 1), 32, 602)
    run_block("cba6x6s2", ref_utils.ConvBnAct(3, 32, 6, 2, 2), 3, 603, hw=16)
    run_block("bneck", ref_utils.BasicBottleneck(32, 32, True, expand_ratio=1.0), 32, 604)
    run_block("c3", ref_utils.C3BottleneckCSP(64, 64, shortcut=True, num_block=2), 64, 605)
    run_block("c3ns", ref_utils.C3BottleneckCSP(128, 64, shortcut=False, num_block=1), 128, 606)
    run_block("sppf", ref_utils.FastSPP(64, 64), 64, 607)
    conv = torch.nn.Conv2d(16, 32, 3, 1, 1, bias=False); bn = torch.nn.BatchNorm2d(32, eps=1e-3)
    cb = ref_utils.ConvBnAct(16, 32, 3, 1, 1); fill_state(cb, 608)
    fused = ref_utils.fuse_conv_bn(cb.conv, cb.bn)
    g6["fuse_args"] = np.array([608]); g6["fuse_w"] = fused.weight.detach().numpy(); g6["fuse_b"] = fused.bias.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "g6_blocks.npz"), **g6)

    # ------------------------------------------------------------------ G7 full model (seeded init)
    g7 = {}
    for name, cls in (("s", ref_models.YOLOV5Small), ("l", ref_models.YOLOV5Large)):
        torch.manual_seed(0)
        m = cls(3, 80)
        sd = m.state_dict()
        g7[f"{name}_keys"] = np.array(list(sd.keys()))
        g7[f"{name}_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        g7[f"{name}_psum"] = np.array([v.double().sum().item() for v in sd.values()])
        g7[f"{name}_pabs"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        if name == "s":
            x = torch.from_numpy(np.random.RandomState(70).rand(2, 3, 64, 64).astype(np.float32))
            m.eval()
            with torch.no_grad():
                outs = m(x)
            for i, o in enumerate(outs):
                g7[f"s_eval64_out{i}"] = o.numpy()
            # train mode at 256x256 (>= 128 samples per channel for every BatchNorm; at 64x64 the deepest
            # stage has 8 samples per channel, which makes bf16-vs-fp32 comparisons ill-conditioned)
            m.train()
            x2 = torch.from_numpy(np.random.RandomState(71).rand(2, 3, 256, 256).astype(np.float32))
            outs = m(x2)
            for i, o in enumerate(outs):
                flat = o.detach().numpy().reshape(-1)
                idx = np.random.RandomState(72 + i).randint(0, flat.size, 4096)
                g7[f"s_train256_idx{i}"] = idx.astype(np.int64)
                g7[f"s_train256_val{i}"] = flat[idx]
                g7[f"s_train256_shape{i}"] = np.array(o.shape)
            g7["s_train256_rm_focus"] = m.focus.bn.running_mean.numpy().copy()
            g7["s_train256_rv_focus"] = m.focus.bn.running_var.numpy().copy()
            g7["s_train256_rv_last"] = m.head_stage4_bscp.cba3.bn.running_var.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g7_model.npz"), **g7)
    # ------------------------------------------------------------------ G8 YOLOX (model init, loss, decode, NMS)
    g8 = {}
    from yoloseries_amd.utils.synth import synth_yolox_heads
    hypx = make_hyp(img=128, focal=False, iou_loss_scale=5.0, use_l1=True, l1_loss_scale=1.0, cls_loss_scale=1.0, cof_loss_scale=1.0,
                    num_anchors=1, iou_type="ciou", topk=13, center_radius=3, num_stage=3)
    for key, focal, itype, seed in (("plain_ciou", False, "ciou", 81), ("focal_giou", True, "giou", 83), ("plain_iou", False, "iou", 85)):
        hx = dict(hypx); hx["use_focal_loss"] = focal; hx["iou_type"] = itype
        # the reference's select_grid falls back to torch.randperm when no cell centre lies in any gt box
        # (yolox_loss.py:270-278); such inputs are not reproducible, so pick the first seed that never gets there
        _orig_randperm = torch.randperm
        while True:
            hits = []
            torch.randperm = lambda *a, **k: (hits.append(1), _orig_randperm(*a, **k))[1]
            lx_probe = ref_loss.YOLOXLoss(hx)
            for call in range(2):
                lx_probe({k: torch.from_numpy(v) for k, v in synth_yolox_heads(2, 128, 80, seed=seed + 10 + call).items()},
                         torch.from_numpy(synth_targets(2, 128, 80, 5, seed=seed + call, min_boxes=2)))
            torch.randperm = _orig_randperm
            # also skip inputs where an exact tie in the SimOTA cost (e.g. a gt with zero IoU to every candidate)
            # decides the assignment: torch.topk's choice among equal costs is an implementation accident
            from oracle.yoloxloss import YOLOXLossOracle
            tie_free = not hits
            for call in range(2 if not hits else 0):
                o1, o2 = YOLOXLossOracle(dict(hx)), YOLOXLossOracle(dict(hx), stable_ties=True)
                for o in (o1, o2):
                    o({k: torch.from_numpy(v) for k, v in synth_yolox_heads(2, 128, 80, seed=seed + 10 + call).items()},
                      torch.from_numpy(synth_targets(2, 128, 80, 5, seed=seed + call, min_boxes=2)))
                tie_free &= all(bool((a == b).all()) for a, b in zip(o1.last_fg, o2.last_fg))
            if not hits and tie_free:
                break
            seed += 100
        lx = ref_loss.YOLOXLoss(hx)
        g8[f"{key}_args"] = np.array([128, 2, int(focal), seed], np.float64)
        g8[f"{key}_itype"] = np.array(itype)
        for call in range(2):
            tnp = synth_targets(2, 128, 80, 5, seed=seed + call, min_boxes=2)
            heads = synth_yolox_heads(2, 128, 80, seed=seed + 10 + call)
            preds = {k: torch.from_numpy(v).requires_grad_(True) for k, v in heads.items()}
            tt = torch.from_numpy(tnp.copy())
            out = lx(preds, tt)
            grads = torch.autograd.grad(out["tot_loss"], list(preds.values()))
            g8[f"{key}_c{call}_vals"] = np.array([out["tot_loss"].item(), out["iou_loss"], out["l1_loss"], out["cls_loss"], out["cof_loss"],
                                                  out["fg_nums"], out["tar_nums"]], np.float64)
            g8[f"{key}_c{call}_balances"] = np.array(lx.balances, np.float64)
            g8[f"{key}_c{call}_tars_after"] = tt.numpy()
            for s, gk in enumerate(preds.keys()):
                g8[f"{key}_c{call}_grad{s}"] = grads[s].numpy()
    # assignment pinned separately: foreground masks of one label_assign call per stage
    lx = ref_loss.YOLOXLoss(hypx)
    tnp = synth_targets(2, 128, 80, 5, seed=91, min_boxes=2)
    heads = synth_yolox_heads(2, 128, 80, seed=92)
    tt = torch.from_numpy(tnp.copy()); tt[..., :4] = ref_utils.xyxy2xywh(tt[..., :4])
    for s, (k, v) in enumerate(heads.items()):
        h, w = v.shape[-2:]
        stride = 128 / h
        grid = lx._make_grid(h, w, tt.type()).unsqueeze(0).expand(1, -1, -1, -1).reshape(-1, 2)
        p = torch.from_numpy(v).permute(0, 1, 3, 4, 2).contiguous().reshape(2, h * w, -1)
        tb, tcof, tcls, tl1, fg, nfg, ngt = lx.label_assign(tt.float(), p.float(), grid.float(), stride)
        g8[f"assign_s{s}_fg"] = fg.numpy(); g8[f"assign_s{s}_tbox"] = tb.numpy(); g8[f"assign_s{s}_tcls"] = tcls.detach().numpy()
        g8[f"assign_s{s}_tl1"] = tl1.numpy(); g8[f"assign_s{s}_n"] = np.array([nfg, int(ngt)])
    g8["assign_args"] = np.array([128, 2, 91, 92])
    # model: state_dict keys + seeded init + eval forward
    torch.manual_seed(0)
    mx = ref_models.YOLOXSmall(1, 3, 80, 0.01)
    sdx = mx.state_dict()
    g8["m_keys"] = np.array(list(sdx.keys())); g8["m_shapes"] = np.array([str(tuple(v.shape)) for v in sdx.values()])
    g8["m_psum"] = np.array([v.double().sum().item() for v in sdx.values()]); g8["m_pabs"] = np.array([v.double().abs().sum().item() for v in sdx.values()])
    xx = torch.from_numpy(np.random.RandomState(93).rand(2, 3, 64, 64).astype(np.float32))
    mx.eval()
    with torch.no_grad():
        oo = mx(xx)
    for s, (k, v) in enumerate(oo.items()):
        g8[f"m_eval64_out{s}"] = v.numpy()
    # evaluator: decode + post-processing
    hypx2 = make_hyp(img=64, num_anchors=1, num_stage=3)
    heads = synth_yolox_heads(2, 64, 80, seed=94, scale=2.0)

    class StubX:
        def __call__(self, x):
            from collections import OrderedDict
            return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in heads.items())
    evx = ref_trainer.YOLOXEvaluator(StubX(), hypx2)
    decx = evx.do_inference(torch.zeros(2, 3, 64, 64))
    g8["dec_args"] = np.array([2, 64, 80, 94, 2.0]); g8["decoded"] = decx.numpy()
    nmsd = clustered_decoded(95, 2, 10, 10)
    hx3 = make_hyp(num_class=nc5, img=320, num_anchors=1, num_stage=3)
    evn = ref_trainer.YOLOXEvaluator(None, hx3)
    res = evn.numba_nms(torch.from_numpy(nmsd.copy()))
    g8["nms_dec"] = nmsd; g8["nms_n"] = np.array([-1 if r is None else len(r) for r in res])
    for i, r in enumerate(res):
        if r is not None:
            g8[f"nms_out{i}"] = np.asarray(r, np.float32)
    np.savez_compressed(os.path.join(OUT, "g8_yolox.npz"), **g8)

    # ------------------------------------------------------------------ G9 mAP_v2
    import tempfile
    from utils.mAP import mAP_v2 as ref_map
    rs9 = np.random.RandomState(97)
    gts, preds = [], []
    for i in range(12):
        n = rs9.randint(0, 7)
        c = rs9.uniform(50, 590, (n, 2)); wh = rs9.uniform(20, 200, (n, 2))
        gt = np.concatenate([c - wh / 2, c + wh / 2, rs9.randint(0, 5, (n, 1))], 1).astype(np.float32)
        pr = []
        for row in gt:          # detections: jittered copies (some with wrong class) + false positives
            for _ in range(rs9.randint(0, 3)):
                box = row[:4] + rs9.uniform(-15, 15, 4)
                pr.append(np.concatenate([box, [rs9.uniform(0.05, 1.0)], [row[4] if rs9.rand() < 0.8 else rs9.randint(0, 5)]]))
        for _ in range(rs9.randint(0, 4)):
            c2 = rs9.uniform(50, 590, 2); w2 = rs9.uniform(20, 120, 2)
            pr.append(np.concatenate([c2 - w2 / 2, c2 + w2 / 2, [rs9.uniform(0.05, 0.6)], [rs9.randint(0, 5)]]))
        gts.append(gt); preds.append(np.array(pr, np.float32).reshape(-1, 6))
    with tempfile.TemporaryDirectory() as td:
        mm = ref_map(gts, preds, td)
        met = mm.compute_ap_per_class()
        mean = ref_map(gts, preds, td).get_mean_metrics()
    g9 = {"n": np.array(len(gts)), "ap": met["ap"], "precision": met["precision"], "recall": met["recall"], "f1": met["f1"],
          "unique_cls": met["unique_cls"], "mean": np.array(mean, np.float64)}
    for i, (a, b) in enumerate(zip(gts, preds)):
        g9[f"gt{i}"] = a; g9[f"pred{i}"] = b
    np.savez_compressed(os.path.join(OUT, "g9_map.npz"), **g9)

    gen_g10(ref_utils)
    gen_g11(ref_utils, ref_models)
    gen_g12(ref_models, ref_trainer)
    gen_g13(ref_trainer)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden written:", sorted(os.listdir(OUT)), f"{total / 1e6:.2f} MB")


def fill_state_rs(mod, seed):
    """state_dict filled from a NumPy RandomState (reproducible on the test side without torch's RNG): same rule as
    main()'s fill_state / tests/test_gpu_model.py::fill_state"""
    import torch
    r = np.random.RandomState(seed)
    sd = mod.state_dict()
    for k2, v in sd.items():
        if k2.endswith("num_batches_tracked"):
            continue
        shape = tuple(v.shape)
        if k2.endswith("running_var"):
            a = r.uniform(0.5, 1.5, shape)
        elif k2.endswith("bn.weight"):
            a = r.uniform(0.7, 1.3, shape)
        elif k2.endswith(("running_mean", "bn.bias", ".bias")):
            a = r.randn(*shape) * 0.2
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            a = r.randn(*shape) / np.sqrt(fan_in)
        sd[k2] = torch.from_numpy(a.astype(np.float32))
    mod.load_state_dict(sd)


def _dev_stats(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    lim = 0.06 * np.abs(b).max() + 0.06 * np.abs(b)
    return np.array([(np.abs(a - b) > lim).mean(), np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30)])


def _backward_sig(g, key, make, x, outs_of, with_corr=False):
    """full-model backward of a TRAINING-mode model (seeded default init; G11, G14): per-parameter gradient signatures + sampled
    input gradient, with the deviation of the reference's own bf16-autocast backward as calibration"""
    import torch

    def run(lowp):
        torch.manual_seed(0)
        m = make().train()
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        if lowp:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                outs = [o.float() for o in outs_of(m(xt))]
        else:
            outs = outs_of(m(xt))
        r = np.random.RandomState(1200 + len(key))
        gos = [torch.from_numpy((r.randn(*o.shape) * 0.1).astype(np.float32)) for o in outs]
        grads = torch.autograd.grad(outs, [xt] + list(m.parameters()), gos)
        return m, [o.shape for o in outs], [gr.detach().double().numpy().reshape(-1) for gr in grads]
    m, shapes, gr = run(False)
    _, _, gr_lo = run(True)
    for i, sh in enumerate(shapes):
        g[f"{key}_gout_shape{i}"] = np.array(sh)
    gx = gr[0]
    idx = np.random.RandomState(1300).randint(0, gx.size, 8192)
    g[f"{key}_gx_idx"] = idx.astype(np.int64)
    g[f"{key}_gx_val"] = gx[idx]
    g[f"{key}_gx_norm"] = np.array([np.sqrt((gx ** 2).sum())])
    g[f"{key}_gx_cal"] = _dev_stats(gr_lo[0][idx], gx[idx])
    names, sig, samp, cal, corr = [], [], [], [], []
    for pi, ((n, p), gf, gl) in enumerate(zip(m.named_parameters(), gr[1:], gr_lo[1:])):
        names.append(n)
        sig.append([gf.sum(), np.abs(gf).sum(), np.sqrt((gf ** 2).sum()), float(gf.size)])
        si = np.random.RandomState(1400 + pi).randint(0, gf.size, 64)
        samp.append(gf[si])
        corr.append(float(np.corrcoef(gl[si], gf[si])[0, 1]) if gf.size >= 64 and np.std(gf[si]) > 0 and np.std(gl[si]) > 0 else np.nan)
        nr = np.sqrt((gf ** 2).sum()) + 1e-30
        cal.append([abs(np.sqrt((gl ** 2).sum()) - nr) / nr, np.sqrt(((gl[si] - gf[si]) ** 2).mean()) / (np.sqrt((gf[si] ** 2).mean()) + 1e-30),
                    np.sqrt(((gl - gf) ** 2).sum()) / nr])
    g[f"{key}_pnames"] = np.array(names)
    g[f"{key}_psig"] = np.array(sig)
    g[f"{key}_psamp"] = np.array(samp)
    g[f"{key}_pcal"] = np.array(cal)
    if with_corr:           # correlation of the reference's own bf16-autocast samples with its fp32 ones, per parameter (G14)
        g[f"{key}_pcorr"] = np.array(corr)


def _frozen_bwd(g, key, make, seed, x, outs_of, nsamp=256):
    """full-graph gradients with BatchNorm in evaluation mode (G12, G14): output samples, per-parameter gradient signature +
    `nsamp` sampled elements, and the reference's own bf16-autocast deviation as calibration"""
    import torch
    m = make()
    fill_state_rs(m, seed)
    m.eval()
    xt = torch.from_numpy(x.copy())
    outs = outs_of(m(xt))
    r = np.random.RandomState(seed + 1)
    gos = [torch.from_numpy((r.randn(*o.shape) * 0.1).astype(np.float32)) for o in outs]
    grads = torch.autograd.grad(outs, list(m.parameters()), gos)
    g[f"{key}_seed"] = np.array([seed])
    for i, o in enumerate(outs):
        flat = o.detach().numpy().reshape(-1)
        idx = np.random.RandomState(seed + 10 + i).randint(0, flat.size, 4096)
        g[f"{key}_out_shape{i}"] = np.array(o.shape)
        g[f"{key}_out_idx{i}"], g[f"{key}_out_val{i}"] = idx.astype(np.int64), flat[idx]
    names, sig, sidx, sval = [], [], [], []
    for pi, ((n, p), gr) in enumerate(zip(m.named_parameters(), grads)):
        gf = gr.detach().double().numpy().reshape(-1)
        names.append(n)
        sig.append([gf.sum(), np.abs(gf).sum(), np.sqrt((gf ** 2).sum()), float(gf.size), np.abs(gf).max()])
        si = np.random.RandomState(seed + 100 + pi).randint(0, gf.size, nsamp)
        sidx.append(si)
        sval.append(gf[si])
    g[f"{key}_pnames"] = np.array(names)
    g[f"{key}_psig"] = np.array(sig)
    g[f"{key}_pidx"] = np.array(sidx, np.int64)
    g[f"{key}_pval"] = np.array(sval)
    # calibration: the reference ITSELF under torch bf16 autocast, same state and inputs — per parameter the largest sampled
    # element error relative to the largest gradient element, and the relative error of the norm
    m2 = make()
    fill_state_rs(m2, seed)
    m2.eval()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        outs2 = [o.float() for o in outs_of(m2(torch.from_numpy(x.copy())))]
    grads2 = torch.autograd.grad(outs2, list(m2.parameters()), gos)
    cal = []
    for pi, gr in enumerate(grads2):
        gl = gr.detach().double().numpy().reshape(-1)
        cal.append([np.abs(gl[sidx[pi]] - sval[pi]).max() / (sig[pi][4] + 1e-30), abs(np.sqrt((gl ** 2).sum()) - sig[pi][2]) / (sig[pi][2] + 1e-30)])
    g[f"{key}_pcal"] = np.array(cal)


def gen_g11(ref_utils, ref_models):
    """G11 (round 2): YOLOv5 m / l / x forward (BASELINE configs #4 / #5 widths and depths, models/normal/yolov5{m,l,x}.py),
    full-model backward of YOLOv5s and YOLOXs (train_yolov5.py:334-337: gradients of every parameter and of the input from
    a fixed output gradient), and the function-level NMS variants of utils/nms.py (gpu_nms with giou/diou/ciou, soft-NMS)."""
    import torch
    g = {}
    # ---- models: eval forward at 64^2 on a RandomState-filled state (running statistics: well conditioned), train forward
    # at 256^2 on the seeded default init.  A train-mode forward through 100-170 BatchNorm layers with 128 samples per
    # channel in the deepest stage amplifies bf16 rounding chaotically — the reference ITSELF under torch bf16 autocast
    # deviates from its fp32 run by 2-18 % of the elements outside a 6 % band for these depths — so the fixture also stores
    # that deviation (`*_cal_*`): the parity bar for the bf16 HIP path is "no further from the fp32 reference than the
    # reference's own bf16 run" (tests/test_gpu_model.py).
    def dev_stats(a, b):
        a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
        lim = 0.06 * np.abs(b).max() + 0.06 * np.abs(b)
        return np.array([(np.abs(a - b) > lim).mean(), np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30)])
    bn_probes = ["focus", "backbone_stage2_conv", "backbone_stage4_conv", "head_stage4_bscp.cba3"]
    for name, cls, seed in (("m", ref_models.YOLOV5Middle, 1101), ("l", ref_models.YOLOV5Large, 1102), ("x", ref_models.YOLOV5XLarge, 1103)):
        torch.manual_seed(0)
        m = cls(3, 80)
        sd = m.state_dict()
        g[f"{name}_keys"] = np.array(list(sd.keys()))
        g[f"{name}_psum"] = np.array([v.double().sum().item() for v in sd.values()])
        g[f"{name}_pabs"] = np.array([v.double().abs().sum().item() for v in sd.values()])
        x2 = torch.from_numpy(np.random.RandomState(seed + 11).rand(2, 3, 256, 256).astype(np.float32))
        m.train()
        with torch.no_grad():
            outs = m(x2)
        probes = {pn: (dict(m.named_modules())[pn].bn.running_mean.numpy().copy(), dict(m.named_modules())[pn].bn.running_var.numpy().copy())
                  for pn in bn_probes}
        torch.manual_seed(0)
        m2 = cls(3, 80).train()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            outs_lo = [o.float() for o in m2(x2)]
        for i, o in enumerate(outs):
            flat = o.numpy().reshape(-1)
            idx = np.random.RandomState(seed + 20 + i).randint(0, flat.size, 8192)
            g[f"{name}_train256_idx{i}"] = idx.astype(np.int64)
            g[f"{name}_train256_val{i}"] = flat[idx]
            g[f"{name}_train256_shape{i}"] = np.array(o.shape)
            g[f"{name}_train256_cal{i}"] = dev_stats(outs_lo[i].numpy().reshape(-1)[idx], flat[idx])
        for pn, (rm, rv) in probes.items():
            g[f"{name}_rm_{pn}"], g[f"{name}_rv_{pn}"] = rm, rv
            bn2 = dict(m2.named_modules())[pn].bn
            g[f"{name}_rvcal_{pn}"] = dev_stats(bn2.running_var.numpy(), rv)
        del m2
        fill_state_rs(m, seed)
        g[f"{name}_seed"] = np.array([seed])
        x = torch.from_numpy(np.random.RandomState(seed + 10).rand(2, 3, 64, 64).astype(np.float32))
        m.eval()
        with torch.no_grad():
            outs = m(x)
        for i, o in enumerate(outs):
            g[f"{name}_eval64_out{i}"] = o.numpy()
        del m

    backward_sig = lambda *a, **k: _backward_sig(g, *a, **k)        # noqa: E731
    xs = np.random.RandomState(1112).rand(2, 3, 256, 256).astype(np.float32)
    backward_sig("v5s_bwd", lambda: ref_models.YOLOV5Small(3, 80), xs, lambda o: list(o))
    xx = np.random.RandomState(1122).rand(2, 3, 256, 256).astype(np.float32)
    backward_sig("yolox_bwd", lambda: ref_models.YOLOXSmall(1, 3, 80, 0.01), xx, lambda o: list(o.values()))

    # ---- utils/nms.py function level: gpu_nms with the pairwise IoU kinds, soft-NMS
    rs = np.random.RandomState(1131)
    c = rs.uniform(20, 180, (12, 2)); wh = rs.uniform(20, 70, (12, 2))
    rows = []
    for k in range(12):
        for _ in range(10):
            cc = c[k] + rs.uniform(-8, 8, 2); ww = wh[k] * rs.uniform(0.8, 1.25, 2)
            rows.append(np.concatenate([cc - ww / 2, cc + ww / 2]))
    nb = np.array(rows, np.float32)
    ns = rs.uniform(0.05, 1.0, len(nb)).astype(np.float32); ns[::13] = 0.0
    g["nms_boxes"], g["nms_scores"] = nb, ns
    for kind in ("giou", "diou", "ciou"):
        g[f"nms_keep_{kind}_0.3"] = np.array(ref_utils.gpu_nms(torch.from_numpy(nb), torch.from_numpy(ns), kind, 0.3))
    sb, ss = nb[:40], ns[:40].reshape(-1, 1).copy()
    g["soft_boxes"], g["soft_scores"] = sb, ss
    for kind in ("giou", "diou", "ciou"):
        g[f"soft_linear_{kind}"] = ref_utils.gpu_linear_soft_nms(torch.from_numpy(sb), torch.from_numpy(ss.copy()), kind, 0.3, 0.001).numpy()
    eb, es = nb[:12], ns[:12].reshape(-1, 1).copy()
    g["softexp_boxes"], g["softexp_scores"] = eb, es
    g["soft_exp_giou"] = ref_utils.gpu_exponential_soft_nms(torch.from_numpy(eb), torch.from_numpy(es.copy()), "giou", 0.3, 0.5, 0.001).numpy()
    np.savez_compressed(os.path.join(OUT, "g11_round2.npz"), **g)
    print("g11 written", os.path.getsize(os.path.join(OUT, "g11_round2.npz")) / 1e6, "MB")


def gen_g12(ref_models, ref_trainer):
    """G12 (round 3): (a) a WELL-CONDITIONED full-graph gradient check — YOLOv5s and YOLOXs with BatchNorm in evaluation mode
    (model.eval(): running statistics, no batch coupling) under autograd, state filled from a RandomState: output samples and,
    for every parameter, the gradient's signature plus 256 sampled elements.  Isolates the concat / upsample / residual / stacked
    GEMM gradient wiring from the batch-statistics chaos of train mode, so the bar can be tight.
    (b) YOLOV5Evaluator.numba_nms with hyp['mutil_label'] = True (trainer/eval_yolov5.py:276-279)."""
    import torch
    g = {}

    frozen_bwd = lambda *a, **k: _frozen_bwd(g, *a, **k)            # noqa: E731
    xs = np.random.RandomState(1201).rand(2, 3, 256, 256).astype(np.float32)
    frozen_bwd("v5s_frozen", lambda: ref_models.YOLOV5Small(3, 80), 1210, xs, lambda o: list(o))
    xx = np.random.RandomState(1202).rand(2, 3, 256, 256).astype(np.float32)
    frozen_bwd("yolox_frozen", lambda: ref_models.YOLOXSmall(1, 3, 80, 0.01), 1220, xx, lambda o: list(o.values()))

    # ---- multi-label candidates: every class whose cls*obj reaches the threshold makes a row of its own
    from yoloseries_amd.utils.synth import COCO_ANCHORS
    nc = 6
    r = np.random.RandomState(1230)
    B, N = 3, 400
    dec = np.zeros((B, N, 5 + nc), np.float32)
    for b in range(B):
        c = r.uniform(30, 290, (12, 2)); wh = r.uniform(20, 80, (12, 2))
        for i in range(N):
            k = r.randint(12)
            cls = r.uniform(0.0, 0.25, nc)
            hot = r.choice(nc, size=r.randint(1, 4), replace=False)
            cls[hot] = r.uniform(0.45, 1.0, len(hot))
            dec[b, i] = np.concatenate([c[k] + r.uniform(-6, 6, 2), wh[k] * r.uniform(0.85, 1.2, 2), [r.uniform(0.05, 1.0)], cls])
    dec[2, :, 4] = 0.01                          # an image without candidates
    h = make_hyp(num_class=nc, img=320, mutil_label=True)
    e = ref_trainer.YOLOV5Evaluator(None, torch.from_numpy(COCO_ANCHORS.copy()), h, compute_metric=False)
    res = e.numba_nms(torch.from_numpy(dec.copy()))
    g["ml_dec"] = dec
    g["ml_n"] = np.array([-1 if q is None else len(q) for q in res])
    for i, q in enumerate(res):
        if q is not None:
            g[f"ml_out{i}"] = np.asarray(q, np.float32)
    np.savez_compressed(os.path.join(OUT, "g12_round3.npz"), **g)
    print("g12 written", os.path.getsize(os.path.join(OUT, "g12_round3.npz")) / 1e6, "MB", g["ml_n"])


def gen_g13(ref_trainer):
    """G13 (round 4): YOLOXEvaluator.numba_nms with hyp['mutil_label'] = True (trainer/eval_yolox.py:218-221) — one candidate per
    (prediction, class) among the predictions with obj * max(cls) >= conf_threshold — on decoded rows with several hot classes."""
    import torch
    nc = 6
    r = np.random.RandomState(1330)
    B, N = 3, 400
    dec = np.zeros((B, N, 5 + nc), np.float32)
    for b in range(B):
        c = r.uniform(30, 290, (12, 2)); wh = r.uniform(20, 80, (12, 2))
        for i in range(N):
            k = r.randint(12)
            cls = r.uniform(0.0, 0.25, nc)
            hot = r.choice(nc, size=r.randint(1, 4), replace=False)
            cls[hot] = r.uniform(0.45, 1.0, len(hot))
            dec[b, i] = np.concatenate([c[k] + r.uniform(-6, 6, 2), wh[k] * r.uniform(0.85, 1.2, 2), [r.uniform(0.05, 1.0)], cls])
    dec[1, :, 4] = 0.01                          # an image without candidates
    h = make_hyp(num_class=nc, img=320, mutil_label=True)
    e = ref_trainer.YOLOXEvaluator(None, h, compute_metric=False)
    res = e.numba_nms(torch.from_numpy(dec.copy()))
    g = {"mlx_dec": dec, "mlx_n": np.array([-1 if q is None else len(q) for q in res]),
         "mlx_thr": np.array([h["conf_threshold"], h["cls_threshold"], h["iou_threshold"]], np.float64)}
    for i, q in enumerate(res):
        if q is not None:
            g[f"mlx_out{i}"] = np.asarray(q, np.float32)
    np.savez_compressed(os.path.join(OUT, "g13_round4.npz"), **g)
    print("g13 written", os.path.getsize(os.path.join(OUT, "g13_round4.npz")) / 1e3, "KB", g["mlx_n"], g["mlx_thr"])


def gen_g14(ref_models, ref_trainer):
    """G14 (round 5): BASELINE config #4's model at the gradient level — YOLOv5l (models/normal/yolov5l.py:16-44) full-model
    backward in training mode (per-parameter signatures, calibrated like G11's YOLOv5s) and with evaluation-mode BatchNorm
    (well conditioned, like G12) — and the evaluators' `bbox_iou` (trainer/eval_yolov5.py:237-258, trainer/eval_yolox.py:177-199)
    with what the reference's `do_nms` (trainer/eval_yolov5.py:94-150) does on 0 / 1 / 2 candidates."""
    import torch
    g = {}
    # batch 4 at 320^2: 400 samples per channel in the deepest stage (at 2 x 256^2 — 128 samples — the reference's OWN bf16 run
    # decorrelates from its fp32 run through the 100 training-mode BatchNorms of this depth: sampled rel. rms 1.12)
    xl = np.random.RandomState(1412).rand(4, 3, 320, 320).astype(np.float32)
    _backward_sig(g, "v5l_bwd", lambda: ref_models.YOLOV5Large(3, 80), xl, lambda o: list(o), with_corr=True)
    xf = np.random.RandomState(1401).rand(2, 3, 256, 256).astype(np.float32)
    _frozen_bwd(g, "v5l_frozen", lambda: ref_models.YOLOV5Large(3, 80), 1410, xf, lambda o: list(o), nsamp=96)

    # ---- bbox_iou: clusters of overlapping boxes, pairs apart on one axis (negative side, zero after the product's sign), pairs apart
    # on BOTH axes (two negative sides: a positive "intersection"), zero-area boxes (0 / 0)
    r = np.random.RandomState(1430)
    c = r.uniform(20, 300, (10, 2)); wh = r.uniform(4, 60, (10, 2))
    rows = []
    for k in range(10):
        for _ in range(6):
            cc = c[k] + r.uniform(-10, 10, 2); ww = wh[k] * r.uniform(0.7, 1.3, 2)
            rows.append(np.concatenate([cc - ww / 2, cc + ww / 2]))
    b = np.array(rows, np.float32)
    tiny = np.array([[10, 10, 11, 11], [12, 12.5, 13, 13.5], [10.5, 30, 11.5, 31], [5, 5, 5, 5], [5, 5, 5, 5], [0, 0, 2, 2], [3, 3, 4, 4]], np.float32)
    b1 = np.concatenate([b[:24], tiny]); b2 = np.concatenate([b, tiny, b[:8] + np.float32(4096.0)])
    g["iou_b1"], g["iou_b2"] = b1, b2
    g["iou_v5"] = ref_trainer.YOLOV5Evaluator.bbox_iou(torch.from_numpy(b1), torch.from_numpy(b2)).numpy()
    g["iou_yolox"] = ref_trainer.YOLOXEvaluator.bbox_iou(torch.from_numpy(b1), torch.from_numpy(b2)).numpy()

    # ---- do_nms of the reference on 0, 1 and 2 candidates: rows it returns, or the exception it raises (utils/nms.py:62-63)
    from yoloseries_amd.utils.synth import COCO_ANCHORS
    nc = 4
    h = make_hyp(num_class=nc, img=320)
    e = ref_trainer.YOLOV5Evaluator(None, torch.from_numpy(COCO_ANCHORS.copy()), h, compute_metric=False)
    dec = np.zeros((3, 5, 5 + nc), np.float32)
    dec[:, :, 4] = 0.01
    dec[1, 2] = [100, 120, 40, 30, 0.9, 0.1, 0.8, 0.2, 0.05]                     # image 1: one candidate
    dec[2, 1] = [100, 120, 40, 30, 0.9, 0.1, 0.8, 0.2, 0.05]                     # image 2: two candidates
    dec[2, 3] = [104, 122, 42, 28, 0.8, 0.1, 0.7, 0.2, 0.05]
    g["donms_dec"] = dec
    status = []
    for i in range(3):
        try:
            res = e.do_nms(torch.from_numpy(dec[i:i + 1].copy()))
            status.append(0 if res[0] is None else 1)
            if res[0] is not None:
                g[f"donms_out{i}"] = res[0].numpy()
        except IndexError:
            status.append(-1)
    g["donms_status"] = np.array(status)                   # 0: None, 1: rows stored, -1: the reference raised IndexError
    np.savez_compressed(os.path.join(OUT, "g14_round5.npz"), **g)
    print("g14 written", os.path.getsize(os.path.join(OUT, "g14_round5.npz")) / 1e6, "MB; reference do_nms status", status)


def gen_g10(ref_utils):
    """G10 letterbox + collate format (utils/data_aug.py:21-70, utils/bbox_tools.py:38-49, dataset/data_collater.py:20-64).
    OpenCV is absent here, so only geometries the reference handles without cv2 are run (scale == 1, training=True);
    the nearest-neighbour resize itself stays unpinned."""
    import importlib.util
    import torch
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.setdefault("torchvision", tv)
    spec = importlib.util.spec_from_file_location("ref_data_collater", os.path.join(REF, "dataset", "data_collater.py"))
    coll = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(coll)
    from yoloseries_amd.dataset.synthetic import SyntheticDetectionDataset
    g = {}
    shapes = [(480, 640), (640, 640), (640, 512), (640, 576)]
    g["shapes"] = np.array(shapes)
    items = []
    for i, hw in enumerate(shapes):
        ds = SyntheticDetectionDataset(8, img_hw=hw, num_class=80, max_boxes=12, seed=40 + i)
        img, ann, iid = ds[i]
        if i == 3:
            ann = {'bboxes': np.zeros((0, 4), np.float32), 'classes': []}     # image without objects
        items.append((img, ann, iid))
        out, info = ref_utils.letter_resize_img(img, [640, 640])
        g[f"lr{i}_info"] = np.array([info['scale'], info['pad_top'], info['pad_left'], info['pad_bottom'], info['pad_right'],
                                      info['org_shape'][0], info['org_shape'][1]], np.float64)
        g[f"lr{i}_sum"] = out.astype(np.float64).sum(axis=(0, 1))
        g[f"lr{i}_sample"] = out.reshape(-1)[::997].copy()
        if len(ann['classes']):
            g[f"lb{i}"] = ref_utils.letter_resize_bbox(np.array(ann['bboxes'], np.float64).copy(), info)
    g["dataset_args"] = np.array([8, 80, 12, 40])      # length, num_class, max_boxes, seed base (seed = base + i, item i)
    batch = coll.fixed_imgsize_collate_fn(items, [640, 640])
    g["c_ann"] = batch['ann'].numpy()
    g["c_img_sum"] = batch['img'].double().sum(dim=(2, 3)).numpy()
    g["c_img_sample"] = batch['img'].reshape(-1)[::9973].numpy()
    g["c_info"] = np.array([[r['scale'], r['pad_top'], r['pad_left'], r['pad_bottom'], r['pad_right']] for r in batch['resize_info']], np.float64)
    g["c_ids"] = np.array(batch['img_id'])
    tb = coll.test_dataset_collate_fn([(torch.full((3, 64, 96), float(k)), {'scale': 1.0 + k}) for k in range(3)])
    g["t_img_sum"] = tb['img'].double().sum(dim=(1, 2, 3)).numpy()
    np.savez_compressed(os.path.join(OUT, "g10_collate.npz"), **g)


if __name__ == "__main__":
    main()
