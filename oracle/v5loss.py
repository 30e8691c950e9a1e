"""ORACLE (test infrastructure, never imported by the product path).

CPU restatement of the reference's YOLOv5 loss, /root/reference loss/yolov5_loss.py:
  match()      — NumPy float32 / int64, bit-exact target assignment (:142-214)
  V5LossOracle — the loss itself written against torch CPU fp32 (a floating-point
                 kernel keeps a torch fp32 reference; autograd supplies d tot / d preds)
Pinned against golden vectors generated from the reference itself
(tools/gen_golden.py -> tests/golden/g2_match.npz, g3_loss.npz).
"""
import math

import numpy as np

from .bbox import F32, xyxy2xywhn


def match(targets_batch, anchors_stage_px, fm_w, fm_h, img_size, anchor_thr=4.0):
    """YOLOV5Loss.match for ONE stage, starting from the raw batch targets.

    targets_batch: (B, maxbox, 6) float32 [xmin,ymin,xmax,ymax,cls,img_id], padding rows -1
    anchors_stage_px: (A,2) anchor (w,h) in pixels; img_size = hyp['input_img_size'] (2,)
    Returns tar_box (N,4) f32, cls, img, anc, gy, gx (int64), in the reference's row order:
    offset group k (0..4) major, then (anchor, image, box) row-major (:187-189).
    """
    t = np.asarray(targets_batch, dtype=F32).copy()
    B, MB, _ = t.shape
    A = anchors_stage_px.shape[0]
    t[..., :4] = xyxy2xywhn(t[..., :4], img_size)                          # :47-48
    t = np.broadcast_to(t[None], (A, B, MB, 6))
    anc_id = np.broadcast_to(np.arange(A, dtype=F32)[:, None, None, None], (A, B, MB, 1))
    t = np.concatenate([t, anc_id], axis=-1).astype(F32)                   # :50-56
    ds = F32(img_size[1]) / F32(fm_w)                                       # :66
    anchor_stage = (np.asarray(anchors_stage_px, dtype=F32) / ds).astype(F32)
    g = np.ones(7, dtype=F32)
    g[:4] = np.array([fm_w, fm_h, fm_w, fm_h], dtype=F32)
    ts = (t * g).astype(F32)                                               # :151-153
    wh = ts[..., 2:4]
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (wh / anchor_stage[:, None, None, :] + F32(1e-16)).astype(F32)
        ar_mask = np.maximum(ratio, F32(1) / ratio).max(axis=-1) < F32(anchor_thr)   # :166-170
    ts = ts[ar_mask]                                                        # (X,7) row-major over (a,b,j)
    xy = ts[:, 0:2]
    off_xy = (np.array([fm_w, fm_h], dtype=F32)[None] - xy).astype(F32)     # :175
    thr = F32(0.5)
    m1 = (np.mod(xy, F32(1.0)) < thr) & (xy > F32(1.))
    m2 = (np.mod(off_xy, F32(1.0)) < thr) & (off_xy > F32(1.))
    mask = np.stack([np.ones(len(ts), bool), m1[:, 0], m1[:, 1], m2[:, 0], m2[:, 1]], axis=0)   # (5,X)
    offset = (np.array([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=F32) * thr).astype(F32)
    ts5 = np.broadcast_to(ts[None], (5,) + ts.shape)[mask]                  # (N,7)
    off5 = np.broadcast_to(offset[:, None, :], (5, len(ts), 2))[mask]
    gxy = ts5[:, 0:2]
    coors = np.trunc(gxy - off5).astype(np.int64)                           # .long() :196
    tar_off = (gxy - coors.astype(F32)).astype(F32)                         # before the clamp :198
    tar_box = np.concatenate([tar_off, ts5[:, 2:4]], axis=-1).astype(F32)
    cls = ts5[:, 4].astype(np.int64)
    img = ts5[:, 5].astype(np.int64)
    anc = ts5[:, 6].astype(np.int64)
    gx = np.clip(coors[:, 0], 0, fm_w - 1)
    gy = np.clip(coors[:, 1], 0, fm_h - 1)
    return tar_box, cls, img, anc, gy, gx


class V5LossOracle:
    """Restatement of YOLOV5Loss.__call__ (:30-140) with torch CPU fp32 ops.

    stage_preds follow the reference layout (B, A*(5+nc), h, w)."""

    def __init__(self, anchors, hyp, stage_num=3):
        self.anchors = np.asarray(anchors, dtype=F32)            # (S,A,2)
        self.hyp = hyp
        self.balances = [4., 1., 0.4] if stage_num == 3 else [4., 1., 0.4, 0.1]

    def focal(self, pred, target):
        import torch
        prob = torch.sigmoid(pred)
        acc = target * prob + (1.0 - target) * (1.0 - prob)
        gamma = self.hyp.get('focal_loss_gamma', 1.5)
        alpha = self.hyp.get('focal_loss_alpha', 0.25)
        return (1.0 - acc) ** gamma * (target * alpha + (1.0 - target) * (1.0 - alpha))

    def __call__(self, stage_preds, targets_batch):
        import torch
        import torch.nn.functional as Fn
        hyp = self.hyp
        tb = np.asarray(targets_batch.detach().cpu().numpy() if hasattr(targets_batch, "detach") else targets_batch, dtype=F32)
        B = tb.shape[0]
        S = len(stage_preds)
        s = 3 / S
        nc = hyp['num_class']
        cls_loss = torch.zeros(1); iou_loss = torch.zeros(1); cof_loss = torch.zeros(1)
        tot_n = 0
        pw_cls = torch.tensor(float(hyp['cls_pos_weight'])); pw_cof = torch.tensor(float(hyp['cof_pos_weight']))
        for i, sp in enumerate(stage_preds):
            bn, _, fh, fw = sp.shape
            A = self.anchors.shape[1]
            ds = F32(hyp['input_img_size'][1]) / F32(fw)
            anchor_stage = torch.from_numpy((self.anchors[i] / ds).astype(F32))
            preds = sp.reshape(bn, A, -1, fh, fw).permute(0, 1, 3, 4, 2).contiguous()                  # :71
            tbox, tcls, timg, tanc, gy, gx = match(tb, self.anchors[i], fw, fh, hyp['input_img_size'], hyp['anchor_match_thr'])
            N = tbox.shape[0]
            tot_n += N
            ti, ta, tgy, tgx = (torch.from_numpy(v) for v in (timg, tanc, gy, gx))
            cur = preds[ti, ta, tgy, tgx]                                                             # :79
            if nc > 1:
                t_cls = torch.zeros_like(cur[:, 5:])
                t_cls[torch.arange(N), torch.from_numpy(tcls)] = hyp['class_smooth_factor']
                fac = self.focal(cur[:, 5:], t_cls) if hyp['use_focal_loss'] else torch.ones_like(t_cls)
                cls_loss = cls_loss + (Fn.binary_cross_entropy_with_logits(cur[:, 5:], t_cls, pos_weight=pw_cls, reduction='none') * fac).mean()
            t_cof = torch.zeros_like(preds[..., 4])
            if N > 0:
                pxy = cur[:, :2].sigmoid() * 2. - 0.5
                pwh = (cur[:, 2:4].sigmoid() * 2.) ** 2 * anchor_stage[ta]
                pb = torch.cat((pxy, pwh), dim=1)
                tbt = torch.from_numpy(tbox)
                iou = _ciou_t(_xywh2xyxy_t(pb), _xywh2xyxy_t(tbt))
                iou_loss = iou_loss + (1.0 - iou).mean()
                t_cof[ti, ta, tgy, tgx] = iou.detach().clamp(0).type_as(t_cof)                        # :114 last writer wins
            fac = self.focal(preds[..., 4], t_cof) if hyp['use_focal_loss'] else torch.ones_like(t_cof)
            tmp = (Fn.binary_cross_entropy_with_logits(preds[..., 4], t_cof, pos_weight=pw_cof, reduction='none') * fac).mean()
            tmp = tmp * self.balances[i]
            self.balances[i] = self.balances[i] * 0.9999 + 0.0001 / tmp.detach().item()              # :124
            cof_loss = cof_loss + tmp
        self.balances = [x / self.balances[1] for x in self.balances]                                # :127
        iou_loss = iou_loss * (hyp['iou_loss_scale'] * s)
        cof_loss = cof_loss * (hyp['cof_loss_scale'] * s * (1. if S == 3 else 1.4))
        cls_loss = cls_loss * (hyp['cls_loss_scale'] * s)
        tot = (iou_loss + cof_loss + cls_loss) * B
        return {'tot_loss': tot, 'iou_loss': iou_loss.item() * B, 'cof_loss': cof_loss.item() * B,
                'cls_loss': cls_loss.item() * B, 'tar_nums': tot_n}


def _xywh2xyxy_t(b):
    import torch
    x, y, w, h = b.chunk(4, -1)
    return torch.cat((x - w / 2, y - h / 2, x + w / 2, y + h / 2), dim=-1)


def _ciou_t(b1, b2):
    """utils/bbox_tools.py:286-339 in torch (alpha under no_grad)."""
    import torch
    eps = 1e-9
    x1, y1, x2, y2 = b1.chunk(4, -1)
    X1, Y1, X2, Y2 = b2.chunk(4, -1)
    w1, h1, w2, h2 = x2 - x1, y2 - y1, X2 - X1, Y2 - Y1
    iw = torch.clamp(torch.min(x2, X2) - torch.max(x1, X1), min=0.)
    ih = torch.clamp(torch.min(y2, Y2) - torch.max(y1, Y1), min=0.)
    inter = iw * ih
    union = torch.clamp(w1 * h1 + w2 * h2 - inter, eps)
    iou = inter / union
    c_hs = torch.max(y2, Y2) - torch.min(y1, Y1)
    c_ws = torch.max(x2, X2) - torch.min(x1, X1)
    c_diag = torch.pow(c_ws, 2) + torch.pow(c_hs, 2)
    cw = (x1 + x2) / 2 - (X1 + X2) / 2
    ch = (y1 + y2) / 2 - (Y1 + Y2) / 2
    ctr = ch ** 2 + cw ** 2
    v = (4 / (math.pi ** 2)) * (torch.atan(w1 / torch.clamp(h1, eps)) - torch.atan(w2 / torch.clamp(h2, eps))).pow(2)
    with torch.no_grad():
        alpha = v / torch.clamp(1 - iou + v, eps)
    c_diag = torch.clamp(c_diag, min=eps)
    return (iou - (ctr / c_diag + v * alpha)).squeeze(-1)
