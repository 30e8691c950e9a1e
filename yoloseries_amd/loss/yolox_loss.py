"""YOLOXLoss — host-side mirror of the reference class (loss/yolox_loss.py:11-458) over csrc/loss_yolox.hip.
Same constructor / call signature / returned dict / stateful ``balances``; like the reference it converts the
caller's target tensor from xyxy to xywh IN PLACE (:42).  SimOTA assignment and every loss term run on the GPU
without host synchronisation (the reference loops over images and ground truths in Python)."""
import ctypes as C

import torch

from .. import _lib
from .._lib import YoloxDesc, check, lib
from ..layout import to_cell_major

__all__ = ["YOLOXLoss"]

_IOU_TYPES = {"iou": 0, "giou": 1, "ciou": 2}


def _canon5(p):
    """(B, 1, E, h, w) or (B, E, h, w) -> cell-major 4-D view (B, E, h, w), ld"""
    if p.dim() == 5:
        if p.shape[1] != 1:
            raise NotImplementedError("YOLOXLoss on the HIP path supports num_anchors=1")
        p = p[:, 0]
    return to_cell_major(p)


class _YoloxLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, targets, *preds):
        desc, canon = owner._make_desc(preds, targets)
        L = lib()
        dev = targets.device
        saved = torch.empty(L.yh_yolox_saved_bytes(C.byref(desc)), dtype=torch.uint8, device=dev)
        ws = owner._workspace(L.yh_yolox_ws_bytes(C.byref(desc)), dev)
        result = torch.empty(8, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * 4)(*[c.data_ptr() for c in canon], *([None] * (4 - len(canon))))
        check(L.yh_yolox_loss_fwd(C.byref(desc), ptrs, targets.data_ptr(), owner._balances.data_ptr(), result.data_ptr(),
                                  saved.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "yh_yolox_loss_fwd")
        ctx.owner, ctx.desc, ctx.canon, ctx.saved, ctx.targets = owner, desc, canon, saved, targets
        ctx.in_meta = [(p.shape, p.dtype) for p in preds]
        owner._last = (desc, saved)
        ctx.mark_non_differentiable(result)
        return result[0:1].clone(), result

    @staticmethod
    def backward(ctx, gtot, _gres):
        L = lib()
        desc, canon = ctx.desc, ctx.canon
        gout = gtot.to(torch.float32).contiguous()
        gbufs = [torch.empty(c.shape[0], c.shape[2], c.shape[3], c.stride(3), dtype=c.dtype, device=c.device) for c in canon]
        ptrs = (C.c_void_p * 4)(*[c.data_ptr() for c in canon], *([None] * (4 - len(canon))))
        gptrs = (C.c_void_p * 4)(*[g.data_ptr() for g in gbufs], *([None] * (4 - len(canon))))
        check(L.yh_yolox_loss_bwd(C.byref(desc), ptrs, ctx.targets.data_ptr(), gout.data_ptr(), ctx.saved.data_ptr(), gptrs,
                                  _lib.stream_ptr()), "yh_yolox_loss_bwd")
        outs = []
        for g, c, (shape, dtype) in zip(gbufs, canon, ctx.in_meta):
            Bn, Ct, h, w = c.shape
            ld = c.stride(3)
            if len(shape) == 5:
                v = g.as_strided((Bn, 1, Ct, h, w), (h * w * ld, h * w * ld, 1, w * ld, ld))
            else:
                v = g.as_strided((Bn, Ct, h, w), (h * w * ld, 1, w * ld, ld))
            outs.append(v if v.dtype == dtype else v.to(dtype))
        return (None, None, *outs)


class YOLOXLoss:

    def __init__(self, hyp) -> None:
        self.hyp = hyp
        self.num_anchors = hyp['num_anchors']
        self.num_stage = hyp.get('num_stage', 3)
        self.img_sz = hyp['input_img_size']
        self.num_class = hyp['num_class']
        self.use_l1 = hyp.get('use_l1', True)
        self.iou_loss_scale = hyp.get('iou_loss_scale', 0.5)
        self.cls_loss_scale = hyp.get('cls_loss_scale', 1.0)
        self.l1_loss_scale = hyp.get('l1_loss_scale', 1.0)
        self.cof_loss_scale = hyp.get('cof_loss_scale', 1.0)
        self.device = hyp['device']
        self.cls_smoothness = hyp['class_smooth_factor']
        self._init_balances = [4., 1., 0.4] if self.num_stage == 3 else [4., 1., 0.4, 0.1]
        self._balances = None
        self._ws = None
        self._last = None
        if self.num_anchors != 1:
            raise NotImplementedError("YOLOXLoss on the HIP path supports num_anchors=1 (the shipped configuration)")
        # class part of the SimOTA cost: the reference evaluates it on zero logits (label_assign :111-147), so it is one
        # constant; computed here with the reference's own fp32 expression
        t = torch.zeros(self.num_class); t[0] = float(self.cls_smoothness)
        pc = torch.sqrt(torch.sigmoid(torch.zeros(self.num_class)) * torch.sigmoid(torch.zeros(1)))
        self._cls_cost_const = float((-(t * torch.log(pc) + (1 - t) * torch.log(1 - pc))).sum(-1))

    @property
    def balances(self):
        if self._balances is None:
            return list(self._init_balances)
        return self._balances.cpu().tolist()[:self.num_stage]

    @balances.setter
    def balances(self, v):
        self._init_balances = [float(x) for x in v]
        if self._balances is not None:
            self._balances[:len(v)] = torch.tensor(self._init_balances, dtype=torch.float64, device=self._balances.device)

    def _workspace(self, nbytes, dev):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._ws

    def _make_desc(self, preds, targets):
        hyp = self.hyp
        d = YoloxDesc()
        d.B, d.maxbox, d.num_class, d.num_stage = targets.shape[0], targets.shape[1], self.num_class, len(preds)
        canon = []
        for s, p in enumerate(preds):
            c, ld = _canon5(p)
            canon.append(c)
            d.H[s], d.W[s], d.ldp[s] = c.shape[2], c.shape[3], ld
            assert c.shape[1] == 5 + self.num_class
        if len({c.dtype for c in canon}) != 1:
            canon = [to_cell_major(c.float())[0] for c in canon]
        d.pred_is_f32 = int(canon[0].dtype == torch.float32)
        d.img_size0 = float(self.img_sz[0])
        d.use_focal = int(bool(hyp['use_focal_loss']))
        d.focal_gamma, d.focal_alpha = float(hyp.get('focal_loss_gamma', 1.5)), float(hyp.get('focal_loss_alpha', 0.25))
        d.use_l1 = int(bool(self.use_l1))
        d.iou_scale, d.cls_scale, d.cof_scale, d.l1_scale = (float(self.iou_loss_scale), float(self.cls_loss_scale),
                                                           float(self.cof_loss_scale), float(self.l1_loss_scale))
        d.cls_smooth = float(self.cls_smoothness)
        d.cls_pos_weight, d.cof_pos_weight = float(hyp.get("cls_pos_weight", 1.)), float(hyp.get("cof_pos_weight", 1.))
        d.iou_type = _IOU_TYPES[hyp['iou_type']]
        d.topk, d.center_radius = int(hyp['topk']), float(hyp['center_radius'])
        d.cls_cost_const = self._cls_cost_const
        return d, canon

    def __call__(self, preds, tars):
        """preds: dict {'pred_s','pred_m','pred_l'} of (N, num_anchors, 5+nc, h, w); tars: (N, bbox_num, 6)
        [xmin, ymin, xmax, ymax, class_id, img_id] — converted to [x_ctr, y_ctr, w, h, ...] in place."""
        plist = list(preds.values())
        if not plist[0].is_cuda:
            raise _lib.YoloHipError("YOLOXLoss: predictions must live on an MI355X device (no CPU path in the product)")
        dev = plist[0].device
        b = tars[..., :4].clone()
        tars[..., 0:2] = (b[..., 0:2] + b[..., 2:4]) / 2
        tars[..., 2:4] = b[..., 2:4] - b[..., 0:2]
        if self._balances is None or self._balances.device != dev:
            self._balances = torch.zeros(4, dtype=torch.float64, device=dev)
            self._balances[:len(self._init_balances)] = torch.tensor(self._init_balances, dtype=torch.float64)
        targets = tars.detach().to(device=dev, dtype=torch.float32).contiguous()
        tot, result = _YoloxLossFn.apply(self, targets, *plist)
        if self.hyp.get('loss_items_on_device', False):
            return {'tot_loss': tot, 'iou_loss': result[1], 'l1_loss': result[2], 'cls_loss': result[3], 'cof_loss': result[4],
                    'fg_nums': result[5], 'tar_nums': result[6]}
        r = result.tolist()
        return {'tot_loss': tot, 'iou_loss': r[1], 'l1_loss': r[2], 'cls_loss': r[3], 'cof_loss': r[4],
                'fg_nums': int(r[5]), 'tar_nums': int(r[6])}

    def foreground_masks(self):
        """(debug / tests) per-stage bool masks (N*h*w,) of the cells SimOTA selected in the last call."""
        desc, saved = self._last
        lay = torch.zeros(8, dtype=torch.int64)
        check(lib().yh_yolox_layout(C.byref(desc), lay.data_ptr()), "yh_yolox_layout")
        lay = lay.tolist()
        raw = saved.cpu().numpy()
        import numpy as np
        outs = []
        for s in range(desc.num_stage):
            n = desc.H[s] * desc.W[s]
            cnt = np.frombuffer(raw[lay[0]:lay[0] + 4 * 4 * desc.B].tobytes(), dtype=np.int32)[s * desc.B:(s + 1) * desc.B]
            cells = np.frombuffer(raw[lay[1]:].tobytes(), dtype=np.int32, count=lay[4 + s] + desc.B * n)[lay[4 + s]:]
            m = np.zeros(desc.B * n, dtype=bool)
            for b in range(desc.B):
                m[b * n + cells[b * n:b * n + cnt[b]]] = True
            outs.append(m)
        return outs
