"""YOLOV5Loss — host-side mirror of the reference class (loss/yolov5_loss.py:8-235)
over the HIP kernels of csrc/loss_v5.hip.  Same constructor, call signature, returned
dict and stateful ``balances``; the arithmetic runs on the GPU only.

Prediction tensors are consumed in place when they are the cell-major ("NHWC") views
the yoloseries_amd models return (shape (B, A*(5+nc), h, w), strides (h*w*ld, 1, w*ld, ld));
any other tensor is first copied into that layout (torch copy, off the fast path).
"""
import ctypes as C

import torch

from .. import _lib
from .._lib import V5LossDesc, check, lib

__all__ = ["YOLOV5Loss"]


def _canon(p):
    """Return (tensor_in_cell_major_layout, ld, was_view). Accepts (B, Ctot, h, w)."""
    B, Ct, h, w = p.shape
    st = p.stride()
    if p.dtype in (torch.bfloat16, torch.float32) and st[1] == 1 and st[3] % 8 == 0 and st[3] >= Ct \
            and st[2] == w * st[3] and st[0] == h * w * st[3] and p.data_ptr() % 16 == 0:
        return p, st[3], True
    ld = ((Ct + 7) // 8) * 8
    dt = p.dtype if p.dtype in (torch.bfloat16, torch.float32) else torch.float32
    buf = torch.zeros(B, h, w, ld, dtype=dt, device=p.device)
    buf[..., :Ct] = p.detach().permute(0, 2, 3, 1)
    return buf.as_strided((B, Ct, h, w), (h * w * ld, 1, w * ld, ld)), ld, False


class _V5LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, targets, *preds):
        desc, canon, lds = owner._make_desc(preds, targets)
        L = lib()
        dev = targets.device
        saved = torch.empty(L.yh_v5loss_saved_bytes(C.byref(desc)), dtype=torch.uint8, device=dev)
        tptr = targets.data_ptr()
        ws = owner._workspace(L.yh_v5loss_ws_bytes(C.byref(desc)), dev)
        result = torch.empty(8, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * 4)(*[c.data_ptr() for c in canon], *([None] * (4 - len(canon))))
        check(L.yh_v5_loss_fwd(C.byref(desc), ptrs, tptr, owner._balances.data_ptr(), result.data_ptr(),
                               saved.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "yh_v5_loss_fwd")
        ctx.owner, ctx.desc, ctx.canon, ctx.saved = owner, desc, canon, saved
        ctx.in_meta = [(p.shape, p.dtype, p.stride()) for p in preds]
        ctx.mark_non_differentiable(result)
        return result[0:1].clone(), result

    @staticmethod
    def backward(ctx, gtot, _gres):
        L = lib()
        desc, canon = ctx.desc, ctx.canon
        gout = gtot.to(torch.float32).contiguous()
        gbufs = []
        for c in canon:
            B, Ct, h, w = c.shape
            ld = c.stride(3)
            gbufs.append(torch.empty(B, h, w, ld, dtype=c.dtype, device=c.device))
        ptrs = (C.c_void_p * 4)(*[c.data_ptr() for c in canon], *([None] * (4 - len(canon))))
        gptrs = (C.c_void_p * 4)(*[g.data_ptr() for g in gbufs], *([None] * (4 - len(canon))))
        ws = ctx.owner._workspace(L.yh_v5loss_ws_bytes(C.byref(desc)), gout.device)
        check(L.yh_v5_loss_bwd(C.byref(desc), ptrs, gout.data_ptr(), ctx.saved.data_ptr(), gptrs, ws.data_ptr(),
                               _lib.stream_ptr()), "yh_v5_loss_bwd")
        outs = []
        for g, c, (shape, dtype, stride) in zip(gbufs, canon, ctx.in_meta):
            B, Ct, h, w = c.shape
            ld = c.stride(3)
            v = g.as_strided((B, Ct, h, w), (h * w * ld, 1, w * ld, ld))
            outs.append(v if v.dtype == dtype else v.to(dtype))
        return (None, None, *outs)


class YOLOV5Loss:

    def __init__(self, anchors, hyp, stage_num=3):
        """:param anchors: tensor (3, 3, 2) in pixels; :param hyp: flat config dict (config/config.py:14-20)"""
        self.anchors = anchors
        self.hyp = hyp
        self.device = hyp['device']
        self.input_img_size = hyp['input_img_size']
        self.stage_num = stage_num
        self._init_balances = [4., 1., 0.4] if stage_num == 3 else [4., 1., 0.4, 0.1]
        self._balances = None
        self._ws = None
        self._anchors_host = [[[float(v) for v in a] for a in st] for st in anchors.detach().cpu().tolist()]

    # ---- state -----------------------------------------------------------------------------
    @property
    def balances(self):
        if self._balances is None:
            return list(self._init_balances)
        return self._balances.cpu().tolist()[:self.stage_num]

    @balances.setter
    def balances(self, v):
        self._init_balances = [float(x) for x in v]
        if self._balances is not None:
            self._balances[:len(v)] = torch.tensor(self._init_balances, dtype=torch.float64, device=self._balances.device)

    def _workspace(self, nbytes, dev):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._ws

    def _make_desc(self, preds, targets, canon_given=None):
        hyp = self.hyp
        d = V5LossDesc()
        d.B, d.maxbox = targets.shape[0], targets.shape[1]
        d.num_class = hyp['num_class']
        d.num_anchor = len(self._anchors_host[0])
        d.num_stage = len(preds)
        d.img_size0, d.img_size1 = float(self.input_img_size[0]), float(self.input_img_size[1])
        flat = [v for st in self._anchors_host for a in st for v in a]
        for s in range(d.num_stage):
            for a in range(d.num_anchor):
                d.anchors[(s * 3 + a) * 2 + 0] = self._anchors_host[s][a][0]
                d.anchors[(s * 3 + a) * 2 + 1] = self._anchors_host[s][a][1]
        del flat
        d.anchor_thr = float(hyp['anchor_match_thr'])
        d.cls_smooth = float(hyp['class_smooth_factor'])
        d.cls_pos_weight = float(hyp['cls_pos_weight'])
        d.cof_pos_weight = float(hyp['cof_pos_weight'])
        d.use_focal = int(bool(hyp['use_focal_loss']))
        d.focal_gamma = float(hyp.get('focal_loss_gamma', 1.5))
        d.focal_alpha = float(hyp.get('focal_loss_alpha', 0.25))
        d.iou_scale, d.cof_scale, d.cls_scale = float(hyp['iou_loss_scale']), float(hyp['cof_loss_scale']), float(hyp['cls_loss_scale'])
        canon, lds = [], []
        dts = set()
        for s, p in enumerate(preds):
            c, ld, _ = _canon(p)
            canon.append(c); lds.append(ld); dts.add(c.dtype)
            d.H[s], d.W[s], d.ldp[s] = p.shape[2], p.shape[3], ld
            assert p.shape[1] == d.num_anchor * (5 + d.num_class)
        if len(dts) != 1:
            canon = [c.float() if c.dtype != torch.float32 else c for c in canon]
            canon = [_canon(c)[0] for c in canon]
        d.pred_is_f32 = int(canon[0].dtype == torch.float32)
        d.targets_xywhn = 0
        return d, canon, lds

    # ---- reference API ---------------------------------------------------------------------
    def __call__(self, stage_preds, targets_batch):
        """:param stage_preds: (small, mid, large) each (bn, A*(5+nc), h, w)
        :param targets_batch: (bn, bbox_num, 6) [xmin, ymin, xmax, ymax, cls, img_id], padding rows -1"""
        assert isinstance(stage_preds, (list, tuple))
        assert isinstance(targets_batch, torch.Tensor), f"targets's type should be torch.Tensor but we got {type(targets_batch)}"
        assert stage_preds[0].size(0) == targets_batch.size(0), "the length of predictions and targets should be the same"
        if not stage_preds[0].is_cuda:
            raise _lib.YoloHipError("YOLOV5Loss: predictions must live on an MI355X device (no CPU path in the product)")
        dev = stage_preds[0].device
        if self._balances is None or self._balances.device != dev:
            self._balances = torch.zeros(4, dtype=torch.float64, device=dev)
            self._balances[:len(self._init_balances)] = torch.tensor(self._init_balances, dtype=torch.float64)
        targets = targets_batch.detach().to(device=dev, dtype=torch.float32).contiguous()
        tot, result = _V5LossFn.apply(self, targets, *stage_preds)
        batch_size = targets_batch.size(0)
        del batch_size
        if self.hyp.get('loss_items_on_device', False):
            # no host sync: the scalars stay on the device (bench / graph-captured training loops)
            return {'tot_loss': tot, 'iou_loss': result[1], 'cof_loss': result[2], 'cls_loss': result[3], 'tar_nums': result[4]}
        r = result.tolist()
        return {'tot_loss': tot, 'iou_loss': r[1], 'cof_loss': r[2], 'cls_loss': r[3], 'tar_nums': int(r[4])}

    def match(self, targets, anchor_stage, fm_shape):
        """Reference signature (loss/yolov5_loss.py:142): targets (A, bn, bbox_num, 7) normalised
        [x, y, w, h, cls, img_id, anchor_id]; anchor_stage (A,2) in grid units; fm_shape [w, h].
        Returns tar_box (N,4), cls, img_idx, anc_idx, gy, gx (int64)."""
        L = lib()
        t = targets[0][..., :6].detach().to(torch.float32).contiguous()
        if not t.is_cuda:
            raise _lib.YoloHipError("YOLOV5Loss.match: tensors must live on an MI355X device")
        fw, fh = int(fm_shape[0]), int(fm_shape[1])
        d = V5LossDesc()
        d.B, d.maxbox, d.num_class, d.num_anchor, d.num_stage = t.shape[0], t.shape[1], self.hyp['num_class'], anchor_stage.shape[0], 1
        d.H[0], d.W[0], d.ldp[0] = fh, fw, ((anchor_stage.shape[0] * (5 + self.hyp['num_class']) + 7) // 8) * 8
        d.img_size0, d.img_size1 = float(fw), float(fw)      # ds = img_size1 / fw = 1: anchors are already in grid units
        av = anchor_stage.detach().cpu().tolist()
        for a in range(d.num_anchor):
            d.anchors[a * 2], d.anchors[a * 2 + 1] = float(av[a][0]), float(av[a][1])
        d.anchor_thr = float(self.hyp['anchor_match_thr'])
        d.targets_xywhn = 1
        cap = 5 * d.num_anchor * d.B * d.maxbox
        count = torch.zeros(1, dtype=torch.int32, device=t.device)
        tbox = torch.empty(cap, 4, dtype=torch.float32, device=t.device)
        tidx = torch.empty(cap, 5, dtype=torch.int32, device=t.device)
        check(L.yh_v5_assign(C.byref(d), t.data_ptr(), count.data_ptr(), tbox.data_ptr(), tidx.data_ptr(), None,
                             _lib.stream_ptr()), "yh_v5_assign")
        n = int(count.item())
        ti = tidx[:n].long()
        return tbox[:n], ti[:, 0], ti[:, 1], ti[:, 2], ti[:, 3], ti[:, 4]

    def assign(self, targets_batch, fm_sizes):
        """All stages at once from the raw batch targets (xyxy pixels): list of match() tuples."""
        L = lib()
        t = targets_batch.detach().to(torch.float32).contiguous()
        d = V5LossDesc()
        d.B, d.maxbox, d.num_class, d.num_anchor, d.num_stage = t.shape[0], t.shape[1], self.hyp['num_class'], len(self._anchors_host[0]), len(fm_sizes)
        d.img_size0, d.img_size1 = float(self.input_img_size[0]), float(self.input_img_size[1])
        for s, (fh, fw) in enumerate(fm_sizes):
            d.H[s], d.W[s], d.ldp[s] = fh, fw, 256
            for a in range(d.num_anchor):
                d.anchors[(s * 3 + a) * 2], d.anchors[(s * 3 + a) * 2 + 1] = self._anchors_host[s][a]
        d.anchor_thr = float(self.hyp['anchor_match_thr'])
        cap = 5 * d.num_anchor * d.B * d.maxbox
        S = d.num_stage
        count = torch.zeros(S, dtype=torch.int32, device=t.device)
        tbox = torch.empty(S, cap, 4, dtype=torch.float32, device=t.device)
        tidx = torch.empty(S, cap, 5, dtype=torch.int32, device=t.device)
        check(L.yh_v5_assign(C.byref(d), t.data_ptr(), count.data_ptr(), tbox.data_ptr(), tidx.data_ptr(), None,
                             _lib.stream_ptr()), "yh_v5_assign")
        outs = []
        for s, n in enumerate(count.tolist()):
            ti = tidx[s, :n].long()
            outs.append((tbox[s, :n], ti[:, 0], ti[:, 1], ti[:, 2], ti[:, 3], ti[:, 4]))
        return outs

    def focal_loss_factor(self, pred, target):
        """loss/yolov5_loss.py:216-235 (kept for API completeness; the kernels fuse it)."""
        prob = torch.sigmoid(pred)
        acc_scale = target * prob + (1.0 - target) * (1.0 - prob)
        gamma = self.hyp.get('focal_loss_gamma', 1.5)
        alpha = self.hyp.get('focal_loss_alpha', 0.25)
        return (1.0 - acc_scale) ** gamma * (target * alpha + (1.0 - target) * (1.0 - alpha))
