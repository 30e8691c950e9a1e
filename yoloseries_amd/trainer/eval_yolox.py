"""YOLOXEvaluator — mirror of trainer/eval_yolox.py:11-259 over csrc/postproc.hip (yolox decode / filter variants)."""
import ctypes as C

import torch

from .. import _lib
from .._lib import DecodeDesc, check, lib
from ..layout import to_cell_major
from .eval_yolov5 import YOLOV5Evaluator, _decode_ws

__all__ = ['YOLOXEvaluator']


class YOLOXEvaluator(YOLOV5Evaluator):

    def __init__(self, yolo, hyp, compute_metric=False):
        self.yolo = yolo
        self.hyp = hyp
        self.device = hyp['device']
        self.num_class = hyp['num_class']
        self.num_stage = hyp.get('num_stage', 3)
        self.anchor_num = 1
        self.ds_scales = [8, 16, 32]
        self.inp_h, self.inp_w = hyp['input_img_size']
        self.use_tta = hyp['use_tta']
        self.iou_threshold = hyp['compute_metric_iou_threshold'] if compute_metric else hyp['iou_threshold']
        self.cls_threshold = hyp['compute_metric_cls_threshold'] if compute_metric else hyp['cls_threshold']
        self.conf_threshold = hyp['compute_metric_conf_threshold'] if compute_metric else hyp['conf_threshold']
        self._yolox = 1

    def _stage_list(self, stage_preds):
        vals = list(stage_preds.values()) if isinstance(stage_preds, dict) else list(stage_preds)
        out = []
        for p in vals:
            if p.dim() == 5:
                if p.shape[1] != 1:
                    raise NotImplementedError("YOLOXEvaluator on the HIP path supports num_anchors=1")
                p = p[:, 0]
            out.append(p)
        return out

    def _desc(self, stage_preds):
        stage_preds = self._stage_list(stage_preds)
        d = DecodeDesc()
        d.B = stage_preds[0].shape[0]
        d.num_class, d.num_anchor, d.num_stage = self.num_class, 1, len(stage_preds)
        canon = []
        for s, p in enumerate(stage_preds):
            c, ld = to_cell_major(p)
            canon.append(c)
            d.H[s], d.W[s], d.ldp[s] = p.shape[2], p.shape[3], ld
        if len({c.dtype for c in canon}) != 1:
            canon = [to_cell_major(c.float())[0] for c in canon]
        d.pred_is_f32 = int(canon[0].dtype == torch.float32)
        d.yolox = 1
        self._cur_img_h = None
        ptrs = (C.c_void_p * 4)(*[c.data_ptr() for c in canon], *([None] * (4 - len(canon))))
        return d, canon, ptrs

    def _set_strides(self, d, img_h):
        for s in range(d.num_stage):
            d.stride[s] = float(img_h) / float(d.H[s])        # input_img_h / fm_h (eval_yolox.py:142-144)

    @torch.no_grad()
    def do_inference(self, inputs):
        stage_preds = self.yolo(inputs)
        return self.decode(stage_preds, inputs.size(2))

    def decode(self, stage_preds, img_h=None):
        sp = self._stage_list(stage_preds)
        if not sp[0].is_cuda:
            raise _lib.YoloHipError("YOLOXEvaluator: tensors must live on an MI355X device")
        d, canon, ptrs = self._desc(sp)
        self._set_strides(d, img_h if img_h is not None else self.inp_h)
        n = sum(p.shape[2] * p.shape[3] for p in sp)
        out = torch.empty(d.B, n, 5 + self.num_class, dtype=torch.float32, device=sp[0].device)
        check(lib().yh_decode_full(C.byref(d), ptrs, out.data_ptr(), _lib.stream_ptr()), "yh_decode_full")
        return out

    @torch.no_grad()
    def __call__(self, inputs):
        if self.use_tta:
            merge_preds_out, _ = self.test_time_augmentation(inputs)
            outs = self.numba_nms(merge_preds_out)
        else:
            outs = self._nms_from_heads(self.yolo(inputs), inputs.size(2))
        return [torch.from_numpy(x) if x is not None else None for x in outs]

    def _nms_from_heads(self, stage_preds, img_h=None):
        if self.hyp.get('mutil_label', False):          # one candidate per (prediction, class): through the decoded tensor (eval_yolox.py:218-221)
            return self.numba_nms(self.decode(stage_preds, img_h))
        sp = self._stage_list(stage_preds)
        d, canon, ptrs = self._desc(sp)
        self._set_strides(d, img_h if img_h is not None else self.inp_h)
        dev = sp[0].device
        B = d.B
        n = sum(p.shape[2] * p.shape[3] for p in sp)
        cap = ((n + 3) // 4) * 4
        cand = torch.empty(B, cap, 6, dtype=torch.float32, device=dev)
        ncand = torch.zeros(B, dtype=torch.int32, device=dev)
        check(lib().yh_decode_filter(C.byref(d), ptrs, float(self.conf_threshold), float(self.cls_threshold),
                                     cand.data_ptr(), ncand.data_ptr(), cap, _decode_ws(self, d, dev), _lib.stream_ptr()), "yh_decode_filter")
        return self._run_nms(cand, ncand, B, cap)

    # candidate rule of this evaluator (eval_yolox.py:206-231): pre-filter on obj * max(cls) >= conf, class confidence >= cls
    # threshold (inclusive, unlike the v5 evaluator); hyp['mutil_label']: one candidate per (prediction, class) (:218-221).
    # numba_nms (eval_yolox.py:201-259), bbox_iou (:177-199) and do_nms are inherited: the reference's YOLOX evaluator has no
    # do_nms of its own — here it is the v5 method over THIS candidate rule
    _FILTER_MODES = (1, 3)
