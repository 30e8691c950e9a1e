"""YOLOV5Evaluator — host-side mirror of trainer/eval_yolov5.py:10-317 over csrc/postproc.hip.

forward -> decode -> candidate filter -> class-aware greedy NMS -> merge filter all run on
the GPU; only the final (n,6) rows per image cross PCIe (the reference copies the whole
decoded (B, 25200, 85) tensor to the host, :265).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib
from .._lib import DecodeDesc, check, lib
from ..layout import to_cell_major

__all__ = ['YOLOV5Evaluator']



def _decode_ws(owner, d, dev):
    """workspace of the two-pass decode + filter (yh_decode_filter_ws_bytes), kept on the evaluator between calls"""
    need = int(lib().yh_decode_filter_ws_bytes(C.byref(d)))
    ws = getattr(owner, "_decode_ws_buf", None)
    if ws is None or ws.numel() < need or ws.device != torch.device(dev):
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        owner._decode_ws_buf = ws
    return ws.data_ptr()

class YOLOV5Evaluator:

    def __init__(self, yolo, anchors, hyp, compute_metric=False):
        self.yolo = yolo
        self.hyp = hyp
        self.device = hyp['device']
        self.num_class = hyp['num_class']
        self.anchor_num = anchors.size(1)
        self.anchors = anchors
        self.num_stage = len(anchors)
        self.ds_scales = [8, 16, 32]
        self.inp_h, self.inp_w = hyp['input_img_size']
        self.use_tta = hyp['use_tta']
        self.iou_threshold = hyp['compute_metric_iou_threshold'] if compute_metric else hyp['iou_threshold']
        self.cls_threshold = hyp['compute_metric_cls_threshold'] if compute_metric else hyp['cls_threshold']
        self.conf_threshold = hyp['compute_metric_conf_threshold'] if compute_metric else hyp['conf_threshold']
        self._anchors_host = anchors.detach().cpu().tolist()

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def __call__(self, inputs):
        """:param inputs: (b, 3, h, w) -> list (len b) of FloatTensor (n, 6) [xmin, ymin, xmax, ymax, conf, cls] on CPU, or None"""
        if self.use_tta:
            merge_preds_out, _ = self.test_time_augmentation(inputs)
            if self.hyp.get("wfb", False):
                raise NotImplementedError("weighted box fusion is outside the HIP hot path (wfb: false in every shipped config)")
            outs = self.numba_nms(merge_preds_out)
        else:
            stage_preds = self.yolo(inputs)
            outs = self._nms_from_heads(stage_preds)
        return [torch.from_numpy(x) if x is not None else None for x in outs]

    def _desc(self, stage_preds):
        d = DecodeDesc()
        d.B = stage_preds[0].shape[0]
        d.num_class, d.num_anchor, d.num_stage = self.num_class, self.anchor_num, len(stage_preds)
        canon = []
        for s, p in enumerate(stage_preds):
            c, ld = to_cell_major(p)
            canon.append(c)
            d.H[s], d.W[s], d.ldp[s] = p.shape[2], p.shape[3], ld
            d.stride[s] = float(self.ds_scales[s])
            for a in range(self.anchor_num):
                d.anchors[(s * 3 + a) * 2], d.anchors[(s * 3 + a) * 2 + 1] = self._anchors_host[s][a]
        if len({c.dtype for c in canon}) != 1:
            canon = [to_cell_major(c.float())[0] for c in canon]
        d.pred_is_f32 = int(canon[0].dtype == torch.float32)
        d.yolox = 0
        ptrs = (C.c_void_p * 4)(*[c.data_ptr() for c in canon], *([None] * (4 - len(canon))))
        return d, canon, ptrs

    @torch.no_grad()
    def do_inference(self, inputs):
        """forward + decode: (bs, sum_s A*h*w, 5+nc) float32 in the reference's order
        (stage small->large, anchor, y, x) (:182-209)"""
        stage_preds = self.yolo(inputs)
        return self.decode(stage_preds)

    def decode(self, stage_preds):
        if not stage_preds[0].is_cuda:
            raise _lib.YoloHipError("YOLOV5Evaluator: tensors must live on an MI355X device")
        d, canon, ptrs = self._desc(stage_preds)
        n = sum(self.anchor_num * p.shape[2] * p.shape[3] for p in stage_preds)
        out = torch.empty(d.B, n, 5 + self.num_class, dtype=torch.float32, device=stage_preds[0].device)
        check(lib().yh_decode_full(C.byref(d), ptrs, out.data_ptr(), _lib.stream_ptr()), "yh_decode_full")
        return out

    def _run_nms(self, cand, ncand, B, cap):
        dev = cand.device
        L = lib()
        max_keep = int(self.hyp['max_predictions_per_img'])
        out = torch.empty(B, max_keep, 6, dtype=torch.float32, device=dev)
        nkeep = torch.zeros(B, dtype=torch.int32, device=dev)
        keep = torch.empty(B, max_keep, dtype=torch.int32, device=dev)
        ws = torch.empty(L.yh_nms_ws_bytes(B, cap), dtype=torch.uint8, device=dev)
        check(L.yh_nms_batched(cand.data_ptr(), ncand.data_ptr(), B, cap, float(self.iou_threshold),
                               int(bool(self.hyp['agnostic'])), 1, max_keep, int(bool(self.hyp['postprocess_bbox'])),
                               out.data_ptr(), nkeep.data_ptr(), keep.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "yh_nms_batched")
        nc_h = ncand.cpu().tolist()
        nk_h = nkeep.cpu().tolist()
        self.last_ncand = nc_h                 # candidates that entered NMS per image (bench.py reports boxes/s from it)
        out_h = out.cpu().numpy()
        return [None if nc_h[b] == 0 else out_h[b, :nk_h[b]].copy() for b in range(B)]

    def _nms_from_heads(self, stage_preds):
        """fused decode + filter + NMS straight from the head tensors (no decoded tensor is materialised)"""
        if self.hyp.get('mutil_label', False):          # one candidate per (prediction, class): through the decoded tensor (:276-279)
            return self.numba_nms(self.decode(stage_preds))
        d, canon, ptrs = self._desc(stage_preds)
        dev = stage_preds[0].device
        B = d.B
        n = sum(self.anchor_num * p.shape[2] * p.shape[3] for p in stage_preds)
        cap = min(((n + 3) // 4) * 4, 16384)
        while True:
            cand = torch.empty(B, cap, 6, dtype=torch.float32, device=dev)
            ncand = torch.zeros(B, dtype=torch.int32, device=dev)
            check(lib().yh_decode_filter(C.byref(d), ptrs, float(self.conf_threshold), float(self.cls_threshold),
                                         cand.data_ptr(), ncand.data_ptr(), cap, _decode_ws(self, d, dev), _lib.stream_ptr()), "yh_decode_filter")
            if cap >= n or int(ncand.max().item()) <= cap:
                break
            cap = ((n + 3) // 4) * 4
        return self._run_nms(cand, ncand, B, cap)

    _FILTER_MODES = (0, 2)        # yh_filter_decoded mode of the single-label / hyp['mutil_label'] candidate rule (:266-286)

    def _filter_decoded(self, preds_out):
        """decoded (bs, N, 5+nc) -> candidate tables (bs, cap, 6) [xmin, ymin, xmax, ymax, conf, cls] + counts, on the device:
        obj >= conf threshold, cls * obj, arg-max > cls threshold or — hyp['mutil_label'] — one candidate per (prediction, class)
        in np.nonzero order (:266-286; yh_filter_decoded)"""
        p = preds_out.detach().to(torch.float32).contiguous()
        if not p.is_cuda:
            p = p.to(self.device if str(self.device).startswith("cuda") else "cuda:0")
        B, n, E = p.shape
        assert E == 5 + self.num_class
        cap = ((n + 3) // 4) * 4
        multi = bool(self.hyp.get('mutil_label', False))
        mode = self._FILTER_MODES[1 if multi else 0]
        while True:
            cand = torch.empty(B, cap, 6, dtype=torch.float32, device=p.device)
            ncand = torch.zeros(B, dtype=torch.int32, device=p.device)
            check(lib().yh_filter_decoded(p.data_ptr(), B, n, self.num_class, float(self.conf_threshold), float(self.cls_threshold), mode,
                                          cand.data_ptr(), ncand.data_ptr(), cap, _lib.stream_ptr()), "yh_filter_decoded")
            most = int(ncand.max().item()) if multi else 0          # multi-label: up to num_class candidates per prediction
            if most <= cap:
                break
            cap = ((most + 3) // 4) * 4
        return cand, ncand, B, cap

    def do_nms(self, preds_out):
        """"Do NMS with torch" (trainer/eval_yolov5.py:94-150): decoded (bs, X, 5+nc) -> list of (n, 6) device tensors or None.
        Candidate rule as numba_nms (yh_filter_decoded); suppression by utils.gpu_nms — hyp['iou_type'] one of iou / giou /
        diou / ciou, EXCLUSIVE threshold — on boxes offset by cls * 4096 when hyp['agnostic'] is set; at most
        max_predictions_per_img rows; with hyp['postprocess_bbox'] a kept box survives only if MORE than one candidate
        overlaps it by > threshold under `bbox_iou` (the unclamped IoU below; the box merge the reference computes there is
        written to a temporary it never returns, :143, so the rows are the candidates' own).  The reference's gpu_nms raises
        IndexError for more than one candidate (utils/nms.py:62-63 indexes a 1-D score tensor with a (1, M) mask); this follows
        the semantics its loop spells — the single-candidate and empty cases agree with it literally."""
        from ..utils.nms import gpu_nms
        cand, ncand, B, _cap = self._filter_decoded(preds_out)
        outputs = []
        for i, bbox_num in enumerate(ncand.cpu().tolist()):
            if not bbox_num:
                outputs.append(None)
                continue
            x = cand[i, :bbox_num]
            box_offset = x[:, 5] * 4096 if self.hyp['agnostic'] else x[:, 5] * 0.
            bboxes_offseted = x[:, :4] + box_offset[:, None]
            keep_index = gpu_nms(bboxes_offseted, x[:, 4].contiguous(), self.hyp['iou_type'], self.iou_threshold)
            if len(keep_index) > self.hyp['max_predictions_per_img']:
                keep_index = keep_index[:self.hyp['max_predictions_per_img']]
            if self.hyp['postprocess_bbox'] and 1 < bbox_num < 3000:
                iou = self.bbox_iou(bboxes_offseted[keep_index], bboxes_offseted)      # (N, M)
                keep_index = torch.tensor(keep_index)[((iou > self.iou_threshold).float().sum(dim=1) > 1).cpu()]
            outputs.append(x[keep_index])
        return outputs

    @staticmethod
    def bbox_iou(bbox1, bbox2):
        """(N, 4), (M, 4) xyxy -> (N, M) IoU exactly as trainer/eval_yolov5.py:237-258 spells it: NO clamp on the intersection
        sides (two boxes apart on BOTH axes get a positive "intersection", the product of two negative sides) and none on the
        union (0 / 0 = NaN).  yh_iou_matrix with a negative clamp."""
        assert bbox1.ndim == 2
        assert bbox2.ndim == 2
        from ..utils.bbox_tools import _iou_matrix
        return _iou_matrix(bbox1, bbox2, -1.0)

    def numba_nms(self, preds_out):
        """:param preds_out: decoded (bs, N, 5+nc) tensor -> list of np.ndarray (n,6) or None (:261-317)"""
        return self._run_nms(*self._filter_decoded(preds_out))

    def test_time_augmentation(self, inputs):
        """3 passes (1.0/none, 0.83/flip-y, 0.67/flip-x), un-scaled / un-flipped, concatenated (:152-179)"""
        img_h, img_w = inputs.size(2), inputs.size(3)
        aug_preds = []
        for s, f in zip([1, 0.83, 0.67], [None, 2, 3]):
            img = inputs.flip(dims=(f,)) if f else inputs
            img = self.scale_img(img, s)
            ripe = self.do_inference(img)
            ripe[..., :4] /= s
            if f == 2:
                ripe[..., 1] = img_h - ripe[..., 1]
            if f == 3:
                ripe[..., 0] = img_w - ripe[..., 0]
            aug_preds.append(ripe)
        return torch.cat(aug_preds, dim=1).contiguous(), aug_preds

    @staticmethod
    def scale_img(img, scale_factor):
        """:160-227 bilinear down-scale then pad to a multiple of 32 with 0.447"""
        if scale_factor == 1.0:
            return img
        h, w = img.shape[2], img.shape[3]
        new_h, new_w = int(scale_factor * h), int(scale_factor * w)
        img = F.interpolate(img, size=(new_h, new_w), align_corners=False, mode='bilinear')
        out_h, out_w = int(np.ceil(h / 32) * 32), int(np.ceil(w / 32) * 32)
        return F.pad(img, [0, out_w - new_w, 0, out_h - new_h], value=0.447)

    def make_grid(self, row_num, col_num):
        y, x = torch.meshgrid([torch.arange(row_num, device=self.device), torch.arange(col_num, device=self.device)], indexing='ij')
        return torch.stack((x, y), dim=2).reshape(row_num, col_num, 2)[None, None, ...].contiguous()
