"""ORACLE (test infrastructure, never imported by the product path).

torch-CPU fp32 restatement of the reference's YOLOX loss with SimOTA assignment,
/root/reference loss/yolox_loss.py:11-458, including its observable quirks:
  * `tars[..., :4]` is converted to xywh IN PLACE on the caller's tensor (:42);
  * inside label_assign the class/objectness part of the cost is evaluated on a zero tensor
    (`preds_ = zeros_like(preds)` only receives the decoded box, :111-113), so the class cost is the
    same constant for every (gt, candidate) pair and the assignment is driven by IoU + geometry;
  * the bare `torch.no_grad()` at :92 is an expression, not a decorator: matched_iou (hence the
    class TARGET) carries gradient back into the box predictions (:150);
  * select_grid's fallback for "no cell inside any gt box" uses torch.randperm (:270-278) and is
    therefore not reproducible; inputs that reach it are outside the pinned domain.
Pinned by tests/golden/g8_yolox.npz (generated from the reference, tools/gen_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _xywh2xyxy(b):
    x, y, w, h = b.chunk(4, -1)
    return torch.cat((x - w / 2, y - h / 2, x + w / 2, y + h / 2), -1)


def _gpu_iou(b1, b2):
    a1 = torch.prod(b1[:, [2, 3]] - b1[:, [0, 1]], dim=-1)
    a2 = torch.prod(b2[:, [2, 3]] - b2[:, [0, 1]], dim=-1)
    ymax = torch.min(b1[:, None, 3], b2[None, :, 3]); xmax = torch.min(b1[:, None, 2], b2[None, :, 2])
    ymin = torch.max(b1[:, None, 1], b2[None, :, 1]); xmin = torch.max(b1[:, None, 0], b2[None, :, 0])
    w = torch.clamp(xmax - xmin, min=0.0); h = torch.clamp(ymax - ymin, min=0.0)
    inter = w * h
    return inter / (a1[:, None] + a2[None, :] - inter).clamp(1e-9)


class YOLOXLossOracle:
    def __init__(self, hyp, stable_ties=False):
        # stable_ties: resolve exact cost ties towards the lower candidate index (what the HIP kernel does); the
        # reference's torch.topk leaves the choice among EQUAL costs to libstdc++'s introselect, which is not part
        # of the contract — golden inputs are chosen such that both variants agree (tools/gen_golden.py)
        self.stable_ties = stable_ties
        self.hyp = hyp
        self.nc = hyp['num_class']
        self.A = hyp['num_anchors']
        self.use_l1 = hyp.get('use_l1', True)
        self.balances = [4., 1., 0.4]

    # ---- select_grid (:235-303) -----------------------------------------------------------------
    def select_grid(self, tar_box, grid, stride):
        eps = 1e-9
        gt = tar_box.clone().detach()
        offs = gt.new_tensor([-1, -1, 1, 1]).unsqueeze(0) * 0.5
        gt_xyxy = gt[:, :2].repeat(1, 2) + gt[:, 2:].repeat(1, 2) * offs
        gt_xyxy = gt_xyxy * gt_xyxy.new_tensor([-1, -1, 1, 1]).unsqueeze(0)
        ctr = (grid + 0.5) * stride
        sctr = ctr.repeat(1, 2) * ctr.new_tensor([1, 1, -1, -1]).unsqueeze(0)
        in_box = (gt_xyxy.unsqueeze(1) + sctr.unsqueeze(0)).min(2).values > eps
        in_box_all = in_box.sum(0) > eps
        if in_box_all.sum() == 0:
            raise RuntimeError("select_grid random fallback (reference :270-278) is outside the oracle's pinned domain")
        coff = gt.new_tensor([-1, -1, 1, 1]) * self.hyp['center_radius']
        gco = gt[:, :2].repeat(1, 2) + coff.unsqueeze(0)
        gco = gco * gco.new_tensor([-1, -1, 1, 1]).unsqueeze(0)
        in_ctr = (sctr.unsqueeze(0) + gco.unsqueeze(1)).min(2).values > eps
        in_ctr_all = in_ctr.sum(0) > eps
        if in_ctr_all.sum() == 0:
            in_ctr_all = in_box_all
        either = (in_box_all.float() + in_ctr_all.float()) > eps
        both = (in_box[:, either].float() + in_ctr[:, either].float()) > 1.
        return either, both

    # ---- simple_ota (:305-359) ------------------------------------------------------------------
    def simple_ota(self, cost, iou, fg_mask):
        mm = torch.zeros_like(cost, dtype=torch.uint8)
        k = min(self.hyp['topk'], iou.size(1))
        topk_iou = torch.topk(iou, k, dim=1)[0]
        dyn_k = torch.clamp(topk_iou.sum(1).int(), 1, cost.size(1)).tolist()
        for i in range(cost.size(0)):
            if self.stable_ties:
                pos = torch.sort(cost[i], stable=True)[1][:dyn_k[i]]
            else:
                pos = torch.topk(cost[i], k=dyn_k[i], largest=False)[1]
            mm[i][pos] = 1
        allm = mm.sum(0)
        if allm.max() > 1:
            amin = torch.min(cost[:, allm > 1], dim=0)[1]
            mm[:, allm > 1] = 0
            mm[amin, allm > 1] = 1
        fg = mm.sum(0) > 0
        num_fg = fg.sum().item()
        fg_mask[fg_mask.clone()] = fg
        matched_gt = mm[:, fg].argmax(0)
        matched_iou = (mm * iou).sum(0)[fg]
        return fg_mask, num_fg, matched_iou, matched_gt

    # ---- label_assign (:93-178) -----------------------------------------------------------------
    def label_assign(self, tars, preds, grid, stride):
        tcls, tbox, tcof, fgs, tl1 = [], [], [], [], []
        tot_gt, tot_fg = 0, 0
        preds_ = torch.zeros_like(preds)
        preds_[..., :2] = (preds[..., :2] + grid[None]) * stride
        preds_[..., 2:4] = torch.exp(preds[..., 2:4]) * stride
        for i in range(tars.size(0)):
            tar, pred = tars[i], preds_[i]
            gm = tar[:, 4] >= 0
            tot_gt += int(gm.sum())
            if gm.sum() == 0:
                c_i = tar.new_zeros((0, self.nc)); b_i = tar.new_zeros((0, 4))
                o_i = tar.new_zeros((pred.size(0), 1)); fgm = tar.new_zeros(pred.size(0)).bool(); l1_i = tar.new_zeros((0, 4))
            else:
                b_i = tar[gm, :4]
                c_i = F.one_hot(tar[gm, 4].long(), num_classes=self.nc) * self.hyp['class_smooth_factor']
                fgm, both = self.select_grid(b_i, grid, stride)
                pb = pred[fgm, :4]
                iou = _gpu_iou(_xywh2xyxy(b_i), _xywh2xyxy(pb))
                iou_l = -torch.log(iou + 1e-9)
                pco = torch.sigmoid(pred[fgm, 4]).unsqueeze(1)
                pcl = torch.sigmoid(pred[fgm, 5:])
                pcl = torch.sqrt((pcl * pco).unsqueeze(0).expand(c_i.size(0), -1, -1))
                tcl = c_i.unsqueeze(1).expand(-1, pcl.size(1), -1)
                cls_l = -(tcl * torch.log(pcl) + (1 - tcl) * torch.log(1 - pcl)).sum(-1)
                cost = cls_l.detach() + 3 * iou_l.detach() + 100000 * (~both)
                fgm, nfg, miou, mgt = self.simple_ota(cost, iou, fgm.clone())
                tot_fg += nfg
                c_i = c_i[mgt] * miou.unsqueeze(-1)            # carries grad into the box predictions (quirk)
                b_i = b_i[mgt]
                o_i = fgm.unsqueeze(-1).float()
                l1_i = tar.new_zeros((nfg, 4))
                if self.use_l1:
                    l1_i[:, 0] = b_i[:, 0] / stride - grid[fgm, 0]
                    l1_i[:, 1] = b_i[:, 1] / stride - grid[fgm, 1]
                    l1_i[:, 2] = torch.log(b_i[:, 2] / stride + 1e-16)
                    l1_i[:, 3] = torch.log(b_i[:, 3] / stride + 1e-16)
            tcls.append(c_i); tbox.append(b_i); tcof.append(o_i); fgs.append(fgm); tl1.append(l1_i)
        return torch.cat(tbox, 0), torch.cat(tcof, 0), torch.cat(tcls, 0), torch.cat(tl1, 0), torch.cat(fgs, 0), tot_fg, tot_gt

    def focal(self, pred, target):
        prob = torch.sigmoid(pred)
        acc = target * prob + (1.0 - target) * (1.0 - prob)
        return (1.0 - acc) ** self.hyp.get('focal_loss_gamma', 1.5) * (
            target * self.hyp.get('focal_loss_alpha', 0.25) + (1.0 - target) * (1.0 - self.hyp.get('focal_loss_alpha', 0.25)))

    def iou_loss(self, pb, tb, iou_type):
        eps = 1e-9
        x1, y1, w1, h1 = pb.chunk(4, -1); x2, y2, w2, h2 = tb.chunk(4, -1)
        ax0, ay0, ax1, ay1 = x1 - w1 / 2, y1 - h1 / 2, x1 + w1 / 2, y1 + h1 / 2
        bx0, by0, bx1, by1 = x2 - w2 / 2, y2 - h2 / 2, x2 + w2 / 2, y2 + h2 / 2
        union = (w1 * h1).clamp(0.) + (w2 * h2).clamp(0.)
        inter = (ax1.minimum(bx1) - ax0.maximum(bx0)).clamp(0.) * (ay1.minimum(by1) - ay0.maximum(by0)).clamp(0.)
        iou = inter / (union - inter + eps)
        if iou_type == 'iou':
            return 1 - iou ** 2
        if iou_type == 'giou':
            convex = (ax1.maximum(bx1) - ax0.minimum(bx0)).clamp(0.) * (ay1.maximum(by1) - ay0.minimum(by0)).clamp(0.)
            return 1 - (iou - torch.abs(convex - union) / (convex + eps)).clamp(min=-1., max=1.)
        c_hs = (ay1.maximum(by1) - ay0.minimum(by0)).clamp(0.)
        c_ws = (ax1.maximum(bx1) - ax0.minimum(bx0)).clamp(0.)
        c_d = torch.pow(c_ws, 2) + torch.pow(c_hs, 2) + eps
        ctr = (x1 - x2) ** 2 + (y1 - y2) ** 2
        v = (4 / (math.pi ** 2)) * (torch.atan(w1 / h1) - torch.atan(w2 / h2)) ** 2
        with torch.no_grad():
            alpha = v / (1 - iou + v).clamp(eps)
        return 1 - (iou - ctr / c_d - v * alpha)

    def stage(self, tars, preds, grid, stride):
        tbox, tcof, tcls, tl1, fg, nfg, ngt = self.label_assign(tars, preds, grid, stride)
        hyp = self.hyp
        if self.use_l1:
            l1 = F.l1_loss(preds[..., :4].reshape(-1, 4)[fg], tl1, reduction='none').mean(-1)
        else:
            l1 = preds.new_tensor([0.0])
        dec = torch.cat(((preds[..., :2] + grid[None]) * stride, torch.exp(preds[..., 2:4]) * stride, preds[..., 4:]), -1)
        nfg = max(nfg, 1)
        iou_l = self.iou_loss(dec[..., :4].reshape(-1, 4)[fg], tbox, hyp['iou_type'])
        pw_o = torch.tensor(float(hyp.get('cof_pos_weight', 1.))); pw_c = torch.tensor(float(hyp.get('cls_pos_weight', 1.)))
        obj = dec[..., 4].reshape(-1, 1)
        cof = F.binary_cross_entropy_with_logits(obj, tcof, pos_weight=pw_o, reduction='none')
        if hyp['use_focal_loss']:
            cof = cof * self.focal(obj, tcof)
        pc = dec[..., 5:].reshape(-1, self.nc)[fg]
        cls = F.binary_cross_entropy_with_logits(pc, tcls, pos_weight=pw_c, reduction='none')
        if hyp['use_focal_loss']:
            cls = cls * self.focal(pc, tcls)
        cls = cls.mean(-1)
        return dict(iou_loss=iou_l.sum() / nfg, l1_loss=l1.sum() / nfg, cls_loss=cls.sum() / nfg, cof_loss=cof.sum() / nfg,
                    num_fg=nfg, num_gt=ngt, fg=fg)

    def __call__(self, preds, tars):
        """preds: dict name -> (N, A, 5+nc, h, w); tars: (N, nbox, 6) xyxy, converted to xywh in place"""
        hyp = self.hyp
        b = tars[..., :4].clone()
        tars[..., 0:2] = (b[..., 0:2] + b[..., 2:4]) / 2
        tars[..., 2:4] = b[..., 2:4] - b[..., 0:2]
        nfg, ngt = 0, 0
        tc = torch.zeros(1); ti = torch.zeros(1); to = torch.zeros(1); tl = torch.zeros(1)
        self.last_fg = []
        for i, k in enumerate(preds.keys()):
            h, w = preds[k].shape[-2:]
            stride = hyp['input_img_size'][0] / h
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
            grid = torch.stack((xs, ys), dim=2).float().unsqueeze(0).expand(self.A, -1, -1, -1).reshape(-1, 2)
            p = preds[k].permute(0, 1, 3, 4, 2).contiguous().reshape(tars.size(0), self.A * h * w, -1).float()
            out = self.stage(tars.float(), p, grid, stride)
            self.last_fg.append(out['fg'])
            tmp = out['cof_loss'] * self.balances[i]
            self.balances[i] = self.balances[i] * 0.9999 + 0.0001 / tmp.detach().item()
            to = to + tmp
            nfg += out['num_fg']; ngt += out['num_gt']
            tc = tc + out['cls_loss']; ti = ti + out['iou_loss']; tl = tl + out['l1_loss']
        self.balances = [x / self.balances[1] for x in self.balances]
        ti = ti * hyp.get('iou_loss_scale', 0.5); tc = tc * hyp.get('cls_loss_scale', 1.0)
        to = to * hyp.get('cof_loss_scale', 1.0); tl = tl * hyp.get('l1_loss_scale', 1.0)
        tot = ti + tc + to + tl
        return {'tot_loss': tot, 'iou_loss': ti.item(), 'l1_loss': tl.item(), 'cls_loss': tc.item(), 'cof_loss': to.item(),
                'fg_nums': int(nfg), 'tar_nums': int(ngt)}
