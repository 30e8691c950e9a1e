"""mAP_v2 — mirror of the reference's COCO-style metric (utils/mAP.py:44-276): per-image TP matching at the
10 IoU thresholds 0.50:0.05:0.95 (:70-100), per-class AP with 101-point interpolation (:171-189),
precision / recall / F1 at the confidence that maximises the smoothed mean F1 (:145-163).
Host-side NumPy bookkeeping on the evaluator's CPU outputs, as in the reference; the curve plots are optional."""
import numpy as np

__all__ = ['mAP_v2', 'iou_np', 'smooth']


def smooth(y, f=0.05):
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode='valid')


def iou_np(box1, box2):
    """(M,4),(N,4) xyxy -> (M,N), union clipped to [1e-6, 1e7] (utils/mAP.py:18-42)"""
    box1 = np.expand_dims(box1, axis=1)
    a1 = np.prod(box1[..., [2, 3]] - box1[..., [0, 1]], axis=-1)
    a2 = np.prod(box2[:, [2, 3]] - box2[:, [0, 1]], axis=-1)
    w = np.maximum(0., np.minimum(box1[..., 2], box2[:, 2]) - np.maximum(box1[..., 0], box2[:, 0]))
    h = np.maximum(0., np.minimum(box1[..., 3], box2[:, 3]) - np.maximum(box1[..., 1], box2[:, 1]))
    inter = w * h
    return inter / np.clip(a1 + a2 - inter, a_min=1e-6, a_max=10000000)


class mAP_v2:

    def __init__(self, ground_truth, predict, plot_save_dir=None, type='coco'):
        """predict: list of (M,6) [xmin,ymin,xmax,ymax,conf,cls]; ground_truth: list of (N,5) [xmin,ymin,xmax,ymax,cls].
        Images whose ground truth or prediction list is empty are dropped (:59-63)."""
        assert len(predict) == len(ground_truth)
        self.pred, self.gt = [], []
        for i in range(len(ground_truth)):
            if len(ground_truth[i]) > 0 and len(predict[i]) > 0:
                self.gt.append(np.asarray(ground_truth[i]))
                self.pred.append(np.asarray(predict[i]))
        self.iou_thr = np.linspace(0.5, 0.95, 10)
        self.save_dir = plot_save_dir
        self.type = type

    def compute_tp(self, gt, pred):
        tp = np.zeros(shape=(pred.shape[0], len(self.iou_thr)), dtype=bool)
        ious = iou_np(gt[:, :4], pred[:, :4])
        mask = (ious >= self.iou_thr[0]) & (gt[:, [4]] == pred[:, 5])
        if mask.sum() > 0:
            gt_i, pred_i = np.nonzero(mask)
            match = np.concatenate((np.stack((gt_i, pred_i), axis=1), ious[mask][:, None]), axis=1)
            if mask.sum() > 1:
                match = match[match[:, 2].argsort()[::-1]]
                match = match[np.unique(match[:, 1], return_index=True)[1]]      # one gt per prediction
                match = match[np.unique(match[:, 0], return_index=True)[1]]      # one prediction per gt
            tp[match[:, 1].astype(np.int32)] = match[:, [2]] >= self.iou_thr
        return tp

    def compute_ap(self, recall, precision, type):
        rec = np.concatenate(([0.], recall, [1.]))
        pre = np.concatenate(([1.], precision, [0.]))
        pre = np.flip(np.maximum.accumulate(np.flip(pre)))
        if type == 'coco':
            xs = np.linspace(0, 1, 101)
            ys = np.interp(xs, rec, pre)
            ap = np.sum((ys[1:] + ys[:-1]) / 2 * np.diff(xs))                    # np.trapz
        else:
            i = np.where(rec[1:] != rec[:-1])[0]
            ap = np.sum((rec[i + 1] - rec[i]) * pre[i + 1])
        return ap, rec, pre

    def compute_ap_per_class(self):
        tps = np.concatenate([self.compute_tp(g, p) for g, p in zip(self.gt, self.pred)], axis=0)
        pred_all = np.concatenate(self.pred, axis=0)
        gt_all = np.concatenate(self.gt, axis=0)
        conf, pcls, tcls = pred_all[:, 4], pred_all[:, 5], gt_all[:, 4]
        sort_i = np.argsort(conf)[::-1]
        stp, scof, scls = tps[sort_i], conf[sort_i], pcls[sort_i]
        tot_cls = np.unique(tcls)
        ap = np.zeros((len(tot_cls), stp.shape[1]))
        precision = np.zeros(shape=[len(tot_cls), 1000])
        recall = np.zeros(shape=[len(tot_cls), 1000])
        xs = np.linspace(0, 1, 1000)
        for i, c in enumerate(tot_cls):
            mi = scls == c
            num_tar = (tcls == c).sum()
            if mi.sum() > 0 and num_tar > 0:
                cfp = (~stp[mi]).cumsum(0)
                ctp = stp[mi].cumsum(0)
                crec = ctp / (num_tar + 1e-16)
                cpre = ctp / (ctp + cfp + 1e-16)
                recall[i] = np.interp(-xs, -scof[mi], crec[:, 0], left=0)
                precision[i] = np.interp(-xs, -scof[mi], cpre[:, 0], left=1)
                for j in range(stp.shape[1]):
                    ap[i, j], _, _ = self.compute_ap(crec[:, j], cpre[:, j], self.type)
        f1 = 2 * precision * recall / (precision + recall + 1e-16)
        best_i = smooth(f1.mean(0), 0.1).argmax()
        return {"precision": precision[:, best_i], "recall": recall[:, best_i], "ap": ap, "f1": f1[:, best_i], "unique_cls": tot_cls}

    def get_mean_metrics(self):
        """-> (mAP@[.5:.95], mAP@.5, mean precision, mean recall)  (:263-276)"""
        try:
            m = self.compute_ap_per_class()
            ap = m['ap']
            return ap.mean(axis=1).mean(), ap[:, 0].mean(), m['precision'].mean(), m['recall'].mean()
        except Exception as err:     # the reference swallows errors the same way (:272-275)
            print(err)
            return 0., 0., 0., 0.
