"""Synthetic detection dataset in the reference's __getitem__ format (image (h,w,3) uint8, annotation dict, id):
there is no dataset on the build / bench machines, so the drop-in drivers run on this (SURVEY §8d synthetic inputs:
1..20 boxes per image, log-uniform sizes, 80 classes)."""
import numpy as np

__all__ = ['SyntheticDetectionDataset']


class SyntheticDetectionDataset:

    def __init__(self, length, img_hw=(480, 640), num_class=80, max_boxes=20, seed=0):
        self.length, self.img_hw, self.nc, self.max_boxes, self.seed = length, tuple(img_hw), num_class, max_boxes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        rs = np.random.RandomState(self.seed * 100003 + i)
        h, w = self.img_hw
        img = rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        n = int(rs.randint(1, self.max_boxes + 1))
        cx, cy = rs.uniform(0.1, 0.9, n) * w, rs.uniform(0.1, 0.9, n) * h
        bw = np.exp(rs.uniform(np.log(8), np.log(0.5 * w), n))
        bh = np.exp(rs.uniform(np.log(8), np.log(0.5 * h), n))
        x0, y0 = np.clip(cx - bw / 2, 0, w - 1), np.clip(cy - bh / 2, 0, h - 1)
        x1, y1 = np.clip(cx + bw / 2, 1, w), np.clip(cy + bh / 2, 1, h)
        bboxes = np.stack([x0, y0, np.maximum(x1, x0 + 1), np.maximum(y1, y0 + 1)], 1).astype(np.float32)
        classes = rs.randint(0, self.nc, n).tolist()
        return img, {'bboxes': bboxes, 'classes': classes}, i
