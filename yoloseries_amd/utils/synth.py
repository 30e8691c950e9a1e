"""Synthetic COCO-shaped inputs: the same generator feeds bench.py, the
tests and tools/gen_golden.py, so that fixtures only need to store seeds.
numpy.random.RandomState streams are frozen across NumPy versions."""
import numpy as np

COCO_ANCHORS = np.array([[[10, 13], [16, 30], [33, 23]],
                         [[30, 61], [62, 45], [59, 119]],
                         [[116, 90], [156, 198], [373, 326]]], dtype=np.float32)


def synth_targets(batch, img_size=640, num_class=80, max_boxes=20, seed=1, min_boxes=1):
    """(B, maxbox, 6) float32 [xmin,ymin,xmax,ymax,cls,img_idx], padded with -1
    (the collate format of dataset/data_collater.py:20-64)."""
    rs = np.random.RandomState(seed)
    counts = rs.randint(min_boxes, max_boxes + 1, size=batch)
    mb = int(counts.max()) if batch else 0
    out = -np.ones((batch, max(mb, 1), 6), dtype=np.float32)
    for b in range(batch):
        n = counts[b]
        cx = rs.uniform(0.1, 0.9, n) * img_size
        cy = rs.uniform(0.1, 0.9, n) * img_size
        w = np.exp(rs.uniform(np.log(8), np.log(img_size / 2), n))
        h = np.exp(rs.uniform(np.log(8), np.log(img_size / 2), n))
        x1 = np.clip(cx - w / 2, 0, img_size); x2 = np.clip(cx + w / 2, 0, img_size)
        y1 = np.clip(cy - h / 2, 0, img_size); y2 = np.clip(cy + h / 2, 0, img_size)
        out[b, :n, 0], out[b, :n, 1], out[b, :n, 2], out[b, :n, 3] = x1, y1, x2, y2
        out[b, :n, 4] = rs.randint(0, num_class, n)
        out[b, :n, 5] = b
    return out


def synth_head_outputs(batch, img_size=640, num_class=80, num_anchor=3, seed=3, scale=1.0, strides=(8, 16, 32)):
    """Random raw head tensors in the reference layout: list of (B, A*(5+nc), h, w) float32."""
    rs = np.random.RandomState(seed)
    return [(rs.randn(batch, num_anchor * (5 + num_class), img_size // s, img_size // s) * scale).astype(np.float32)
            for s in strides]


def synth_nms_heads(batch, img_size=640, num_class=80, num_anchor=3, seed=2, frac=0.01, clusters=50, strides=(8, 16, 32), wh_shift=0.0):
    """Head tensors whose decode yields ~`frac` of the anchors above conf 0.001, grouped in
    overlapping clusters: objectness logits are shifted far negative except on
    cells near `clusters` random centres; class logits ~N(0,1) with one boosted class."""
    rs = np.random.RandomState(seed)
    E = 5 + num_class
    outs = []
    centres = rs.uniform(0.1, 0.9, (batch, clusters, 2)) * img_size
    ccls = rs.randint(0, num_class, (batch, clusters))
    for s in strides:
        h = w = img_size // s
        t = rs.randn(batch, num_anchor, E, h, w).astype(np.float32)
        t[:, :, 4] -= 12.0
        if wh_shift:
            t[:, :, 2:4] = t[:, :, 2:4] * 0.5 + wh_shift       # boxes of about (2*sigmoid(wh_shift))^2 anchors: distinct objects survive NMS
        n_pick = max(1, int(frac * num_anchor * h * w / clusters))
        for b in range(batch):
            for k in range(clusters):
                gx = int(np.clip(centres[b, k, 0] / s, 0, w - 1)); gy = int(np.clip(centres[b, k, 1] / s, 0, h - 1))
                for _ in range(n_pick):
                    a = rs.randint(num_anchor)
                    yy = int(np.clip(gy + rs.randint(-1, 2), 0, h - 1)); xx = int(np.clip(gx + rs.randint(-1, 2), 0, w - 1))
                    t[b, a, 4, yy, xx] = rs.uniform(0.0, 4.0)
                    t[b, a, 5 + ccls[b, k], yy, xx] += 4.0
        outs.append(t.reshape(batch, num_anchor * E, h, w))
    return outs


def synth_yolox_heads(batch, img_size=640, num_class=80, seed=5, scale=1.0, strides=(8, 16, 32)):
    """Random YOLOX head tensors in the reference layout: OrderedDict pred_s/m/l of (B, 1, 5+nc, h, w) float32.
    Box logits are kept small (xy ~ N(0.5, 0.5) cells, wh ~ N(log 3, 0.5)) so that IoUs with the targets are non-trivial."""
    from collections import OrderedDict
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for name, s in zip(("pred_s", "pred_m", "pred_l"), strides):
        h = w = img_size // s
        t = (rs.randn(batch, 1, 5 + num_class, h, w) * scale).astype(np.float32)
        t[:, :, 0:2] = (0.5 + 0.5 * rs.randn(batch, 1, 2, h, w)).astype(np.float32)
        t[:, :, 2:4] = (np.log(3.0) + 0.5 * rs.randn(batch, 1, 2, h, w)).astype(np.float32)
        out[name] = t
    return out


# learnable synthetic detection task: filled rectangles on a dark noisy background, colour = class.  Used by the training
# drivers (--data shapes) and tests/test_gpu_driver.py to show that the whole stack (forward, loss, backward, optimizer, EMA,
# evaluator, mAP) LEARNS — random-noise images (synth_targets alone) can only check that it runs.
SHAPE_PALETTE = np.array([[1.0, 0.1, 0.1], [0.1, 1.0, 0.1], [0.15, 0.25, 1.0], [1.0, 1.0, 0.1],
                          [1.0, 0.1, 1.0], [0.1, 1.0, 1.0], [1.0, 0.55, 0.1], [0.9, 0.9, 0.9]], dtype=np.float32)


def synth_shapes_batch(batch, img_size=320, num_class=4, max_boxes=4, seed=1):
    """-> (img (B,3,H,W) float32 in [0,1], ann (B, max_boxes, 6) float32 [xmin,ymin,xmax,ymax,cls,img_idx] padded with -1).
    Boxes do not overlap by more than a little (rejection sampling), sizes 1/8 .. 1/2 of the image, class < min(num_class, 8)."""
    rs = np.random.RandomState(seed)
    nc = min(num_class, len(SHAPE_PALETTE))
    img = rs.uniform(0.0, 0.25, size=(batch, 3, img_size, img_size)).astype(np.float32)
    ann = -np.ones((batch, max_boxes, 6), dtype=np.float32)
    for b in range(batch):
        placed = []
        for _ in range(rs.randint(1, max_boxes + 1)):
            for _try in range(20):
                w, h = rs.uniform(img_size / 8, img_size / 2, 2)
                x1, y1 = rs.uniform(0, img_size - w), rs.uniform(0, img_size - h)
                box = np.array([x1, y1, x1 + w, y1 + h])
                ok = True
                for q in placed:
                    iw = min(box[2], q[2]) - max(box[0], q[0])
                    ih = min(box[3], q[3]) - max(box[1], q[1])
                    if iw > 0 and ih > 0 and iw * ih > 0.05 * min(w * h, (q[2] - q[0]) * (q[3] - q[1])):
                        ok = False
                        break
                if ok:
                    break
            else:
                continue
            c = rs.randint(0, nc)
            xi1, yi1, xi2, yi2 = int(round(box[0])), int(round(box[1])), int(round(box[2])), int(round(box[3]))
            img[b, :, yi1:yi2, xi1:xi2] = SHAPE_PALETTE[c][:, None, None] * rs.uniform(0.8, 1.0)
            ann[b, len(placed)] = [xi1, yi1, xi2, yi2, c, b]
            placed.append(box)
    return img, ann
