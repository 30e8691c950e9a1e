"""Letterbox geometry: the transform between a dataset image and the (3, H, W) network input, and its record
(`scale`, `pad_top/left/bottom/right`, `org_shape`) that the evaluators use to map predictions back.  Behaviour of the
reference's utils/data_aug.py:21-70 (letter_resize_img) and utils/bbox_tools.py:38-49 (letter_resize_bbox).

The reference resizes with OpenCV (`cv2.resize(..., interpolation=0)`, INTER_NEAREST) and pads with
`cv2.copyMakeBorder`; OpenCV is not a dependency here: nearest sampling is restated from its index rule
``src = min(floor(dst * src_size / dst_size), src_size - 1)`` and the border is a constant fill."""
import numpy as np

__all__ = ['letter_resize_img', 'letter_resize_bbox', 'resize_nearest']


def resize_nearest(img, resize_w, resize_h):
    """nearest-neighbour resize with OpenCV's INTER_NEAREST index rule"""
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(resize_h) * (h / resize_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(resize_w) * (w / resize_w)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def _round_up(v, stride):
    rem = int(np.remainder(v, stride))
    return int(v + (stride - rem if rem > 0 else 0))


def letter_resize_img(img, dst_size, stride=64, fill_value=128, only_ds=False, training=True):
    """Scale `img` (h, w, 3) uint8 by one factor so that it fits `dst_size` (int or [h, w], rounded up to a multiple of
    `stride`) and pad with `fill_value`.

    training=True: the output always has the full target size (batches need one shape), padding split evenly, the odd
    pixel at the bottom / right.  training=False: only as much padding as the next multiple of `stride` needs.
    only_ds=True never enlarges.  Returns (uint8 image, record dict)."""
    target = [dst_size, dst_size] if isinstance(dst_size, int) else list(dst_size)
    target = [_round_up(target[0], stride), _round_up(target[1], stride)]
    src_h, src_w = img.shape[:2]
    scale = float(np.min([target[0] / src_h, target[1] / src_w]))
    if only_ds:
        scale = min(scale, 1.0)
    if scale != 1.:
        new_h, new_w = int(src_h * scale), int(src_w * scale)
        body = resize_nearest(img, new_w, new_h)
    else:
        new_h, new_w = src_h, src_w
        body = img
    slack_h, slack_w = target[0] - new_h, target[1] - new_w
    if training:
        top, left = slack_h // 2, slack_w // 2
        out_h, out_w = target
    else:
        slack_h, slack_w = int(np.remainder(slack_h, stride)), int(np.remainder(slack_w, stride))
        top, left = int(round(slack_h / 2)), int(round(slack_w / 2))
        out_h, out_w = new_h + slack_h, new_w + slack_w
    bottom, right = slack_h - top, slack_w - left
    canvas = np.empty((out_h, out_w, 3), dtype=np.uint8)
    canvas[...] = np.asarray(fill_value, dtype=np.int64).astype(np.uint8)
    canvas[top:top + new_h, left:left + new_w] = body
    record = {'scale': scale, 'pad_top': top, 'pad_left': left, 'pad_bottom': bottom, 'pad_right': right,
              'org_shape': (src_h, src_w)}
    return canvas, record


def letter_resize_bbox(bboxes, letter_info):
    """xyxy boxes of the original image -> of the letterboxed image"""
    boxes = np.asarray(bboxes) * letter_info['scale']
    boxes[:, [0, 2]] += letter_info['pad_left']
    boxes[:, [1, 3]] += letter_info['pad_top']
    return boxes
