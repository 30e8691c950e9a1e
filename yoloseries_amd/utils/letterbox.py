"""Letterbox geometry — mirror of utils/data_aug.py:21-70 (letter_resize_img) and utils/bbox_tools.py:38-49
(letter_resize_bbox), the format transform between dataset images and the (B,3,H,W) network input; the
``resize_info`` dict it returns is what the evaluators use to map predictions back to the original image.

The reference resizes with ``cv2.resize(..., interpolation=0)`` (INTER_NEAREST) and pads with
``cv2.copyMakeBorder``; OpenCV is not a dependency here: nearest sampling is restated with its index rule
``src = min(floor(dst * src_size / dst_size), src_size - 1)`` and the border is a constant pad."""
import numpy as np

__all__ = ['letter_resize_img', 'letter_resize_bbox', 'resize_nearest']


def resize_nearest(img, resize_w, resize_h):
    """cv2.resize(img, (resize_w, resize_h), interpolation=cv2.INTER_NEAREST)"""
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(resize_h) * (h / resize_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(resize_w) * (w / resize_w)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def letter_resize_img(img, dst_size, stride=64, fill_value=128, only_ds=False, training=True):
    """:param img: (h,w,3) uint8;  :param dst_size: int or [h, w] (rounded up to a multiple of ``stride``)
    :return: letterboxed uint8 image and {'scale','pad_top','pad_left','pad_bottom','pad_right','org_shape'}"""
    if isinstance(dst_size, int):
        dst_size = [dst_size, dst_size]
    dst_del_h, dst_del_w = np.remainder(dst_size[0], stride), np.remainder(dst_size[1], stride)
    dst_pad_h = stride - dst_del_h if dst_del_h > 0 else 0
    dst_pad_w = stride - dst_del_w if dst_del_w > 0 else 0
    dst_size = [int(dst_size[0] + dst_pad_h), int(dst_size[1] + dst_pad_w)]

    org_h, org_w = img.shape[:2]
    scale = float(np.min([dst_size[0] / org_h, dst_size[1] / org_w]))
    if only_ds:
        scale = min(scale, 1.0)
    if scale != 1.:
        resize_h, resize_w = int(org_h * scale), int(org_w * scale)
        img_resize = resize_nearest(img, resize_w, resize_h)
    else:
        resize_h, resize_w = img.shape[:2]
        img_resize = img.copy()

    if not training:      # as few padding pixels as possible at test time
        pad_h, pad_w = dst_size[0] - resize_h, dst_size[1] - resize_w
        pad_h, pad_w = int(np.remainder(pad_h, stride)), int(np.remainder(pad_w, stride))
        top = int(round(pad_h / 2))
        left = int(round(pad_w / 2))
        bottom = pad_h - top
        right = pad_w - left
        fv = fill_value if not isinstance(fill_value, int) else (fill_value, fill_value, fill_value)
        img_out = np.empty((resize_h + pad_h, resize_w + pad_w, 3), dtype=img_resize.dtype)
        img_out[...] = np.asarray(fv, dtype=img_resize.dtype)
        img_out[top:top + resize_h, left:left + resize_w] = img_resize
    else:
        img_out = np.full(shape=dst_size + [3], fill_value=fill_value)
        pad_h, pad_w = dst_size[0] - resize_h, dst_size[1] - resize_w
        top, left = pad_h // 2, pad_w // 2
        bottom, right = pad_h - top, pad_w - left
        img_out[top:(top + resize_h), left:(left + resize_w)] = img_resize
    letter_info = {'scale': scale, 'pad_top': top, 'pad_left': left, "pad_bottom": bottom, "pad_right": right,
                   "org_shape": (org_h, org_w)}
    return img_out.astype(np.uint8), letter_info


def letter_resize_bbox(bboxes, letter_info):
    """[xmin,ymin,xmax,ymax] of the original image -> of the letterboxed image (utils/bbox_tools.py:38-49)"""
    bboxes = np.asarray(bboxes) if not isinstance(bboxes, np.ndarray) else bboxes
    letter_bbox = bboxes * letter_info['scale']
    letter_bbox[:, [1, 3]] += letter_info['pad_top']
    letter_bbox[:, [0, 2]] += letter_info['pad_left']
    return letter_bbox
