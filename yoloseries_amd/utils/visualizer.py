"""cv2_save_img — mirror of utils/visualizer.py:167-214: draw [xmin, ymin, xmax, ymax] boxes with 'label:score' captions on an
RGB uint8 image and write it to `save_path`.  OpenCV when it is installed (the reference's only path), Pillow otherwise."""
from pathlib import Path

import numpy as np

__all__ = ["cv2_save_img"]


def cv2_save_img(img, bboxes, labels, scores, save_path):
    assert isinstance(img, np.ndarray)
    assert len(bboxes) == len(labels)
    Path(save_path).parent.mkdir(parents=True, exist_ok=True)
    img = np.ascontiguousarray(img)
    caps = [f"{labels[i]}:{scores[i]:.1f}" if scores is not None and len(scores) == len(labels) else f"{labels[i]}" for i in range(len(bboxes))]
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None and hasattr(cv2, "imwrite"):
        for box, cap in zip(bboxes, caps):
            lt, rb = (round(box[0]), round(box[1])), (round(box[2]), round(box[3]))
            img = cv2.rectangle(img, pt1=lt, pt2=rb, color=[0, 238, 238], thickness=1)
            img = cv2.rectangle(img, pt1=lt, pt2=(lt[0] + int(box[2] - box[0]), lt[1] + 12), color=[200, 0, 0], thickness=-1)
            img = cv2.putText(img, text=cap, org=(lt[0], lt[1] + 9), fontFace=cv2.FONT_HERSHEY_SIMPLEX, fontScale=0.35,
                              color=[255, 255, 255], thickness=1, lineType=cv2.LINE_AA)
        cv2.imwrite(str(save_path), img[:, :, ::-1])
        return
    from PIL import Image, ImageDraw
    im = Image.fromarray(img.astype(np.uint8))
    dr = ImageDraw.Draw(im)
    for box, cap in zip(bboxes, caps):
        lt, rb = (round(box[0]), round(box[1])), (round(box[2]), round(box[3]))
        dr.rectangle([lt, rb], outline=(0, 238, 238), width=1)
        dr.rectangle([lt, (lt[0] + int(box[2] - box[0]), lt[1] + 12)], fill=(200, 0, 0))
        dr.text((lt[0] + 1, lt[1]), cap, fill=(255, 255, 255))
    im.save(str(save_path))
