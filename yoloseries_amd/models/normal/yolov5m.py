"""YOLOV5Middle — mirror of models/normal/yolov5m.py:7-116 (see _yolov5_base.py)."""
from ._yolov5_base import YOLOV5Base

__all__ = ['YOLOV5Middle']


class YOLOV5Middle(YOLOV5Base):
    WIDTH = 48
    DEPTHS = (2,4,6,2)
    HEAD_DEPTH = 2
