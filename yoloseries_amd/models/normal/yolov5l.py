"""YOLOV5Large — mirror of models/normal/yolov5l.py:7-116 (see _yolov5_base.py)."""
from ._yolov5_base import YOLOV5Base

__all__ = ['YOLOV5Large']


class YOLOV5Large(YOLOV5Base):
    WIDTH = 64
    DEPTHS = (3,6,9,3)
    HEAD_DEPTH = 3
