"""YOLOV5XLarge — mirror of models/normal/yolov5x.py:7-116 (see _yolov5_base.py)."""
from ._yolov5_base import YOLOV5Base

__all__ = ['YOLOV5XLarge']


class YOLOV5XLarge(YOLOV5Base):
    WIDTH = 80
    DEPTHS = (4,12,12,4)
    HEAD_DEPTH = 4
