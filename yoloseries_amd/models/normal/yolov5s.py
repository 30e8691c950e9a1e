"""YOLOV5Small — mirror of models/normal/yolov5s.py:7-116 (see _yolov5_base.py)."""
from ._yolov5_base import YOLOV5Base

__all__ = ['YOLOV5Small']


class YOLOV5Small(YOLOV5Base):
    WIDTH = 32
    DEPTHS = (1,2,3,1)
    HEAD_DEPTH = 1
