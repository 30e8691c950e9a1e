from .yolov5_loss import YOLOV5Loss  # noqa: F401
from .yolox_loss import YOLOXLoss  # noqa: F401
