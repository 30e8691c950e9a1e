from .yolov5_loss import YOLOV5Loss  # noqa: F401
