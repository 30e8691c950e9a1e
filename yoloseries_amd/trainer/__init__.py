from .eval_yolov5 import *  # noqa: F401,F403
from .ema_model import *  # noqa: F401,F403
from .eval_yolox import *  # noqa: F401,F403
