from .yolov5m import *  # noqa: F401,F403
from .yolov5s import *  # noqa: F401,F403
from .yolov5l import *  # noqa: F401,F403
from .yolov5x import *  # noqa: F401,F403
from .yolox_s import *  # noqa: F401,F403
