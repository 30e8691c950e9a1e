from .synth import COCO_ANCHORS, synth_head_outputs, synth_nms_heads, synth_targets, synth_yolox_heads  # noqa: F401
from .bbox_tools import *  # noqa: F401,F403
from .nms import *  # noqa: F401,F403
from .layer_tools import *  # noqa: F401,F403
from .optim import *  # noqa: F401,F403
from .dist import *  # noqa: F401,F403
from .mAP import mAP_v2  # noqa: F401
from .letterbox import letter_resize_bbox, letter_resize_img  # noqa: F401
from .common import *  # noqa: F401,F403
from .setup_env import *  # noqa: F401,F403
from .gpu import *  # noqa: F401,F403
from .meter import *  # noqa: F401,F403
from .model_utils import *  # noqa: F401,F403
from .launch import *  # noqa: F401,F403
from .logger import *  # noqa: F401,F403
from .visualizer import *  # noqa: F401,F403
from .graph import GraphedStep  # noqa: F401
