from .synth import COCO_ANCHORS, synth_head_outputs, synth_nms_heads, synth_targets  # noqa: F401
