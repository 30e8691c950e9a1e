"""adjust_status / summary_model — mirror of utils/model_utils.py:12-64."""
import contextlib
import warnings

import torch
from torch import nn

__all__ = ["adjust_status", "summary_model"]


@contextlib.contextmanager
def adjust_status(module: nn.Module, training: bool = False):
    """temporarily put every sub-module into train / eval mode (utils/model_utils.py:12-38)"""
    saved = {m: m.training for m in module.modules()}
    for m in saved:
        m.training = training
    try:
        yield module
    finally:
        for m, flag in saved.items():
            m.training = flag


def summary_model(model, input_img_size=[640, 640], verbose=False, prefix=""):
    """parameter / layer counts and the conv work of one image (utils/model_utils.py:41-64).  The reference asks `thop` for the
    multiply-accumulates and reports them divided by 2e9; here they come from the engine's own graph census (the same convs,
    no extra package), reported with the same scaling; '' when the model is not an engine model."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        number_params = sum(p.numel() for p in model.parameters())
        number_gradients = sum(p.numel() for p in model.parameters() if p.requires_grad)
        number_layers = len(list(model.modules()))
        flops = ""
        if hasattr(model, "_yh_build"):
            from ..engine import Builder, ConvOp
            b = Builder()
            model._yh_build(b, 1, int(input_img_size[0]), int(input_img_size[1]))
            macs = sum(o.Ho * o.Wo * o.N * o.k * o.k * (12 if o.focus else o.Ctot) for o in b.ops if isinstance(o, ConvOp))
            flops = macs / (1e9 * 2)
        if verbose:
            print(f"Model Summary: {prefix} {number_layers} layers; {number_params} parameters; {number_gradients} gradients; {flops} GFLOPs")
        return {"number_params": number_params, "number_gradients": number_gradients, "flops": flops, "number_layers": number_layers}
