"""ExponentialMovingAverageModel — mirror of trainer/ema_model.py:7-28.
decay = decay_ratio * (1 - exp(-n/2000)); every floating tensor of the state_dict follows
e = d*e + (1-d)*p.  On a model with a flat parameter arena the update is one HIP launch."""
import math
from copy import deepcopy

import torch

__all__ = ['ExponentialMovingAverageModel']


class ExponentialMovingAverageModel:

    def __init__(self, model, decay_ratio=0.9999, update_num=0):
        self.ema = deepcopy(model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model).eval()
        self.update_num = update_num
        self.get_decay_weight = lambda x: decay_ratio * (1 - math.exp(-x / 2000))
        for parm in self.ema.parameters():
            parm.requires_grad_(False)

    def update(self, model):
        with torch.no_grad():
            self.update_num += 1
            decay_weight = self.get_decay_weight(self.update_num)
            src = model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model
            state = src.state_dict()
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    if v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and state[k].is_contiguous():
                        from .. import hipk
                        hipk.ema_update(v, state[k].detach(), decay_weight)
                    else:
                        v *= decay_weight
                        v += (1. - decay_weight) * state[k].detach()
