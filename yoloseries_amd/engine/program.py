"""Program: the pre-built kernel launches of one (batch, input shape) — forward (engine.forward), backward (engine.backward), tuning (engine.tune)."""
import torch

from .._lib import lib
from .backward import BackwardMixin
from .forward import ForwardMixin
from .tune import TunerMixin


class Program(ForwardMixin, BackwardMixin, TunerMixin):
    """Pre-built kernel launches for one (batch, input shape)."""

    def __init__(self, builder, pack, B, outputs, bn_eps_of=None):
        self.B, self.pack = B, pack
        self.ops, self.bufs = builder.ops, builder.bufs
        self.outputs = outputs          # list of ConvOp (plain) or Ref whose buffers are returned
        dev = pack.device
        self.dev = dev
        self.L = lib()
        for b in self.bufs:
            # raw conv outputs (".y") are training-only and allocated by _build_train(); head buffers are fresh per forward
            if not b.name.endswith(".y") and not getattr(b, "is_head", False):
                b.t = torch.zeros(B, b.H, b.W, b.C, dtype=torch.bfloat16, device=dev)
        self.generation = 0
        self.profile = None             # {(kernel family, algorithmic flops): [(start_event, end_event)]} when profiling
        self._compiled = {}             # 'train' | 'eval' | ('bwd', two_streams, hooked) -> CompiledCmds (yh_exec replay)
        self.bwd_ready = False
        self._keep = []                 # keeps ctypes structs / tensors alive
        self._build_forward()
