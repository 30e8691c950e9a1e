"""hipGraph capture of a whole training step.

A YOLOv5s step enqueues ~700 kernel launches / events through Python + ctypes (4-6 ms of host time): hidden behind the
GPU at batch 64, the bound at batch 16.  Every entry point of the C ABI only enqueues on the stream it is given — no
allocation, no synchronisation, no host copies (include/yolohip.h) — and the per-step scalars of the optimizer and of the
EMA live in device memory (FlatSGD.scal, yh_ema_advance), so the whole step (forward, loss, backward with its side-stream
branch, clip, SGD, EMA) is captured ONCE into a hipGraph and replayed with one launch per step.

    stepper = GraphedStep(step_fn, pre_replay=[opt.graph_pre_replay, ema.graph_pre_replay])
    out = stepper()          # eager for the first `warmup` calls (builds / tunes the programs), then capture, then replay

Contract for `step_fn`: it reads its inputs from STATIC tensors (refill them in place, e.g. x.copy_(batch) — shapes are
fixed), performs no host synchronisation (.item(), .cpu(), hyp['loss_items_on_device'] = True), and returns tensors /
dicts of tensors: the returned objects are the static outputs, refreshed by every replay.  `pre_replay` callables run on
the host before each replay: the Python-side bookkeeping that the captured kernels cannot do (step counters, pushing a
changed learning rate into the device scalars).
"""
import warnings

import torch

from .._lib import YoloHipError

__all__ = ["GraphedStep"]


class GraphedStep:

    def __init__(self, step_fn, pre_replay=(), warmup=3, enabled=True):
        self.step_fn, self.pre_replay, self.warmup = step_fn, list(pre_replay), int(warmup)
        self.enabled = enabled and torch.cuda.is_available()
        self.calls = 0
        self.graph = None
        self.static_out = None
        self.failed = None              # message of a failed capture (the stepper keeps running eagerly)

    @property
    def mode(self):
        return "hipGraph replay" if self.graph is not None else "eager"

    def _owners(self):
        return [o for o in (getattr(fn, "__self__", None) for fn in self.pre_replay) if o is not None]

    def _capture(self):
        g = torch.cuda.CUDAGraph()
        # host -> device traffic the captured kernels depend on (a learning rate the warm-up has just moved) is pushed now,
        # outside the capture; the owners' host counters are remembered so that a failed capture — which ran step_fn's Python
        # bookkeeping but no kernel — can be undone
        owners = self._owners()
        for o in owners:
            if hasattr(o, "graph_pre_capture"):
                o.graph_pre_capture()
        snaps = [(o, o.graph_snapshot()) for o in owners if hasattr(o, "graph_snapshot")]
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g):
                out = self.step_fn()
        except Exception as e:                      # leave the process usable: fall back to eager launches
            self.failed = f"{type(e).__name__}: {e}"
            self.enabled = False
            torch.cuda.synchronize()
            for o, snap in snaps:
                o.graph_restore(snap)
            warnings.warn(f"hipGraph capture of the train step failed, continuing with eager launches: {self.failed}", RuntimeWarning)
            return None
        self.graph, self.static_out = g, out
        return out

    def __call__(self):
        self.calls += 1
        if not self.enabled or self.calls <= self.warmup:
            return self.step_fn()
        if self.graph is None:
            out = self._capture()                   # the capture itself does not execute the step ...
            if out is None:
                return self.step_fn()
            self.graph.replay()                     # ... its first replay does (host bookkeeping was done by step_fn while capturing)
            return self.static_out
        for fn in self.pre_replay:
            fn()
        self.graph.replay()
        return self.static_out

    def reset(self):
        """drop the captured graph (input shapes / model changed): the next call captures again"""
        self.graph, self.static_out, self.calls = None, None, self.warmup
