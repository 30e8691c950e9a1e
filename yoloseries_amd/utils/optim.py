"""Flat-arena SGD for the HIP models: the three parameter groups of the reference's
``_init_optimizer`` (train_yolov5.py:258-280 — BN weights | conv weights + weight decay | biases),
torch.optim.SGD(nesterov) semantics, clip_grad_norm_(10) (:344) and zero_grad in three launches
over the model's flat parameter arena instead of ~3x177 per-tensor kernels.

``param_groups`` is a list of dicts with 'lr' / 'initial_lr' / 'momentum' / 'weight_decay', so the
reference's warm-up code (train_yolov5.py:437-456) can mutate it unchanged."""
import torch
from torch import nn

from .. import hipk
from .._lib import YoloHipError

__all__ = ['FlatSGD']


class FlatSGD:

    def __init__(self, model, lr, momentum=0.937, weight_decay=1e-4, nesterov=True):
        self.model = model
        model.flat_grads_only = True
        self.nesterov = nesterov
        self.param_groups = [
            {"name": "bn_weight", "lr": lr, "initial_lr": lr, "momentum": momentum, "weight_decay": 0.0},
            {"name": "weight", "lr": lr, "initial_lr": lr, "momentum": momentum, "weight_decay": weight_decay},
            {"name": "bias", "lr": lr, "initial_lr": lr, "momentum": momentum, "weight_decay": 0.0},
        ]
        self._built_for = None
        self._pending_buf = None        # momentum buffer loaded before the parameter arena exists (resume)
        self.steps = 0
        self._gacc = None
        self._nacc = 0
        model._yh_grad_hook_opt = self._on_grad

    def _build(self, pack):
        gid = {}
        for m in self.model.modules():
            if hasattr(m, "bias") and isinstance(m.bias, nn.Parameter):
                gid[id(m.bias)] = 2
            if isinstance(m, nn.BatchNorm2d):
                gid[id(m.weight)] = 0
            elif hasattr(m, 'weight') and isinstance(m.weight, nn.Parameter):
                gid[id(m.weight)] = 1
        dev = pack.device
        group = torch.empty(pack.n, dtype=torch.uint8, device=dev)
        o = 0
        for p in pack.params:
            group[o:o + p.numel()] = gid.get(id(p), 1)
            o += p.numel()
        self.group = group
        self.buf = torch.zeros(pack.n, dtype=torch.float32, device=dev)
        if self._pending_buf is not None:
            if self._pending_buf.numel() != pack.n:
                raise YoloHipError(f"FlatSGD: checkpointed momentum buffer has {self._pending_buf.numel()} elements, the model {pack.n}")
            self.buf.copy_(self._pending_buf.to(dev))
            self._pending_buf = None
        # per-step scalars live in device memory (lr[3] | wd[3] | momentum | first-step flag): the step kernel takes no
        # host value as a launch argument, so a captured hipGraph of the step replays correctly (utils/graph.py)
        self.scal = torch.zeros(8, dtype=torch.float32, device=dev)
        self.part = torch.zeros(4096, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scale = torch.ones(1, dtype=torch.float32, device=dev)
        self._lr_host = None
        self._built_for = pack
        self._use_scale = False

    def _pack(self):
        st = self.model.__dict__.get('_yh')
        if not st or st['pack'] is None:
            raise YoloHipError("FlatSGD: run a forward pass first (the parameter arena is created lazily)")
        if self._built_for is not st['pack']:
            self._build(st['pack'])
        return st['pack']

    def _on_grad(self, flat_g):
        # gradient accumulation over several backward() calls (train_yolov5.py:327-337)
        if self._nacc == 0:
            self._gacc = flat_g
        else:
            self._gacc = self._gacc + flat_g
        self._nacc += 1

    def replace_accumulated(self):
        """the data-parallel exchange averaged the ACCUMULATED gradient at an accumulation boundary and hands it over as
        the next flat gradient: drop the local sum so that _on_grad() takes the average as is (utils/dist.py)"""
        self._gacc, self._nacc = None, 0

    def _grad(self):
        g = self._gacc if self._gacc is not None else getattr(self.model, "_yh_last_flat_grad", None)
        if g is None:
            raise YoloHipError("FlatSGD.step(): no gradient (call backward() first)")
        return g

    def clip_grad_norm_(self, max_norm):
        self._pack()
        g = self._grad()
        hipk.sumsq(g, self.part, self.sumsq)
        hipk.clip_scale(self.sumsq, max_norm, self.scale)
        self._use_scale = True
        return self.sumsq          # device scalar: total_norm ** 2 (no host sync)

    def _sync_scalars(self):
        """device copy of lr / weight decay / momentum / first-step flag, rewritten only when a value changed (warm-up,
        per-epoch schedule).  The copy is an ordinary stream-ordered H2D from pageable memory: staged when it is issued,
        so the host may run ahead; it must be issued outside a graph capture (GraphedStep calls this before each replay)."""
        vals = tuple(float(pg["lr"]) for pg in self.param_groups) + tuple(float(pg["weight_decay"]) for pg in self.param_groups) + \
            (float(self.param_groups[0]["momentum"]), 1.0 if self.steps == 0 else 0.0)
        if vals != self._lr_host:
            if torch.cuda.is_current_stream_capturing():
                raise YoloHipError("FlatSGD: learning-rate / momentum changed inside a graph capture; call graph_pre_replay() "
                                   "(or step once eagerly) before capturing")
            self.scal.copy_(torch.tensor(vals, dtype=torch.float32))
            self._lr_host = vals

    def step(self):
        pack = self._pack()
        g = self._grad()
        self._sync_scalars()
        hipk.sgd_step_dev(pack.flat, g, self.buf, self.group, self.scal, self.nesterov, self.scale if self._use_scale else None)
        self.steps += 1
        self._use_scale = False

    def graph_pre_replay(self):
        """host-side bookkeeping of one step that runs as a graph replay: refresh the device scalars if the schedule moved
        them, count the step"""
        self._pack()
        self._sync_scalars()
        self.steps += 1

    def graph_pre_capture(self):
        """push the device scalars outside the capture that is about to start"""
        self._pack()
        self._sync_scalars()

    def graph_snapshot(self):
        return (self.steps, self._use_scale)

    def graph_restore(self, snap):
        """a capture failed after step() had advanced the host counters without running a kernel"""
        self.steps, self._use_scale = snap
        self._lr_host = None                      # rewrite the device scalars at the next step

    def zero_grad(self, set_to_none=True):
        self._gacc, self._nacc = None, 0
        self.model._yh_last_flat_grad = None

    def state_dict(self):
        buf = self.buf if self._built_for is not None else self._pending_buf
        return {"momentum_buffer": buf, "steps": self.steps, "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        """Training.load_model() runs before the first forward, i.e. before the flat parameter arena (and with it the
        momentum buffer) exists: the loaded buffer is kept and copied in when the arena is built."""
        self.param_groups = [dict(g) for g in sd["param_groups"]]
        self.steps = sd["steps"]
        mb = sd.get("momentum_buffer")
        if mb is None:
            if self.steps > 0:
                raise YoloHipError("FlatSGD.load_state_dict: steps > 0 but the checkpoint holds no momentum buffer")
            return
        if self._built_for is not None:
            if mb.numel() != self.buf.numel():
                raise YoloHipError(f"FlatSGD: checkpointed momentum buffer has {mb.numel()} elements, the model {self.buf.numel()}")
            self.buf.copy_(mb)
        else:
            self._pending_buf = mb.detach().clone()
