"""The autograd node of the whole conv graph and the mixin of nn.Modules whose forward runs on the engine."""
import torch

from .._lib import YoloHipError
from .graph import Builder, ConvOp, ParamPack
from .program import Program


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole conv graph: forward runs the train program, backward
    returns (a) nothing for the image and (b) per-parameter gradient views of one flat buffer."""

    @staticmethod
    def forward(ctx, host, prog, x, *params):
        ctx.frozen = not host.training                # model.eval() under autograd: BatchNorm on its running statistics
        gen = prog.forward(True, frozen=ctx.frozen)
        ctx.prog, ctx.gen, ctx.host = prog, gen, host
        outs = host._yh_outputs(prog)
        ctx.out_meta = [(o.shape, o.stride()) for o in outs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        prog = ctx.prog
        if prog.generation != ctx.gen:
            raise YoloHipError("backward() after another forward() on the same model/shape: activations were overwritten "
                               "(run forward/backward alternately)")
        from ..layout import is_cell_major
        head_grads = []
        for o, g, op in zip(ctx.out_meta, gouts, prog.outputs):
            shape, stride = o
            if isinstance(op, ConvOp):
                Bn, Ct, h, w = prog.B, op.N, op.Ho, op.Wo
                if g is not None and g.dim() == 5:          # (B, anchors=1, C, h, w) views of the YOLOX head
                    g = g[:, 0]
                ld = op.y.C
                if g is None:
                    gb = torch.zeros(Bn, h, w, ld, dtype=torch.bfloat16, device=prog.dev)
                elif g.dtype == torch.bfloat16 and is_cell_major(g) and g.stride(3) == ld:
                    gb = g.as_strided((Bn, h, w, ld), (h * w * ld, w * ld, ld, 1))
                else:
                    gb = torch.zeros(Bn, h, w, ld, dtype=torch.bfloat16, device=prog.dev)
                    gb[..., :Ct] = g.permute(0, 2, 3, 1)
                head_grads.append(gb)
            else:
                head_grads.append(g.permute(0, 2, 3, 1).to(torch.bfloat16) if g is not None else torch.zeros(
                    shape[0], shape[2], shape[3], shape[1], dtype=torch.bfloat16, device=prog.dev))
        bucket_hook = getattr(ctx.host, "_yh_bucket_hook", None)   # data-parallel all-reduce, overlapped (utils/dist.py)
        owner = getattr(bucket_hook, "__self__", None)
        if owner is not None and not getattr(owner, "buckets_active", True):
            bucket_hook = None                                      # no_sync / accumulation boundary: no per-bucket segmentation
        flat_only = bool(getattr(ctx.host, "flat_grads_only", False))
        flat_g, pgrads = prog.backward(head_grads, bucket_hook, frozen=ctx.frozen, param_views=not flat_only)
        ctx.host._yh_last_flat_grad = flat_g
        # whole-gradient hook of the data-parallel exchange: all-reduces flat_g when no bucket hook ran, keeps the
        # books of un-exchanged accumulation steps, and at an accumulation boundary swaps in the averaged total
        hook = getattr(ctx.host, "_yh_grad_hook", None)
        if hook is not None:
            hook(flat_g, bucketed=bucket_hook is not None)
        hook = getattr(ctx.host, "_yh_grad_hook_opt", None)    # flat-arena optimizer
        if hook is not None:
            hook(flat_g)
        if flat_only:
            # the flat-arena optimizer (utils/optim.py FlatSGD) consumes flat_g directly; skip 177 AccumulateGrad nodes
            pgrads = [None] * len(prog.pack.params)
        gx = None
        if ctx.needs_input_grad[2] and prog.in_buf.needs_grad and prog.in_buf.g is not None:
            gx = prog.in_buf.g.permute(0, 3, 1, 2).float()
        return (None, None, gx, *pgrads)


class HipModuleMixin:
    """Mixed into nn.Modules whose forward runs on the engine."""

    def _yh_state(self):
        st = self.__dict__.get('_yh')
        if st is None:
            st = {'pack': None, 'progs': {}}
            self.__dict__['_yh'] = st
        return st

    def __getstate__(self):
        s = dict(self.__dict__)
        s.pop('_yh', None)
        s.pop('_yh_last_flat_grad', None)
        s.pop('_yh_grad_hook', None)
        s.pop('_yh_bucket_hook', None)
        s.pop('_yh_grad_hook_opt', None)
        s.pop('flat_grads_only', None)
        return s

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__['_yh'] = None

    def _yh_reset(self):
        self.__dict__['_yh'] = None

    # subclasses implement:  _yh_build(builder, B, H, W) -> (input_kind, outputs)
    def _yh_program(self, B, H, W):
        st = self._yh_state()
        if st['pack'] is not None and not st['pack'].valid_for(self):
            st = {'pack': None, 'progs': {}}
            self.__dict__['_yh'] = st
        key = (B, H, W)
        prog = st['progs'].get(key)
        if prog is None:
            b = Builder()
            outputs = self._yh_build(b, B, H, W)
            if st['pack'] is None:
                st['pack'] = ParamPack(self, b.ops)
            prog = Program(b, st['pack'], B, outputs)
            prog.in_buf = b.bufs[0]
            st['progs'][key] = prog
            if len(st['progs']) > 4:      # bound the number of cached shapes
                st['progs'].pop(next(iter(st['progs'])))
        return prog

    def _yh_outputs(self, prog):
        from ..layout import cell_major_view
        outs = []
        for o in prog.outputs:
            if isinstance(o, ConvOp):
                outs.append(cell_major_view(o.y.t, o.N))
            else:
                t = o.buf.t[..., o.coff:o.coff + o.C]
                outs.append(t.permute(0, 3, 1, 2))
        return outs

    def _yh_forward(self, prog, x):
        pk = prog.pack
        pk.repack()
        # a differentiable forward runs the training program (raw conv outputs kept for the backward); in evaluation mode its
        # BatchNorms use the running statistics, exactly as nn.BatchNorm2d.eval() does under autograd
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in pk.params)):
            if not self.training and not getattr(self, "_yh_warned_eval_grad", False):
                self.__dict__['_yh_warned_eval_grad'] = True
                import warnings
                warnings.warn("yoloseries_amd: model.eval() called with gradients enabled: this runs the TRAINING program with frozen "
                              "BatchNorm (raw conv outputs kept, no folded inference kernels) and is several times slower and larger "
                              "than the inference program; wrap evaluation in torch.no_grad() unless the gradients are wanted "
                              "(INTEGRATION.md, 'Evaluation under autograd')", stacklevel=3)
            return _NetFn.apply(self, prog, x, *pk.params)
        prog.forward(self.training)
        return tuple(self._yh_outputs(prog))
