"""CNN building blocks — mirror of the hot-path classes of utils/layer_tools.py
(ConvBnAct :82-94, BasicBottleneck :97-114, C3BottleneckCSP :152-169, FastSPP :270-288,
Concat :64-72, Upsample :444-451, Detect :454-470, fuse_conv_bn :26-53, autopad :75-79).

The classes keep the reference's constructor signatures and sub-module names (so state_dict
keys are identical) but their arithmetic runs on the HIP engine (yoloseries_amd/engine.py):
each block describes itself to an engine ``Builder`` through ``_emit``; a block called on its
own (``block(x)`` with an NCHW tensor) builds a one-block program.
"""
import torch
import torch.nn as nn

from ..engine import Builder, HipModuleMixin, Ref

__all__ = ['fuse_conv_bn', 'Concat', 'autopad', 'ConvBnAct', 'BasicBottleneck', 'C3BottleneckCSP', 'FastSPP',
           'Upsample', 'Detect']


def autopad(kernel, padding):
    if padding is None:
        return kernel // 2 if isinstance(kernel, int) else [p // 2 for p in kernel]
    return padding


def fuse_conv_bn(conv_layer, bn_layer):
    """W' = diag(g/sqrt(var+eps)) W ; b' = g (b - mean)/sqrt(var+eps) + beta  (utils/layer_tools.py:26-53).
    Host-side parameter algebra (a handful of tiny tensors), returns a plain nn.Conv2d."""
    fuseconv = nn.Conv2d(conv_layer.in_channels, conv_layer.out_channels, conv_layer.kernel_size, conv_layer.stride,
                         conv_layer.padding, groups=conv_layer.groups, bias=True).requires_grad_(False).to(conv_layer.weight.device)
    conv_w = conv_layer.weight.clone().view(conv_layer.out_channels, -1)
    bn_w = torch.diag(bn_layer.weight.div(torch.sqrt(bn_layer.running_var + bn_layer.eps)))
    fuseconv.weight.copy_(torch.mm(bn_w, conv_w).view(fuseconv.weight.shape))
    conv_b = torch.zeros(conv_layer.weight.size(0), device=conv_layer.weight.device) if conv_layer.bias is None else conv_layer.bias
    bn_b = bn_layer.bias - (bn_layer.weight.mul(bn_layer.running_mean).div(torch.sqrt(bn_layer.eps + bn_layer.running_var)))
    fuseconv.bias.copy_(torch.mm(bn_w, conv_b.reshape(-1, 1)).reshape(-1) + bn_b)
    return fuseconv


class _Block(HipModuleMixin, nn.Module):
    """stand-alone execution of one block: NCHW tensor in, NCHW (bf16 view) out"""

    def _yh_build(self, b, B, H, W):
        cin = self._in_channels()
        xin = b.buf("input", H, W, cin, needs_grad=True)
        out = self._emit(b, "", [Ref(xin)])
        return [out]

    def forward(self, x):
        B, Cin, H, W = x.shape
        prog = self._yh_program(B, H, W)
        prog.in_buf.t.copy_(x.detach().permute(0, 2, 3, 1))
        (y,) = self._yh_forward(prog, x)
        return y


class Concat(nn.Module):
    """kept for state_dict/API compatibility; the engine reads concatenated inputs in place"""

    def __init__(self, dimension=1):
        super().__init__()
        self.dim = dimension

    def forward(self, x):
        assert isinstance(x, (list, tuple))
        return torch.cat(x, dim=self.dim).contiguous()


class Upsample(nn.Module):
    """nearest x2; inside the models it is an addressing mode of the consumer conv"""

    def __init__(self, size=None, scale_factor=2, mode='nearest'):
        super().__init__()
        self.upsample = nn.Upsample(size, scale_factor, mode)

    def forward(self, x):
        return self.upsample(x)


class ConvBnAct(_Block):

    def __init__(self, in_channel, out_channel, kernel, stride, padding=None, groups=1, bias=False, act=True, inplace=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channel, out_channel, kernel, stride, padding=autopad(kernel, padding), groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channel, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU(inplace=inplace) if act else nn.Identity()

    def _in_channels(self):
        return self.conv.in_channels

    def _is_stem(self):
        c = self.conv
        return c.in_channels <= 4 and c.kernel_size == (6, 6) and c.stride == (2, 2) and c.padding == (2, 2)

    def _yh_build(self, b, B, H, W):
        if self._is_stem():      # the 6x6/s2/p2 image stem runs as a 3x3 conv on the space-to-depth tensor
            x0 = b.buf("input_s2d", H // 2, W // 2, 16, needs_grad=False)
            return b.cba("cba", [self], [Ref(x0)], focus=True)
        return super()._yh_build(b, B, H, W)

    def forward(self, x):
        if self._is_stem():
            from .. import hipk
            B, Cin, H, W = x.shape
            prog = self._yh_program(B, H, W)
            hipk.input_s2d(x.detach().float().contiguous(), prog.in_buf.t)
            (y,) = self._yh_forward(prog, x)
            return y
        return super().forward(x)

    def forward_fuse(self, x):
        """utils/layer_tools.py:93-94: `act(conv(x))` of a module whose conv is the biased result of fuse_conv_bn and whose `bn` was
        deleted (detect_yolov5.py:110-116).  Runs the inference program: the bias rides in the conv epilogue where the folded
        BatchNorm normally sits (engine.bn_of)."""
        if hasattr(self, 'bn'):
            raise RuntimeError("forward_fuse: fuse first (m.conv = fuse_conv_bn(m.conv, m.bn); delattr(m, 'bn')), as detect_yolov5.py:110-116 does")
        if self.training or torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("forward_fuse is an inference path: call .eval() and freeze the fused parameters (fuse_conv_bn returns them frozen)")
        return ConvBnAct.forward(self, x)

    def _emit(self, b, name, segs, dst=None, res=None):
        (out,) = b.cba(name + "cba" if name == "" else name, [self], segs, dsts=[dst] if dst is not None else None, res=res)
        return out


class BasicBottleneck(_Block):

    def __init__(self, in_channel, out_channel, shorcut, groups=1, expand_ratio=0.5):
        super().__init__()
        mid_channel = int(in_channel * expand_ratio)
        self.conv_bn_act_1 = ConvBnAct(in_channel, mid_channel, 1, 1)
        self.conv_bn_act_2 = ConvBnAct(mid_channel, out_channel, 3, 1, 1, groups=groups)
        self.residual = shorcut and (in_channel == out_channel)

    def _in_channels(self):
        return self.conv_bn_act_1.conv.in_channels

    def _emit(self, b, name, segs, dst=None):
        assert len(segs) == 1
        u = self.conv_bn_act_1._emit(b, name + "conv_bn_act_1", segs)
        return self.conv_bn_act_2._emit(b, name + "conv_bn_act_2", [u], dst=dst, res=segs[0] if self.residual else None)


class C3BottleneckCSP(_Block):
    """3 convolution layers with BottleneckCSP (utils/layer_tools.py:152-169); cba1 and cba2 read the
    same input and run as ONE GEMM with stacked output channels."""

    def __init__(self, in_channel, out_channel, shortcut=True, num_block=1, groups=1, bias=False):
        super().__init__()
        mid_channel = out_channel // 2
        self.cba1 = ConvBnAct(in_channel, mid_channel, 1, 1, groups=groups, bias=bias)
        self.cba2 = ConvBnAct(in_channel, mid_channel, 1, 1, groups=groups, bias=bias)
        self.cba3 = ConvBnAct(mid_channel * 2, out_channel, 1, 1, groups=groups, bias=bias)
        self.blocks = nn.Sequential(*[BasicBottleneck(mid_channel, mid_channel, shortcut, expand_ratio=1.0) for _ in range(num_block)])
        self.concat = Concat()

    def _in_channels(self):
        return self.cba1.conv.in_channels

    def _emit(self, b, name, segs, dst=None):
        mid = self.cba1.conv.out_channels
        Hi, Wi = segs[0].buf.H << segs[0].ups, segs[0].buf.W << segs[0].ups
        a12 = b.buf(name + "a12", Hi, Wi, 2 * mid)
        b.cba(name + "cba12", [self.cba1, self.cba2], segs, dsts=[Ref(a12, 0, mid), Ref(a12, mid, mid)])
        t = Ref(a12, 0, mid)
        for i, blk in enumerate(self.blocks):
            t = blk._emit(b, f"{name}blocks.{i}.", [t])
        return self.cba3._emit(b, name + "cba3", [t, Ref(a12, mid, mid)], dst=dst)


class FastSPP(_Block):
    """SPPF: cba1 -> 3 chained 5x5 max-pools -> concat(4) -> cba2 (utils/layer_tools.py:270-288)"""

    def __init__(self, in_channel, out_channel, kernel=5):
        super().__init__()
        mid_channel = in_channel // 2
        self.cba1 = ConvBnAct(in_channel, mid_channel, 1, 1, 0)
        self.cba2 = ConvBnAct(mid_channel * 4, out_channel, 1, 1)
        self.maxpool = nn.MaxPool2d(kernel_size=kernel, stride=1, padding=kernel // 2)
        if kernel != 5:
            raise NotImplementedError("FastSPP: only the 5x5 pool of the shipped models is implemented on the HIP path")

    def _in_channels(self):
        return self.cba1.conv.in_channels

    def _emit(self, b, name, segs, dst=None):
        mid = self.cba1.conv.out_channels
        H, W = segs[0].buf.H, segs[0].buf.W
        cat = b.buf(name + "cat", H, W, 4 * mid)
        self.cba1._emit(b, name + "cba1", segs, dst=Ref(cat, 0, mid))
        for i in range(3):
            b.pool(f"{name}pool{i}", Ref(cat, i * mid, mid), Ref(cat, (i + 1) * mid, mid))
        return self.cba2._emit(b, name + "cba2", [Ref(cat, 0, 4 * mid)], dst=dst)


class Detect(nn.Module):

    def __init__(self, in_channels=None, out_channel=3 * 85):
        super().__init__()
        if in_channels is None:
            in_channels = [256, 512, 1024]
        self.detect_small = nn.Conv2d(in_channels[0], out_channel, (1, 1), (1, 1), (0, 0))
        self.detect_mid = nn.Conv2d(in_channels[1], out_channel, (1, 1), (1, 1), (0, 0))
        self.detect_large = nn.Conv2d(in_channels[2], out_channel, (1, 1), (1, 1), (0, 0))

    def _emit(self, b, name, refs):
        return [b.plain(name + n, conv, r) for n, conv, r in
                zip(("detect_small", "detect_mid", "detect_large"), (self.detect_small, self.detect_mid, self.detect_large), refs)]
