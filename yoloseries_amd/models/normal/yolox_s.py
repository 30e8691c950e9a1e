"""YOLOXSmall — mirror of models/normal/yolox_s.py:10-212.

Same CSPDarknet + PAN-FPN as YOLOv5s (``neck.*`` keys) and a decoupled head per stage
(``detect.pred_{small,middle,large}.{stem,conv,cls,reg,cof}``, :112-137).  Sub-module names and
creation order follow the reference, so state_dict keys and seeded default initialisation are identical.

On the HIP engine the two 3x3 branch convs of a head (cls[0] and conv[0]) share one GEMM (stacked output
channels), and the three biased 1x1 predictors (reg | cof | cls) are one block-diagonal GEMM that writes the
(4+1+nc)-channel prediction in the reference's channel order — ``torch.cat((reg, cof, cls))`` never runs.
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from ... import hipk
from ...engine import HipModuleMixin, Ref
from ...utils.layer_tools import C3BottleneckCSP, Concat, ConvBnAct, FastSPP, Upsample

__all__ = ['YOLOXSmall']


class SmallYOLOXBackboneAndNeck(nn.Module):

    def __init__(self, in_channel=3):
        super().__init__()
        self.focus = ConvBnAct(in_channel, 32, 6, 2, 2)
        self.backbone_stage1_conv = ConvBnAct(32, 64, 3, 2, 1)
        self.backbone_stage1_bscp = C3BottleneckCSP(64, 64, shortcut=True, num_block=1)
        self.backbone_stage2_conv = ConvBnAct(64, 128, 3, 2, 1)
        self.backbone_stage2_bscp = C3BottleneckCSP(128, 128, shortcut=True, num_block=2)
        self.backbone_stage3_conv = ConvBnAct(128, 256, 3, 2, 1)
        self.backbone_stage3_bscp = C3BottleneckCSP(256, 256, shortcut=True, num_block=3)
        self.backbone_stage4_conv = ConvBnAct(256, 512, 3, 2, 1)
        self.backbone_stage4_bscp = C3BottleneckCSP(512, 512, shortcut=True, num_block=1)
        self.backbone_stage4_spp = FastSPP(512, 512, kernel=5)
        self.head_upsample = Upsample()
        self.head_concat = Concat()
        self.head_stage1_conv = ConvBnAct(512, 256, 1, 1, 0)
        self.head_stage1_bscp = C3BottleneckCSP(512, 256, shortcut=False, num_block=1)
        self.head_stage2_conv = ConvBnAct(256, 128, 1, 1, 0)
        self.head_stage2_bscp = C3BottleneckCSP(256, 128, shortcut=False, num_block=1)
        self.head_stage3_conv = ConvBnAct(128, 128, 3, 2, 1)
        self.head_stage3_bscp = C3BottleneckCSP(256, 256, shortcut=False, num_block=1)
        self.head_stage4_conv = ConvBnAct(256, 256, 3, 2, 1)
        self.head_stage4_bscp = C3BottleneckCSP(512, 512, shortcut=False, num_block=1)
        self.output_features = ['stage_3', 'stage_4', 'stage_5']

    def _emit(self, b, x0):
        p = "neck."
        (x,) = b.cba(p + "focus", [self.focus], [Ref(x0)], focus=True)
        x = self.backbone_stage1_conv._emit(b, p + "backbone_stage1_conv", [x])
        x = self.backbone_stage1_bscp._emit(b, p + "backbone_stage1_bscp.", [x])
        x = self.backbone_stage2_conv._emit(b, p + "backbone_stage2_conv", [x])
        stage1_x = self.backbone_stage2_bscp._emit(b, p + "backbone_stage2_bscp.", [x])
        x = self.backbone_stage3_conv._emit(b, p + "backbone_stage3_conv", [stage1_x])
        stage2_x = self.backbone_stage3_bscp._emit(b, p + "backbone_stage3_bscp.", [x])
        x = self.backbone_stage4_conv._emit(b, p + "backbone_stage4_conv", [stage2_x])
        x = self.backbone_stage4_bscp._emit(b, p + "backbone_stage4_bscp.", [x])
        x = self.backbone_stage4_spp._emit(b, p + "backbone_stage4_spp.", [x])
        head1_x = self.head_stage1_conv._emit(b, p + "head_stage1_conv", [x])
        x = self.head_stage1_bscp._emit(b, p + "head_stage1_bscp.", [Ref(head1_x.buf, head1_x.coff, head1_x.C, ups=1), stage2_x])
        head2_x = self.head_stage2_conv._emit(b, p + "head_stage2_conv", [x])
        small_x = self.head_stage2_bscp._emit(b, p + "head_stage2_bscp.", [Ref(head2_x.buf, head2_x.coff, head2_x.C, ups=1), stage1_x])
        x = self.head_stage3_conv._emit(b, p + "head_stage3_conv", [small_x])
        mid_x = self.head_stage3_bscp._emit(b, p + "head_stage3_bscp.", [x, head2_x])
        x = self.head_stage4_conv._emit(b, p + "head_stage4_conv", [mid_x])
        large_x = self.head_stage4_bscp._emit(b, p + "head_stage4_bscp.", [x, head1_x])
        return small_x, mid_x, large_x


class Detect(nn.Module):
    """decoupled YOLOX head (models/normal/yolox_s.py:82-162)"""

    def __init__(self, num_anchors=1, in_channels=(256, 512, 1024), mid_channel=256, wid_mul=1.0, num_classes=80):
        super().__init__()
        self.num_anchors = num_anchors
        self.num_classes = num_classes
        self.pred_small = self._make_layers(int(in_channels[0] * wid_mul), int(mid_channel * wid_mul))
        self.pred_middle = self._make_layers(int(in_channels[1] * wid_mul), int(mid_channel * wid_mul))
        self.pred_large = self._make_layers(int(in_channels[2] * wid_mul), int(mid_channel * wid_mul))

    def _make_layers(self, in_c, mid_c):
        stem = ConvBnAct(in_c, mid_c, 3, 1, 1, act=True)
        cls = nn.Sequential(ConvBnAct(mid_c, mid_c, 3, 1, 1, act=True),
                            nn.Conv2d(mid_c, int(self.num_anchors * self.num_classes), 1, 1))
        conv = nn.Sequential(ConvBnAct(mid_c, mid_c, 3, 1, 1, act=True))
        reg = nn.Conv2d(mid_c, self.num_anchors * 4, 1, 1)
        cof = nn.Conv2d(mid_c, int(self.num_anchors * 1), 1, 1)
        return nn.ModuleDict({'stem': stem, 'conv': conv, 'cls': cls, 'reg': reg, 'cof': cof})

    def _emit(self, b, feats):
        outs = []
        for name, layers, x in zip(("pred_small", "pred_middle", "pred_large"), (self.pred_small, self.pred_middle, self.pred_large), feats):
            p = f"detect.{name}."
            stem = layers['stem']._emit(b, p + "stem", [x])
            fcls, freg = b.cba(p + "cls0_conv0", [layers['cls'][0], layers['conv'][0]], [stem])
            outs.append(b.plain_multi(p + "pred", [(layers['reg'], 0), (layers['cof'], 0), (layers['cls'][1], 1)], [freg, fcls]))
        return outs


class YOLOXSmall(HipModuleMixin, nn.Module):

    def __init__(self, num_anchors=1, in_channel=3, num_classes=80, prior_prob=0.01):
        super().__init__()
        if num_anchors != 1:
            raise NotImplementedError("YOLOXSmall on the HIP path supports num_anchors=1 (the shipped configuration)")
        self.neck = SmallYOLOXBackboneAndNeck(in_channel)
        self.detect = Detect(num_anchors=num_anchors, in_channels=[128, 256, 512], mid_channel=128, wid_mul=1.0, num_classes=num_classes)
        self.num_anchor = num_anchors
        self.num_classes = num_classes
        self._init_bias(prior_prob)

    def _init_bias(self, p):
        """bias prior -log((1-p)/p) on the cls and reg predictors, not on cof (models/normal/yolox_s.py:174-198)"""
        for layers in (self.detect.pred_small, self.detect.pred_middle, self.detect.pred_large):
            for m in layers['cls']:
                if isinstance(m, nn.Conv2d):
                    bias = m.bias.view(self.num_anchor, -1)
                    bias.data.fill_(-math.log((1 - p) / p))
                    m.bias = torch.nn.Parameter(bias.view(-1), requires_grad=True)
        for layers in (self.detect.pred_small, self.detect.pred_middle, self.detect.pred_large):
            m = layers['reg']
            bias = m.bias.view(self.num_anchor, -1)
            bias.data.fill_(-math.log((1 - p) / p))
            m.bias = torch.nn.Parameter(bias.view(-1), requires_grad=True)

    def _yh_build(self, b, B, H, W):
        if H % 32 or W % 32:
            raise ValueError(f"input size must be a multiple of 32, got {H}x{W}")
        x0 = b.buf("input_s2d", H // 2, W // 2, 16, needs_grad=False)
        feats = self.neck._emit(b, x0)
        return self.detect._emit(b, feats)

    def _yh_outputs(self, prog):
        outs = []
        E = 5 + self.num_classes
        for o in prog.outputs:
            Bn, h, w, ld = o.y.t.shape
            outs.append(o.y.t.as_strided((Bn, 1, E, h, w), (h * w * ld, h * w * ld, 1, w * ld, ld)))
        return outs

    def forward(self, x):
        """:return: OrderedDict pred_s / pred_m / pred_l, each (bn, num_anchors, 5+nc, H/s, W/s) [x, y, w, h, cof, cls...]"""
        if not x.is_cuda:
            raise RuntimeError("yoloseries_amd models run on an MI355X device only (no CPU path in the product)")
        B, Cin, H, W = x.shape
        prog = self._yh_program(B, H, W)
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        hipk.input_s2d(xin, prog.in_buf.t)
        s, m, l = self._yh_forward(prog, x)
        return OrderedDict((("pred_s", s), ("pred_m", m), ("pred_l", l)))
