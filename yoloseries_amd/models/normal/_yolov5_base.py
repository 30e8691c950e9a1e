"""Shared implementation of YOLOV5Small/Middle/Large/XLarge — mirror of
models/normal/yolov5{s,m,l,x}.py:7-116 (identical topology, widths/depths differ).

Module attribute names and registration order follow the reference exactly, so
``state_dict()`` keys (348 for v5s) and the seeded ``_init_bias`` initialisation are
identical; ``forward`` runs the whole CSPDarknet + PAN-FPN + Detect graph on the HIP
engine and returns cell-major views shaped (B, A*(5+nc), H/s, W/s).
"""
import math

import torch
from torch import nn

from ...engine import HipModuleMixin, Ref
from ...utils.layer_tools import C3BottleneckCSP, Concat, ConvBnAct, Detect, FastSPP, Upsample
from ... import hipk


class YOLOV5Base(HipModuleMixin, nn.Module):
    WIDTH = 32                  # stem channels
    DEPTHS = (1, 2, 3, 1)       # bottlenecks in the 4 backbone C3 blocks
    HEAD_DEPTH = 1

    def __init__(self, anchor_num, num_class, in_channel=3):
        super().__init__()
        self.num_class = num_class
        w = self.WIDTH
        d = self.DEPTHS
        hd = self.HEAD_DEPTH
        # ============================== backbone ==============================
        self.focus = ConvBnAct(in_channel, w, 6, 2, 2)
        self.backbone_stage1_conv = ConvBnAct(w, 2 * w, 3, 2, 1)
        self.backbone_stage1_bscp = C3BottleneckCSP(2 * w, 2 * w, shortcut=True, num_block=d[0])
        self.backbone_stage2_conv = ConvBnAct(2 * w, 4 * w, 3, 2, 1)
        self.backbone_stage2_bscp = C3BottleneckCSP(4 * w, 4 * w, shortcut=True, num_block=d[1])
        self.backbone_stage3_conv = ConvBnAct(4 * w, 8 * w, 3, 2, 1)
        self.backbone_stage3_bscp = C3BottleneckCSP(8 * w, 8 * w, shortcut=True, num_block=d[2])
        self.backbone_stage4_conv = ConvBnAct(8 * w, 16 * w, 3, 2, 1)
        self.backbone_stage4_bscp = C3BottleneckCSP(16 * w, 16 * w, shortcut=True, num_block=d[3])
        self.backbone_stage4_spp = FastSPP(16 * w, 16 * w, kernel=5)
        # ============================== head ==============================
        self.head_upsample = Upsample()
        self.head_concat = Concat()
        self.head_stage1_conv = ConvBnAct(16 * w, 8 * w, 1, 1, 0)
        self.head_stage1_bscp = C3BottleneckCSP(16 * w, 8 * w, shortcut=False, num_block=hd)
        self.head_stage2_conv = ConvBnAct(8 * w, 4 * w, 1, 1, 0)
        self.head_stage2_bscp = C3BottleneckCSP(8 * w, 4 * w, shortcut=False, num_block=hd)
        self.head_stage3_conv = ConvBnAct(4 * w, 4 * w, 3, 2, 1)
        self.head_stage3_bscp = C3BottleneckCSP(8 * w, 8 * w, shortcut=False, num_block=hd)
        self.head_stage4_conv = ConvBnAct(8 * w, 8 * w, 3, 2, 1)
        self.head_stage4_bscp = C3BottleneckCSP(16 * w, 16 * w, shortcut=False, num_block=hd)
        # detect layers
        self.num_anchor = anchor_num
        self.detect = Detect(in_channels=[4 * w, 8 * w, 16 * w], out_channel=self.num_anchor * (num_class + 5))
        self._init_bias()

    def _init_bias(self):
        """models/normal/yolov5s.py:47-85: kaiming-normal(fan_out, relu) on every conv, zero biases, then the
        RetinaNet-style prior on the detect biases.  The reference obtains the strides from a 128x128
        dummy forward in train mode, which also leaves every BatchNorm with running_var = 0.97
        and num_batches_tracked = 1 (all activations are zero); both side effects are reproduced
        analytically here (strides are 8/16/32 by construction)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                m.eps = 1e-3
                m.momentum = 0.03
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.running_var.mul_(1.0 - m.momentum)       # (1-mom)*1 + mom*var(zeros)
                    m.num_batches_tracked.add_(1)
            detect_layer = [self.detect.detect_small, self.detect.detect_mid, self.detect.detect_large]
            for m, stride in zip(detect_layer, (8.0, 16.0, 32.0)):
                bias = m.bias.view(self.num_anchor, -1)
                bias[:, 4] += math.log(8 / (512 / stride) ** 2)
                bias[:, 5:] += math.log(0.6 / (self.num_class - 0.99))
                m.bias = torch.nn.Parameter(bias.view(-1), requires_grad=True)

    # ------------------------------------------------------------------ engine graph
    def _yh_build(self, b, B, H, W):
        if H % 32 or W % 32:
            raise ValueError(f"input size must be a multiple of 32, got {H}x{W}")
        x0 = b.buf("input_s2d", H // 2, W // 2, 16, needs_grad=False)
        (x,) = b.cba("focus", [self.focus], [Ref(x0)], focus=True)
        x = self.backbone_stage1_conv._emit(b, "backbone_stage1_conv", [x])
        x = self.backbone_stage1_bscp._emit(b, "backbone_stage1_bscp.", [x])
        x = self.backbone_stage2_conv._emit(b, "backbone_stage2_conv", [x])
        stage1_x = self.backbone_stage2_bscp._emit(b, "backbone_stage2_bscp.", [x])
        x = self.backbone_stage3_conv._emit(b, "backbone_stage3_conv", [stage1_x])
        stage2_x = self.backbone_stage3_bscp._emit(b, "backbone_stage3_bscp.", [x])
        x = self.backbone_stage4_conv._emit(b, "backbone_stage4_conv", [stage2_x])
        x = self.backbone_stage4_bscp._emit(b, "backbone_stage4_bscp.", [x])
        x = self.backbone_stage4_spp._emit(b, "backbone_stage4_spp.", [x])
        head1_x = self.head_stage1_conv._emit(b, "head_stage1_conv", [x])
        # upsample + concat are addressing modes of the consumer (never materialised)
        x = self.head_stage1_bscp._emit(b, "head_stage1_bscp.", [Ref(head1_x.buf, head1_x.coff, head1_x.C, ups=1), stage2_x])
        head2_x = self.head_stage2_conv._emit(b, "head_stage2_conv", [x])
        small_x = self.head_stage2_bscp._emit(b, "head_stage2_bscp.", [Ref(head2_x.buf, head2_x.coff, head2_x.C, ups=1), stage1_x])
        x = self.head_stage3_conv._emit(b, "head_stage3_conv", [small_x])
        mid_x = self.head_stage3_bscp._emit(b, "head_stage3_bscp.", [x, head2_x])
        x = self.head_stage4_conv._emit(b, "head_stage4_conv", [mid_x])
        large_x = self.head_stage4_bscp._emit(b, "head_stage4_bscp.", [x, head1_x])
        return self.detect._emit(b, "detect.", [small_x, mid_x, large_x])

    def forward(self, x):
        """:param x: (bn, 3, H, W) float, H and W multiples of 32
        :return: (small, mid, large) each (bn, A*(5+nc), H/s, W/s), bf16, cell-major memory"""
        if not x.is_cuda:
            raise RuntimeError("yoloseries_amd models run on an MI355X device only (no CPU path in the product)")
        B, Cin, H, W = x.shape
        prog = self._yh_program(B, H, W)
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        hipk.input_s2d(xin, prog.in_buf.t)
        return self._yh_forward(prog, x)
