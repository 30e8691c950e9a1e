"""ORACLE (test infrastructure, never imported by the product path).

Plain torch-CPU fp32 restatement of the YOLOv5 s/m/l/x forward graph
(/root/reference models/normal/yolov5s.py:87-116 with the blocks of utils/layer_tools.py:
ConvBnAct :82-94, BasicBottleneck :97-114, C3BottleneckCSP :152-169, FastSPP :270-288,
Detect :454-470).  Functional: it reads the tensors of a ``state_dict`` with the reference's
key names, so it can be run on the weights of a yoloseries_amd model.  Pinned by
tests/golden/g7_model.npz (outputs of the reference model itself, tests/test_oracle_net.py).
"""
import torch
import torch.nn.functional as F


class V5NetOracle:
    def __init__(self, state_dict, train=False, momentum=0.03, eps=1e-3, grad=None):
        """train: BatchNorm on batch statistics (and running-statistics update); grad: parameters require gradients
        (default: == train; grad=True with train=False is model.eval() under autograd — BatchNorm on its running statistics)"""
        self.sd = {k: v.detach().to("cpu", torch.float32).clone() for k, v in state_dict.items() if v.dtype.is_floating_point}
        self.train = train
        self.grad = train if grad is None else grad
        self.momentum, self.eps = momentum, eps
        self.params = {}

    def p(self, key, requires_grad=False):
        if key not in self.params:
            t = self.sd[key]
            if requires_grad:
                t = t.clone().requires_grad_(True)
            self.params[key] = t
        return self.params[key]

    def cba(self, x, name, k, s, pad):
        w = self.p(name + ".conv.weight", self.grad)
        y = F.conv2d(x, w, None, s, pad)
        g, b = self.p(name + ".bn.weight", self.grad), self.p(name + ".bn.bias", self.grad)
        rm, rv = self.sd[name + ".bn.running_mean"], self.sd[name + ".bn.running_var"]
        y = F.batch_norm(y, rm, rv, g, b, self.train, self.momentum, self.eps)
        return F.silu(y)

    def bottleneck(self, x, name, shortcut):
        y = self.cba(self.cba(x, name + ".conv_bn_act_1", 1, 1, 0), name + ".conv_bn_act_2", 3, 1, 1)
        return y + x if shortcut else y

    def c3(self, x, name, shortcut):
        y1 = self.cba(x, name + ".cba1", 1, 1, 0)
        i = 0
        while f"{name}.blocks.{i}.conv_bn_act_1.conv.weight" in self.sd:
            y1 = self.bottleneck(y1, f"{name}.blocks.{i}", shortcut)
            i += 1
        y2 = self.cba(x, name + ".cba2", 1, 1, 0)
        return self.cba(torch.cat((y1, y2), 1), name + ".cba3", 1, 1, 0)

    def sppf(self, x, name):
        x = self.cba(x, name + ".cba1", 1, 1, 0)
        x2 = F.max_pool2d(x, 5, 1, 2)
        x3 = F.max_pool2d(x2, 5, 1, 2)
        x4 = F.max_pool2d(x3, 5, 1, 2)
        return self.cba(torch.cat((x, x2, x3, x4), 1), name + ".cba2", 1, 1, 0)

    def __call__(self, x):
        x = self.cba(x, "focus", 6, 2, 2)
        x = self.cba(x, "backbone_stage1_conv", 3, 2, 1)
        x = self.c3(x, "backbone_stage1_bscp", True)
        s1 = self.c3(self.cba(x, "backbone_stage2_conv", 3, 2, 1), "backbone_stage2_bscp", True)
        s2 = self.c3(self.cba(s1, "backbone_stage3_conv", 3, 2, 1), "backbone_stage3_bscp", True)
        x = self.cba(s2, "backbone_stage4_conv", 3, 2, 1)
        x = self.c3(x, "backbone_stage4_bscp", True)
        x = self.sppf(x, "backbone_stage4_spp")
        h1 = self.cba(x, "head_stage1_conv", 1, 1, 0)
        x = torch.cat((F.interpolate(h1, scale_factor=2, mode="nearest"), s2), 1)
        x = self.c3(x, "head_stage1_bscp", False)
        h2 = self.cba(x, "head_stage2_conv", 1, 1, 0)
        x = torch.cat((F.interpolate(h2, scale_factor=2, mode="nearest"), s1), 1)
        small = self.c3(x, "head_stage2_bscp", False)
        x = torch.cat((self.cba(small, "head_stage3_conv", 3, 2, 1), h2), 1)
        mid = self.c3(x, "head_stage3_bscp", False)
        x = torch.cat((self.cba(mid, "head_stage4_conv", 3, 2, 1), h1), 1)
        large = self.c3(x, "head_stage4_bscp", False)
        outs = []
        for n, t in (("detect.detect_small", small), ("detect.detect_mid", mid), ("detect.detect_large", large)):
            outs.append(F.conv2d(t, self.p(n + ".weight", self.grad), self.p(n + ".bias", self.grad)))
        return tuple(outs)

    def sgd_step(self, lr=0.01):
        """plain SGD on every tensor that received a gradient (keeps the CPU baseline a full train step)"""
        with torch.no_grad():
            for k, t in self.params.items():
                if t.grad is not None:
                    t -= lr * t.grad
                    t.grad = None
