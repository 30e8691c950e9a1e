"""Host-side executor of the YOLO conv graph on the HIP kernels (include/yolohip.h).

A model (or a single block used stand-alone) describes itself once to a ``Builder`` as a
list of ops over NHWC bf16 buffers:

  ConvOp  — implicit-GEMM conv over a virtual channel-concat of up to two slices (one may be
            read through a nearest-2x upsample), followed by training-mode BatchNorm + SiLU
            (stats from the conv epilogue, finalize, apply) or, for Detect, a bias only.
            Sibling 1x1 convs that read the same input (C3's cba1/cba2,
            utils/layer_tools.py:165-168) are one GEMM with stacked output channels.
  PoolOp  — SPPF 5x5/s1 max-pool writing into a channel slice of the concat buffer.

torch.cat / nn.Upsample / x.clone() of the reference are never materialised: concat and
upsample are addressing modes of the consumer's loader, residuals are fused into the apply.
``Program`` holds the pre-built kernel descriptors for one input shape and runs
forward (train / eval) and backward; ``ParamPack`` keeps the fp32 master parameters in one
flat arena (the nn.Parameters are views of it, so state_dict/optimizers are unchanged) and
maps them to the packed bf16 weight images and back (packed fp32 grads -> parameter grads)
with one index-gather launch each.
"""
import ctypes as C
import json
import os
import struct

import numpy as np
import torch

from . import hipk
from ._lib import (BnFoldItem, BnPart, Cmd, ConvDesc, YH_BN_MAX_PARTS, WgradDesc, YH_CMD_EVENT_RECORD, YH_CMD_SLOTS, YH_CMD_STREAM_WAIT, YH_ACT_NONE, YH_ACT_SILU, YH_CONV_DGRAD, YH_CONV_FWD, YoloHipError, check, lib)
from .hipk import Slice

BN_EPS_DEFAULT = 1e-3


def _rup(x, m):
    return ((x + m - 1) // m) * m


def _tune_cache_path():
    """per-machine timings live in the user's cache directory, not in the package (YH_TUNE_CACHE overrides)"""
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    return os.environ.get("YH_TUNE_CACHE", os.path.join(base, "yoloseries_amd", "tune_cache.json"))


class _TuneTable:
    """launch parameters per layer shape.  Two layers, never mixed: the shipped table for the BASELINE configurations
    (tune_defaults.json, timed on an MI355X with tools/make_tune_defaults.sh: the same choices on every box, no tuning launches in
    the first steps; read-only) and what THIS machine timed itself for other shapes (a small JSON file, kept across processes).
    A lookup asks the local layer first, then the shipped one; only locally timed keys are ever written back, so a later
    release of tune_defaults.json is not shadowed by a frozen copy of the old one.  YH_TUNE_DEFAULTS=0 ignores the shipped table."""

    def __init__(self):
        self.shipped, self.local, self.dirty = {}, {}, False
        if os.environ.get("YH_TUNE_DEFAULTS", "1") != "0":
            self.shipped = self._read(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_defaults.json"))
        self.local = self._read(_tune_cache_path())
        self.hits_shipped = self.hits_local = self.timed = 0

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return dict(json.load(f))
        except (OSError, ValueError):
            return {}

    def __contains__(self, key):
        return key in self.local or key in self.shipped

    def __getitem__(self, key):
        if key in self.local:
            self.hits_local += 1
            return self.local[key]
        self.hits_shipped += 1
        return self.shipped[key]

    def __setitem__(self, key, value):
        self.local[key] = value
        self.timed += 1
        self.dirty = True

    def save(self):
        if not self.dirty:
            return
        self.dirty = False
        path = _tune_cache_path()
        try:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(self.local, f, indent=0, sort_keys=True)
            os.replace(tmp, path)
        except OSError:
            pass                               # read-only install: tune again next time


def _tune_cache():
    if _tune_cache.data is None:
        _tune_cache.data = _TuneTable()
    return _tune_cache.data


_tune_cache.data = None


def _tune_cache_save():
    if _tune_cache.data is not None:
        _tune_cache.data.save()


def tuning_source():
    """where the launch parameters of this process came from (reported by bench.py)"""
    t = _tune_cache()
    return {"shipped_table": t.hits_shipped, "local_cache": t.hits_local, "timed_now": t.timed}


TUNE_ITERS = max(1, int(os.environ.get("YH_TUNE_ITERS", "3")))   # launches timed per candidate (tools/make_tune_defaults.sh: 12)
MERGE_PARTS = os.environ.get("YH_MERGE_PARTS", "1") != "0"   # stacked ConvBnAct layers: one BN+SiLU pass for all parts
# YH_WGRAD_PARTIAL=1: the weight gradients' split-M partial tiles go to a workspace with plain stores and are summed in split
# order by a second kernel (yh_wgrad_desc.partial) instead of fp32 atomics: BIT-REPRODUCIBLE gradients.  Measured on the YOLOv5s
# step: the weight-gradient kernels themselves get 4 % faster (3.67 -> 3.52 ms), the step 1.6 % slower (the 2.5 GB of partial
# tiles are written and read back next to an HBM-bound main chain) — so the atomic form stays the default.
# YH_FUSE_STEM_BWD: the BatchNorm backward apply of a layer without a data gradient (the stem) runs inside its weight gradient's
# operand staging (yh_wgrad_desc.bn_*; the staged gz is the apply pass's gz bit for bit, tests/test_gpu_conv.py): the last pass of
# the backward's critical path and the gz round trip through HBM disappear.  1 (default): where the patch form of the weight
# gradient takes the layer (conv_wgpf_kernel: <= 64 output channels) — measured on the YOLOv5s step 12.70 -> 12.58..12.64 ms (+0.8 %,
# profiles/r03_step_experiments.txt m); 2: also through the im2col form (conv_wgrad_kernel<..., FBN>: 0.55 ms against 0.22 + 0.25 —
# the sigmoid of 210 M elements is hidden behind HBM time in a streaming pass but not between the barriers of a 15-wave-per-CU GEMM;
# the step gets 1 % slower); 0: never.
FUSE_STEM_BWD = int(os.environ.get("YH_FUSE_STEM_BWD", "1"))
HEAD_COLSUM_SIDE = os.environ.get("YH_HEAD_COLSUM_SIDE", "1") != "0"    # bias gradients of the head layers on the weight-gradient stream
SPPF_FUSE = os.environ.get("YH_SPPF_FUSE", "1") != "0"      # FastSPP's three pools in one launch per direction (csrc/sppf.hip)
WG_WS_BYTES = (256 << 20) if os.environ.get("YH_WGRAD_PARTIAL", "0") == "1" else 0
NGZ = int(os.environ.get("YH_GZ_RING", "3"))   # gz buffers the side-stream weight gradients may lag behind by


# wide weight-gradient tilings that also exist with 64-pixel k-steps (yh_wgrad_desc.tile_k = 64): 32-pixel name -> 64-pixel name
_WGRAD_TK64 = {
    "conv_wgrad_kernel<1, 5, 1, 1, 32, 3, true, false>": "conv_wgrad_kernel<1, 5, 1, 1, 64, 3, true, false>",
    "conv_wgrad_kernel<1, 4, 1, 2, 32, 3, true, false>": "conv_wgrad_kernel<1, 4, 1, 2, 64, 2, true, false>",
    "conv_wgrad_kernel<1, 4, 1, 3, 32, 3, false, false>": "conv_wgrad_kernel<1, 4, 1, 3, 64, 2, false, false>",
    "conv_wgrad_kernel<1, 4, 2, 1, 32, 4, false, false>": "conv_wgrad_kernel<1, 4, 2, 1, 64, 2, false, false>",
    "conv_wgrad_kernel<1, 4, 2, 2, 32, 3, false, false>": "conv_wgrad_kernel<1, 4, 2, 2, 64, 2, false, false>",
}


# version prefixes of the tuning-table keys: bumped when the candidates or the meaning of a tuned value change, so that stale
# entries of a shipped / cached table are not applied.  Stride-2 data gradients carry their own version (conv_dg2_kernel, algo 7,
# joined their candidates in round 3), and so do weight gradients that leave through the partial-tile workspace.
KEY_CONV, KEY_CONV_S2D, KEY_CONV_P3, KEY_CONV_EVAL, KEY_WGRAD, KEY_WGRAD_WS = "conv6", "conv7", "conv8", "conv9", "wgrad10", "wgrad8"
KEY_CONV_C80 = "conv10"        # inference 3x3 layers with 80 -> 160 channels: conv_c80_kernel (algo 12) joined their candidates in round 4
KEY_CONV_PT = "conv11"         # training 1x1 layers with 128 / 256 / 512 input channels: conv_pt_kernel (algo 13) joined their candidates in round 5
TUNE_KEY_VERSIONS = frozenset((KEY_CONV, KEY_CONV_S2D, KEY_CONV_P3, KEY_CONV_EVAL, KEY_CONV_C80, KEY_CONV_PT, KEY_WGRAD, KEY_WGRAD_WS, KEY_WGRAD + "f", KEY_WGRAD_WS + "f"))

# YH_SKIP_ALGOS=<n>[,<n>]: leave these kernel families (yh_conv_desc.algo) out of the per-layer timing — A/B runs of a new family on
# one box (use a YH_TUNE_CACHE of its own and YH_TUNE_DEFAULTS=0 for the layers concerned)
SKIP_ALGOS = frozenset(x for x in os.environ.get("YH_SKIP_ALGOS", "").split(",") if x)

# YH_ABL_SKIP=<entry point>[,...|wgrad]: TIMING EXPERIMENTS ONLY (results are wrong) — the named launches are left out of the
# compiled programs, which gives the wall time a step would have if that family were free (profiles/r03_step_ablation.txt)
ABL_SKIP = frozenset(x for x in os.environ.get("YH_ABL_SKIP", "").split(",") if x)

# YH_EXEC=0: launch every kernel of a program from Python (one ctypes call each) instead of replaying the compiled command array
# with one yh_exec call (csrc/exec.hip)
USE_EXEC = os.environ.get("YH_EXEC", "1") != "0"


def _slot(v):
    """a command argument widened to the 8-byte slot yh_exec expects"""
    if v is None:
        return 0
    if isinstance(v, float):
        return struct.unpack("<Q", struct.pack("<d", v))[0]
    if isinstance(v, (C.Structure, C.Array)):
        return C.addressof(v)
    return int(v) & 0xFFFFFFFFFFFFFFFF


class CompiledCmds:
    """a command list as a yh_cmd array (include/yolohip.h): built once per program, replayed with one call per segment"""

    def __init__(self, L, capacity):
        self.L = L
        self.arr = (Cmd * max(capacity, 1))()
        self.n = 0
        self.names = []
        self.source = None              # the Python command list this array was compiled from

    def call(self, fn, args, stream=0, label=""):
        nargs = C.c_int32(0)
        op = self.L.yh_exec_op(fn.__name__.encode(), C.byref(nargs))
        if op < 0 or nargs.value != len(args) + 1 or nargs.value > YH_CMD_SLOTS:
            raise YoloHipError(f"{fn.__name__} [{label}] cannot be compiled into a program ({len(args)} arguments)")
        c = self.arr[self.n]
        c.op, c.nslots, c.stream = op, nargs.value, stream
        for i, v in enumerate(args):
            c.slots[i] = _slot(v)
        self.names.append(f"{fn.__name__} [{label}]")
        self.n += 1
        return self.n - 1

    def event(self, kind, handle, stream):
        c = self.arr[self.n]
        c.op, c.nslots, c.stream = kind, 1, stream
        c.slots[0] = int(handle)
        self.names.append("event record" if kind == YH_CMD_EVENT_RECORD else "stream wait")
        self.n += 1

    def run(self, streams, lo=0, hi=None):
        hi = self.n if hi is None else hi
        if hi <= lo:
            return
        failed = C.c_int32(-1)
        arr = (C.c_void_p * len(streams))(*streams)
        rc = self.L.yh_exec(C.cast(C.byref(self.arr, lo * C.sizeof(Cmd)), C.POINTER(Cmd)), hi - lo, arr, len(streams), C.byref(failed))
        if rc != 0:
            check(rc, self.names[lo + failed.value] if failed.value >= 0 else "yh_exec")


def plan_grad_buckets(marks, gsize, nbuckets):
    """Buckets of the packed gradient arena for the data-parallel exchange.  `marks` lists, in backward order,
    (command index after which an op's gradients are complete, start of the op's slice); ops are laid out in
    forward order, so the finished region grows downwards from `gsize` and every bucket is one contiguous slice.
    Returns [(cmd_index, lo, hi)]: after command cmd_index-1 the slice [lo, hi) can be all-reduced while the
    rest of the backward runs.  Cuts are made when a bucket holds >= gsize/nbuckets elements."""
    out, hi, last = [], gsize, gsize
    target = max(1, gsize // max(1, nbuckets))
    for i, (ci, lo) in enumerate(marks):
        if lo > last:
            raise YoloHipError("gradient arena is not laid out in forward op order")
        last = lo
        if hi - lo >= target or i == len(marks) - 1:
            lo_cut = 0 if i == len(marks) - 1 else lo
            out.append((ci, lo_cut, hi))
            hi = lo_cut
    return out


def _pick_bn(n):
    """output-channel tile the conv kernel picks (csrc/conv_igemm.hip pick_bn)"""
    return 32 if n <= 32 else (64 if n <= 64 else 128)


class TBuf:
    """An NHWC bf16 activation buffer (allocated per Program)."""

    def __init__(self, name, H, W, Cn, needs_grad=True):
        self.name, self.H, self.W, self.C = name, H, W, Cn
        self.needs_grad = needs_grad
        self.t = None       # (B,H,W,C) bf16
        self.g = None       # gradient buffer, same shape
        self.ginit = None   # per-channel "gradient already written" flags while planning the backward


class Ref:
    """Channel slice of a TBuf, optionally read through a 2x nearest upsample."""

    def __init__(self, buf, coff=0, Cn=None, ups=0):
        self.buf, self.coff, self.C, self.ups = buf, coff, (buf.C - coff if Cn is None else Cn), ups

    def sl(self, grad=False):
        return Slice(self.buf.g if grad else self.buf.t, self.coff, self.C, self.ups)


class ConvOp:
    def __init__(self, name, segs, parts, k, stride, pad, Hi, Wi, kind, outs, res, focus=False):
        self.name, self.segs, self.parts = name, segs, parts
        self.k, self.stride, self.pad, self.Hi, self.Wi = k, stride, pad, Hi, Wi
        self.Ho = (Hi + 2 * pad - k) // stride + 1
        self.Wo = (Wi + 2 * pad - k) // stride + 1
        self.kind, self.outs, self.res, self.focus = kind, outs, res, focus
        self.Ctot = sum(s.C for s in segs)
        self.Ktot = k * k * self.Ctot
        self.part_N = [c.out_channels for c, _ in parts]
        self.N = sum(self.part_N)
        self.Npad = _rup(self.N, 128)
        self.y = None            # raw conv output buffer (cba) / head buffer (plain)


class PoolOp:
    def __init__(self, name, src, dst):
        self.name, self.src, self.dst = name, src, dst
        self.idx = None


def sppf_chain(ops, i, L):
    """ops[i], ops[i+1], ops[i+2] = FastSPP's three max-pools chained through slices of one concat buffer (utils/layer_tools.py:282-288)
    on a map the fused kernels take (csrc/sppf.hip)?  YH_SPPF_FUSE=0: never."""
    if not SPPF_FUSE or i + 2 >= len(ops) or not all(isinstance(o, PoolOp) for o in ops[i:i + 3]):
        return False
    a, b_, c = ops[i:i + 3]
    same = lambda r1, r2: r1.buf is r2.buf and r1.coff == r2.coff and r1.C == r2.C and not r1.ups and not r2.ups   # noqa: E731
    if not (same(a.dst, b_.src) and same(b_.dst, c.src)):
        return False
    if not (a.src.buf is a.dst.buf and a.src.C == a.dst.C == b_.dst.C == c.dst.C and not a.src.ups):
        return False
    return bool(L.yh_sppf_pool3_ok(a.src.buf.H, a.src.buf.W, a.src.C))


def bn_of(m):
    """the BatchNorm of a ConvBnAct as the engine sees it.  A module that went through the reference's deployment fusion
    (detect_yolov5.py:110-116: `m.conv = fuse_conv_bn(m.conv, m.bn); delattr(m, 'bn'); m.forward = m.forward_fuse`) has a biased
    conv and no `bn`: the engine's inference program folds (gamma, beta, mean, var) into the conv epilogue anyway, so the fused
    module is described to it by a stand-in BatchNorm with gamma 1, mean 0, var 1 - eps and beta = the fused conv's bias —
    conv(x) * 1 + bias, then SiLU, exactly utils/layer_tools.py:93-94.  The stand-in is not a registered sub-module (the
    state_dict stays the fused one) and shares the conv's bias tensor."""
    bn = getattr(m, 'bn', None)
    if bn is not None:
        return bn
    conv = m.conv
    if conv.bias is None:
        raise YoloHipError("a ConvBnAct without `bn` must carry the fused conv of fuse_conv_bn (bias=True)")
    st = m.__dict__.get('_yh_fused_bn')
    if st is None or st.bias is not conv.bias:
        n, dev = conv.out_channels, conv.weight.device
        st = torch.nn.BatchNorm2d(n, eps=BN_EPS_DEFAULT).to(dev).eval()
        st.weight.requires_grad_(False)
        with torch.no_grad():
            st.running_var.fill_(1.0 - BN_EPS_DEFAULT)
        st.bias = conv.bias                       # the SAME Parameter: a later load_state_dict of the fused conv is seen
        m.__dict__['_yh_fused_bn'] = st
    return st


class Builder:
    """Collects buffers and ops for one batch/input shape."""

    def __init__(self):
        self.bufs, self.ops = [], []

    def buf(self, name, H, W, Cn, needs_grad=True):
        if Cn % 8:
            raise YoloHipError(f"{name}: channel count {Cn} must be a multiple of 8 on the HIP path")
        b = TBuf(name, H, W, Cn, needs_grad)
        self.bufs.append(b)
        return b

    def cba(self, name, mods, segs, dsts=None, res=None, focus=False):
        """ConvBnAct(s) sharing one input; returns the activation Refs (one per module)."""
        c0 = mods[0].conv
        k, s = c0.kernel_size[0], c0.stride[0]
        p = c0.padding[0]
        Hi, Wi = segs[0].buf.H << segs[0].ups, segs[0].buf.W << segs[0].ups
        if focus:                      # 6x6/s2/p2 on the image == 3x3/s1/p1 on the space-to-depth tensor
            k, s, p = 3, 1, 1
        for m in mods:
            if m.conv.groups != 1 or (m.conv.bias is not None and hasattr(m, 'bn')):
                raise YoloHipError(f"{name}: grouped / biased ConvBnAct is outside the HIP hot path")
        op = ConvOp(name, segs, [(m.conv, bn_of(m)) for m in mods], k, s, p, Hi, Wi, 'cba', None, res, focus)
        op.y = self.buf(name + ".y", op.Ho, op.Wo, op.N)
        if dsts is None:
            dsts = [Ref(self.buf(name + f".a{i}" if len(mods) > 1 else name + ".a", op.Ho, op.Wo, n)) for i, n in enumerate(op.part_N)]
        op.outs = dsts
        self.ops.append(op)
        return dsts

    def plain(self, name, conv, seg):
        """Detect 1x1 conv with bias (utils/layer_tools.py:454-470): output buffer padded to ld 256-multiple."""
        Hi, Wi = seg.buf.H, seg.buf.W
        op = ConvOp(name, [seg], [(conv, None)], conv.kernel_size[0], conv.stride[0], conv.padding[0], Hi, Wi, 'plain', None, None)
        op.y = self.buf(name + ".out", op.Ho, op.Wo, _rup(op.N, 8))
        op.y.is_head = True
        self.ops.append(op)
        return op

    def plain_multi(self, name, parts, segs):
        """Several biased 1x1 convs with DIFFERENT inputs evaluated as one block-diagonal GEMM whose output
        columns are the concatenation of the convs' outputs (YOLOX head: reg | cof | cls, yolox_s.py:128-137).
        parts: list of (conv, index of its input segment in `segs`)."""
        Hi, Wi = segs[0].buf.H, segs[0].buf.W
        c0 = parts[0][0]
        op = ConvOp(name, segs, [(c, None) for c, _ in parts], c0.kernel_size[0], c0.stride[0], c0.padding[0], Hi, Wi, 'plain', None, None)
        op.part_seg = [si for _, si in parts]
        op.y = self.buf(name + ".out", op.Ho, op.Wo, _rup(op.N, 8))
        op.y.is_head = True
        self.ops.append(op)
        return op

    def pool(self, name, src, dst):
        self.ops.append(PoolOp(name, src, dst))


# ------------------------------------------------------------------------------------------
class ParamPack:
    """Flat fp32 parameter arena + index maps to/from the packed kernel layouts."""

    def __init__(self, module, ops, host_only=False):
        """host_only=True builds only the (NumPy) index maps — used by the CPU tests of the packing logic."""
        params = list(module.parameters())
        if not params:
            raise YoloHipError("model has no parameters")
        dev = params[0].device
        if dev.type != "cuda" and not host_only:
            raise YoloHipError("yoloseries_amd models run on an MI355X device only (no CPU path in the product); call .to('cuda') first")
        self.device = dev
        self.params = params
        if any(p.dtype != torch.float32 for p in params):
            raise YoloHipError("master parameters must be float32 (bf16 copies are made by the engine)")
        sizes = [p.numel() for p in params]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.n = int(offs[-1])
        fbufs = [b for b in module.buffers() if b.dtype == torch.float32]
        nb = sum(b.numel() for b in fbufs)
        self.nbuf = nb
        self.flat = self.fbuf = None
        if not host_only:
            flat = torch.empty(self.n, dtype=torch.float32, device=dev)
            for p, o in zip(params, offs[:-1]):
                flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat[o:o + p.numel()].view(p.shape)
            self.flat = flat
            # float buffers (BatchNorm running statistics) share one arena too: EMA / DP averaging are single launches
            self.fbuf = torch.empty(max(nb, 1), dtype=torch.float32, device=dev)
            o = 0
            for b in fbufs:
                self.fbuf[o:o + b.numel()].copy_(b.data.reshape(-1))
                b.data = self.fbuf[o:o + b.numel()].view(b.shape)
                o += b.numel()
        self.off = {id(p): int(o) for p, o in zip(params, offs[:-1])}
        pack_idx, self.wloc = [], {}
        fpack_idx, self.bias_loc = [], {}
        fcur = 0
        cur = 0
        gcur = 0
        unpack = np.full(self.n, -1, dtype=np.int64)
        self.gloc, self.bn_g, self.bias_g = {}, {}, {}
        self.fused_ops = set()

        def widx(conv):
            w = conv.weight
            return self.off[id(w)] + np.arange(w.numel(), dtype=np.int64).reshape(tuple(w.shape))

        for op in ops:
            if not isinstance(op, ConvOp):
                continue
            part_seg = getattr(op, "part_seg", None)
            if part_seg is None:
                Wall = np.concatenate([widx(c) for c, _ in op.parts], axis=0)      # [N, I, kh, kw]
            else:       # block-diagonal: every part only sees the channels of its own input segment
                Wall = np.full((op.N, op.Ctot, op.k, op.k), -1, dtype=np.int64)
                seg_c0 = np.concatenate([[0], np.cumsum([sg.C for sg in op.segs])])
                r = 0
                for (conv, _), si in zip(op.parts, part_seg):
                    o = conv.out_channels
                    Wall[r:r + o, seg_c0[si]:seg_c0[si + 1]] = widx(conv)
                    r += o
            N = op.N
            if op.focus:
                n_, ci, kh6, kw6 = Wall.shape
                assert kh6 == 6 and kw6 == 6 and 4 * ci <= 16
                P = np.full((N, 3, 3, 16), -1, dtype=np.int64)
                for dy in range(2):
                    for dx in range(2):
                        for c in range(ci):
                            P[:, :, :, (dy * 2 + dx) * ci + c] = Wall[:, c, dy::2, dx::2]
                P = P.reshape(N, 9 * 16)
            else:
                P = Wall.transpose(0, 2, 3, 1).reshape(N, op.Ktot)
            Pp = np.full((op.Npad, op.Ktot), -1, dtype=np.int64)
            Pp[:N] = P
            self.wloc[(op.name, 'fwd')] = (cur, op.Npad, op.Ktot)
            pack_idx.append(Pp.reshape(-1)); cur += Pp.size
            # packed gradient image [N][Ktot] and the map back to parameter positions
            pos = gcur + np.arange(N * op.Ktot, dtype=np.int64).reshape(N, op.Ktot)
            valid = P >= 0
            unpack[P[valid]] = pos[valid]
            self.gloc[op.name] = gcur
            gcur += _rup(N * op.Ktot, 8)
            # dgrad images, one per input segment that needs a gradient
            if not op.focus:
                Nk = _rup(N, 8)
                Wk = Wall
                if Nk != N:
                    Wk = np.concatenate([Wall, np.full((Nk - N,) + Wall.shape[1:], -1, dtype=np.int64)], axis=0)
                c0 = 0
                for si, sg in enumerate(op.segs):
                    if sg.buf.needs_grad:
                        D = Wk[:, c0:c0 + sg.C].transpose(1, 2, 3, 0).reshape(sg.C, op.k * op.k * Nk)
                        Cp = _rup(sg.C, 128)
                        Dp = np.full((Cp, D.shape[1]), -1, dtype=np.int64)
                        Dp[:sg.C] = D
                        self.wloc[(op.name, 'dgrad', si)] = (cur, Cp, D.shape[1])
                        pack_idx.append(Dp.reshape(-1)); cur += Dp.size
                    c0 += sg.C
            # BN affine / bias gradients live in the packed-gradient arena too
            if op.kind == 'plain':
                # biases of all parts gathered into one fp32 vector (output-column order), gradient = column sums
                brow = np.full(_rup(N, 8), -1, dtype=np.int64)
                r = 0
                for conv, _ in op.parts:
                    if conv.bias is not None:
                        brow[r:r + conv.out_channels] = self.off[id(conv.bias)] + np.arange(conv.out_channels)
                        unpack[self.off[id(conv.bias)]:self.off[id(conv.bias)] + conv.out_channels] = gcur + r + np.arange(conv.out_channels)
                    r += conv.out_channels
                self.bias_loc[op.name] = fcur
                fpack_idx.append(brow); fcur += len(brow)
                self.bias_g[(op.name, 0)] = gcur
                gcur += _rup(N, 8)
            for pi, (conv, bn) in enumerate(op.parts):
                if op.kind == 'plain':
                    break
                if bn is not None and id(bn.weight) not in self.off:
                    self.fused_ops.add(op.name)       # stand-in BatchNorm of a deployment-fused module (bn_of): inference only
                elif bn is not None:
                    Cn = bn.weight.numel()
                    unpack[self.off[id(bn.weight)]:self.off[id(bn.weight)] + Cn] = gcur + np.arange(Cn)
                    unpack[self.off[id(bn.bias)]:self.off[id(bn.bias)] + Cn] = gcur + Cn + np.arange(Cn)
                    self.bn_g[(op.name, pi)] = (gcur, gcur + Cn)
                    gcur += _rup(2 * Cn, 8)
        if cur >= 2 ** 31 or gcur >= 2 ** 31:
            raise YoloHipError("parameter arena too large for int32 index maps")
        self.pack_idx_np = np.concatenate(pack_idx).astype(np.int32)
        self.unpack_idx_np = unpack.astype(np.int32)
        self.fpack_idx_np = np.concatenate(fpack_idx).astype(np.int32) if fpack_idx else None
        self.gsize = gcur
        if host_only:
            return
        self.pack_idx = torch.from_numpy(self.pack_idx_np).to(dev)
        self.unpack_idx = torch.from_numpy(self.unpack_idx_np).to(dev)
        self.wpack = torch.zeros(cur, dtype=torch.bfloat16, device=dev)
        self.fpack_idx = torch.from_numpy(self.fpack_idx_np).to(dev) if fpack_idx else None
        self.fpack = torch.zeros(max(fcur, 8), dtype=torch.float32, device=dev)
        self.gsize = gcur
        self.gpack = torch.zeros(max(gcur, 8), dtype=torch.float32, device=dev)
        self.packed_version = -1

    def valid_for(self, module):
        ps = list(module.parameters())
        return len(ps) == len(self.params) and all(a is b for a, b in zip(ps, self.params)) and \
            ps[0].data_ptr() == self.flat.data_ptr() and ps[0].device == self.device

    def still_valid(self):
        """cheap form of valid_for() for a module this pack was already validated against: its first and last parameter still
        alias the arena where the pack put them (a .to() / load with assign= / re-created parameter moves them)"""
        p0, p1 = self.params[0], self.params[-1]
        return p0.data_ptr() == self.flat.data_ptr() and p1.data_ptr() == self.flat.data_ptr() + 4 * (self.n - p1.numel())

    def repack(self):
        hipk.pack_bf16(self.flat, self.pack_idx, self.wpack)
        if self.fpack_idx is not None:
            hipk.gather_f32(self.flat, self.fpack_idx, self.fpack)

    def wptr(self, key):
        off, rows, K = self.wloc[key]
        return self.wpack.data_ptr() + 2 * off, rows, K

    def grads_to_params(self):
        """packed fp32 gradients -> one flat gradient in parameter order (fresh tensor per call)."""
        flat_g = torch.empty(self.n, dtype=torch.float32, device=self.device)
        hipk.gather_f32(self.gpack, self.unpack_idx, flat_g)
        outs, o = [], 0
        for p in self.params:
            outs.append(flat_g[o:o + p.numel()].view(p.shape))
            o += p.numel()
        return flat_g, outs


# ------------------------------------------------------------------------------------------
class Program:
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

    # -- helpers -------------------------------------------------------------------------
    def _conv_desc(self, op, train, segs=None):
        pk = self.pack
        wp, npad, K = pk.wptr((op.name, 'fwd'))
        assert K == op.Ktot
        d = ConvDesc()
        segs = op.segs if segs is None else segs
        for i, sg in enumerate(segs):
            d.seg[i] = hipk.make_seg(sg.sl())
        d.nseg, d.mode = len(segs), YH_CONV_FWD
        d.B, d.Ho, d.Wo, d.Hi, d.Wi = self.B, op.Ho, op.Wo, op.Hi, op.Wi
        d.KH = d.KW = op.k
        d.stride, d.pad = op.stride, op.pad
        d.w, d.N, d.Npad = wp, op.N, npad
        return d

    def _tune_conv(self, d, kind, name, stats_ok=False):
        """Launch parameters of one conv / dgrad launch — kernel family (register-staged conv_v2 or LDS-DMA conv_v3 with one
        of its tiles), k-step width, cap on persistent blocks — timed once when the program is built: the best setting
        differs per layer shape by 5-40 % (YH_CONV_TUNE=0: library defaults).  Results never change (identical math);
        only the number of BatchNorm partial-sum rows follows the grid."""
        if os.environ.get("YH_CONV_TUNE", "1") == "0":
            return
        small3 = d.KH == 3 and d.stride == 1 and d.nseg == 1 and d.seg[0].C <= 128 and d.N <= 128 and kind != 'eval'
        c80 = kind == 'eval' and d.KH == 3 and d.nseg == 1 and d.seg[0].C == 80 and d.N == 160
        ctot = d.seg[0].C + (d.seg[1].C if d.nseg > 1 else 0)
        pt = kind != 'eval' and d.KH == 1 and d.stride == 1 and ctot in (128, 256, 512) and (d.nseg == 1 or d.seg[0].C == d.seg[1].C)
        key = f"{KEY_CONV_S2D if d.mode == YH_CONV_DGRAD and d.stride == 2 else (KEY_CONV_P3 if small3 else ((KEY_CONV_C80 if c80 else KEY_CONV_EVAL) if kind == 'eval' else (KEY_CONV_PT if pt else KEY_CONV)))}:{kind}:" + ",".join(str(int(v)) for v in (
            d.mode, d.B, d.Ho, d.Wo, d.Hi, d.Wi, d.KH, d.stride, d.pad, d.N, d.nseg, d.seg[0].C, d.seg[0].ld, d.seg[0].ups,
            d.seg[1].C if d.nseg > 1 else 0, d.seg[1].ups if d.nseg > 1 else 0, d.ld0, d.nsplit, d.accumulate, int(bool(d.stats or stats_ok)),
            int(bool(d.res)), d.act, int(bool(d.bias)), int(bool(d.scale)), int(bool(d.bnr_part)), 0))
        cache = _tune_cache()
        if key in cache:
            d.tile_k, d.grid_cap, d.algo = (int(v) for v in cache[key])
            return
        L = self.L
        saved = (d.seg[0].ptr, d.stats)
        if not d.seg[0].ptr:
            d.seg[0].ptr = self.gy_scratch.data_ptr()
        # candidates: (algo, tile_k, grid_cap)
        d.tile_k = d.grid_cap = 0
        d.algo = 1
        base = L.yh_conv_stat_blocks(C.byref(d))
        cands = []
        tks = (0, 32) if all(d.seg[i].C % 64 == 0 for i in range(d.nseg)) and d.N > 64 else (0,)
        for tk in tks:
            for cap in (0, 2 * base):
                d.tile_k, d.grid_cap = tk, cap
                if cap and L.yh_conv_stat_blocks(C.byref(d)) == base:
                    continue                       # fewer tiles than blocks: the cap changes nothing
                cands.append((1, tk, cap))
        d.tile_k = d.grid_cap = 0
        if os.environ.get("YH_CONV_V3", "1") != "0":
            for algo in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13):
                if str(algo) in SKIP_ALGOS:
                    continue
                d.algo = algo
                kn = self._kernel_name(d)
                if ("conv_v3" in kn and algo < 5) or ("conv_halo_kernel" in kn and algo == 5) or ("conv_halo160" in kn and algo == 6) or \
                        ("conv_dg2" in kn and algo == 7) or ("conv_p3" in kn and algo == 8) or ("conv_h80" in kn and algo == 9) or ("conv_pw" in kn and algo == 10) or ("conv_c80" in kn and algo == 12) or \
                        ("conv_pt" in kn and algo == 13):
                    cands.append((algo, 0, 0))
                    if algo < 5 and kn.endswith(", true>") and all(d.seg[i].C % 32 == 0 for i in range(d.nseg)):
                        cands.append((algo, 32, 0))    # ragged last channel block: 32-channel steps instead of 64 + tail
        rows_max, bnr_max = 1, 1
        for algo, tk, cap in cands:
            d.algo, d.tile_k, d.grid_cap = algo, tk, cap
            rows_max = max(rows_max, L.yh_conv_stat_blocks(C.byref(d)))
            if d.bnr_part:
                bnr_max = max(bnr_max, L.yh_conv_bnr_rows(C.byref(d)))
        tmp_stats = None
        if stats_ok:
            tmp_stats = torch.zeros(rows_max + 8, 2, d.Npad, dtype=torch.float32, device=self.dev)
            d.stats = tmp_stats.data_ptr()
        saved_part, tmp_part = d.bnr_part, None
        if d.bnr_part:                   # a slab big enough for every grid tried below
            tmp_part = torch.zeros((bnr_max + 8) * 2 * d.N, dtype=torch.float32, device=self.dev)
            d.bnr_part = tmp_part.data_ptr()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        best, best_ms = (0, 0, 1), None
        for algo, tk, cap in cands:
            d.algo, d.tile_k, d.grid_cap = algo, tk, cap
            check(L.yh_conv_igemm(C.byref(d), st), f"yh_conv_igemm tune [{name}]")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(TUNE_ITERS):
                L.yh_conv_igemm(C.byref(d), st)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            if best_ms is None or ms < best_ms * (0.97 if TUNE_ITERS < 8 else 0.99):   # keep the earlier candidate unless clearly better
                best, best_ms = (tk, cap, algo), ms
        d.tile_k, d.grid_cap, d.algo = best
        d.seg[0].ptr, d.stats = saved
        d.bnr_part = saved_part
        cache[key] = [int(best[0]), int(best[1]), int(best[2])]

    def _eval_concat_plan(self):
        """Inference only: a conv that reads concat(t, upper half of a buffer `cat`) — C3's cba3 (utils/layer_tools.py:152-169) — where
        t is produced by ONE conv (the last bottleneck's 3x3) and cat's lower half (cba1's output) has no reader behind that conv:
        the producer writes over the lower half instead, and the reader becomes a conv over ONE 2*mid-channel segment (whole
        cache lines per pixel, every kernel family eligible; with 80 + 80 channels the two-segment form fell back to the generic
        kernel).  A producer that reads the lower half as its residual does so element by element before it stores the same element.
        Training keeps both buffers: cba1's activation is needed by the backward.  YH_EVAL_INPLACE_CAT=0: off.
        Returns ({producer op: Ref}, {reader op: [Ref]})."""
        outs, segs = {}, {}
        if os.environ.get("YH_EVAL_INPLACE_CAT", "1") == "0":
            return outs, segs
        ops = self.ops
        for ci, c3 in enumerate(ops):
            if not isinstance(c3, ConvOp) or c3.kind != 'cba' or len(c3.segs) != 2:
                continue
            s0, s1 = c3.segs
            cat, tb = s1.buf, s0.buf
            if s0.ups or s1.ups or tb is cat or s0.coff or tb.C != s0.C or s1.coff != s0.C or cat.C != s0.C + s1.C:
                continue
            prods = [(i, o) for i, o in enumerate(ops) if isinstance(o, ConvOp) and o.kind == 'cba' and any(r.buf is tb for r in o.outs)]
            if len(prods) != 1 or len(prods[0][1].outs) != 1 or prods[0][0] >= ci:
                continue
            pi, prod = prods[0]
            lower = lambda r: r is not None and r.buf is cat and r.coff < s0.C      # noqa: E731
            ok = True
            for i, o in enumerate(ops):
                if isinstance(o, PoolOp):
                    ok &= not (o.src.buf is tb or o.dst.buf is tb or lower(o.src) or lower(o.dst))
                    continue
                if o is not c3 and (any(sg.buf is tb for sg in o.segs) or (o.res is not None and o.res.buf is tb)):
                    ok = False                                   # t has another reader
                if any(lower(sg) for sg in o.segs) and i >= pi:
                    ok = False                                   # the lower half is a conv input at or behind the producer
                if lower(o.res) and (i > pi or (i == pi and not (o.res.coff == 0 and o.res.C == s0.C))):
                    ok = False
                if o.kind == 'cba' and any(lower(r) for r in o.outs) and i >= pi:
                    ok = False
            if ok:
                outs[prod] = Ref(cat, 0, s0.C)
                segs[c3] = [Ref(cat, 0, cat.C)]
        return outs, segs

    def _build_forward(self):
        """inference program (folded BatchNorm + SiLU in the conv epilogue).  The training program — raw conv outputs,
        statistics, BatchNorm work buffers, pool arg-max — is built by _build_train() at the first training forward, so an
        evaluation-only model never allocates the pre-activation tensors (half of the activation memory)."""
        B, pk, L = self.B, self.pack, self.L
        self.cmd_train, self.cmd_eval = None, []
        self.op_state = {}
        fold_items = []                 # every BatchNorm of the net is folded to (scale, shift) by ONE launch ahead of the convs
        skip = set()
        cat_outs, cat_segs = self._eval_concat_plan()
        for oi, op in enumerate(self.ops):
            if oi in skip:
                continue
            if isinstance(op, PoolOp) and sppf_chain(self.ops, oi, L):
                p1, p2, p3 = self.ops[oi:oi + 3]
                s = p1.src.sl()
                d1, d2, d3 = p1.dst.sl(), p2.dst.sl(), p3.dst.sl()
                Hs, Ws = p1.src.buf.H, p1.src.buf.W
                self.cmd_eval.append((L.yh_sppf_pool3_fwd, (s.ptr(), s.ld, B, Hs, Ws, s.C, d1.ptr(), d2.ptr(), d3.ptr(), d1.ld, None, None, None),
                                      p1.name, ('yh_sppf_pool3_fwd', 0, 8.0 * B * Hs * Ws * s.C)))
                skip.update((oi + 1, oi + 2))
                continue
            if isinstance(op, PoolOp):
                s, dd = op.src.sl(), op.dst.sl()
                args = (s.ptr(), s.ld, B, op.src.buf.H, op.src.buf.W, s.C, dd.ptr(), dd.ld, None)
                self.cmd_eval.append((L.yh_maxpool5_fwd, args, op.name, ('yh_maxpool5_fwd', 0, 4.0 * B * op.src.buf.H * op.src.buf.W * s.C)))
                continue
            st = {}
            self.op_state[op.name] = st
            if op.kind == 'plain':
                d = self._conv_desc(op, True)
                d.bias = pk.fpack.data_ptr() + 4 * pk.bias_loc[op.name]
                d.act = YH_ACT_NONE
                d.out0, d.ld0, d.nsplit = pk.wpack.data_ptr(), op.y.C, op.N   # placeholder: head buffers are fresh tensors per forward
                st['desc'] = d
                st['fam'] = self._fam_conv(op, d)
                self.cmd_eval.append((L.yh_conv_igemm, (d,), op.name, st['fam']))
                continue
            # folded BN + SiLU (+ residual) in the conv epilogue
            de = self._conv_desc(op, False, cat_segs.get(op))
            op_outs = [cat_outs[op]] if op in cat_outs else op.outs
            st['fold'] = torch.zeros(2, op.N, dtype=torch.float32, device=self.dev)
            c0 = 0
            for (conv, bn), n in zip(op.parts, op.part_N):
                it = BnFoldItem()
                it.gamma, it.beta, it.rm, it.rv = bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                it.scale, it.shift = st['fold'].data_ptr() + 4 * c0, st['fold'].data_ptr() + 4 * (op.N + c0)
                it.eps, it.C = float(bn.eps), n
                fold_items.append(it)
                c0 += n
            de.scale, de.shift = st['fold'].data_ptr(), st['fold'].data_ptr() + 4 * op.N
            de.act = YH_ACT_SILU
            o0 = op_outs[0].sl()
            de.out0, de.ld0, de.nsplit = o0.ptr(), o0.ld, op.part_N[0] if len(op_outs) > 1 else op.N
            if len(op_outs) > 1:
                o1 = op_outs[1].sl()
                de.out1, de.ld1 = o1.ptr(), o1.ld
                assert len(op_outs) == 2
            if op.res is not None:
                r = op.res.sl()
                de.res, de.ldr = r.ptr(), r.ld
            st['desc_eval'] = de
            self._tune_conv(de, 'eval', op.name)
            self.cmd_eval.append((L.yh_conv_igemm, (de,), op.name, self._fam_conv(op, de)))
        if fold_items:
            arr = (BnFoldItem * len(fold_items))(*fold_items)
            self.fold_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
            self.cmd_eval.insert(0, (L.yh_bn_fold_batch, (self.fold_table.data_ptr(), len(fold_items)), "bn_fold", ('yh_bn_fold', 0, 0.0)))
        _tune_cache_save()

    def _build_train(self):
        """training program: conv (+ per-block BatchNorm partial sums) -> finalize -> BN+SiLU apply (+ residual)"""
        B, pk, L = self.B, self.pack, self.L
        if pk.fused_ops:
            raise YoloHipError(f"a model whose ConvBnAct layers went through fuse_conv_bn ({len(pk.fused_ops)} of them) has no BatchNorm "
                               "left to train: it runs the inference program only (call .eval() under torch.no_grad())")
        for b in self.bufs:
            if b.t is None and not getattr(b, "is_head", False):
                b.t = torch.zeros(B, b.H, b.W, b.C, dtype=torch.bfloat16, device=self.dev)
        self.cmd_train = []
        self.cmd_frozen = None           # derived from cmd_train (evaluation-mode BatchNorm under autograd): rebuilt with it
        self.cmd_bwd_frozen = None
        skip = set()
        for oi, op in enumerate(self.ops):
            if oi in skip:
                continue
            if isinstance(op, PoolOp) and sppf_chain(self.ops, oi, L):
                p1, p2, p3 = self.ops[oi:oi + 3]
                Hs, Ws = p1.src.buf.H, p1.src.buf.W
                for q in (p1, p2, p3):
                    q.idx = torch.zeros(B, Hs, Ws, q.src.C, dtype=torch.int8, device=self.dev)
                s = p1.src.sl()
                d1, d2, d3 = p1.dst.sl(), p2.dst.sl(), p3.dst.sl()
                self.cmd_train.append((L.yh_sppf_pool3_fwd, (s.ptr(), s.ld, B, Hs, Ws, s.C, d1.ptr(), d2.ptr(), d3.ptr(), d1.ld,
                                                             p1.idx.data_ptr(), p2.idx.data_ptr(), p3.idx.data_ptr()),
                                       p1.name, ('yh_sppf_pool3_fwd', 0, 11.0 * B * Hs * Ws * s.C)))
                skip.update((oi + 1, oi + 2))
                continue
            if isinstance(op, PoolOp):
                op.idx = torch.zeros(B, op.src.buf.H, op.src.buf.W, op.src.C, dtype=torch.int8, device=self.dev)
                s, dd = op.src.sl(), op.dst.sl()
                args = (s.ptr(), s.ld, B, op.src.buf.H, op.src.buf.W, s.C, dd.ptr(), dd.ld, op.idx.data_ptr())
                self.cmd_train.append((L.yh_maxpool5_fwd, args, op.name, ('yh_maxpool5_fwd', 0, 5.0 * B * op.src.buf.H * op.src.buf.W * s.C)))
                continue
            M = B * op.Ho * op.Wo
            st = self.op_state[op.name]
            if op.kind == 'plain':
                self.cmd_train.append((L.yh_conv_igemm, (st['desc'],), op.name, st['fam']))
                continue
            d = self._conv_desc(op, True)
            d.act = YH_ACT_NONE
            d.out0, d.ld0, d.nsplit = op.y.t.data_ptr(), op.y.C, op.N
            self._tune_conv(d, 'fwd', op.name, stats_ok=True)
            nblk = L.yh_conv_stat_blocks(C.byref(d))
            st['stats'] = torch.zeros(nblk, 2, op.Npad, dtype=torch.float32, device=self.dev)
            d.stats = st['stats'].data_ptr()
            st['desc_train'] = d
            self.cmd_train.append((L.yh_conv_igemm, (d,), op.name, self._fam_conv(op, d)))
            st['ws'] = []
            c0 = 0
            merged = MERGE_PARTS and 2 <= len(op.parts) <= YH_BN_MAX_PARTS and op.res is None
            parts_arr = (BnPart * len(op.parts))() if merged else None
            for pi, ((conv, bn), n) in enumerate(zip(op.parts, op.part_N)):
                ws = torch.zeros(4 * n, dtype=torch.float32, device=self.dev)
                st['ws'].append(ws)
                mom = bn.momentum if bn.momentum is not None else 0.1
                dst = op.outs[pi].sl()
                res = op.res.sl() if (op.res is not None and pi == 0) else None
                if merged:
                    pa = parts_arr[pi]
                    pa.ws, pa.C, pa.out, pa.ldo = ws.data_ptr(), n, dst.ptr(), dst.ld
                    pa.slab, pa.nblk, pa.ldslab = st['stats'].data_ptr() + 4 * c0, nblk, op.Npad
                    pa.gamma, pa.beta, pa.eps, pa.momentum = bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(mom)
                    pa.running_mean, pa.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    pa.num_batches = bn.num_batches_tracked.data_ptr()
                    c0 += n
                    continue
                self.cmd_train.append((L.yh_bn_finalize, (
                    st['stats'].data_ptr() + 4 * c0, nblk, op.Npad, n, M, bn.weight.data_ptr(), bn.bias.data_ptr(),
                    bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                    float(bn.eps), float(mom), ws.data_ptr()), op.name, ('yh_bn_finalize', 0, 8.0 * nblk * n)))
                self.cmd_train.append((L.yh_bn_silu_apply, (
                    op.y.t.data_ptr() + 2 * c0, op.y.C, ws.data_ptr(), n, M, dst.ptr(), dst.ld,
                    res.ptr() if res else None, res.ld if res else 0), op.name, ('yh_bn_silu_apply', 0, (6.0 if res else 4.0) * M * n)))
                c0 += n
            if merged:          # one finalize launch and one pass over the whole rows of y for all parts
                self._keep.append(parts_arr)
                self.cmd_train.append((L.yh_bn_finalize_parts, (parts_arr, len(op.parts), M), op.name,
                                       ('yh_bn_finalize', 0, 8.0 * nblk * op.N)))
                self.cmd_train.append((L.yh_bn_silu_apply_parts, (op.y.t.data_ptr(), op.y.C, M, parts_arr, len(op.parts)), op.name,
                                       ('yh_bn_silu_apply_parts', 0, 4.0 * M * op.N)))

    # -- forward ---------------------------------------------------------------------------
    def _kernel_name(self, d):
        """instantiation yh_conv_igemm launches for descriptor d, spelled as rocprofv3 prints it"""
        buf = C.create_string_buffer(96)
        saved = d.seg[0].ptr
        if not saved:                      # head gradient pointer is filled in at run time
            d.seg[0].ptr = self.gy_scratch.data_ptr()
        rc = self.L.yh_conv_kernel_name(C.byref(d), buf, 96)
        d.seg[0].ptr = saved
        check(rc, "yh_conv_kernel_name")
        return buf.value.decode()

    @staticmethod
    def _conv_bytes(d):
        """algorithmic HBM bytes of one yh_conv_igemm launch: every input segment read once, the output written once (read too when
        it accumulates), residual / fused-reduction operands read once; weights are negligible next to the activations"""
        rd = sum(2.0 * d.B * (d.Hi >> d.seg[i].ups) * (d.Wi >> d.seg[i].ups) * d.seg[i].C for i in range(d.nseg))
        out = 2.0 * d.B * d.Ho * d.Wo * d.N
        return rd + out * (2.0 if d.accumulate else 1.0) + (out * min(1.0, d.nsplit / max(d.N, 1)) if d.res else 0.0) + (out if d.bnr_part else 0.0)

    def _head_on_side(self, op, two):
        """does a head layer's bias gradient (column sums of the head gradient) run on the weight-gradient stream?  Only with a
        scratch of its own: `part_scratch` belongs to the main stream's BatchNorm reductions (a layer wider than `head_scratch` was
        sized for — it is sized from the widest plain op of the graph, so none today — falls back to the main stream, never to a
        shared buffer on another stream)"""
        return bool(two and HEAD_COLSUM_SIDE and op.y.C * 1024 * 2 <= self.head_scratch.numel())

    def _head_scratch(self, op, two):
        """partial-sum scratch of a head layer's bias gradient: its own buffer when the column sums run on the side stream"""
        return self.head_scratch.data_ptr() if self._head_on_side(op, two) else self.part_scratch.data_ptr()

    def _is_fused_stem(self, op):
        """a ConvBnAct without a data gradient (the stem) whose BatchNorm backward apply runs inside its weight gradient (YH_FUSE_STEM_BWD)"""
        if not isinstance(op, ConvOp) or op.kind != 'cba':
            return False
        Kseg0 = op.k * op.k * op.segs[0].C
        return bool(FUSE_STEM_BWD and len(op.parts) == 1 and len(op.segs) == 1 and op.res is None and
                    not op.segs[0].buf.needs_grad and op.N % 8 == 0 and
                    ((op.N <= 32 and Kseg0 <= 256) or (op.N > 32 and 128 < Kseg0 <= 256)) and
                    (FUSE_STEM_BWD >= 2 or (self.wg_ws is None and self._stem_patch_ok(op))))

    def _stem_patch_ok(self, op):
        """does the patch form of the weight gradient (conv_wgpf_kernel) take this layer with the fused BatchNorm backward?"""
        wd = WgradDesc()
        wd.gy, wd.ldg, wd.N = op.y.t.data_ptr(), op.N, op.N
        wd.bn_z, wd.bn_ldz = op.y.t.data_ptr(), op.y.C
        wd.bn_ws = wd.bn_gamma = wd.bn_coef = op.y.t.data_ptr()          # placeholders: eligibility only looks at null / alignment
        wd.seg = hipk.make_seg(op.segs[0].sl())
        wd.coff_k, wd.Ctot = 0, op.Ctot
        wd.B, wd.Ho, wd.Wo, wd.Hi, wd.Wi = self.B, op.Ho, op.Wo, op.Hi, op.Wi
        wd.KH = wd.KW = op.k
        wd.stride, wd.pad = op.stride, op.pad
        wd.tile_k = 40
        return bool(self.L.yh_conv_wgrad_patch_ok(C.byref(wd)))

    def _fam_conv(self, op, d):
        M = self.B * op.Ho * op.Wo
        return (self._kernel_name(d), 2.0 * M * op.N * op.k * op.k * (12 if op.focus else op.Ctot), self._conv_bytes(d))

    def _compile(self, cmds):
        cc = CompiledCmds(self.L, len(cmds))
        for fn, args, name, meta in cmds:
            if getattr(fn, "__name__", "") in ABL_SKIP:
                continue
            cc.call(fn, args, 0, name)
        return cc

    def _run(self, cmds):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        prof = self.profile
        if prof is None and USE_EXEC:
            key = 'train' if cmds is self.cmd_train else ('frozen' if cmds is getattr(self, "cmd_frozen", None) else 'eval')
            cc = self._compiled.get(key)
            if cc is None or cc.source is not cmds:
                cc = self._compiled[key] = self._compile(cmds)
                cc.source = cmds
            cc.run([st.value])
            return
        for fn, args, name, meta in cmds:
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn(*args, st)
            if rc != 0:
                check(rc, f"{getattr(fn, '__name__', fn)} [{name}]")
            if prof is not None:
                e1.record()
                prof.setdefault(meta + (name,), []).append((e0, e1))

    def _frozen_cmds(self):
        """the training program with every BatchNorm in evaluation mode (model.eval() under autograd): the finalize launches —
        batch statistics -> constants, running-statistics update — give way to yh_bn_frozen (constants from the running
        statistics); the conv kernels still emit their partial sums, nobody reads them"""
        L, out = self.L, []
        for cmd in self.cmd_train:
            fn, args = cmd[0], cmd[1]
            if fn is L.yh_bn_finalize:
                _stats, _nblk, _ld, n, _M, gamma, beta, rm, rv, _nbt, eps, _mom, ws = args
                out.append((L.yh_bn_frozen, (gamma, beta, rm, rv, eps, n, ws), cmd[2], ('yh_bn_frozen', 0, 0.0)))
            elif fn is L.yh_bn_finalize_parts:
                parts, nparts, _M = args
                for i in range(nparts):
                    q = parts[i]
                    out.append((L.yh_bn_frozen, (q.gamma, q.beta, q.running_mean, q.running_var, float(q.eps), int(q.C), q.ws), cmd[2],
                                ('yh_bn_frozen', 0, 0.0)))
            else:
                out.append(cmd)
        return out

    def forward(self, train, frozen=False):
        self.generation += 1
        # fresh head buffers every call: the returned views must not be overwritten by the next forward
        for o in self.outputs:
            if isinstance(o, ConvOp):
                o.y.t = torch.empty(self.B, o.y.H, o.y.W, o.y.C, dtype=torch.bfloat16, device=self.dev)
                self.op_state[o.name]['desc'].out0 = o.y.t.data_ptr()
        if train and self.cmd_train is None:
            self._build_train()
        if train and frozen:
            if getattr(self, "cmd_frozen", None) is None:
                self.cmd_frozen = self._frozen_cmds()
            self._run(self.cmd_frozen)
        else:
            self._run(self.cmd_train if train else self.cmd_eval)
        return self.generation

    # -- backward --------------------------------------------------------------------------
    def _build_backward(self):
        B, pk, L = self.B, self.pack, self.L
        cmds = []
        self._wgrad_on_main = set()      # ids of the weight-gradient descriptors that stay on the main stream
        max_gy = 0
        for b in self.bufs:
            b.ginit = np.zeros(b.C, dtype=bool)
            if b.needs_grad and b.g is None and not b.name.endswith(".y") and not getattr(b, "is_head", False):
                b.g = torch.zeros(B, b.H, b.W, b.C, dtype=torch.bfloat16, device=self.dev)
        for op in self.ops:
            if isinstance(op, ConvOp) and op.kind == 'cba':
                max_gy = max(max_gy, B * op.Ho * op.Wo * op.N)
        self.gy_scratch = torch.zeros(max(max_gy, 8), dtype=torch.bfloat16, device=self.dev)
        # the weight gradients run on a side stream next to the data-gradient / BatchNorm chain (they only share the
        # layer's gz): consecutive layers alternate between two gz buffers so that a layer's wgrad may still be
        # reading its gz while the next layer's BN backward writes the other one
        self.two_streams = os.environ.get("YH_BWD_STREAMS", "1") != "0"
        self.gy_ring = [self.gy_scratch] + ([torch.zeros(max(max_gy, 8), dtype=torch.bfloat16, device=self.dev) for _ in range(NGZ - 1)] if self.two_streams else [self.gy_scratch] * (NGZ - 1))
        n_cba = 0
        self.part_scratch = torch.zeros(1024 * 2 * 2048, dtype=torch.float32, device=self.dev)
        # partial sums of the head layers' bias gradients (column sums of the head gradients): these run on the SIDE stream (they feed
        # nothing but the packed gradient arena; the largest takes 76 us on YOLOv5s) and may not share a scratch with the main stream
        head_c = max([256] + [op.y.C for op in self.ops if isinstance(op, ConvOp) and op.kind == 'plain'])
        self.head_scratch = torch.zeros(1024 * 2 * head_c, dtype=torch.float32, device=self.dev)
        self.coef_scratch = {}
        self.ups_scratch = {}
        self.wgrad_tuned = {}
        # workspace of the weight gradients' split-M partial tiles (plain stores + a deterministic reduce instead of fp32
        # atomics; YH_WGRAD_PARTIAL=0: atomics).  One buffer serves every launch: they all run on one stream, in order.
        self.wg_ws = torch.empty(WG_WS_BYTES // 4, dtype=torch.float32, device=self.dev) if WG_WS_BYTES > 0 else None

        writes_seen = {}

        def claim(ref):
            """returns accumulate flag for a write into grad(ref) and marks it written"""
            k3 = (ref.buf.name, ref.coff, ref.C)
            writes_seen[k3] = writes_seen.get(k3, 0) + 1
            flags = ref.buf.ginit[ref.coff:ref.coff + ref.C]
            if flags.all():
                return 1
            if flags.any():
                raise YoloHipError(f"partial gradient overlap on {ref.buf.name}")
            flags[:] = True
            return 0

        def require(ref, who):
            if not ref.buf.ginit[ref.coff:ref.coff + ref.C].all():
                raise YoloHipError(f"{who}: gradient of {ref.buf.name}[{ref.coff}:{ref.coff + ref.C}] is never produced")

        # head outputs receive their gradient from the caller
        for o in self.outputs:
            if isinstance(o, ConvOp):
                o.y.ginit[:] = True
            else:
                o.buf.ginit[o.coff:o.coff + o.C] = True

        # Which ConvBnAct outputs get the LAST contribution to their gradient from a data-gradient launch (a conv reads exactly
        # that slice, not upsampled; residual adds / pools / other convs that read it come later in the forward, so their
        # gradient is already in the buffer)?  For those the BatchNorm-backward reduction is taken in that dgrad's epilogue
        # (yh_conv_desc.bnr_*, accumulating where it is not the only writer) and the separate reduce pass is dropped.
        fuse_ok = os.environ.get("YH_FUSE_BNR", "1") != "0"
        fuse_acc = os.environ.get("YH_FUSE_BNR_ACC", "1") != "0"     # ... also when that launch accumulates onto earlier writers
        uses, producer_of = {}, {}
        for o2 in self.ops:
            if isinstance(o2, PoolOp):
                uses.setdefault((o2.src.buf.name, o2.src.coff, o2.src.C), []).append(('pool',))
                continue
            for sg2 in o2.segs:
                uses.setdefault((sg2.buf.name, sg2.coff, sg2.C), []).append(('seg', sg2.ups))
            if o2.res is not None:
                uses.setdefault((o2.res.buf.name, o2.res.coff, o2.res.C), []).append(('res',))
            if o2.kind == 'cba':
                c0_ = 0
                for pi2, n2 in enumerate(o2.part_N):
                    r2 = o2.outs[pi2]
                    producer_of[(r2.buf.name, r2.coff, r2.C)] = (o2, pi2, c0_)
                    c0_ += n2
        for o2 in self.outputs:
            if not isinstance(o2, ConvOp):
                uses.setdefault((o2.buf.name, o2.coff, o2.C), []).append(('out',))

        def last_writer(key):
            """asked by a data-gradient launch that has just claimed `key` (a non-upsampled conv segment): was that the LAST write
            into this gradient slice — every other consumer (conv segments, residual adds, pools) comes later in the forward
            and so earlier in this program — with no differently-sliced use of the same buffer overlapping it?"""
            u = uses.get(key, [])
            if any(x[0] == 'out' for x in u) or writes_seen.get(key, 0) != len(u):
                return False
            return not any(k2[0] == key[0] and k2 != key and not (k2[1] + k2[2] <= key[1] or k2[1] >= key[1] + key[2]) for k2 in uses)
        self.bnr_fused = {}
        marks = []
        skip_bwd = set()
        nops = len(self.ops)
        for ri, op in enumerate(reversed(self.ops)):
            oi = nops - 1 - ri
            if oi in skip_bwd:
                continue
            if isinstance(op, PoolOp) and oi >= 2 and sppf_chain(self.ops, oi - 2, L):
                # the chain's backward in one launch; needs every pool's input gradient to exist already (cba2's data gradient wrote
                # all four slices of the concat buffer earlier in this program) — else the three single launches below
                p1, p2, p3 = self.ops[oi - 2:oi + 1]
                flags = lambda r: r.buf.ginit[r.coff:r.coff + r.C]        # noqa: E731
                if flags(p3.dst).all() and flags(p3.src).all() and flags(p2.src).all():
                    acc1 = claim(p1.src)
                    claim(p2.src); claim(p3.src)
                    g1, g2, g3, gx_ = p1.dst.sl(True), p2.dst.sl(True), p3.dst.sl(True), p1.src.sl(True)
                    Hs, Ws = p1.src.buf.H, p1.src.buf.W
                    cmds.append((L.yh_sppf_pool3_bwd, (g1.ptr(), g2.ptr(), g3.ptr(), g1.ld, p1.idx.data_ptr(), p2.idx.data_ptr(), p3.idx.data_ptr(),
                                                       B, Hs, Ws, g1.C, gx_.ptr(), gx_.ld, acc1), p1.name,
                                 ('yh_sppf_pool3_bwd', 0, (13.0 if acc1 else 11.0) * B * Hs * Ws * g1.C)))
                    skip_bwd.update((oi - 1, oi - 2))
                    continue
            if isinstance(op, PoolOp):
                require(op.dst, op.name)
                acc = claim(op.src)
                go, gi = op.dst.sl(True), op.src.sl(True)
                cmds.append((L.yh_maxpool5_bwd, (go.ptr(), go.ld, op.idx.data_ptr(), B, op.src.buf.H, op.src.buf.W, go.C,
                                                 gi.ptr(), gi.ld, acc), op.name, ('yh_maxpool5_bwd', 0, (7.0 if acc else 5.0) * B * op.src.buf.H * op.src.buf.W * go.C)))
                continue
            M = B * op.Ho * op.Wo
            st = self.op_state[op.name]
            gdw = pk.gpack.data_ptr() + 4 * pk.gloc[op.name]
            gys = self.gy_scratch
            if op.kind == 'cba':
                gys = self.gy_ring[n_cba % NGZ]
                cmds.append(('gz_begin', n_cba % NGZ, None, ('sync', 0, 0.0)))       # main stream: wait until this gz buffer's last wgrad is done
                n_cba += 1
            if op.kind == 'plain':
                # gradient arrives in op.y.g (set per call); bias grad = column sums
                st['gy_ref'] = 'head'
                gy_ptr_holder = st
                conv = op.parts[0][0]
                cmds.append(('head_colsum', op, pk.bias_g.get((op.name, 0)), ('yh_colsum', 0, 2.0 * M * op.y.C)))
                gy_ld, gyN = op.y.C, op.N
                gy_sl = None
            else:
                c0 = 0
                # a layer without a data gradient (the stem: its input is the image) hands gz to nobody but its own weight
                # gradient: that kernel forms gz from (ga, z) in its operand loader (yh_wgrad_desc.bn_*), the apply pass — the
                # last 0.2 ms of the backward's critical path on YOLOv5s — and the gz round trip through HBM disappear
                fused_stem = self._is_fused_stem(op)
                merged = (MERGE_PARTS and 2 <= len(op.parts) <= YH_BN_MAX_PARTS and
                          not (op.res is not None and op.res.buf.needs_grad))
                bwd_parts = (BnPart * len(op.parts))() if merged else None
                scratch_off = 0
                for pi, ((conv, bn), n) in enumerate(zip(op.parts, op.part_N)):
                    require(op.outs[pi], op.name)
                    ga = op.outs[pi].sl(True)
                    ws = st['ws'][pi]
                    coef = torch.zeros(2 * n, dtype=torch.float32, device=self.dev)
                    self.coef_scratch[(op.name, pi)] = coef
                    nblk = L.yh_ew_blocks(M)
                    ypart = op.y.t.data_ptr() + 2 * c0
                    part_ptr = self.part_scratch.data_ptr()
                    fused = self.bnr_fused.get((op.name, pi))
                    if fused is not None:          # the consumer's data gradient already left the partial sums in its own slab
                        part_ptr, nblk = fused[0].data_ptr(), fused[1]
                    else:
                        if bwd_parts is not None:  # the merged finalize reads every part's rows: they may not share the scratch slab
                            part_ptr += 4 * scratch_off
                            scratch_off += nblk * 2 * n
                            assert scratch_off <= self.part_scratch.numel()
                        cmds.append((L.yh_bn_silu_bwd_reduce, (ga.ptr(), ga.ld, ypart, op.y.C, ws.data_ptr(), n, M,
                                                               part_ptr), op.name, ('yh_bn_silu_bwd_reduce', 0, 4.0 * M * n)))
                    goff, boff = pk.bn_g[(op.name, pi)]
                    if bwd_parts is not None:
                        pa = bwd_parts[pi]
                        pa.slab, pa.nblk = part_ptr, nblk
                        pa.dgamma, pa.dbeta = pk.gpack.data_ptr() + 4 * goff, pk.gpack.data_ptr() + 4 * boff
                    else:
                        cmds.append((L.yh_bn_bwd_finalize, (part_ptr, nblk, n, M, ws.data_ptr(),
                                                            pk.gpack.data_ptr() + 4 * goff, pk.gpack.data_ptr() + 4 * boff,
                                                            coef.data_ptr()), op.name, ('yh_bn_bwd_finalize', 0, 8.0 * nblk * n)))
                    gres_ptr, gres_ld, gres_acc = None, 0, 0
                    if op.res is not None and pi == 0 and op.res.buf.needs_grad:
                        gres_acc = claim(op.res)
                        gr = op.res.sl(True)
                        gres_ptr, gres_ld = gr.ptr(), gr.ld
                    if bwd_parts is not None:
                        pa = bwd_parts[pi]
                        pa.ws, pa.C, pa.ga, pa.ldga = ws.data_ptr(), n, ga.ptr(), ga.ld
                        pa.gamma, pa.coef = bn.weight.data_ptr(), coef.data_ptr()
                    elif fused_stem:
                        st['fused_bwd'] = (ga, ws, bn, coef)
                    else:
                        cmds.append((L.yh_bn_silu_bwd_apply, (ga.ptr(), ga.ld, ypart, op.y.C, ws.data_ptr(), bn.weight.data_ptr(),
                                                              coef.data_ptr(), n, M, gys.data_ptr() + 2 * c0, op.N,
                                                              gres_ptr, gres_ld, gres_acc), op.name,
                                     ('yh_bn_silu_bwd_apply', 0, (6.0 + (4.0 if gres_acc else 2.0) * (gres_ptr is not None)) * M * n)))
                    c0 += n
                if bwd_parts is not None:          # the parts' reductions are done: one finalize, one pass writes gz of the whole stacked layer
                    self._keep.append(bwd_parts)
                    cmds.append((L.yh_bn_bwd_finalize_parts, (bwd_parts, len(op.parts), M), op.name,
                                 ('yh_bn_bwd_finalize', 0, 8.0 * sum(int(q.nblk) * int(q.C) for q in bwd_parts))))
                    cmds.append((L.yh_bn_silu_bwd_apply_parts, (op.y.t.data_ptr(), op.y.C, M, bwd_parts, len(op.parts), gys.data_ptr(), op.N),
                                 op.name, ('yh_bn_silu_bwd_apply_parts', 0, 6.0 * M * op.N)))
                gy_ld, gyN = op.N, op.N
            # wgrad per segment (side stream: starts when gz is ready).  The fused stem's weight gradient is the LAST link of the backward's
            # critical chain (it waits for the finalize behind the last data gradient): it stays on the main stream, beside the side
            # stream's last weight gradient instead of behind it
            on_main = op.kind == 'cba' and fused_stem
            if not on_main:
                cmds.append(('wg_begin', None, None, ('sync', 0, 0.0)))
            def wgrad_desc_for(sg, coff_k):
                wd = WgradDesc()
                wd.gy = gys.data_ptr() if op.kind == 'cba' else 0
                wd.ldg, wd.N = gy_ld, gyN
                if op.kind == 'cba' and fused_stem:
                    ga_, ws_, bn_, coef_ = st['fused_bwd']
                    wd.gy, wd.ldg = ga_.ptr(), ga_.ld
                    wd.bn_z, wd.bn_ldz = op.y.t.data_ptr(), op.y.C
                    wd.bn_ws, wd.bn_gamma, wd.bn_coef = ws_.data_ptr(), bn_.weight.data_ptr(), coef_.data_ptr()
                wd.seg = hipk.make_seg(sg.sl())
                wd.coff_k, wd.Ctot = coff_k, op.Ctot
                wd.B, wd.Ho, wd.Wo, wd.Hi, wd.Wi = B, op.Ho, op.Wo, op.Hi, op.Wi
                wd.KH = wd.KW = op.k
                wd.stride, wd.pad = op.stride, op.pad
                wd.dw = gdw
                if self.wg_ws is not None:
                    wd.partial, wd.partial_bytes = self.wg_ws.data_ptr(), self.wg_ws.numel() * 4
                return wd
            coff_k = 0
            for si, sg in enumerate(op.segs):
                wd = wgrad_desc_for(sg, coff_k)
                ntile = L.yh_conv_wgrad_tiles(gyN, op.k * op.k * sg.C)
                kcols = op.k * op.k * (12 if op.focus else sg.C)
                nbytes_x = 2.0 * B * (op.Hi >> sg.ups) * (op.Wi >> sg.ups) * sg.C
                wd.splits = self._tune_wgrad_splits(wd, M, ntile, op)
                self._keep.append(wd)
                if on_main:
                    self._wgrad_on_main.add(id(wd))
                cmds.append(('wgrad', op, wd, (self._wgrad_name(L, wd), 2.0 * M * op.N * kcols,
                                               2.0 * M * gy_ld * (2 if wd.bn_z else 1) + nbytes_x)))
                coff_k += sg.C
            if not on_main:
                cmds.append(('wg_end', (n_cba - 1) % NGZ if op.kind == 'cba' else None, None, ('sync', 0, 0.0)))
            # every gradient of this op's parameters has been enqueued: its slice of the packed arena is final
            marks.append((len(cmds), pk.gloc[op.name]))
            # dgrad per segment
            for si, sg in enumerate(op.segs):
                if not sg.buf.needs_grad:
                    continue
                wp, cpad, Kd = pk.wptr((op.name, 'dgrad', si))
                Nk = _rup(op.N, 8)
                d = ConvDesc()
                d.seg[0].ptr = gys.data_ptr() if op.kind == 'cba' else 0
                d.seg[0].ld, d.seg[0].C, d.seg[0].ups = gy_ld, Nk, 0
                d.nseg, d.mode = 1, YH_CONV_DGRAD
                d.B, d.Ho, d.Wo, d.Hi, d.Wi = B, op.Hi, op.Wi, op.Ho, op.Wo
                d.KH = d.KW = op.k
                d.stride, d.pad = op.stride, op.pad
                d.w, d.N, d.Npad = wp, sg.C, cpad
                d.act = YH_ACT_NONE
                d.nsplit = sg.C
                if sg.ups:
                    tmp = torch.zeros(B, op.Hi, op.Wi, sg.C, dtype=torch.bfloat16, device=self.dev)
                    self.ups_scratch[(op.name, si)] = tmp
                    d.out0, d.ld0, d.accumulate = tmp.data_ptr(), sg.C, 0
                    acc = claim(Ref(sg.buf, sg.coff, sg.C))
                    gl = Slice(sg.buf.g, sg.coff, sg.C)
                    self._keep.append(d)
                    self._tune_conv(d, 'dgrad', op.name)
                    cmds.append(('dgrad', op, d, (self._kernel_name(d), 2.0 * M * op.N * op.k * op.k * sg.C, self._conv_bytes(d))))
                    cmds.append((L.yh_upsample2_bwd, (tmp.data_ptr(), sg.C, B, sg.buf.H, sg.buf.W, sg.C, gl.ptr(), gl.ld, acc), op.name,
                                 ('yh_upsample2_bwd', 0, (2.0 + (1.0 if acc else 0.5)) * B * op.Hi * op.Wi * sg.C)))
                else:
                    acc = claim(Ref(sg.buf, sg.coff, sg.C))
                    gl = Slice(sg.buf.g, sg.coff, sg.C)
                    d.out0, d.ld0, d.accumulate = gl.ptr(), gl.ld, acc
                    self._keep.append(d)
                    key = (sg.buf.name, sg.coff, sg.C)
                    if fuse_ok and key in producer_of and last_writer(key) and (acc == 0 or fuse_acc):
                        rows = L.yh_conv_bnr_rows(C.byref(d))
                        if rows > 0:
                            po, ppi, pc0 = producer_of[key]
                            d.bnr_z, d.bnr_ldz = po.y.t.data_ptr() + 2 * pc0, po.y.C
                            d.bnr_ws, d.bnr_C = self.op_state[po.name]['ws'][ppi].data_ptr(), sg.C
                            slab = torch.zeros(rows * 2 * sg.C, dtype=torch.float32, device=self.dev)
                            d.bnr_part = slab.data_ptr()
                            self.bnr_fused[(po.name, ppi)] = (slab, rows)
                    self._tune_conv(d, 'dgrad', op.name)
                    if d.bnr_part and L.yh_conv_bnr_rows(C.byref(d)) != self.bnr_fused[(po.name, ppi)][1]:
                        # the tuned block cap changed the grid: size the slab for it
                        rows = L.yh_conv_bnr_rows(C.byref(d))
                        slab = torch.zeros(rows * 2 * sg.C, dtype=torch.float32, device=self.dev)
                        d.bnr_part = slab.data_ptr()
                        self.bnr_fused[(po.name, ppi)] = (slab, rows)
                    cmds.append(('dgrad', op, d, (self._kernel_name(d), 2.0 * M * op.N * op.k * op.k * sg.C, self._conv_bytes(d))))
        self.cmd_bwd = cmds
        self.cmd_bwd_frozen = None
        self.bwd_buckets = plan_grad_buckets(marks, pk.gsize, int(os.environ.get("YH_DP_BUCKETS", "4")))
        self.bwd_ready = True
        _tune_cache_save()

    def _tune_wgrad_splits(self, wd, M, ntile, op):
        """Split-M factor (and, for the wide 64-row tilings, the pixels per k-step) of one weight-gradient launch.  The best
        total block count depends on the tile configuration's residency and on how the atomics of the epilogue amortise
        (measured 256..1024 blocks, up to 1.6x apart), so it is timed once per layer when the backward program is built
        (YH_WGRAD_TUNE=0: fixed 512-block rule).  Sets wd.tile_k, returns the split factor."""
        Kseg = wd.KH * wd.KW * wd.seg.C

        def splits_for(total, tk=0):
            nt = self.L.yh_conv_wgrad_tiles2(wd.N, Kseg, tk) if tk == 128 else ntile
            return max(1, min((M + 255) // 256, (total + nt - 1) // nt))
        if os.environ.get("YH_WGRAD_TUNE", "1") == "0":
            return splits_for(512)
        key = f"{KEY_WGRAD_WS if wd.partial else KEY_WGRAD}{'f' if wd.bn_z else ''}:" + ",".join(str(int(v)) for v in (wd.N, wd.ldg, wd.seg.C, wd.seg.ld, wd.seg.ups, wd.Ctot, wd.B, wd.Ho, wd.Wo,
                                                          wd.Hi, wd.Wi, wd.KH, wd.stride, wd.pad))
        cache = _tune_cache()
        if key in cache:
            sp, wd.tile_k = (int(v) for v in cache[key])
            return sp
        gy_saved = wd.gy
        if not wd.gy:                      # head gradient arrives at run time: time against the scratch buffer
            if self.gy_scratch.numel() < M * wd.ldg:
                return splits_for(512)
            wd.gy = self.gy_scratch.data_ptr()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        name = self.L.yh_conv_wgrad_kernel_name(wd.N, wd.KH * wd.KW * wd.seg.C).decode()
        tks = (0, 64) if name in _WGRAD_TK64 else (0,)
        if name.startswith("conv_wgrad_kernel<4, 2, 1, 2, 64"):
            tks = tks + (32, 35)            # the general tiling with 32-pixel k-steps (two blocks per CU): 8 waves of 32 x 64 / 4 of 64 x 64
        if 128 <= Kseg <= 384 and wd.N > 32 and not wd.bn_z:
            tks = tks + (128,)              # the general 128-column tiling on a layer that defaults to a wide one
        best, best_ms = None, None
        wd.tile_k = 40
        if not wd.partial and self.L.yh_conv_wgrad_patch_ok(C.byref(wd)):
            tks = tks + (40,)               # patch form (conv_wgp_kernel): the input patch of a pixel region staged once in LDS
        wtiles = 0 if (wd.partial or wd.bn_z or os.environ.get("YH_WGRAD_WAVE", "1") == "0") else self.L.yh_conv_wgrad_wave_tiles(C.byref(wd))
        if wtiles > 0:
            tks = tks + (129,)              # wave-private 128 x 128 tiles + stream-K (conv_wgs_kernel): `splits` = workgroups, one per CU
        for tk in tks:
            wd.tile_k = tk
            if tk == 129:                   # an exact tiles x splits grid where it fills the chip, else 256 workgroups dealt (tile, 32 pixels) units
                # Workgroups (= CUs: the form holds a whole CU) a weight gradient may take.  Alone on the chip 256 is fastest; in the
                # two-stream backward the main chain runs beside it, and its short latency-bound kernels (finalize launches, small
                # layers) wait for a CU while a weight gradient holds all of them: the YOLOv5s step is shortest when the weight
                # gradients leave a quarter of the CUs alone (12.00 -> 11.90 ms), the YOLOv5l step — long kernels on both streams —
                # when its big layers take the whole chip (43.36 -> 42.82 ms): layers under 60 GFLOP get 192, the others 256
                # (profiles/r04_step_experiments.txt d).  Half of the budget is timed too: on the small layers the atomics (one
                # partial tile per workgroup) dominate.
                wflops = 2.0 * M * wd.N * wd.KH * wd.KW * wd.seg.C
                gmax = int(os.environ.get("YH_WGS_G", "256" if wflops >= 60e9 else "192"))
                sps = set()
                for g in (gmax, gmax // 2):
                    sps |= {g} | ({wtiles * (g // wtiles)} if wtiles <= g else set())
                sps = sorted(sps)
            else:
                sps = [1024] if tk == 40 else sorted({splits_for(t, tk) for t in (256, 512, 768, 1024, 1536)})
            for sp in sps:
                wd.splits = sp
                if wd.partial and self.L.yh_conv_wgrad_ws_bytes(C.byref(wd)) > wd.partial_bytes:
                    continue                   # more partial tiles than the workspace holds
                check(self.L.yh_conv_wgrad(C.byref(wd), st), f"yh_conv_wgrad tune [{op.name}]")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(TUNE_ITERS):
                    self.L.yh_conv_wgrad(C.byref(wd), st)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                if best_ms is None or ms < best_ms:
                    best, best_ms = (sp, tk), ms
        wd.gy = gy_saved
        wd.tile_k = best[1]
        self.wgrad_tuned[(op.name, wd.coff_k)] = (best[0], best_ms / TUNE_ITERS)
        cache[key] = [int(best[0]), int(best[1])]
        return best[0]

    @staticmethod
    def _wgrad_name(L, wd):
        """instantiation yh_conv_wgrad launches for this descriptor, profiler spelling (64-pixel k-steps on the wide tilings:
        csrc/conv_wgrad.hip, yh_conv_wgrad)"""
        name = L.yh_conv_wgrad_kernel_name2(wd.N, wd.KH * wd.KW * wd.seg.C, wd.tile_k).decode()
        if wd.tile_k == 64:
            name = _WGRAD_TK64.get(name, name)
        if wd.bn_z:                        # last template argument: BatchNorm backward fused into the operand loader
            name = name[:-len(", false>")] + ", true>"
        if wd.tile_k == 129 and L.yh_conv_wgrad_wave_tiles(C.byref(wd)) > 0:
            return L.yh_conv_wgrad_wave_name(C.byref(wd)).decode()
        if wd.tile_k == 40 and L.yh_conv_wgrad_patch_ok(C.byref(wd)):
            buf = C.create_string_buffer(64)
            L.yh_conv_wgrad_patch_name(C.byref(wd), buf, 64)
            name = buf.value.decode() or name
        return name

    def _bucket_ready(self, bucket_hook, bucket, main, side):
        """hand a finished gradient bucket to the data-parallel hook.  With the side stream, the bucket's weight gradients
        were enqueued THERE and its BatchNorm / bias gradients on the main stream: the hook is called in the side stream's
        context after it has been made to wait for the main stream's position, so the collective is ordered behind both
        without stalling the dgrad / BatchNorm chain."""
        part = self.pack.gpack[bucket[1]:bucket[2]]
        if side is None:
            return bucket_hook(part)
        self._ev_gz.record(main)
        side.wait_event(self._ev_gz)
        with torch.cuda.stream(side):
            return bucket_hook(part)

    def _frozen_bwd_cmds(self):
        """backward of the evaluation-mode-BatchNorm forward: the same launches, with the batch-mean coefficients every
        yh_bn_bwd_finalize leaves for the apply pass zeroed (gz = gamma * invstd * dz).  Command indices are preserved
        for the gradient buckets: the fill rides in the finalize's slot as a pair."""
        L, out = self.L, []
        for cmd in self.cmd_bwd:
            fn = cmd[0]
            if fn is L.yh_bn_bwd_finalize:
                coef_ptr = cmd[1][7]
                out.append(('pair', [cmd, (L.yh_fill_u32, (coef_ptr, 0, 2 * cmd[1][2]), cmd[2], ('yh_fill_u32', 0, 0.0))], cmd[2], ('sync', 0, 0.0)))
            elif fn is L.yh_bn_bwd_finalize_parts:
                parts, nparts = cmd[1][0], cmd[1][1]
                fills = [(L.yh_fill_u32, (parts[i].coef, 0, 2 * int(parts[i].C)), cmd[2], ('yh_fill_u32', 0, 0.0)) for i in range(nparts)]
                out.append(('pair', [cmd] + fills, cmd[2], ('sync', 0, 0.0)))
            else:
                out.append(cmd)
        return out

    def _compile_backward(self, two, buckets, cmd_bwd=None):
        """the backward command list as a yh_cmd array: kernels on stream 0 (main) / 1 (side: weight gradients), the event
        records and stream waits of the gz ring in between; returns (array, positions at which a gradient bucket is complete,
        per-call patches for the head gradients)"""
        L = self.L
        cmd_bwd = self.cmd_bwd if cmd_bwd is None else cmd_bwd
        cc = CompiledCmds(L, 2 * len(cmd_bwd) + 8 + 4 * sum(1 for c in cmd_bwd if c[0] == 'pair'))
        cc.source = cmd_bwd
        breaks, patches = [], []
        pending = [False] * NGZ
        if two:
            for ev in [self._ev_gz] + self._ev_wg:        # materialise the raw event handles
                ev.record(self._side)
            ev_gz, ev_wg = self._ev_gz.cuda_event, [e.cuda_event for e in self._ev_wg]
        nb = 0
        for ci, cmd in enumerate(cmd_bwd):
            while nb < len(buckets) and buckets[nb][0] == ci:
                breaks.append(cc.n)
                nb += 1
            fn = cmd[0]
            if fn == 'pair':
                for sub in cmd[1]:
                    cc.call(sub[0], sub[1], 0, sub[2])
                continue
            if fn == 'gz_begin':
                if two and pending[cmd[1]]:
                    cc.event(YH_CMD_STREAM_WAIT, ev_wg[cmd[1]], 0)
                    pending[cmd[1]] = False
            elif fn == 'wg_begin':
                if two:
                    cc.event(YH_CMD_EVENT_RECORD, ev_gz, 0)
                    cc.event(YH_CMD_STREAM_WAIT, ev_gz, 1)
            elif fn == 'wg_end':
                if two and cmd[1] is not None:
                    cc.event(YH_CMD_EVENT_RECORD, ev_wg[cmd[1]], 1)
                    pending[cmd[1]] = True
            elif fn == 'head_colsum':
                _, op, boff, _m = cmd
                if boff is not None:
                    i = cc.call(L.yh_colsum, (0, op.y.C, op.y.C, self.B * op.Ho * op.Wo, self._head_scratch(op, two),
                                              self.pack.gpack.data_ptr() + 4 * boff), 1 if self._head_on_side(op, two) else 0, op.name)
                    patches.append(('colsum', None, op.name, i))
            elif fn == 'wgrad':
                _, op, wd, _m = cmd
                if "wgrad" in ABL_SKIP:
                    continue
                cc.call(L.yh_conv_wgrad, (wd,), 1 if two and id(wd) not in self._wgrad_on_main else 0, op.name)
                if op.kind == 'plain':
                    patches.append(('wgrad', wd, op.name, -1))
            elif fn == 'dgrad':
                _, op, d, _m = cmd
                cc.call(L.yh_conv_igemm, (d,), 0, op.name)
                if op.kind == 'plain':
                    patches.append(('dgrad', d, op.name, -1))
            else:
                _, args, name, _m = cmd
                if fn.__name__ in ABL_SKIP:
                    continue
                cc.call(fn, args, 0, name)
        while nb < len(buckets):
            breaks.append(cc.n)
            nb += 1
        return cc, breaks, patches

    def backward(self, head_grads, bucket_hook=None, frozen=False):
        """head_grads: list of [B,h,w,ld] bf16 gradient buffers matching self.outputs (plain ops).
        bucket_hook(slice of the packed fp32 gradient arena) -> finisher or None: called as soon as a bucket of
        gradients is complete (data-parallel all-reduce overlapped with the remaining backward); finishers run
        before the gradients are scattered to parameter order."""
        if not self.bwd_ready:
            self._build_backward()
        cmd_bwd = self.cmd_bwd
        if frozen:
            if self.cmd_bwd_frozen is None:
                self.cmd_bwd_frozen = self._frozen_bwd_cmds()
            cmd_bwd = self.cmd_bwd_frozen
        buckets = self.bwd_buckets if bucket_hook is not None else []
        nb, finishers = 0, []
        pk, L = self.pack, self.L
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        hipk.fill_zero(pk.gpack)
        heads = {}
        for o, g in zip(self.outputs, head_grads):
            if isinstance(o, ConvOp):
                heads[o.name] = g
            else:
                o.buf.g[..., o.coff:o.coff + o.C].copy_(g)
        prof = self.profile
        two = self.two_streams
        if two:
            if getattr(self, "_side", None) is None:
                # YH_SIDE_PRIO: priority of the weight-gradient stream (ROCm: -1 high, 0 normal, 1 low)
                self._side = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("YH_SIDE_PRIO", "0")))
                self._ev_gz = torch.cuda.Event()
                self._ev_wg = [torch.cuda.Event() for _ in range(NGZ)]
                self._ev_all = torch.cuda.Event()
            main = torch.cuda.current_stream()
            side = self._side
            st_side = C.c_void_p(side.cuda_stream)
            self._ev_gz.record(main)               # packed arena zeroed, head gradients in place
            side.wait_event(self._ev_gz)
            pending = [False] * NGZ
        if prof is None and USE_EXEC:
            # replay the compiled command array (yh_exec): one call per bucket segment instead of one ctypes call per launch
            key = ('bwd', two, bucket_hook is not None, frozen)
            comp = self._compiled.get(key)
            if comp is None or comp[0].source is not cmd_bwd:
                comp = self._compiled[key] = self._compile_backward(two, buckets, cmd_bwd)
            cc, breaks, patches = comp
            for kind, obj, opname, slot_idx in patches:          # head gradients arrive per call
                ptr = heads[opname].data_ptr()
                if kind == 'wgrad':
                    obj.gy = ptr
                elif kind == 'dgrad':
                    obj.seg[0].ptr = ptr
                else:
                    cc.arr[slot_idx].slots[0] = ptr
            streams = [st.value, st_side.value] if two else [st.value]
            lo = 0
            for pos in breaks:
                cc.run(streams, lo, pos)
                lo = pos
                finishers.append(self._bucket_ready(bucket_hook, buckets[nb], main if two else None, side if two else None))
                nb += 1
            cc.run(streams, lo, cc.n)
            if two:
                self._ev_all.record(side)
                main.wait_event(self._ev_all)
            for f in finishers:
                if f is not None:
                    f()
            return pk.grads_to_params()
        for ci, cmd in enumerate(cmd_bwd):
            while nb < len(buckets) and buckets[nb][0] == ci:
                finishers.append(self._bucket_ready(bucket_hook, buckets[nb], main if two else None, side if two else None))
                nb += 1
            fn = cmd[0]
            if fn == 'pair':
                for sub in cmd[1]:
                    rc = sub[0](*sub[1], st)
                    if rc:
                        check(rc, f"{sub[0].__name__} bwd [{sub[2]}]")
                continue
            if fn == 'gz_begin':
                if two and pending[cmd[1]]:
                    main.wait_event(self._ev_wg[cmd[1]])
                    pending[cmd[1]] = False
                continue
            if fn == 'wg_begin':
                if two:
                    self._ev_gz.record(main)
                    side.wait_event(self._ev_gz)
                continue
            if fn == 'wg_end':
                if two:
                    if cmd[1] is not None:
                        self._ev_wg[cmd[1]].record(side)
                        pending[cmd[1]] = True
                continue
            on_side = two and ((fn == 'wgrad' and id(cmd[2]) not in self._wgrad_on_main) or (fn == 'head_colsum' and self._head_on_side(cmd[1], two)))
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side if on_side else None)
            if fn == 'head_colsum':
                _, op, boff, _m = cmd
                g = heads[op.name]
                if boff is not None:
                    rc = L.yh_colsum(g.data_ptr(), op.y.C, op.y.C, self.B * op.Ho * op.Wo, self._head_scratch(op, two),
                                     pk.gpack.data_ptr() + 4 * boff, st_side if on_side else st)
                    if rc:
                        check(rc, "yh_colsum")
            elif fn == 'wgrad':
                _, op, wd, _m = cmd
                if op.kind == 'plain':
                    wd.gy = heads[op.name].data_ptr()
                rc = L.yh_conv_wgrad(C.byref(wd), st_side if on_side else st)
                if rc:
                    check(rc, f"yh_conv_wgrad [{op.name}]")
            elif fn == 'dgrad':
                _, op, d, _m = cmd
                if op.kind == 'plain':
                    d.seg[0].ptr = heads[op.name].data_ptr()
                rc = L.yh_conv_igemm(C.byref(d), st)
                if rc:
                    check(rc, f"yh_conv_igemm dgrad [{op.name}]")
            else:
                _, args, name, _m = cmd
                rc = fn(*args, st)
                if rc:
                    check(rc, f"{fn.__name__} bwd [{name}]")
            if prof is not None:
                e1.record(side if on_side else None)
                prof.setdefault(cmd[3] + (cmd[1].name if hasattr(cmd[1], 'name') else cmd[2],), []).append((e0, e1))
        while nb < len(buckets):
            finishers.append(self._bucket_ready(bucket_hook, buckets[nb], main if two else None, side if two else None))
            nb += 1
        if two:
            self._ev_all.record(side)
            main.wait_event(self._ev_all)
        for f in finishers:
            if f is not None:
                f()
        return pk.grads_to_params()


# ------------------------------------------------------------------------------------------
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
        from .layout import is_cell_major
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
        flat_g, pgrads = prog.backward(head_grads, bucket_hook, frozen=ctx.frozen)
        ctx.host._yh_last_flat_grad = flat_g
        # whole-gradient hook of the data-parallel exchange: all-reduces flat_g when no bucket hook ran, keeps the
        # books of un-exchanged accumulation steps, and at an accumulation boundary swaps in the averaged total
        hook = getattr(ctx.host, "_yh_grad_hook", None)
        if hook is not None:
            hook(flat_g, bucketed=bucket_hook is not None)
        hook = getattr(ctx.host, "_yh_grad_hook_opt", None)    # flat-arena optimizer
        if hook is not None:
            hook(flat_g)
        if getattr(ctx.host, "flat_grads_only", False):
            # the flat-arena optimizer (utils/optim.py FlatSGD) consumes flat_g directly; skip 177 AccumulateGrad nodes
            pgrads = [None] * len(pgrads)
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
        from .layout import cell_major_view
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
