"""Planner: the op graph a model describes to a Builder (buffers, ConvOp / PoolOp over virtual concats), the flat parameter arena
with its pack / unpack index maps (ParamPack), and the gradient-bucket plan of the data-parallel exchange."""
import numpy as np
import torch

from .. import hipk
from .._lib import YoloHipError
from ..hipk import Slice
from . import flags as _flags
from .flags import BN_EPS_DEFAULT


def _rup(x, m):
    return ((x + m - 1) // m) * m


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
    if not _flags.SPPF_FUSE or i + 2 >= len(ops) or not all(isinstance(o, PoolOp) for o in ops[i:i + 3]):
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

    def grads_to_params(self, views=True):
        """packed fp32 gradients -> one flat gradient in parameter order (fresh tensor per call) and, unless views=False, its
        per-parameter views."""
        flat_g = torch.empty(self.n, dtype=torch.float32, device=self.device)
        hipk.gather_f32(self.gpack, self.unpack_idx, flat_g)
        if not views:
            return flat_g, None
        outs, o = [], 0
        for p in self.params:
            outs.append(flat_g[o:o + p.numel()].view(p.shape))
            o += p.numel()
        return flat_g, outs
