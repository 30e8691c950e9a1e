"""Thin tensor-level wrappers over the C ABI (include/yolohip.h).

torch is used here only for device memory and the current stream; all arithmetic is
done by the HIP kernels.  Activations are NHWC bf16 tensors (rows = B*H*W pixels); a
"slice" is (tensor, channel offset, channels) and maps to (ptr + coff, ld).
"""
import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import ConvDesc, Seg, WgradDesc, YH_ACT_NONE, YH_ACT_SILU, YH_CONV_DGRAD, YH_CONV_FWD, check, lib


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, elem_off=0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + elem_off * t.element_size())


@dataclass
class Slice:
    """A channel slice [coff, coff+C) of an NHWC bf16 buffer of shape (B,H,W,Ctot)."""
    buf: torch.Tensor
    coff: int
    C: int
    ups: int = 0

    @property
    def ld(self):
        return self.buf.shape[-1]

    def ptr(self):
        return self.buf.data_ptr() + 2 * self.coff


def full(buf):
    return Slice(buf, 0, buf.shape[-1])


def make_seg(s: Slice) -> Seg:
    return Seg(C.c_void_p(s.ptr()), s.ld, s.C, s.ups, 0)


def pack_weight_fwd(w_oihw: torch.Tensor, npad=None) -> torch.Tensor:
    """OIHW fp32 -> [Npad][KH*KW*C] bf16 (k = tap*C + c). Test/helper path (torch ops)."""
    n, c, kh, kw = w_oihw.shape
    npad = npad or ((n + 127) // 128) * 128
    out = torch.zeros(npad, kh * kw * c, dtype=torch.bfloat16, device=w_oihw.device)
    out[:n] = w_oihw.permute(0, 2, 3, 1).reshape(n, -1).to(torch.bfloat16)
    return out


def pack_weight_dgrad(w_oihw: torch.Tensor, cpad=None) -> torch.Tensor:
    """OIHW fp32 -> [Cpad][KH*KW*N] bf16 (k = tap*N + n): the dgrad GEMM's B operand."""
    n, c, kh, kw = w_oihw.shape
    cpad = cpad or ((c + 127) // 128) * 128
    out = torch.zeros(cpad, kh * kw * n, dtype=torch.bfloat16, device=w_oihw.device)
    out[:c] = w_oihw.permute(1, 2, 3, 0).reshape(c, -1).to(torch.bfloat16)
    return out


def conv_desc(segs, mode, B, Ho, Wo, Hi, Wi, k, stride, pad, w, N, out0, nsplit=None, out1=None,
              bias=None, scale=None, shift=None, act=YH_ACT_NONE, accumulate=0, res=None, stats=None):
    d = ConvDesc()
    for i, s in enumerate(segs):
        d.seg[i] = make_seg(s)
    d.nseg = len(segs)
    d.mode = mode
    d.B, d.Ho, d.Wo, d.Hi, d.Wi = B, Ho, Wo, Hi, Wi
    d.KH = d.KW = k
    d.stride, d.pad = stride, pad
    d.w = w.data_ptr()
    d.N, d.Npad = N, w.shape[0]
    d.bias = bias.data_ptr() if bias is not None else None
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.act, d.accumulate = act, accumulate
    d.out0, d.ld0 = out0.ptr(), out0.ld
    d.nsplit = nsplit if nsplit is not None else N
    if out1 is not None:
        d.out1, d.ld1 = out1.ptr(), out1.ld
    if res is not None:
        d.res, d.ldr = res.ptr(), res.ld
    d.stats = stats.data_ptr() if stats is not None else None
    return d


def conv_stat_blocks(d: ConvDesc) -> int:
    return lib().yh_conv_stat_blocks(C.byref(d))


def conv_launch(d: ConvDesc):
    check(lib().yh_conv_igemm(C.byref(d), _st()), "yh_conv_igemm")


def wgrad_desc(gy: Slice, N, seg: Slice, coff_k, Ctot, B, Ho, Wo, Hi, Wi, k, stride, pad, dw, splits):
    d = WgradDesc()
    d.gy, d.ldg, d.N = gy.ptr(), gy.ld, N
    d.seg = make_seg(seg)
    d.coff_k, d.Ctot = coff_k, Ctot
    d.B, d.Ho, d.Wo, d.Hi, d.Wi = B, Ho, Wo, Hi, Wi
    d.KH = d.KW = k
    d.stride, d.pad = stride, pad
    d.dw = dw.data_ptr()
    d.splits = splits
    return d


def wgrad_launch(d: WgradDesc):
    check(lib().yh_conv_wgrad(C.byref(d), _st()), "yh_conv_wgrad")


def bn_finalize(stats, nblk, ldstat, Cn, count, gamma, beta, rm, rv, nbt, eps, momentum, ws):
    check(lib().yh_bn_finalize(_p(stats), nblk, ldstat, Cn, count, _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt),
                               eps, momentum, _p(ws), _st()), "yh_bn_finalize")


def bn_fold(gamma, beta, rm, rv, eps, Cn, scale, shift):
    check(lib().yh_bn_fold(_p(gamma), _p(beta), _p(rm), _p(rv), eps, Cn, _p(scale), _p(shift), _st()), "yh_bn_fold")


def bn_silu_apply(y: Slice, ws, M, out: Slice, res: Slice = None):
    check(lib().yh_bn_silu_apply(y.ptr(), y.ld, _p(ws), y.C, M, out.ptr(), out.ld,
                                 res.ptr() if res else None, res.ld if res else 0, _st()), "yh_bn_silu_apply")


def ew_blocks(M):
    return lib().yh_ew_blocks(M)


def bn_silu_bwd_reduce(ga: Slice, y: Slice, ws, M, part):
    check(lib().yh_bn_silu_bwd_reduce(ga.ptr(), ga.ld, y.ptr(), y.ld, _p(ws), y.C, M, _p(part), _st()),
          "yh_bn_silu_bwd_reduce")


def bn_bwd_finalize(part, nblk, Cn, M, ws, dgamma, dbeta, coef):
    check(lib().yh_bn_bwd_finalize(_p(part), nblk, Cn, M, _p(ws), _p(dgamma), _p(dbeta), _p(coef), _st()), "yh_bn_bwd_finalize")


def bn_silu_bwd_apply(ga: Slice, y: Slice, ws, gamma, coef, M, gy: Slice, gres: Slice = None, gres_acc=0):
    check(lib().yh_bn_silu_bwd_apply(ga.ptr(), ga.ld, y.ptr(), y.ld, _p(ws), _p(gamma), _p(coef), y.C, M,
                                     gy.ptr(), gy.ld, gres.ptr() if gres else None, gres.ld if gres else 0,
                                     gres_acc, _st()), "yh_bn_silu_bwd_apply")


def _bn_parts(parts):
    """parts: dicts with ws, C and (forward) out: Slice or (backward) ga: Slice, gamma, coef"""
    from ._lib import BnPart
    arr = (BnPart * len(parts))()
    for a, q in zip(arr, parts):
        a.ws, a.C = q["ws"].data_ptr(), q["C"]
        if "out" in q:
            a.out, a.ldo = q["out"].ptr(), q["out"].ld
        if "ga" in q:
            a.ga, a.ldga = q["ga"].ptr(), q["ga"].ld
            a.gamma, a.coef = q["gamma"].data_ptr(), q["coef"].data_ptr()
        if "slab" in q:          # finalize: forward (gamma, beta, rm, rv, nbt, eps, momentum, ldslab) or backward (dgamma, dbeta, coef)
            a.slab, a.nblk, a.ldslab = q["slab"], q["nblk"], q.get("ldslab", q["C"])
            for k in ("gamma", "beta", "running_mean", "running_var", "num_batches", "dgamma", "dbeta", "coef"):
                if k in q:
                    setattr(a, k, q[k].data_ptr())
            a.eps, a.momentum = q.get("eps", 0.0), q.get("momentum", 0.0)
    return arr


def bn_finalize_parts(parts, count):
    arr = _bn_parts(parts)
    check(lib().yh_bn_finalize_parts(arr, len(parts), count, _st()), "yh_bn_finalize_parts")


def bn_bwd_finalize_parts(parts, M):
    arr = _bn_parts(parts)
    check(lib().yh_bn_bwd_finalize_parts(arr, len(parts), M, _st()), "yh_bn_bwd_finalize_parts")


def bn_silu_apply_parts(y: Slice, M, parts):
    arr = _bn_parts(parts)
    check(lib().yh_bn_silu_apply_parts(y.ptr(), y.ld, M, arr, len(parts), _st()), "yh_bn_silu_apply_parts")


def bn_silu_bwd_apply_parts(y: Slice, M, parts, gy: Slice):
    arr = _bn_parts(parts)
    check(lib().yh_bn_silu_bwd_apply_parts(y.ptr(), y.ld, M, arr, len(parts), gy.ptr(), gy.ld, _st()), "yh_bn_silu_bwd_apply_parts")


def colsum(g: Slice, M, part, out):
    check(lib().yh_colsum(g.ptr(), g.ld, g.C, M, _p(part), _p(out), _st()), "yh_colsum")


def maxpool5_fwd(x: Slice, B, H, W, out: Slice, idx):
    check(lib().yh_maxpool5_fwd(x.ptr(), x.ld, B, H, W, x.C, out.ptr(), out.ld, _p(idx), _st()), "yh_maxpool5_fwd")


def maxpool5_bwd(gout: Slice, idx, B, H, W, gin: Slice, accumulate):
    check(lib().yh_maxpool5_bwd(gout.ptr(), gout.ld, _p(idx), B, H, W, gout.C, gin.ptr(), gin.ld, accumulate, _st()),
          "yh_maxpool5_bwd")


def upsample2_bwd(ghi: Slice, B, Hlo, Wlo, glo: Slice, accumulate):
    check(lib().yh_upsample2_bwd(ghi.ptr(), ghi.ld, B, Hlo, Wlo, ghi.C, glo.ptr(), glo.ld, accumulate, _st()),
          "yh_upsample2_bwd")


def input_s2d(x, out):
    B, Cin, H, W = x.shape
    check(lib().yh_input_s2d(_p(x), B, Cin, H, W, _p(out), _st()), "yh_input_s2d")


def fill_zero(t):
    nbytes = t.numel() * t.element_size()
    assert nbytes % 4 == 0
    check(lib().yh_fill_u32(_p(t), 0, nbytes // 4, _st()), "yh_fill_u32")


def pack_bf16(src, idx, dst):
    check(lib().yh_pack_bf16(_p(src), _p(idx), idx.numel(), _p(dst), _st()), "yh_pack_bf16")


def gather_f32(src, idx, dst):
    check(lib().yh_gather_f32(_p(src), _p(idx), idx.numel(), _p(dst), _st()), "yh_gather_f32")


def sgd_step(p, g, buf, group, lr, wd, momentum, nesterov, first, grad_scale=None):
    check(lib().yh_sgd_step(_p(p), _p(g), _p(buf), _p(group), p.numel(), _p(lr), _p(wd), lr.numel(), momentum,
                            int(nesterov), int(first), _p(grad_scale), _st()), "yh_sgd_step")


def sumsq(x, part, out):
    check(lib().yh_sumsq(_p(x), x.numel(), _p(part), _p(out), _st()), "yh_sumsq")


def ema_update(ema, p, decay):
    check(lib().yh_ema_update(_p(ema), _p(p), p.numel(), decay, _st()), "yh_ema_update")


def clip_scale(sumsq_t, max_norm, out):
    check(lib().yh_clip_scale(_p(sumsq_t), float(max_norm), _p(out), _st()), "yh_clip_scale")


def sgd_step_dev(p, g, buf, group, scal, nesterov, grad_scale=None):
    """SGD step with lr | wd | momentum | first-step flag read from the device tensor `scal` (8 floats): graph-replay safe"""
    check(lib().yh_sgd_step_dev(_p(p), _p(g), _p(buf), _p(group), p.numel(), _p(scal), int(nesterov), _p(grad_scale), _st()), "yh_sgd_step_dev")


def ema_update_dev(ema, p, decay_dev):
    check(lib().yh_ema_update_dev(_p(ema), _p(p), p.numel(), _p(decay_dev), _st()), "yh_ema_update_dev")


def ema_advance(counter, decay, ratio, tau):
    check(lib().yh_ema_advance(_p(counter), _p(decay), float(ratio), float(tau), _st()), "yh_ema_advance")
