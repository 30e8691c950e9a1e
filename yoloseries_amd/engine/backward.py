"""Program, backward side: the backward program (BatchNorm+SiLU backward, fused reductions, data / weight gradients, the
gz ring and the weight-gradient stream), its compiled form, gradient buckets for the data-parallel exchange."""
import ctypes as C
import os

import numpy as np
import torch

from .. import hipk
from .._lib import (BnPart, ConvDesc, YH_BN_MAX_PARTS, WgradDesc, YH_CMD_EVENT_RECORD, YH_CMD_STREAM_WAIT, YH_ACT_NONE, YH_CONV_DGRAD,
                    YoloHipError, check)
from ..hipk import Slice
from ..streams import side_stream
from . import flags as _flags
from .executor import CompiledCmds
from .graph import ConvOp, PoolOp, Ref, _rup, plan_grad_buckets, sppf_chain
from .tune import _tune_cache_save


class BackwardMixin:

    def _head_on_side(self, op, two):
        """does a head layer's bias gradient (column sums of the head gradient) run on the weight-gradient stream?  Only with a
        scratch of its own: `part_scratch` belongs to the main stream's BatchNorm reductions (a layer wider than `head_scratch` was
        sized for — it is sized from the widest plain op of the graph, so none today — falls back to the main stream, never to a
        shared buffer on another stream)"""
        return bool(two and _flags.HEAD_COLSUM_SIDE and op.y.C * 1024 * 2 <= self.head_scratch.numel())

    def _head_scratch(self, op, two):
        """partial-sum scratch of a head layer's bias gradient: its own buffer when the column sums run on the side stream"""
        return self.head_scratch.data_ptr() if self._head_on_side(op, two) else self.part_scratch.data_ptr()

    def _is_fused_stem(self, op):
        """a ConvBnAct without a data gradient (the stem) whose BatchNorm backward apply runs inside its weight gradient (YH_FUSE_STEM_BWD)"""
        if not isinstance(op, ConvOp) or op.kind != 'cba':
            return False
        Kseg0 = op.k * op.k * op.segs[0].C
        return bool(_flags.FUSE_STEM_BWD and len(op.parts) == 1 and len(op.segs) == 1 and op.res is None and
                    not op.segs[0].buf.needs_grad and op.N % 8 == 0 and
                    ((op.N <= 32 and Kseg0 <= 256) or (op.N > 32 and 128 < Kseg0 <= 256)) and
                    (_flags.FUSE_STEM_BWD >= 2 or (self.wg_ws is None and self._stem_patch_ok(op))))

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
        NGZ = self.ngz = _flags.NGZ
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
        self.wg_ws = torch.empty(_flags.WG_WS_BYTES // 4, dtype=torch.float32, device=self.dev) if _flags.WG_WS_BYTES > 0 else None

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
                merged = (_flags.MERGE_PARTS and 2 <= len(op.parts) <= YH_BN_MAX_PARTS and
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
            launches = []
            for si, sg in enumerate(op.segs):
                wd = wgrad_desc_for(sg, coff_k)
                ntile = L.yh_conv_wgrad_tiles(gyN, op.k * op.k * sg.C)
                kcols = op.k * op.k * (12 if op.focus else sg.C)
                nbytes_x = 2.0 * B * (op.Hi >> sg.ups) * (op.Wi >> sg.ups) * sg.C
                wd.splits = self._tune_wgrad_splits(wd, M, ntile, op)
                self._keep.append(wd)
                if on_main:
                    self._wgrad_on_main.add(id(wd))
                launches.append((op, wd, (self._wgrad_name(L, wd), 2.0 * M * op.N * kcols,
                                          2.0 * M * gy_ld * (2 if wd.bn_z else 1) + nbytes_x)))
                coff_k += sg.C
            slot = (n_cba - 1) % NGZ if op.kind == 'cba' else None
            if not on_main:
                cmds.append(('wg_begin', None, None, ('sync', 0, 0.0)))
            for ln in launches:
                cmds.append(('wgrad',) + ln)
            if not on_main:
                cmds.append(('wg_end', [slot] if slot is not None else None, None, ('sync', 0, 0.0)))
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
        pending = [None] * self.ngz       # per gz buffer: index of the event its last weight-gradient launch recorded
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
                if two and pending[cmd[1]] is not None:
                    e = pending[cmd[1]]
                    cc.event(YH_CMD_STREAM_WAIT, ev_wg[e], 0)
                    pending = [None if q == e else q for q in pending]
            elif fn == 'wg_begin':
                if two:
                    cc.event(YH_CMD_EVENT_RECORD, ev_gz, 0)
                    cc.event(YH_CMD_STREAM_WAIT, ev_gz, 1)
            elif fn == 'wg_end':
                if two and cmd[1]:
                    cc.event(YH_CMD_EVENT_RECORD, ev_wg[cmd[1][0]], 1)
                    for slot in cmd[1]:
                        pending[slot] = cmd[1][0]
            elif fn == 'head_colsum':
                _, op, boff, _m = cmd
                if boff is not None:
                    i = cc.call(L.yh_colsum, (0, op.y.C, op.y.C, self.B * op.Ho * op.Wo, self._head_scratch(op, two),
                                              self.pack.gpack.data_ptr() + 4 * boff), 1 if self._head_on_side(op, two) else 0, op.name)
                    patches.append(('colsum', None, op.name, i))
            elif fn == 'wgrad':
                _, op, wd, _m = cmd
                if "wgrad" in _flags.ABL_SKIP:
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
                if fn.__name__ in _flags.ABL_SKIP:
                    continue
                cc.call(fn, args, 0, name)
        while nb < len(buckets):
            breaks.append(cc.n)
            nb += 1
        return cc, breaks, patches

    def backward(self, head_grads, bucket_hook=None, frozen=False, param_views=True):
        """head_grads: list of [B,h,w,ld] bf16 gradient buffers matching self.outputs (plain ops).
        bucket_hook(slice of the packed fp32 gradient arena) -> finisher or None: called as soon as a bucket of
        gradients is complete (data-parallel all-reduce overlapped with the remaining backward); finishers run
        before the gradients are scattered to parameter order.
        param_views=False: only the flat gradient is wanted (the flat-arena optimizer): the 177 per-parameter views — ~0.5 ms of
        host time between the last backward kernel and the optimizer's first — are not built."""
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
                # a stream PROBED to run beside the compute stream (streams.py: a fresh stream may share its hardware queue)
                self._side = side_stream(self.dev)
                self._ev_gz = torch.cuda.Event()
                self._ev_wg = [torch.cuda.Event() for _ in range(self.ngz)]
                self._ev_all = torch.cuda.Event()
            main = torch.cuda.current_stream()
            side = self._side
            st_side = C.c_void_p(side.cuda_stream)
            self._ev_gz.record(main)               # packed arena zeroed, head gradients in place
            side.wait_event(self._ev_gz)
            pending = [None] * self.ngz
        if prof is None and _flags.USE_EXEC:
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
            return pk.grads_to_params(param_views)
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
                if two and pending[cmd[1]] is not None:
                    e = pending[cmd[1]]
                    main.wait_event(self._ev_wg[e])
                    pending = [None if q == e else q for q in pending]
                continue
            if fn == 'wg_begin':
                if two:
                    self._ev_gz.record(main)
                    side.wait_event(self._ev_gz)
                continue
            if fn == 'wg_end':
                if two and cmd[1]:
                    self._ev_wg[cmd[1][0]].record(side)
                    for slot in cmd[1]:
                        pending[slot] = cmd[1][0]
                continue
            on_side = two and ((fn == 'wgrad' and id(cmd[2]) not in self._wgrad_on_main) or 
                               (fn == 'head_colsum' and self._head_on_side(cmd[1], two)))
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
                label = cmd[1].name if hasattr(cmd[1], 'name') else cmd[2]
                prof.setdefault(cmd[3] + (label,), []).append((e0, e1))
        while nb < len(buckets):
            finishers.append(self._bucket_ready(bucket_hook, buckets[nb], main if two else None, side if two else None))
            nb += 1
        if two:
            self._ev_all.record(side)
            main.wait_event(self._ev_all)
        for f in finishers:
            if f is not None:
                f()
        return pk.grads_to_params(param_views)
