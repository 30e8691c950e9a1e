"""Program, forward side: the inference program (folded BatchNorm + SiLU in the conv epilogue), the training program
(conv + BatchNorm partial sums -> finalize -> BN+SiLU pass), evaluation-mode BatchNorm under autograd, and running a pass."""
import ctypes as C
import os

import torch

from .. import hipk
from .._lib import BnFoldItem, BnPart, ConvDesc, YH_BN_MAX_PARTS, YH_ACT_NONE, YH_ACT_SILU, YH_CONV_FWD, YoloHipError, check
from . import flags as _flags
from .executor import CompiledCmds
from .graph import ConvOp, PoolOp, Ref, sppf_chain
from .tune import _tune_cache_save


class ForwardMixin:

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
        if os.environ.get("YH_BWD_STREAMS", "1") != "0" and not torch.cuda.is_current_stream_capturing():
            # the backward's weight-gradient stream is probed HERE — on the caller's thread, at program build, before the training
            # tensors exist — not inside the first backward() (autograd thread, activations resident): streams.py
            from ..streams import side_stream
            side_stream(self.dev)
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
            merged = _flags.MERGE_PARTS and 2 <= len(op.parts) <= YH_BN_MAX_PARTS and op.res is None
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

    @staticmethod
    def _conv_bytes(d):
        """algorithmic HBM bytes of one yh_conv_igemm launch: every input segment read once, the output written once (read too when
        it accumulates), residual / fused-reduction operands read once; weights are negligible next to the activations"""
        rd = sum(2.0 * d.B * (d.Hi >> d.seg[i].ups) * (d.Wi >> d.seg[i].ups) * d.seg[i].C for i in range(d.nseg))
        out = 2.0 * d.B * d.Ho * d.Wo * d.N
        return rd + out * (2.0 if d.accumulate else 1.0) + (out * min(1.0, d.nsplit / max(d.N, 1)) if d.res else 0.0) + (out if d.bnr_part else 0.0)

    def _fam_conv(self, op, d):
        M = self.B * op.Ho * op.Wo
        return (self._kernel_name(d), 2.0 * M * op.N * op.k * op.k * (12 if op.focus else op.Ctot), self._conv_bytes(d))

    def _compile(self, cmds):
        cc = CompiledCmds(self.L, len(cmds))
        for fn, args, name, meta in cmds:
            if getattr(fn, "__name__", "") in _flags.ABL_SKIP:
                continue
            cc.call(fn, args, 0, name)
        return cc

    def _run(self, cmds):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        prof = self.profile
        if prof is None and _flags.USE_EXEC:
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
