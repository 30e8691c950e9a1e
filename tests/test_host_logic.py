"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares, the
engine's graph builder and weight-packing index maps are correct (checked by emulating the packed GEMMs in
NumPy against torch convolutions), and the synthetic-data generators are deterministic."""
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from yoloseries_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "yolohip.h")).read()
    declared = set(re.findall(r"\b(yh_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"yh_stream", "yh_bf16"}
    L = _lib.lib()                       # raises if any symbol bound in _lib.py is missing
    assert L.yh_version() >= 100
    import ctypes
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, f"declared in include/yolohip.h but not exported: {missing}"
    unbound = [s for s in sorted(declared) if s not in _lib.EXPORTED_SYMBOLS]
    assert not unbound, f"declared but not bound in yoloseries_amd/_lib.py: {unbound}"


def test_ctypes_struct_sizes_match_header_layout():
    """the ctypes mirrors must have the natural C layout of the header structs (pointers 8, ints/floats 4)"""
    import ctypes as C
    from yoloseries_amd import _lib
    assert C.sizeof(_lib.Seg) == 24
    assert C.sizeof(_lib.ConvDesc) == 2 * 24 + 4 * 2 + 4 * 9 + 4 + 8 + 8 + 8 * 3 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 4 + 4 + 8 + 8
    assert C.sizeof(_lib.V5LossDesc) == 4 * 5 + 32 + 8 + 96 + 4 * 4 + 4 * 3 + 12 + 4 + 16 + 4
    assert C.sizeof(_lib.DecodeDesc) == 16 + 32 + 16 + 96 + 4 + 16 + 4


def _builder_for(model, B, H, W):
    from yoloseries_amd.engine import Builder
    b = Builder()
    outs = model._yh_build(b, B, H, W)
    return b, outs


def test_v5s_graph_structure():
    from yoloseries_amd import models
    from yoloseries_amd.engine import ConvOp, PoolOp
    m = models.YOLOV5Small(3, 80)
    b, outs = _builder_for(m, 2, 640, 640)
    convs = [o for o in b.ops if isinstance(o, ConvOp)]
    pools = [o for o in b.ops if isinstance(o, PoolOp)]
    # 60 nn.Conv2d in the reference model; the 8 C3 blocks fuse cba1+cba2 into one GEMM -> 52 conv ops
    assert sum(len(o.parts) for o in convs) == 60 and len(convs) == 52 and len(pools) == 3
    macs = sum(2 * o.Ho * o.Wo * o.N * o.k * o.k * (12 if o.focus else o.Ctot) for o in convs) / 2
    assert abs(macs / 1e9 - 8.217) < 0.01, macs            # 8.217 GMAC / image at 640x640
    assert [(o.Ho, o.N) for o in outs] == [(80, 255), (40, 255), (20, 255)]
    # the neck joins read the low-resolution map through the upsampling addressing mode
    ups = [o.name for o in convs if any(s.ups for s in o.segs)]
    assert ups == ["head_stage1_bscp.cba12", "head_stage2_bscp.cba12"]


def _emulate_conv(pk, op, x_nhwc):
    """NumPy emulation of conv_igemm.hip's math from the packed weight image (fp32, no rounding)"""
    off, npad, K = pk.wloc[(op.name, 'fwd')]
    idx = pk.pack_idx_np[off:off + npad * K].reshape(npad, K)
    flat = np.concatenate([p.detach().numpy().reshape(-1) for p in pk.params])
    Wp = np.where(idx >= 0, flat[np.clip(idx, 0, None)], 0.0)[:op.N]
    Bn, H, W, Cn = x_nhwc.shape
    k, s, p = op.k, op.stride, op.pad
    xp = np.pad(x_nhwc, ((0, 0), (p, p), (p, p), (0, 0)))
    cols = []
    for kh in range(k):
        for kw in range(k):
            cols.append(xp[:, kh:kh + s * op.Ho:s, kw:kw + s * op.Wo:s, :])
    A = np.concatenate(cols, axis=-1).reshape(-1, K)
    return (A @ Wp.T).reshape(Bn, op.Ho, op.Wo, op.N)


def test_weight_packing_maps_cpu():
    """fused C3 dual conv, the space-to-depth stem and a 3x3 conv: packed-image GEMM == torch conv"""
    from yoloseries_amd import models
    from yoloseries_amd.engine import ConvOp, ParamPack
    torch.manual_seed(0)
    m = models.YOLOV5Small(3, 80)
    b, _ = _builder_for(m, 1, 64, 64)
    pk = ParamPack(m, b.ops, host_only=True)
    ops = {o.name: o for o in b.ops if isinstance(o, ConvOp)}
    rs = np.random.RandomState(0)
    # stem: 6x6/s2/p2 on (1,3,64,64) == 3x3/s1/p1 on the space-to-depth tensor
    x = rs.randn(1, 3, 64, 64).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x), m.focus.conv.weight, None, 2, 2).permute(0, 2, 3, 1).detach().numpy()
    s2d = np.zeros((1, 32, 32, 16), np.float32)
    s2d[..., :12] = x.reshape(1, 3, 32, 2, 32, 2).transpose(0, 2, 4, 3, 5, 1).reshape(1, 32, 32, 12)
    np.testing.assert_allclose(_emulate_conv(pk, ops["focus"], s2d), ref, rtol=1e-4, atol=1e-4)
    # fused cba1|cba2 of the first C3
    op = ops["backbone_stage1_bscp.cba12"]
    xin = rs.randn(1, 16, 16, 64).astype(np.float32)
    xt = torch.from_numpy(xin).permute(0, 3, 1, 2)
    c3 = m.backbone_stage1_bscp
    ref = torch.cat([F.conv2d(xt, c3.cba1.conv.weight), F.conv2d(xt, c3.cba2.conv.weight)], 1).permute(0, 2, 3, 1).detach().numpy()
    np.testing.assert_allclose(_emulate_conv(pk, op, xin), ref, rtol=1e-4, atol=1e-4)
    # stride-2 3x3
    op = ops["backbone_stage2_conv"]
    xin = rs.randn(1, 16, 16, 64).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(xin).permute(0, 3, 1, 2), m.backbone_stage2_conv.conv.weight, None, 2, 1).permute(0, 2, 3, 1).detach().numpy()
    np.testing.assert_allclose(_emulate_conv(pk, op, xin), ref, rtol=1e-4, atol=1e-4)
    # gradient un-packing: every parameter element is produced exactly once
    un = pk.unpack_idx_np
    assert (un >= 0).all() and len(np.unique(un)) == len(un) and un.max() < pk.gsize


def test_yolox_block_diagonal_head_packing_cpu():
    from yoloseries_amd import models
    from yoloseries_amd.engine import ConvOp, ParamPack
    torch.manual_seed(0)
    m = models.YOLOXSmall(1, 3, 80, 0.01)
    b, outs = _builder_for(m, 1, 64, 64)
    pk = ParamPack(m, b.ops, host_only=True)
    op = outs[0]
    assert op.N == 85 and op.Ctot == 256 and op.part_seg == [0, 0, 1]
    rs = np.random.RandomState(1)
    freg, fcls = rs.randn(1, 8, 8, 128).astype(np.float32), rs.randn(1, 8, 8, 128).astype(np.float32)
    lay = m.detect.pred_small
    tr, tc = torch.from_numpy(freg).permute(0, 3, 1, 2), torch.from_numpy(fcls).permute(0, 3, 1, 2)
    ref = torch.cat([F.conv2d(tr, lay['reg'].weight), F.conv2d(tr, lay['cof'].weight), F.conv2d(tc, lay['cls'][1].weight)], 1)
    got = _emulate_conv(pk, op, np.concatenate([freg, fcls], -1))
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).detach().numpy(), rtol=1e-4, atol=1e-4)
    # bias gather follows the output-column order reg | cof | cls
    flat = np.concatenate([p.detach().numpy().reshape(-1) for p in pk.params])
    o = pk.bias_loc[op.name]
    bias = flat[pk.fpack_idx_np[o:o + 85]]
    np.testing.assert_array_equal(bias, np.concatenate([lay['reg'].bias.detach().numpy(), lay['cof'].bias.detach().numpy(), lay['cls'][1].bias.detach().numpy()]))
    assert len(m.state_dict()) == 414


def test_synth_generators_are_deterministic():
    from yoloseries_amd.utils.synth import synth_head_outputs, synth_targets
    a, b2 = synth_targets(4, 640, 80, 20, seed=1), synth_targets(4, 640, 80, 20, seed=1)
    np.testing.assert_array_equal(a, b2)
    assert a.shape[2] == 6 and (a[..., 4].max() < 80) and ((a[..., 5] == -1) | (a[..., 5] >= 0)).all()
    pad = a[..., 4] < 0
    assert (a[pad] == -1).all()
    h = synth_head_outputs(1, 64, 80, 3, seed=3)
    assert [x.shape for x in h] == [(1, 255, 8, 8), (1, 255, 4, 4), (1, 255, 2, 2)]


def test_product_refuses_cpu_tensors():
    """no CPU fallback: the product path fails loudly instead of computing on the host"""
    import pytest
    from yoloseries_amd import models
    from yoloseries_amd._lib import YoloHipError
    m = models.YOLOV5Small(3, 80)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.utils.synth import COCO_ANCHORS
    lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS), dict(device="cpu", input_img_size=[64, 64], num_class=80))
    with pytest.raises(YoloHipError):
        lf([torch.zeros(1, 255, 8, 8)], torch.zeros(1, 2, 6))


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under yoloseries_amd/ (nor the drivers) may import or call it"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(root, "yoloseries_amd")) for f in fs if f.endswith(".py")]
    files += [os.path.join(root, f) for f in ("train_yolov5.py", "val_yolov5.py")]
    for path in files:
        src = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
            offenders.append(path)
    assert not offenders, offenders
