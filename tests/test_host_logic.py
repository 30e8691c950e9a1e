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


def test_ctypes_struct_sizes_match_header_layout(tmp_path):
    """the ctypes mirrors (yoloseries_amd/_lib.py) against the C compiler's view of include/yolohip.h: size of every struct and the
    offset of every field, from a probe compiled with gcc"""
    import ctypes as C
    import subprocess
    from yoloseries_amd import _lib
    pairs = [("yh_seg", _lib.Seg), ("yh_conv_desc", _lib.ConvDesc), ("yh_wgrad_desc", _lib.WgradDesc), ("yh_v5loss_desc", _lib.V5LossDesc),
             ("yh_yolox_desc", _lib.YoloxDesc), ("yh_decode_desc", _lib.DecodeDesc), ("yh_bn_fold_item", _lib.BnFoldItem),
             ("yh_bn_part", _lib.BnPart), ("yh_cmd", _lib.Cmd)]
    hdr = open(os.path.join(ROOT, "include", "yolohip.h")).read()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "yolohip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        body = hdr[hdr.index(f"typedef struct {cname} {{"):hdr.index(f"}} {cname};")]
        for fname, _ in cls._fields_:
            if fname.startswith("reserved") and not re.search(rf"\b{fname}\b", body):
                continue
            assert re.search(rf"\b{fname}\b", body), f"{cname}.{fname} is in the ctypes mirror but not in the header"
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("  return 0; }")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    want = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert C.sizeof(cls) == int(want[cname]), f"{cname}: ctypes {C.sizeof(cls)} bytes, C {want[cname]}"
        for fname, _ in cls._fields_:
            key = f"{cname}.{fname}"
            if key in want:
                assert getattr(cls, fname).offset == int(want[key]), f"{key}: ctypes offset {getattr(cls, fname).offset}, C {want[key]}"


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


# names the reference's drivers import (train_yolov5.py:28-44, val_yolov5.py:20-33; SURVEY.md §8b "Runtime helpers the
# drivers import"): with `utils`, `trainer`, `loss`, `models`, `dataset`, `config` aliased to this package every one of them
# must resolve, so the driver scripts only change their import roots
DRIVER_IMPORTS = {
    "config": ["Config"],
    "loss": ["YOLOV5Loss", "YOLOXLoss"],
    "trainer": ["YOLOV5Evaluator", "YOLOXEvaluator", "ExponentialMovingAverageModel"],
    "dataset": ["build_dataloader", "build_test_dataloader", "build_val_dataloader"],
    "models": ["YOLOV5Small", "YOLOV5Middle", "YOLOV5Large", "YOLOV5XLarge", "YOLOXSmall"],
    "utils": ["cv2_save_img", "cv2_save_img_plot_pred_gt", "maybe_mkdir", "clear_dir", "time_synchronize", "summary_model", "mAP_v2", "configure_nccl",
              "configure_omp", "get_local_rank", "print_config", "get_rank", "get_world_size", "occupy_mem", "padding",
              "MeterBuffer", "all_reduce_norm", "is_parallel", "adjust_status", "synchronize", "configure_module", "launch",
              "get_num_devices", "gpu_nms", "gpu_linear_soft_nms", "gpu_exponential_soft_nms", "numba_nms", "gpu_iou",
              "gpu_CIoU", "gpu_DIoU", "gpu_Giou", "xyxy2xywh", "xyxy2xywhn", "xywh2xyxy", "numba_iou", "numba_xywh2xyxy",
              "numba_xyxy2xywh", "letter_resize_img", "letter_resize_bbox"],
}


def _alias_modules():
    import importlib
    import sys
    import yoloseries_amd.dataset, yoloseries_amd.loss, yoloseries_amd.models, yoloseries_amd.trainer, yoloseries_amd.utils  # noqa: F401,E401
    saved = {k: sys.modules.get(k) for k in DRIVER_IMPORTS}
    for k in ("utils", "trainer", "loss", "models", "dataset"):
        sys.modules[k] = importlib.import_module("yoloseries_amd." + k)
    sys.modules["config"] = importlib.import_module("config")
    return saved


def _restore_modules(saved):
    import sys
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_driver_import_surface_resolves():
    saved = _alias_modules()
    try:
        ns = {}
        for mod, names in DRIVER_IMPORTS.items():
            exec(f"from {mod} import {', '.join(names)}", ns)
        exec("from models import *", ns)
        assert callable(ns["launch"]) and callable(ns["YOLOV5Small"])
        # when the reference tree is present (build container), its own import statements are executed against the aliases
        for ref in ("/root/reference/train_yolov5.py", "/root/reference/val_yolov5.py", "/root/reference/train_yolox.py",
                    "/root/reference/val_yolox.py"):
            if os.path.exists(ref):
                import ast
                tree = ast.parse(open(ref).read())
                ours = set(DRIVER_IMPORTS)
                for node in tree.body:
                    if isinstance(node, ast.ImportFrom) and node.module in ours:
                        code = ast.unparse(node)
                        exec(code, {})           # raises ImportError if a name the reference's driver needs is missing
    finally:
        _restore_modules(saved)


def test_runtime_helpers_behaviour(tmp_path):
    """host helpers mirrored from utils/common.py, utils/meter.py, utils/model_utils.py, utils/logger.py, utils/setup_env.py"""
    import torch
    from yoloseries_amd import models, utils as U
    assert U.padding(640) == (640, 640) and U.padding(641) == (672, 672) and U.padding((100, 33), 32) == (128, 64)
    d = tmp_path / "a" / "b"
    U.maybe_mkdir(str(d)); U.maybe_mkdir(d)
    (d / "f.txt").write_text("x")
    U.clear_dir(str(d))
    assert d.exists() and not list(d.iterdir())
    assert not U.is_parallel(torch.nn.Linear(2, 2))
    assert isinstance(U.time_synchronize(), float)
    mb = U.MeterBuffer(window_size=3)
    for v in (1.0, 2.0, 3.0, 4.0):
        mb.update(tot_loss=torch.tensor(v), iter_time=v * 2)
    assert mb["tot_loss"].latest == 4.0 and abs(mb["tot_loss"].avg - 3.0) < 1e-9 and abs(mb["tot_loss"].global_avg - 2.5) < 1e-9
    assert list(mb.get_filtered_meter("time")) == ["iter_time"] and mb["iter_time"].median == 6.0
    mb.clear_meters(); assert mb["tot_loss"].latest is None and mb["tot_loss"].total == 10.0
    m = models.YOLOV5Small(3, 80).train()
    with U.adjust_status(m, training=False) as mm:
        assert not any(x.training for x in mm.modules())
    assert all(x.training for x in m.modules())
    sm = U.summary_model(m, [640, 640])
    assert sm["number_params"] == 7235389 and abs(sm["flops"] * 2 - 8.217) < 0.01      # MACs / 2e9 like the reference's thop line
    # drawing helpers of the val driver (val_yolov5.py:21-22): files are written, boxes change pixels, the blend keeps the size
    import numpy as np
    from PIL import Image
    pic = np.full((64, 96, 3), 90, dtype=np.uint8)
    U.cv2_save_img(pic, [[10, 20, 50, 60]], [3], [0.9], str(tmp_path / "v" / "p.png"))
    U.cv2_save_img_plot_pred_gt(pic, [[10, 20, 50, 60]], [3], [0.9], [[30, 25, 80, 50]], [1], str(tmp_path / "v" / "pg.png"))
    U.cv2_save_img_plot_pred_gt(pic, [], [], [], [], [], str(tmp_path / "v" / "none.png"))
    a, b, c0 = (np.asarray(Image.open(tmp_path / "v" / n)) for n in ("p.png", "pg.png", "none.png"))
    assert a.shape == b.shape == c0.shape == pic.shape and (a != pic).any() and (b != a).any() and (c0 == pic).all()
    assert tuple(a[40, 10]) == (0, 238, 238)                     # left edge of the predicted box
    table = U.print_config({"lr": 0.01, "_hidden": 1, "name": "x"})
    assert "lr" in table and "_hidden" not in table
    import os as _os
    env = dict(_os.environ)
    try:
        U.configure_nccl(); U.configure_omp(); U.configure_module()
        assert _os.environ["NCCL_IB_DISABLE"] == "1" and _os.environ["NCCL_SOCKET_IFNAME"] == "lo"
    finally:
        _os.environ.clear(); _os.environ.update(env)
    import numpy as _np
    U.cv2_save_img(_np.zeros((64, 64, 3), _np.uint8), [[4, 4, 40, 40]], [3], [0.9], str(tmp_path / "o" / "x.png"))
    assert (tmp_path / "o" / "x.png").stat().st_size > 0
    called = []
    U.launch(lambda a: called.append(a), 1, args=(5,))
    assert called == [5]


def _launch_main(tag):
    import torch.distributed as dist
    from yoloseries_amd.utils import get_local_rank, get_rank, get_world_size
    t = __import__("torch").tensor([float(get_rank() + 1)])
    dist.all_reduce(t)
    assert get_world_size() == 2 and t.item() == 3.0 and get_local_rank() == get_rank()
    open(f"{tag}.{get_rank()}", "w").write("ok")


def test_launch_two_ranks_gloo(tmp_path):
    """utils.launch (utils/launch.py:39-139): spawns one process per rank, initialises the process group and the local group"""
    from yoloseries_amd.utils import launch
    tag = str(tmp_path / "done")
    launch(_launch_main, 2, backend="gloo", dist_url="auto", args=(tag,))
    assert os.path.exists(tag + ".0") and os.path.exists(tag + ".1")


def test_shipped_tuning_table_matches_the_engine_key_versions():
    """yoloseries_amd/tune_defaults.json (tools/make_tune_defaults.sh) must be regenerated whenever the meaning of a tuned value
    changes: its keys carry the same version prefixes the engine asks for"""
    import json
    from yoloseries_amd import engine
    path = engine.TUNE_DEFAULTS_PATH
    assert os.path.dirname(path) == os.path.dirname(os.path.dirname(os.path.abspath(engine.__file__)))
    table = json.load(open(path))
    prefixes = {k.split(":", 1)[0] for k in table}
    assert prefixes <= engine.TUNE_KEY_VERSIONS and engine.KEY_CONV in prefixes, (prefixes, engine.TUNE_KEY_VERSIONS)
    assert len(table) > 300
    assert all(isinstance(v, list) and all(isinstance(x, int) for x in v) for v in table.values())


def test_planning_helpers_tolerate_empty_descriptors():
    """yh_conv_stat_blocks / yh_conv_bnr_rows are called while a descriptor is being filled in: a zero dimension answers 0"""
    import ctypes as C
    from yoloseries_amd._lib import ConvDesc, lib
    d = ConvDesc()
    assert lib().yh_conv_stat_blocks(C.byref(d)) == 0
    assert lib().yh_conv_bnr_rows(C.byref(d)) == 0


def test_executor_knows_every_program_entry_point():
    """yh_exec (csrc/exec.hip) replays the engine's command lists: every entry point a Program emits must be in its table, with
    the argument count the ctypes signature declares (stream included)"""
    import ctypes as C
    from yoloseries_amd import _lib
    L = _lib.lib()
    for name in ("yh_conv_igemm", "yh_conv_wgrad", "yh_bn_finalize", "yh_bn_fold_batch", "yh_bn_silu_apply", "yh_bn_silu_bwd_reduce",
                 "yh_bn_bwd_finalize", "yh_bn_silu_bwd_apply", "yh_colsum", "yh_maxpool5_fwd", "yh_maxpool5_bwd", "yh_upsample2_bwd",
                 "yh_fill_u32"):
        n = C.c_int32(0)
        assert L.yh_exec_op(name.encode(), C.byref(n)) >= 0, name
        assert n.value == len(_lib._SIGS[name][1]) <= _lib.YH_CMD_SLOTS, (name, n.value)
    assert L.yh_exec_op(b"yh_nms_batched", None) == -1
    assert C.sizeof(_lib.Cmd) == 16 + 8 * _lib.YH_CMD_SLOTS


def test_fuse_conv_bn_matches_reference_fixture():
    """fuse_conv_bn (utils/layer_tools.py:26-53): weight and bias of the fused conv against the reference's (g6 `fuse_w`, `fuse_b`)
    — host-side parameter algebra, CPU"""
    import torch
    from test_gpu_model import fill_state
    from yoloseries_amd.utils.layer_tools import ConvBnAct, fuse_conv_bn
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_blocks.npz"))
    cb = ConvBnAct(16, 32, 3, 1, 1)
    fill_state(cb, int(g["fuse_args"][0]))
    with torch.no_grad():
        fused = fuse_conv_bn(cb.conv, cb.bn)
    assert not fused.weight.requires_grad and not fused.bias.requires_grad
    np.testing.assert_allclose(fused.weight.numpy(), g["fuse_w"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fused.bias.numpy(), g["fuse_b"], rtol=1e-6, atol=1e-7)


def _asm_pending_reads(text):
    """compiler-generated instructions that read a register an inline-asm load (ds_read / buffer_load into registers) has written,
    in front of the next inline-asm s_waitcnt of the load's counter: [(kernel, line, instruction)]"""
    def regs(tok):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
        return out
    hits, in_asm, pend, kern = [], False, {}, ""
    for i, ln in enumerate(text.split("\n")):
        t = ln.strip()
        if re.match(r"^_Z\w+:", ln):
            kern, pend = ln.split(":")[0], {}
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ln.startswith("\t") or t.startswith((".", ";")):
            continue
        if in_asm:
            m = re.match(r"(ds_read\w*|buffer_load_dword\w*)\s+(\S+),", t)
            if m and not t.endswith(" lds"):
                for r in regs(m[2]):
                    pend[r] = "lgkm" if m[1].startswith("ds_") else "vm"
            if t.startswith("s_waitcnt"):
                pend = {r: c for r, c in pend.items() if not ((c == "lgkm" and "lgkmcnt" in t) or (c == "vm" and "vmcnt" in t))}
        elif pend:
            parts = t.split(None, 1)
            if len(parts) == 2 and parts[0].startswith(("v_", "ds_", "buffer_", "global_")):
                ops = parts[1].split(",")
                srcs = parts[1] if parts[0].startswith(("buffer_store", "global_store", "ds_write", "v_cmp")) else ",".join(ops[1:])
                if regs(srcs) & set(pend):
                    hits.append((kern, i, t))
    return hits


def test_inline_asm_loads_are_not_read_before_their_wait(tmp_path):
    """conv_halo160_kernel requests its fragments by inline-asm ds_read and waits for them by an inline-asm s_waitcnt tied to the
    destination registers.  The compiler believes those registers defined at the ds_read: should it ever copy or use one in front
    of the wait (it did exactly that to conv_pt_kernel's operand loads in round 5), the copy holds stale bytes.  The generated
    code of every kernel of conv_igemm.hip is scanned: no compiler-generated instruction reads such a register before the wait."""
    import subprocess
    src = os.path.join(ROOT, "yoloseries_amd", "csrc", "conv_igemm.hip")
    out = tmp_path / "igemm.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only", "-S",
                    "-o", str(out), src], check=True, capture_output=True)
    text = out.read_text()
    assert "conv_halo160_kernel" in text and len(re.findall(r";;#ASMSTART\n\s*ds_read_b128", text)) >= 10
    hits = _asm_pending_reads(text)
    assert not hits, hits[:5]


def test_conv_pt_kernel_has_no_register_spills(tmp_path):
    """conv_pt_kernel counts its vector-memory instructions by hand (s_waitcnt vmcnt(N) with compile-time N): a register spill would add
    scratch loads / stores the counts do not know about.  Every instantiation the library launches must compile without scratch."""
    import subprocess
    src = os.path.join(ROOT, "yoloseries_amd", "csrc", "conv_pt.hip")
    out = tmp_path / "pt.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only", "-S",
                    "-o", str(out), src], check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN\S*conv_pt_kernel\S*):.*?; ScratchSize: (\d+)", text, flags=re.S | re.M)
    assert len(kernels) >= 20, len(kernels)
    bad = [(k, n) for k, n in kernels if int(n) != 0]
    assert not bad, bad
    # nothing in flight may be held in a register the compiler knows as a C value: every inline-asm load of this file is an LDS-DMA
    # transfer (round 5: operands loaded into registers by asm and "tied" to their wait were copied by the compiler in front of it)
    asm_loads = re.findall(r";;#ASMSTART\n((?:(?!;;#ASMEND).)*?buffer_load_dword\S*[^\n]*)", text, flags=re.S)
    assert asm_loads and all(ln.strip().endswith(" lds") for blk in asm_loads for ln in blk.split("\n") if "buffer_load" in ln), \
        [ln for blk in asm_loads for ln in blk.split("\n") if "buffer_load" in ln and not ln.strip().endswith(" lds")][:3]
