"""The net oracle (oracle/v5net.py, torch-CPU fp32) against the outputs of the reference's own classes: it is the expected
value of every per-element block gradient and of the YOLOv5x 1280^2 evaluator test on the GPU, so it is pinned here, on the
CPU, against the committed golden vectors (tools/gen_golden.py ran /root/reference's models / layer classes):

  * g6_blocks.npz   ConvBnAct / BasicBottleneck / C3 / FastSPP: eval output, train output, input gradient, the reference's
                    signature (sum | abs-sum | norm | first elements) of every parameter gradient, BN running statistics;
  * g7_model.npz    YOLOv5s: eval forward at 64^2, train forward at 256^2 (sampled), running statistics;
  * g11_round2.npz  YOLOv5 m / l / x: eval forward at 64^2 on a RandomState-filled state, train forward at 256^2.

fp32 on both sides (same torch build): 2e-5 relative to the tensor's largest value.
"""
import os

import numpy as np
import pytest
import torch

from oracle.v5net import V5NetOracle
from test_gpu_model import _blocks, fill_state

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5


def _near(got, ref, name, tol=TOL):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref).max() if got.size else 0.0
    assert err <= tol * max(np.abs(ref).max(), 1e-6) + 1e-7, f"{name}: max |diff| {err:.3g} vs max |ref| {np.abs(ref).max():.3g}"


def _block_forward(o, key, xt):
    if key.startswith("cba"):
        k = int(o.sd["blk.conv.weight"].shape[-1])
        s, p = {"cba1x1": (1, 0), "cba3x3s2": (2, 1), "cba6x6s2": (2, 2)}[key]
        return o.cba(xt, "blk", k, s, p)
    if key == "bneck":
        return o.bottleneck(xt, "blk", True)
    if key in ("c3", "c3ns"):
        return o.c3(xt, "blk", key == "c3")
    return o.sppf(xt, "blk")


@pytest.mark.parametrize("key", ["cba1x1", "cba3x3s2", "cba6x6s2", "bneck", "c3", "c3ns", "sppf"])
def test_oracle_blocks_vs_reference(key):
    g = np.load(os.path.join(G, "g6_blocks.npz"))
    seed, cin, hw = (int(v) for v in g[f"{key}_args"])
    mod = _blocks()[key][0]()
    fill_state(mod, seed)
    sd = {"blk." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    x = np.random.RandomState(seed + 1).randn(2, cin, hw, hw).astype(np.float32)
    with torch.no_grad():
        _near(_block_forward(V5NetOracle(sd, train=False), key, torch.from_numpy(x)).numpy(), g[f"{key}_eval"], key + " eval")
    o = V5NetOracle(sd, train=True)
    xt = torch.from_numpy(x).requires_grad_(True)
    y = _block_forward(o, key, xt)
    _near(y.detach().numpy(), g[f"{key}_train"], key + " train")
    names = [n for n, _ in mod.named_parameters()]
    grads = torch.autograd.grad(y, [xt] + [o.params["blk." + n] for n in names], torch.from_numpy(g[f"{key}_gout"]))
    _near(grads[0].numpy(), g[f"{key}_gx"], key + " gx", 1e-4)
    for n, gr in zip(names, grads[1:]):
        sig = g[f"{key}_gp_{n}"]                     # sum | abs-sum | norm | first 29 elements, written by the reference
        gf = gr.double().reshape(-1)
        assert abs(gf.norm().item() - sig[2]) <= 1e-4 * sig[2] + 1e-9, f"{key} {n} norm"
        assert abs(gf.abs().sum().item() - sig[1]) <= 1e-4 * sig[1] + 1e-9, f"{key} {n} abs-sum"
        k = min(29, gf.numel())
        assert np.abs(gf[:k].numpy() - sig[3:3 + k]).max() <= 1e-4 * max(np.abs(sig[3:3 + k]).max(), sig[1] / gf.numel()) + 1e-8, f"{key} {n} head"
    for n, _ in mod.named_buffers():                 # F.batch_norm updated the oracle's copies of the running statistics in place
        if not n.endswith("num_batches_tracked"):
            _near(o.sd["blk." + n].numpy(), g[f"{key}_buf_{n}"], f"{key} buffer {n}", 1e-5)


def _model_cls(name):
    from yoloseries_amd import models
    return {"s": models.YOLOV5Small, "m": models.YOLOV5Middle, "l": models.YOLOV5Large, "x": models.YOLOV5XLarge}[name]


def test_oracle_v5s_vs_reference_model():
    g = np.load(os.path.join(G, "g7_model.npz"))
    torch.manual_seed(0)
    sd = _model_cls("s")(3, 80).state_dict()
    x = torch.from_numpy(np.random.RandomState(70).rand(2, 3, 64, 64).astype(np.float32))
    with torch.no_grad():
        for i, o in enumerate(V5NetOracle(sd, train=False)(x)):
            _near(o.numpy(), g[f"s_eval64_out{i}"], f"v5s eval out{i}")
    net = V5NetOracle(sd, train=True)
    x2 = torch.from_numpy(np.random.RandomState(71).rand(2, 3, 256, 256).astype(np.float32))
    with torch.no_grad():
        outs = net(x2)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(g[f"s_train256_shape{i}"])
        _near(o.reshape(-1).numpy()[g[f"s_train256_idx{i}"]], g[f"s_train256_val{i}"], f"v5s train out{i}", 2e-4)
    _near(net.sd["focus.bn.running_mean"].numpy(), g["s_train256_rm_focus"], "focus running_mean", 1e-5)
    _near(net.sd["focus.bn.running_var"].numpy(), g["s_train256_rv_focus"], "focus running_var", 1e-5)
    _near(net.sd["head_stage4_bscp.cba3.bn.running_var"].numpy(), g["s_train256_rv_last"], "last running_var", 1e-4)


@pytest.mark.parametrize("name", ["m", "l", "x"])
def test_oracle_v5mlx_vs_reference_model(name):
    g = np.load(os.path.join(G, "g11_round2.npz"))
    seed = int(g[f"{name}_seed"][0])
    m2 = _model_cls(name)(3, 80)
    fill_state(m2, seed)
    x = torch.from_numpy(np.random.RandomState(seed + 10).rand(2, 3, 64, 64).astype(np.float32))
    with torch.no_grad():
        for i, o in enumerate(V5NetOracle(m2.state_dict(), train=False)(x)):
            _near(o.numpy(), g[f"{name}_eval64_out{i}"], f"v5{name} eval out{i}")
    torch.manual_seed(0)
    net = V5NetOracle(_model_cls(name)(3, 80).state_dict(), train=True)
    x2 = torch.from_numpy(np.random.RandomState(seed + 11).rand(2, 3, 256, 256).astype(np.float32))
    with torch.no_grad():
        outs = net(x2)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(g[f"{name}_train256_shape{i}"])
        # ~60 stacked train-mode BatchNorms: fp32 summation order (thread count of the host) moves the deepest outputs by ~1e-4
        _near(o.reshape(-1).numpy()[g[f"{name}_train256_idx{i}"]], g[f"{name}_train256_val{i}"], f"v5{name} train out{i}", 5e-4)
    for pn, tol in (("focus", 1e-5), ("backbone_stage2_conv", 1e-4), ("backbone_stage4_conv", 1e-4), ("head_stage4_bscp.cba3", 5e-4)):
        _near(net.sd[pn + ".bn.running_mean"].numpy(), g[f"{name}_rm_{pn}"], f"v5{name} {pn} running_mean", tol)
        _near(net.sd[pn + ".bn.running_var"].numpy(), g[f"{name}_rv_{pn}"], f"v5{name} {pn} running_var", tol)


def test_oracle_v5s_eval_mode_gradients_vs_reference():
    """model.eval() under autograd (BatchNorm on its running statistics, no batch coupling): the net oracle's outputs and the
    gradient of EVERY parameter against the reference's (g12_round3.npz: signature + 256 sampled elements per parameter)"""
    g = np.load(os.path.join(G, "g12_round3.npz"))
    seed = int(g["v5s_frozen_seed"][0])
    m = _model_cls("s")(3, 80)
    fill_state(m, seed)
    net = V5NetOracle(m.state_dict(), train=False, grad=True)
    x = torch.from_numpy(np.random.RandomState(1201).rand(2, 3, 256, 256).astype(np.float32))
    outs = net(x)
    r = np.random.RandomState(seed + 1)
    gos = [torch.from_numpy((r.randn(*o.shape) * 0.1).astype(np.float32)) for o in outs]
    for i, o in enumerate(outs):
        _near(o.detach().reshape(-1).numpy()[g[f"v5s_frozen_out_idx{i}"]], g[f"v5s_frozen_out_val{i}"], f"eval-mode out{i}")
    names = [str(n) for n in g["v5s_frozen_pnames"]]
    assert names == [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad(outs, [net.params[n] for n in names], gos)
    for pi, (n, gr) in enumerate(zip(names, grads)):
        gf = gr.double().reshape(-1).numpy()
        s_sum, s_abs, s_norm, s_size, s_max = g["v5s_frozen_psig"][pi]
        assert gf.size == int(s_size)
        assert abs(np.sqrt((gf ** 2).sum()) - s_norm) <= 1e-4 * s_norm + 1e-9, n
        assert np.abs(gf[g["v5s_frozen_pidx"][pi]] - g["v5s_frozen_pval"][pi]).max() <= 1e-4 * s_max + 1e-9, n
