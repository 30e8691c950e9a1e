"""GPU parity of the conv graph executed by the HIP engine against golden vectors produced by the
reference's own layer / model classes (fp32 on CPU).

The HIP path computes in bf16 with fp32 accumulation (BASELINE.json config #2), so values are
compared with bf16 tolerances: activations  |err| <= 3e-2*max|ref| + 3e-2*|ref|  (a handful of
stacked bf16 roundings), gradients 6e-2 relative to the tensor's max.  Running statistics
are fp32 reductions of bf16-rounded conv outputs: 2e-2.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def fill_state(mod, seed):
    """must mirror tools/gen_golden.py::fill_state (same RandomState call order)"""
    r = np.random.RandomState(seed)
    sd = mod.state_dict()
    for k2, v in sd.items():
        if k2.endswith("num_batches_tracked"):
            continue
        shape = tuple(v.shape)
        if k2.endswith("running_var"):
            a = r.uniform(0.5, 1.5, shape)
        elif k2.endswith("bn.weight"):
            a = r.uniform(0.7, 1.3, shape)
        elif k2.endswith(("running_mean", "bn.bias", ".bias")):
            a = r.randn(*shape) * 0.2
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            a = r.randn(*shape) / np.sqrt(fan_in)
        sd[k2] = torch.from_numpy(a.astype(np.float32))
    mod.load_state_dict(sd)


def _close(got, ref, rel, name, outlier_frac=0.0):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    lim = rel * np.abs(ref).max() + rel * np.abs(ref)
    err = np.abs(got - ref)
    nbad = int((err > lim).sum())
    assert nbad <= outlier_frac * err.size, f"{name}: max err {err.max():.4g} vs ref max {np.abs(ref).max():.4g}; {nbad} of {err.size} out of tol"


def _blocks():
    from yoloseries_amd import utils as U
    return {
        "cba1x1": (lambda: U.ConvBnAct(32, 64, 1, 1), 32),
        "cba3x3s2": (lambda: U.ConvBnAct(32, 64, 3, 2, 1), 32),
        "cba6x6s2": (lambda: U.ConvBnAct(3, 32, 6, 2, 2), 3),
        "bneck": (lambda: U.BasicBottleneck(32, 32, True, expand_ratio=1.0), 32),
        "c3": (lambda: U.C3BottleneckCSP(64, 64, shortcut=True, num_block=2), 64),
        "c3ns": (lambda: U.C3BottleneckCSP(128, 64, shortcut=False, num_block=1), 128),
        "sppf": (lambda: U.FastSPP(64, 64), 64),
    }


@pytest.mark.parametrize("key", ["cba1x1", "cba3x3s2", "cba6x6s2", "bneck", "c3", "c3ns", "sppf"])
def test_block_golden(dev, key):
    g = np.load(os.path.join(G, "g6_blocks.npz"))
    seed, cin, hw = (int(v) for v in g[f"{key}_args"])
    ctor, _ = _blocks()[key]
    mod = ctor()
    fill_state(mod, seed)
    mod = mod.to(dev)
    r = np.random.RandomState(seed + 1)
    x = r.randn(2, cin, hw, hw).astype(np.float32)
    mod.eval()
    with torch.no_grad():
        ye = mod(torch.from_numpy(x).to(dev))
    _close(ye.float().cpu().numpy(), g[f"{key}_eval"], 3e-2, key + " eval")
    mod.train()
    xt = torch.from_numpy(x).to(dev).requires_grad_(cin >= 8)
    y = mod(xt)
    _close(y.detach().float().cpu().numpy(), g[f"{key}_train"], 3e-2, key + " train")
    go = torch.from_numpy(g[f"{key}_gout"]).to(dev)
    params = list(mod.parameters())
    grads = torch.autograd.grad(y, ([xt] if cin >= 8 else []) + params, go)
    if cin >= 8:
        # max-pool routes each gradient to the arg-max pixel: where bf16 rounding makes two window
        # entries (nearly) tie, the route differs from the fp32 reference for a few elements
        _close(grads[0].cpu().numpy(), g[f"{key}_gx"], 6e-2, key + " gx", outlier_frac=0.005 if key == "sppf" else 0.0)
        grads = grads[1:]
    for (n, p), gr in zip(mod.named_parameters(), grads):
        sig = g[f"{key}_gp_{n}"]
        gf = gr.double().reshape(-1).cpu()
        scale = sig[1] / gf.numel() + 1e-12            # mean |grad| of the reference
        got_first = gf[:29].numpy()
        assert np.abs(got_first - sig[3:3 + len(got_first)]).max() <= 6e-2 * max(np.abs(sig[3:]).max(), scale) + 0.15 * scale, f"{key} grad {n}"
        assert abs(gf.norm().item() - sig[2]) <= 5e-2 * sig[2] + 1e-6, f"{key} grad-norm {n}: {gf.norm().item()} vs {sig[2]}"
    for n, bf in mod.named_buffers():
        ref = g[f"{key}_buf_{n}"]
        if n.endswith("num_batches_tracked"):
            assert int(bf.item()) == int(ref)
        else:
            _close(bf.cpu().numpy(), ref, 2e-2, f"{key} buffer {n}")


def test_model_state_dict_and_init():
    """CPU-side: same 348 keys and bit-identical seeded init as the reference."""
    from yoloseries_amd import models
    g = np.load(os.path.join(G, "g7_model.npz"))
    for name, cls in (("s", models.YOLOV5Small), ("l", models.YOLOV5Large)):
        torch.manual_seed(0)
        m = cls(3, 80)
        sd = m.state_dict()
        assert list(sd.keys()) == list(g[f"{name}_keys"])
        assert [str(tuple(v.shape)) for v in sd.values()] == list(g[f"{name}_shapes"])
        np.testing.assert_allclose([v.double().sum().item() for v in sd.values()], g[f"{name}_psum"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose([v.double().abs().sum().item() for v in sd.values()], g[f"{name}_pabs"], rtol=1e-12, atol=1e-12)   # summation order differs between hosts


def test_model_forward_golden(dev):
    from yoloseries_amd import models
    g = np.load(os.path.join(G, "g7_model.npz"))
    torch.manual_seed(0)
    m = models.YOLOV5Small(3, 80).to(dev)
    x = torch.from_numpy(np.random.RandomState(70).rand(2, 3, 64, 64).astype(np.float32)).to(dev)
    m.eval()
    with torch.no_grad():
        outs = m(x)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == g[f"s_eval64_out{i}"].shape
        _close(o.float().cpu().numpy(), g[f"s_eval64_out{i}"], 3e-2, f"eval out{i}")
    m.train()
    x2 = torch.from_numpy(np.random.RandomState(71).rand(2, 3, 256, 256).astype(np.float32)).to(dev)
    outs = m(x2)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(g[f"s_train256_shape{i}"])
        flat = o.detach().float().contiguous().cpu().numpy().reshape(-1)
        _close(flat[g[f"s_train256_idx{i}"]], g[f"s_train256_val{i}"], 6e-2, f"train out{i}", outlier_frac=0.015)
        # calibration: the reference itself under torch bf16 autocast differs from its fp32 run by up to 1.6
        # (of max 10.1) with 0.9% of the deepest-stage elements outside this 6% band (~60 stacked bf16 layers)
    _close(m.focus.bn.running_mean.cpu().numpy(), g["s_train256_rm_focus"], 2e-2, "focus running_mean")
    _close(m.focus.bn.running_var.cpu().numpy(), g["s_train256_rv_focus"], 2e-2, "focus running_var")
    _close(m.head_stage4_bscp.cba3.bn.running_var.cpu().numpy(), g["s_train256_rv_last"], 3e-2, "last running_var")


def test_train_step_loss_decreases(dev):
    """forward + loss + backward + SGD on a fixed synthetic batch: gradients reach every parameter and
    the loss goes down; state_dict round-trips through load_state_dict."""
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    torch.manual_seed(0)
    m = models.YOLOV5Small(3, 80).to(dev)
    hyp = dict(device=dev, num_class=80, input_img_size=[128, 128], use_focal_loss=True, focal_loss_gamma=1.5,
               focal_loss_alpha=0.25, iou_loss_scale=0.05, cls_loss_scale=0.5, cof_loss_scale=1.0, anchor_match_thr=4.0,
               class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0)
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), hyp)
    opt = torch.optim.SGD(m.parameters(), lr=0.005, momentum=0.9, nesterov=True)
    x = torch.from_numpy(np.random.RandomState(3).rand(4, 3, 128, 128).astype(np.float32)).to(dev)
    t = torch.from_numpy(synth_targets(4, 128, 80, 6, seed=4)).to(dev)
    losses = []
    for it in range(10):
        out = lossf(m(x), t)
        opt.zero_grad()
        out["tot_loss"].backward()
        if it == 0:
            missing = [n for n, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all() or p.grad.abs().sum() == 0]
            assert not missing, f"parameters without a finite non-zero gradient: {missing[:8]}"
        opt.step()
        losses.append(out["tot_loss"].item())
    # weight gradients are summed with fp32 atomics (order varies run to run): compare a window, not one step
    assert all(np.isfinite(losses)) and min(losses[-4:]) < losses[0], losses
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = models.YOLOV5Small(3, 80).to(dev)
    m2.load_state_dict(sd)
    m.eval(); m2.eval()
    with torch.no_grad():
        a = m(x); b2 = m2(x)
    for u, v in zip(a, b2):
        assert torch.equal(u, v)
