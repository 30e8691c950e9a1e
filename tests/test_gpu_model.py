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


def _oracle_block_param_grads(key, sd, x, gout, names):
    """parameter gradients of one block from the CPU oracle (train-mode BatchNorm, fp32), in the order of `names`"""
    from oracle.v5net import V5NetOracle
    o = V5NetOracle({"blk." + k: v for k, v in sd.items()}, train=True)
    xt = torch.from_numpy(x)
    if key.startswith("cba"):
        w = sd["conv.weight"]
        k = int(w.shape[-1])
        s, p = {"cba1x1": (1, 0), "cba3x3s2": (2, 1), "cba6x6s2": (2, 2)}[key]
        y = o.cba(xt, "blk", k, s, p)
    elif key == "bneck":
        y = o.bottleneck(xt, "blk", True)
    elif key in ("c3", "c3ns"):
        y = o.c3(xt, "blk", key == "c3")
    else:
        y = o.sppf(xt, "blk")
    return torch.autograd.grad(y, [o.params["blk." + n] for n in names], torch.from_numpy(np.asarray(gout, dtype=np.float32)))


@pytest.mark.parametrize("key", ["cba1x1", "cba3x3s2", "cba6x6s2", "bneck", "c3", "c3ns", "sppf"])
def test_block_golden(dev, key):
    g = np.load(os.path.join(G, "g6_blocks.npz"))
    seed, cin, hw = (int(v) for v in g[f"{key}_args"])
    ctor, _ = _blocks()[key]
    mod = ctor()
    fill_state(mod, seed)
    sd0 = {k: v.detach().clone() for k, v in mod.state_dict().items()}
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
    oracle_grads = _oracle_block_param_grads(key, sd0, x, g[f"{key}_gout"], [n for n, _ in mod.named_parameters()])
    for (n, p), gr, og in zip(mod.named_parameters(), grads, oracle_grads):
        # the reference's own signature of this gradient (sum | abs-sum | norm | first 29 elements) ...
        sig = g[f"{key}_gp_{n}"]
        gf = gr.double().reshape(-1).cpu()
        scale = sig[1] / gf.numel() + 1e-12            # mean |grad| of the reference
        got_first = gf[:29].numpy()
        assert np.abs(got_first - sig[3:3 + len(got_first)]).max() <= 6e-2 * max(np.abs(sig[3:]).max(), scale) + 0.15 * scale, f"{key} grad {n}"
        assert abs(gf.norm().item() - sig[2]) <= 5e-2 * sig[2] + 1e-6, f"{key} grad-norm {n}: {gf.norm().item()} vs {sig[2]}"
        # ... and EVERY element against the fp32 oracle of the block (oracle/v5net.py, itself pinned by the same signatures)
        of = og.double().reshape(-1)
        assert abs(of.norm().item() - sig[2]) <= 1e-4 * sig[2] + 1e-9, f"{key}: oracle gradient of {n} is not the reference's"
        err = (gf - of).abs()
        lim = 6e-2 * of.abs() + 0.25 * scale + 2e-2 * of.abs().max()
        # (SPPF: the arg-max routing of the pools differs on bf16 near-ties, as for gx above)
        assert (err > lim).sum().item() <= (0.01 if key == "sppf" else 0.0) * err.numel(), \
            f"{key} grad {n}: {(err > lim).sum().item()} / {err.numel()} elements off (max {err.max().item():.3g})"
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


# ---------------------------------------------------------------------------------------------------------------------
# round 2: YOLOv5 m / l / x (BASELINE configs #4, #5) and full-model backward against tests/golden/g11_round2.npz
# (reference run by tools/gen_golden.py::gen_g11).
#
# Conditioning.  A TRAIN-mode pass through 100-170 BatchNorm+SiLU layers at random init amplifies bf16 storage rounding
# strongly: the reference ITSELF under torch bf16 autocast is 2-18 % of the elements outside a 6 % band in the forward
# (m / l / x, 256x256) and 35-45 % RMS off in every parameter gradient of the backward, for any output gradient, batch
# or image size (measured when the fixture was made; even rounding only the forward tensors gives 30 %).  Element-wise
# parity of the bf16 path is therefore pinned where it is well conditioned — blocks (test_block_golden), eval-mode
# forwards (running statistics), shallow-to-mid running statistics — and the full-graph train-mode checks use the
# reference's own bf16 deviation, stored in the fixture as `*_cal*`, as the bar: the HIP path must be no further from the
# fp32 reference than that, and bit-consistent between its two backward schedules.

def _dev_stats(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    lim = 0.06 * np.abs(b).max() + 0.06 * np.abs(b)
    return float((np.abs(a - b) > lim).mean()), float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


def test_model_mlx_state_dict_and_init():
    """CPU-side: same keys and bit-identical seeded init as the reference for the m / l / x variants"""
    from yoloseries_amd import models
    g = np.load(os.path.join(G, "g11_round2.npz"))
    for name, cls in (("m", models.YOLOV5Middle), ("l", models.YOLOV5Large), ("x", models.YOLOV5XLarge)):
        torch.manual_seed(0)
        sd = cls(3, 80).state_dict()
        assert list(sd.keys()) == list(g[f"{name}_keys"])
        np.testing.assert_allclose([v.double().sum().item() for v in sd.values()], g[f"{name}_psum"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose([v.double().abs().sum().item() for v in sd.values()], g[f"{name}_pabs"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", ["m", "l", "x"])
def test_model_mlx_forward_golden(dev, name):
    """widths / depths of models/normal/yolov5{m,l,x}.py: eval forward at 64x64 (folded BN, RandomState-filled state) and
    train forward at 256x256 (batch statistics, running-stat update, seeded default init)"""
    from yoloseries_amd import models
    g = np.load(os.path.join(G, "g11_round2.npz"))
    cls = {"m": models.YOLOV5Middle, "l": models.YOLOV5Large, "x": models.YOLOV5XLarge}[name]
    seed = int(g[f"{name}_seed"][0])
    torch.manual_seed(0)
    m = cls(3, 80).to(dev).train()
    x2 = torch.from_numpy(np.random.RandomState(seed + 11).rand(2, 3, 256, 256).astype(np.float32)).to(dev)
    outs = m(x2)
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(g[f"{name}_train256_shape{i}"])
        flat = o.detach().float().contiguous().cpu().numpy().reshape(-1)
        frac, rms = _dev_stats(flat[g[f"{name}_train256_idx{i}"]], g[f"{name}_train256_val{i}"])
        cfrac, crms = g[f"{name}_train256_cal{i}"]
        print(f"v5{name} train256 out{i}: outside the 6% band {frac:.4f} (reference under bf16 autocast {cfrac:.4f}), rel rms {rms:.4f} ({crms:.4f})")
        assert frac <= 1.25 * cfrac + 0.01 and rms <= 1.25 * crms + 0.01, f"v5{name} train out{i}: {frac:.4f}/{rms:.4f} vs calibration {cfrac:.4f}/{crms:.4f}"
    # running statistics after one training forward: shallow layers tight, deep ones with the calibrated slack
    mods = dict(m.named_modules())
    for pn, tol in (("focus", 2e-2), ("backbone_stage2_conv", 2e-2), ("backbone_stage4_conv", 3e-2), ("head_stage4_bscp.cba3", 6e-2)):
        _close(mods[pn].bn.running_mean.cpu().numpy(), g[f"{name}_rm_{pn}"], tol, f"v5{name} {pn} running_mean", outlier_frac=0.01)
        _close(mods[pn].bn.running_var.cpu().numpy(), g[f"{name}_rv_{pn}"], tol, f"v5{name} {pn} running_var", outlier_frac=0.01)
    m2 = cls(3, 80)
    assert len(m2.state_dict()) == len(g[f"{name}_keys"])
    fill_state(m2, seed)
    m2 = m2.to(dev).eval()
    x = torch.from_numpy(np.random.RandomState(seed + 10).rand(2, 3, 64, 64).astype(np.float32)).to(dev)
    with torch.no_grad():
        outs = m2(x)
    for i, o in enumerate(outs):
        ref = g[f"{name}_eval64_out{i}"]
        assert tuple(o.shape) == ref.shape
        _close(o.float().cpu().numpy(), ref, 3e-2, f"v5{name} eval out{i}", outlier_frac=0.002)


def _backward_check(dev, key, make, x_np, outs_of, fixture="g11_round2.npz"):
    """gradients of every parameter and of the input from a fixed output gradient, under both backward schedules
    (weight gradients on the side stream / everything on one stream)"""
    g = np.load(os.path.join(G, fixture))
    torch.manual_seed(0)
    model = make().to(dev).train()
    names = [n for n, _ in model.named_parameters()]
    assert names == list(g[f"{key}_pnames"])
    sig, samp, cal = g[f"{key}_psig"], g[f"{key}_psamp"], g[f"{key}_pcal"]
    results = {}
    for streams in (1, 0):
        # the image gradient is not produced on the HIP path (training never uses it: the stem has no data gradient,
        # DESIGN.md section 3), so only the parameter gradients are compared
        xt = torch.from_numpy(x_np.copy()).to(dev)
        outs = outs_of(model(xt))
        for prog in model._yh_state()['progs'].values():
            if prog.bwd_ready:
                prog.two_streams = bool(streams)
        r = np.random.RandomState(1200 + len(key))
        gos = [torch.from_numpy((r.randn(*o.shape) * 0.1).astype(np.float32)).to(dev) for o in outs]
        for i, go in enumerate(gos):
            assert tuple(go.shape) == tuple(g[f"{key}_gout_shape{i}"])
        grads = torch.autograd.grad(outs, list(model.parameters()), gos)
        prog = next(iter(model._yh_state()['progs'].values()))
        assert prog.bwd_ready and prog.two_streams == bool(streams)
        torch.cuda.synchronize()
        results[streams] = [gr.detach().double().cpu().numpy().reshape(-1) for gr in grads]
    # (a) the two schedules compute the same thing: identical up to the fp32 atomics of the weight-gradient reduction
    for a, b2 in zip(results[1], results[0]):
        assert np.abs(a - b2).max() <= 2e-3 * (np.abs(a).max() + 1e-30)
    # (b) against the fp32 reference, with the reference's own bf16 deviation as the bar (see the note above).  A fixture with
    # `_pcorr` (g14: YOLOv5l) is one where the reference's OWN bf16-autocast backward decorrelates from its fp32 run (training-mode
    # BatchNorm through ~100 layers from a random init: sampled rel. rms 1.3, correlation 0.1) and a parameter's deviation is a
    # random draw that changes from run to run with the order of the fp32 atomics (one bn.bias: 0.22 in one run, 0.44 in the next):
    # there the bars are on the DISTRIBUTION over the 499 parameters — median, 95th percentile and maximum of the gradient-norm
    # deviation against the same statistics of the reference's bf16 run — plus a gross-error bound per parameter; the
    # well-conditioned per-parameter check of the same model is test_full_graph_gradients_eval_mode_bn[v5l_frozen]
    distributional = f"{key}_pcorr" in g.files
    gr = [None] + results[1]
    dev_norm, dev_samp, corr = [], [], []
    for pi, n in enumerate(names):
        gf = gr[1 + pi]
        rsum, rabs, rnorm, numel = sig[pi]
        assert gf.size == int(numel), n
        si = np.random.RandomState(1400 + pi).randint(0, gf.size, 64)
        dn = abs(float(np.sqrt((gf ** 2).sum())) - rnorm) / (rnorm + 1e-30)
        ds = float(np.sqrt(((gf[si] - samp[pi]) ** 2).mean()) / (np.sqrt((samp[pi] ** 2).mean()) + 1e-30))
        dev_norm.append(dn); dev_samp.append(ds)
        if gf.size >= 64 and np.std(samp[pi]) > 0:
            corr.append(float(np.corrcoef(gf[si], samp[pi])[0, 1]))
        assert np.isfinite(gf).all(), n
        # no parameter is grossly wrong (a dropped / doubled contribution shows as O(1) here)
        if distributional:
            assert dn <= 0.75, f"{key} grad-norm {n}: {dn:.3f} (calibration {cal[pi, 0]:.3f})"
            continue
        assert dn <= max(3.0 * cal[pi, 0], 0.0) + 0.25, f"{key} grad-norm {n}: {dn:.3f} (calibration {cal[pi, 0]:.3f})"
        assert ds <= 1.5 * cal[pi, 1] + 0.15, f"{key} grad samples {n}: rel rms {ds:.3f} (calibration {cal[pi, 1]:.3f})"
    dev_norm, dev_samp = np.array(dev_norm), np.array(dev_samp)
    print(f"{key}: per-parameter |norm| deviation median {np.median(dev_norm):.4f} (cal {np.median(cal[:, 0]):.4f}), sampled rel rms median "
          f"{np.median(dev_samp):.3f} (cal {np.median(cal[:, 1]):.3f}), sample correlation median {np.median(corr):.3f}")
    if distributional:
        # the calibration is ONE draw of the reference's bf16 run; this path's draw depends on the kernels the per-box tuning picks for
        # these shapes (summation order -> bf16 rounding -> amplified): medians of 0.04 and 0.09 were seen on two boxes
        p95, c95 = np.percentile(dev_norm, 95), np.percentile(cal[:, 0], 95)
        print(f"{key}: 95th percentile {p95:.3f} (cal {c95:.3f}), max {dev_norm.max():.3f} (cal {cal[:, 0].max():.3f})")
        assert np.median(dev_norm) <= 3.0 * np.median(cal[:, 0]) + 0.02
        assert p95 <= 2.0 * c95 + 0.05
    else:
        assert np.median(dev_norm) <= 1.25 * np.median(cal[:, 0]) + 0.01
    assert np.median(dev_samp) <= 1.1 * np.median(cal[:, 1]) + 0.02
    # the samples point the same way as the reference's: 0.85, or — a fixture that stores it (g14) — no worse than what the reference's
    # own bf16-autocast run reaches against its fp32 run on this model
    want = 0.85
    if f"{key}_pcorr" in g.files:
        want = min(0.85, float(np.nanmedian(g[f"{key}_pcorr"])) - 0.05)
    assert np.median(corr) >= want, f"{key}: sample correlation median {np.median(corr):.3f} < {want:.3f}"


def test_v5s_full_backward_golden(dev):
    from yoloseries_amd import models
    x = np.random.RandomState(1112).rand(2, 3, 256, 256).astype(np.float32)
    _backward_check(dev, "v5s_bwd", lambda: models.YOLOV5Small(3, 80), x, lambda o: list(o))


def test_v5l_full_backward_golden(dev):
    """BASELINE config #4's model (models/normal/yolov5l.py:16-44; 128 ... 1 024-channel layers, three bottlenecks per C3 and more):
    every parameter gradient of a training-mode step against the reference's (g14, calibrated like the YOLOv5s case)"""
    from yoloseries_amd import models
    x = np.random.RandomState(1412).rand(4, 3, 320, 320).astype(np.float32)
    _backward_check(dev, "v5l_bwd", lambda: models.YOLOV5Large(3, 80), x, lambda o: list(o), fixture="g14_round5.npz")


def test_yolox_full_backward_golden(dev):
    from yoloseries_amd import models
    x = np.random.RandomState(1122).rand(2, 3, 256, 256).astype(np.float32)
    _backward_check(dev, "yolox_bwd", lambda: models.YOLOXSmall(1, 3, 80, 0.01), x, lambda o: list(o.values()))


def test_v5x_1280_eval_through_evaluator(dev):
    """BASELINE config #5 shape at batch 1: YOLOV5XLarge eval forward at 1280x1280 -> decode -> filter -> class-aware NMS
    through YOLOV5Evaluator, against the CPU oracle (V5NetOracle fp32 forward, oracle decode) and, on the device-decoded
    tensor, bit-exact post-processing against oracle.postproc.postprocess_v5"""
    from oracle import postproc as opp
    from oracle.v5net import V5NetOracle
    from yoloseries_amd import models
    from yoloseries_amd.trainer import YOLOV5Evaluator
    from yoloseries_amd.utils.synth import COCO_ANCHORS
    m = models.YOLOV5XLarge(3, 80)
    fill_state(m, 1151)
    # detect biases as the reference initialises them (objectness prior, models/normal/yolov5s.py:47-85), so that only a
    # small share of the 100 800 anchors passes the confidence threshold
    with torch.no_grad():
        for det, s in ((m.detect.detect_small, 8), (m.detect.detect_mid, 16), (m.detect.detect_large, 32)):
            b = det.bias.view(3, -1)
            b[:, 4] = -2.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev).eval()
    x = torch.from_numpy(np.random.RandomState(1152).rand(1, 3, 1280, 1280).astype(np.float32))
    hyp = dict(device=dev, num_class=80, input_img_size=[1280, 1280], iou_threshold=0.2, conf_threshold=0.3, cls_threshold=0.3,
               max_predictions_per_img=300, iou_type="iou", mutil_label=False, agnostic=True, postprocess_bbox=True, wfb=False,
               use_tta=False, half=False, compute_metric_conf_threshold=0.001, compute_metric_iou_threshold=0.65,
               compute_metric_cls_threshold=0.001)
    ev = YOLOV5Evaluator(m, torch.from_numpy(COCO_ANCHORS).to(dev), hyp)
    with torch.no_grad():
        heads = m(x.to(dev))
    assert [tuple(h.shape) for h in heads] == [(1, 255, 160, 160), (1, 255, 80, 80), (1, 255, 40, 40)]
    dec = ev.decode(heads)
    assert tuple(dec.shape) == (1, 100800, 85)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        oheads = V5NetOracle(sd, train=False)(x)
    odec = opp.decode_v5([h.numpy() for h in oheads], COCO_ANCHORS, [8, 16, 32])
    got = dec.cpu().numpy()
    # head logits: bf16 conv chain (~100 layers) against fp32; decoded boxes / probabilities follow
    for i, (h, oh) in enumerate(zip(heads, oheads)):
        _close(h.float().cpu().numpy(), oh.numpy(), 5e-2, f"v5x 1280 head {i}", outlier_frac=0.01)
    err = np.abs(got[..., 4:] - odec[..., 4:])
    assert (err > 0.05).mean() < 0.01, f"decoded probabilities: {(err > 0.05).mean():.4f} of the entries differ by more than 0.05"
    # post-processing on identical (device-decoded) input: bit exact, through both entry points
    # thresholds placed on this random net's score distribution so that a few thousand of the 100 800 anchors become candidates
    obj = np.sort(got[0, :, 4])[::-1]
    conf_thr = float(obj[4000])
    cls_thr = 0.25 * conf_thr
    ev.conf_threshold, ev.cls_threshold = conf_thr, cls_thr
    res_nms = ev.numba_nms(dec)
    res_fused = ev._nms_from_heads(heads)
    ref = opp.postprocess_v5(got, conf_thr, cls_thr, 0.2)
    assert ref[0] is not None and len(ref[0]) > 0, "test input produced no detections"
    for a in (res_nms[0], res_fused[0]):
        np.testing.assert_array_equal(a, ref[0])
    assert ev.last_ncand[0] >= 1000
    out = ev(x.to(dev))
    np.testing.assert_array_equal(out[0].numpy(), ref[0])


def test_eval_inplace_concat_plan(dev, monkeypatch):
    """inference: the last bottleneck of every C3 block writes over cba1's half of the concat buffer and cba3 reads ONE
    segment (engine._eval_concat_plan) — same predictions as the two-buffer form (YH_EVAL_INPLACE_CAT=0), and a training
    forward is unaffected (the training program keeps both buffers)"""
    from yoloseries_amd import engine, models
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(7)).to(dev)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("YH_EVAL_INPLACE_CAT", flag)
        torch.manual_seed(0)
        m = models.YOLOV5Small(3, 80).to(dev)
        m.train()
        tr0 = [o.detach().float().cpu() for o in m(x)]
        m.eval()
        with torch.no_grad():
            ev = [o.float().cpu() for o in m(x)]
        prog = m._yh_program(2, 128, 128)
        po, ps = prog._eval_concat_plan()
        nc3 = sum(1 for o in prog.ops if isinstance(o, engine.ConvOp) and o.name.endswith("cba3"))
        assert (len(ps) == nc3 and len(po) == nc3 and nc3 == 8) if flag == "1" else (not po and not ps)
        cba3 = [c for _, args, name, _ in prog.cmd_eval if name.endswith("cba3") for c in args]
        assert len(cba3) == 8 and all(d.nseg == (1 if flag == "1" else 2) for d in cba3)
        outs[flag] = (tr0, ev)
    for a, b in zip(outs["1"][0], outs["0"][0]):
        assert torch.equal(a, b)                            # training forward: untouched
    for a, b in zip(outs["1"][1], outs["0"][1]):
        d = (a - b).abs()
        assert d.max() <= 3e-2 * max(1.0, b.abs().max().item()) and d.mean() <= 2e-3 * max(1.0, b.abs().mean().item())


def test_program_executor_matches_per_launch_calls(dev):
    """the compiled command arrays replayed by yh_exec (one call per pass) against one ctypes call per launch: identical forward
    outputs (same kernels, same order), gradients equal up to the atomics' summation order"""
    from yoloseries_amd import engine, models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    import bench
    res = {}
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(5)).to(dev)
    t = torch.from_numpy(synth_targets(2, 128, 80, 8, seed=6, min_boxes=4)).to(dev)
    for use in (False, True):
        old = engine.flags.USE_EXEC
        engine.flags.USE_EXEC = use
        try:
            torch.manual_seed(0)
            m = models.YOLOV5Small(3, 80).to(dev).train()
            outs = m(x)
            YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, 128, 2))(outs, t)["tot_loss"].backward()
            g = m._yh_last_flat_grad.clone()
            m.eval()
            with torch.no_grad():
                ev = [o.float().cpu() for o in m(x)]
            res[use] = ([o.detach().float().cpu() for o in outs], g.cpu(), ev)
        finally:
            engine.flags.USE_EXEC = old
    (o0, g0, e0), (o1, g1, e1) = res[False], res[True]
    for a, b in zip(o0 + e0, o1 + e1):
        assert torch.equal(a, b)
    assert (g1 - g0).abs().max() <= 1e-4 * g0.abs().max()


@pytest.mark.parametrize("name", ["v5s", "v5l", "yolox"])
def test_step_deterministic_parts_bit_identical_at_judged_shape(dev, name):
    """Forward and the backward's main chain are deterministic (fixed-order reductions; only the weight gradients use fp32 atomics):
    at the judged shape (batch 64, 640 x 640, the shipped launch parameters) five forward + backward passes of one model on one
    input give bit-identical head outputs, loss and BatchNorm weight / bias gradients — those depend on every data-gradient, every
    BatchNorm-backward pass and every fused reduction of the chain, so a synchronisation error anywhere in it shows here on any
    box (round 5: conv_pt_kernel's fused reduction differed in ~1 of 60 launches on some boxes)."""
    from yoloseries_amd import models
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    import bench
    B, img = 64, 640
    torch.manual_seed(0)
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(11)).to(dev)
    t = torch.from_numpy(synth_targets(B, img, 80, 12, seed=12, min_boxes=2)).to(dev)
    if name in ("v5s", "v5l"):
        from yoloseries_amd.loss import YOLOV5Loss
        m = (models.YOLOV5Small if name == "v5s" else models.YOLOV5Large)(3, 80).to(dev).train()
        # (a fresh loss object per pass: YOLOV5Loss carries the running `balances` of the stages from call to call)
        make_loss = lambda: YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))      # noqa: E731
    else:
        from yoloseries_amd.loss import YOLOXLoss
        m = models.YOLOXSmall(1, 3, 80).to(dev).train()
        make_loss = lambda: YOLOXLoss(dict(device=dev, num_class=80, input_img_size=[img, img], batch_size=B, use_focal_loss=False, focal_loss_gamma=1.5,
                               focal_loss_alpha=0.25, iou_loss_scale=5.0, use_l1=True, l1_loss_scale=1.0, cls_loss_scale=1.0,
                               cof_loss_scale=1.0, class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0, num_anchors=1,
                               iou_type="ciou", topk=13, center_radius=3, num_stage=3, loss_items_on_device=True))      # noqa: E731
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
    assert len(bns) >= 50
    first = None
    for r in range(5):
        for p_ in m.parameters():
            p_.grad = None
        outs = m(x)
        loss = make_loss()(outs, t.clone())["tot_loss"]
        loss.backward()
        torch.cuda.synchronize()
        flat = []

        def walk(o):
            if torch.is_tensor(o):
                flat.append(o.detach().clone())
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
        walk(outs)
        assert len(flat) >= 3
        cur = flat + [loss.detach().clone()] + [b.weight.grad.clone() for b in bns] + [b.bias.grad.clone() for b in bns]
        if first is None:
            first = cur
            assert all(torch.isfinite(c.float()).all() for c in cur)
        else:
            bad = [i for i, (a, b) in enumerate(zip(first, cur)) if not torch.equal(a, b)]
            assert not bad, f"pass {r}: tensors {bad[:6]} of {len(cur)} differ from the first pass (0..{len(flat) - 1} outputs, {len(flat)} loss, then BatchNorm gradients)"


@pytest.mark.parametrize("key", ["v5s_frozen", "yolox_frozen", "v5l_frozen"])
def test_full_graph_gradients_eval_mode_bn(dev, key):
    """A WELL-CONDITIONED check of the whole backward graph: the model in evaluation mode under autograd (BatchNorm on its
    running statistics: no batch coupling, so bf16 rounding is not amplified through ~60 batch normalisations) against the
    reference's gradients (g12_round3.npz).  Every parameter: gradient norm within 2 %, 256 sampled elements within
    3e-2 of the parameter's largest gradient element — a mis-scaled branch of a concat / upsample / residual / stacked GEMM
    gradient (5-10 %) cannot pass.  Under both backward schedules."""
    from yoloseries_amd import models
    g = np.load(os.path.join(G, "g14_round5.npz" if key == "v5l_frozen" else "g12_round3.npz"))
    seed = int(g[f"{key}_seed"][0])
    if key == "v5s_frozen":
        model, xs, outs_of = models.YOLOV5Small(3, 80), 1201, (lambda o: list(o))
    elif key == "v5l_frozen":            # config #4's model (g14, round 5): 499 parameters, 96 sampled elements each
        model, xs, outs_of = models.YOLOV5Large(3, 80), 1401, (lambda o: list(o))
    else:
        model, xs, outs_of = models.YOLOXSmall(1, 3, 80, 0.01), 1202, (lambda o: list(o.values()))
    fill_state(model, seed)
    rm0 = {n: b.clone() for n, b in model.named_buffers()}
    model = model.to(dev).eval()
    names = [n for n, _ in model.named_parameters()]
    assert names == [str(n) for n in g[f"{key}_pnames"]]
    x = np.random.RandomState(xs).rand(2, 3, 256, 256).astype(np.float32)
    for streams in (1, 0):
        outs = outs_of(model(torch.from_numpy(x).to(dev)))
        for prog in model._yh_state()['progs'].values():
            if prog.bwd_ready:
                prog.two_streams = bool(streams)
        r = np.random.RandomState(seed + 1)
        gos = [torch.from_numpy((r.randn(*o.shape) * 0.1).astype(np.float32)).to(dev) for o in outs]
        for i, o in enumerate(outs):
            assert tuple(o.shape) == tuple(g[f"{key}_out_shape{i}"])
            flat = o.detach().float().contiguous().cpu().numpy().reshape(-1)
            _close(flat[g[f"{key}_out_idx{i}"]], g[f"{key}_out_val{i}"], 3e-2, f"{key} eval-mode out{i}")
        grads = torch.autograd.grad(outs, list(model.parameters()), gos)
        worst, bad = {}, []
        for pi, (n, gr) in enumerate(zip(names, grads)):
            gf = gr.double().reshape(-1).cpu().numpy()
            s_sum, s_abs, s_norm, s_size, s_max = g[f"{key}_psig"][pi]
            assert gf.size == int(s_size)
            nrm = np.sqrt((gf ** 2).sum())
            err = np.abs(gf[g[f"{key}_pidx"][pi]] - g[f"{key}_pval"][pi]).max()
            # bar: 3e-2 of the parameter's largest gradient element / 2e-2 of its norm — or, where the REFERENCE ITSELF under torch
            # bf16 autocast is further from its fp32 run than that (`pcal`: the BatchNorm affine gradients, sums of dz (* xhat) over
            # every pixel with heavy cancellation, and the deepest 8 x 8 stage), 1.5 x the reference's own deviation
            kind = "bn" if ".bn." in n else "w"
            cal_e, cal_n = g[f"{key}_pcal"][pi]
            tol_e, tol_n = max(3e-2, 1.5 * cal_e), max(2e-2, 1.5 * cal_n)
            if kind == "bn":                      # per-channel sums over all pixels: wider floor (the conv weights carry the wiring check)
                tol_e, tol_n = max(8e-2, 2.0 * cal_e), max(5e-2, 2.0 * cal_n)
            elif key == "v5l_frozen":             # twice the depth of YOLOv5s in front of the 8 x 8 stage (128 pixels per weight-gradient sum): its
                tol_n = max(2.5e-2, 2.0 * cal_n)  # 512 -> 512 bottleneck conv, the parameter the reference's own bf16 run is furthest off on (1.1 %), sits at 2.2 %
            worst[kind] = max(worst.get(kind, 0.0), err / (s_max + 1e-30))
            if err > tol_e * s_max + 1e-7 or abs(nrm - s_norm) > tol_n * s_norm + 1e-7:
                bad.append((n, round(err / (s_max + 1e-30), 4), round(abs(nrm - s_norm) / (s_norm + 1e-30), 4)))
        print(f"{key} streams={streams}: worst sampled error / largest element {worst}")
        assert not bad, f"{key}: {len(bad)} of {len(names)} parameter gradients off (name, element error, norm error): {bad[:8]}"
    # evaluation mode: the running statistics did not move
    for n, b in model.named_buffers():
        assert torch.equal(b.cpu(), rm0[n]), n


def test_forward_fuse_deployment_flow(dev):
    """the reference's deployment fusion (detect_yolov5.py:110-116): `m.conv = fuse_conv_bn(m.conv, m.bn); delattr(m, 'bn');
    m.forward = m.forward_fuse` on every ConvBnAct — stand-alone block (utils/layer_tools.py:93-94) and whole YOLOv5s — gives the
    predictions of the unfused evaluation model"""
    from yoloseries_amd import models
    from yoloseries_amd.utils.layer_tools import ConvBnAct, fuse_conv_bn

    def fuse_all(model):
        for m in model.modules():
            if isinstance(m, ConvBnAct) and hasattr(m, 'bn'):
                m.conv = fuse_conv_bn(m.conv, m.bn)
                delattr(m, 'bn')
                m.forward = m.forward_fuse
    cb = ConvBnAct(32, 64, 3, 1, 1)
    fill_state(cb, 1501)
    cb = cb.to(dev).eval()
    x = torch.from_numpy(np.random.RandomState(1502).randn(2, 32, 16, 16).astype(np.float32)).to(dev)
    with pytest.raises(RuntimeError):
        cb.forward_fuse(x)                                  # not fused yet
    with torch.no_grad():
        y0 = cb(x).float().cpu()
        ref = torch.nn.functional.silu(torch.nn.functional.batch_norm(
            torch.nn.functional.conv2d(x, cb.conv.weight, None, 1, 1), cb.bn.running_mean, cb.bn.running_var, cb.bn.weight, cb.bn.bias, False, 0.0, cb.bn.eps)).cpu()
        fuse_all(cb)
        assert not hasattr(cb, 'bn') and cb.conv.bias is not None and 'bn.weight' not in cb.state_dict()
        y1 = cb(x).float().cpu()
    _close(y0.numpy(), ref.numpy(), 3e-2, "ConvBnAct eval vs torch")
    _close(y1.numpy(), ref.numpy(), 3e-2, "ConvBnAct fused vs torch")
    assert (y1 - y0).abs().max() <= 2e-2 * max(1.0, ref.abs().max().item())
    m = models.YOLOV5Small(3, 80)
    fill_state(m, 1503)
    m = m.to(dev).eval()
    xi = torch.from_numpy(np.random.RandomState(1504).rand(2, 3, 64, 64).astype(np.float32)).to(dev)
    with torch.no_grad():
        o0 = [o.float().cpu() for o in m(xi)]
        fuse_all(m)
        assert not any(k.endswith("bn.weight") for k in m.state_dict())
        o1 = [o.float().cpu() for o in m(xi)]
    for a, b in zip(o1, o0):
        d = (a - b).abs()
        assert d.max() <= 3e-2 * max(1.0, b.abs().max().item()) and d.mean() <= 3e-3 * max(1.0, b.abs().mean().item())


def test_head_wider_than_256_channels_two_streams_vs_serial(dev):
    """81 classes: the head layers have 3 * 86 = 258 output channels (row pitch 264 > 256), so the partial sums of their bias
    gradients — on the weight-gradient stream in the two-stream backward — need the scratch sized from the widest head
    (engine._head_scratch; ADVICE r03: a scratch shared with the main stream's BatchNorm reductions raced).  The two schedules must
    give the same gradients (up to the fp32 atomics of the weight-gradient reduction), for every parameter incl. the head biases."""
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    import bench
    nc, B, img = 81, 2, 128
    torch.manual_seed(0)
    model = models.YOLOV5Small(3, nc).to(dev).train()
    hyp = bench.make_hyp(dev, img, B)
    hyp["num_class"] = nc
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(11)).to(dev)
    t = torch.from_numpy(synth_targets(B, img, nc, 8, seed=12, min_boxes=4)).to(dev)
    grads = {}
    for rep in range(3):                                  # three rounds per schedule: a race shows as run-to-run differences
        for streams in (1, 0):
            outs = model(x)
            assert outs[0].shape[1] == 3 * (5 + nc)
            for prog in model._yh_state()['progs'].values():
                if prog.bwd_ready:
                    prog.two_streams = bool(streams)
            YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), hyp)(outs, t)["tot_loss"].backward()
            prog = next(iter(model._yh_state()['progs'].values()))
            assert prog.two_streams == bool(streams) and prog.head_scratch.numel() >= 1024 * 2 * 264
            if streams:
                assert all(prog._head_on_side(o, True) for o in prog.outputs)
            g = model._yh_last_flat_grad.double().cpu()
            assert torch.isfinite(g).all()
            grads.setdefault(streams, []).append(g)
    ref = grads[0][0]
    scale = ref.abs().max().item()
    for streams in (1, 0):
        for g in grads[streams]:
            assert (g - ref).abs().max().item() <= 2e-3 * scale, f"streams={streams}: {(g - ref).abs().max().item() / scale:.2e}"
    # the head biases specifically (column sums of the head gradients: the launches that moved to the side stream)
    off = 0
    for n, p in model.named_parameters():
        if n.startswith("detect.") and n.endswith(".bias"):
            a, b = grads[1][0][off:off + p.numel()], ref[off:off + p.numel()]
            assert a.abs().max() > 0 and (a - b).abs().max() <= 1e-3 * b.abs().max() + 1e-7, n
        off += p.numel()
