"""End-to-end smoke of the drop-in training driver (train_yolov5.py: Training life-cycle of the reference's
train_yolov5.py:49-806) on synthetic data: warm-up + SGD + EMA + checkpoint + evaluation + mAP."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_driver_smoke(dev, tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    monkeypatch.chdir(tmp_path)
    import train_yolov5
    t = train_yolov5.main(["--epochs", "2", "--img", "128", "--batch", "4", "--steps-per-epoch", "4"])
    losses = [h["tot_loss"] for h in t.history]
    assert len(losses) == 8 and all(np.isfinite(losses))
    assert os.path.exists(t.last_ckpt)
    ck = torch.load(t.last_ckpt, map_location="cpu", weights_only=False)
    assert set(["model_state_dict", "optim_state_dict", "ema", "epoch", "step", "hyp"]) <= set(ck.keys())
    assert len(ck["model_state_dict"]) == 348
    assert set(t.last_metrics) == {"map", "map50", "precision", "recall", "n_pred"}
    # resume: a fresh Training restores weights and epoch from the checkpoint
    t2 = train_yolov5.Training(t.anchors, dict(t.hyp, pretrained_model_path=t.last_ckpt, total_epoch=2))
    assert t2.start_epoch == 2
    for (k, a), (_, b2) in zip(t.model.state_dict().items(), t2.model.state_dict().items()):
        assert torch.equal(a.cpu(), b2.cpu()), k
    # ... and the optimizer state: the momentum buffer is loaded before the parameter arena exists and must be in place
    # once the first forward has built it (FlatSGD.load_state_dict defers the copy)
    mb = ck["optim_state_dict"]["momentum_buffer"]
    assert mb is not None and float(mb.abs().sum()) > 0
    assert t2.optimizer.steps == ck["optim_state_dict"]["steps"] > 0
    t2.model.train()
    t2.model(torch.rand(4, 3, 128, 128, device=dev))
    t2.optimizer._pack()
    assert torch.equal(t2.optimizer.buf.cpu(), mb.cpu())


def test_training_driver_dataset_path(dev, tmp_path, monkeypatch):
    """same driver fed through the reference's data path: DataLoader -> fixed_imgsize_collate_fn -> DataPrefetcher"""
    sys.path.insert(0, ROOT)
    monkeypatch.chdir(tmp_path)
    import train_yolov5
    t = train_yolov5.main(["--epochs", "1", "--img", "128", "--batch", "4", "--steps-per-epoch", "3", "--data", "dataset"])
    losses = [h["tot_loss"] for h in t.history]
    assert len(losses) == 3 and all(np.isfinite(losses)) and all(h["tar_nums"] > 0 for h in t.history)
    assert set(t.last_metrics) == {"map", "map50", "precision", "recall", "n_pred"}


def test_validation_driver(dev, tmp_path, monkeypatch):
    """val_yolov5.py mirror: checkpoint from the training driver -> evaluator -> un-letterboxed boxes -> mAP_v2"""
    sys.path.insert(0, ROOT)
    monkeypatch.chdir(tmp_path)
    import train_yolov5
    import val_yolov5
    t = train_yolov5.main(["--epochs", "1", "--img", "128", "--batch", "4", "--steps-per-epoch", "2"])
    v = val_yolov5.main(["--img", "128", "--batch", "4", "--val-batches", "2", "--ckpt", t.last_ckpt])
    assert v.loaded_ema and v.metrics["images"] == 8
    assert all(np.isfinite([v.metrics[k] for k in ("map", "map50", "precision", "recall")]))
    # frame transforms are inverses of the letterbox (utils/letterbox.py)
    info = [{'scale': 0.5, 'pad_top': 10, 'pad_left': 4, 'pad_bottom': 10, 'pad_right': 4, 'org_shape': (200, 240)}]
    pred = [torch.tensor([[14., 20., 64., 70., 0.9, 3.]], device=dev)]
    out = val_yolov5.Training.preds_postprocess(pred, info)[0]
    assert np.allclose(out[0, :4], [20., 20., 120., 120.]) and out[0, 5] == 3
    ann = torch.tensor([[[14., 20., 64., 70., 3., 0.], [-1., -1., -1., -1., -1., -1.]]])
    bb, cc = val_yolov5.Training.gt_bbox_postprocess(ann, info)
    assert np.allclose(bb[0], [[20., 20., 120., 120.]]) and cc[0].tolist() == [3]


def test_training_learns_a_detection_task(dev, tmp_path, monkeypatch):
    """the whole stack on a LEARNABLE synthetic task (coloured rectangles, colour = class; yoloseries_amd/utils/synth.py):
    forward, loss, backward, clip, SGD-nesterov with warm-up, EMA, evaluator (decode + NMS) and mAP_v2 must make mAP rise from
    zero — random-noise batches can only show that the step runs.  300 steps at 320 x 320, batch 16 (a few seconds)."""
    sys.path.insert(0, ROOT)
    monkeypatch.chdir(tmp_path)
    import train_yolov5
    t = train_yolov5.main(["--data", "shapes", "--epochs", "6", "--img", "320", "--batch", "16", "--steps-per-epoch", "50"])
    first = np.mean([h["tot_loss"] for h in t.history[:10]])
    last = np.mean([h["tot_loss"] for h in t.history[-10:]])
    assert last < 0.6 * first, (first, last)
    # (the fp32 atomics of the weight gradients make every run a slightly different trajectory: observed mAP50 0.25-0.45, recall 0.24-0.6)
    assert t.last_metrics["map50"] > 0.08 and t.last_metrics["recall"] > 0.12, t.last_metrics


def test_yolox_training_and_validation_drivers(dev, tmp_path, monkeypatch):
    """train_yolox.py / val_yolox.py mirrors (the reference's YOLOX drivers differ from the v5 ones in the model table, loss /
    evaluator constructors, log line, checkpoint names and config file)"""
    sys.path.insert(0, ROOT)
    monkeypatch.chdir(tmp_path)
    import train_yolox
    import val_yolox
    t = train_yolox.main(["--epochs", "2", "--img", "128", "--batch", "4", "--steps-per-epoch", "4"])
    assert type(t.loss_fcn).__name__ == "YOLOXLoss" and type(t.validate).__name__ == "YOLOXEvaluator"
    losses = [h["tot_loss"] for h in t.history]
    assert len(losses) == 8 and all(np.isfinite(losses)) and all("l1_loss" in h for h in t.history)
    assert os.path.basename(t.last_ckpt) == "yolox_small_epoch_2.pth"
    ck = torch.load(t.last_ckpt, map_location="cpu", weights_only=False)
    assert len(ck["model_state_dict"]) == 414 and ck["hyp"]["topk"] == 13
    assert set(t.last_metrics) == {"map", "map50", "precision", "recall", "n_pred"}
    v = val_yolox.main(["--img", "128", "--batch", "4", "--val-batches", "2", "--ckpt", t.last_ckpt])
    assert v.loaded_ema and v.metrics["images"] == 8
    assert all(np.isfinite([v.metrics[k] for k in ("map", "map50", "precision", "recall")]))
