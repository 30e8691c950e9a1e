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
