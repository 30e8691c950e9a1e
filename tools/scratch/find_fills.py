import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, traceback
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss
from yoloseries_amd.utils import FlatSGD
from yoloseries_amd.trainer.ema_model import ExponentialMovingAverageModel
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
import bench
dev = torch.device("cuda:0")
B, img = 64, 640
model = models.YOLOV5Small(3, 80).to(dev).train()
lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
ema = ExponentialMovingAverageModel(model)
x = torch.rand(B, 3, img, img, device=dev)
t = torch.from_numpy(synth_targets(B, img, 80, 6, seed=1)).to(dev)
def step():
    out = lossf(model(x), t); out["tot_loss"].backward(); opt.clip_grad_norm_(10.0); opt.step(); opt.zero_grad(); ema.update(model)
for _ in range(3): step()
torch.cuda.synchronize()
seen = {}
orig_zeros, orig_zero_, orig_fill_ = torch.zeros, torch.Tensor.zero_, torch.Tensor.fill_
def note(kind, numel, dtype):
    st = traceback.extract_stack(limit=6)[:-2]
    key = (kind, int(numel), str(dtype), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st)))
    seen[key] = seen.get(key, 0) + 1
def zeros(*a, **k):
    r = orig_zeros(*a, **k); note("zeros", r.numel() * r.element_size(), r.dtype); return r
def zero_(self):
    note("zero_", self.numel() * self.element_size(), self.dtype); return orig_zero_(self)
def fill_(self, v):
    note("fill_", self.numel() * self.element_size(), self.dtype); return orig_fill_(self, v)
torch.zeros, torch.Tensor.zero_, torch.Tensor.fill_ = zeros, zero_, fill_
step()
torch.zeros, torch.Tensor.zero_, torch.Tensor.fill_ = orig_zeros, orig_zero_, orig_fill_
for k, n in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[1]):
    print(n, k)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.key_averages(group_by_input_shape=True):
    if any(s in e.key for s in ("fill", "zero", "copy_", "empty_like", "clone")):
        print(e.key, e.count, e.input_shapes, round(e.device_time_total, 1))
