import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from yoloseries_amd import models
from yoloseries_amd.loss import YOLOV5Loss
from yoloseries_amd.trainer import ExponentialMovingAverageModel
from yoloseries_amd.utils import FlatSGD
from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
dev = torch.device('cuda:0')
for B in (64, 16):
    torch.manual_seed(0)
    m = models.YOLOV5Small(3, 80).to(dev).train()
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, 640, B))
    opt = FlatSGD(m, lr=0.01, momentum=0.937, weight_decay=1e-4, nesterov=True)
    ema = ExponentialMovingAverageModel(m)
    x = torch.rand(B, 3, 640, 640, device=dev)
    t = torch.from_numpy(synth_targets(B, 640, 80, 20, seed=1)).to(dev)
    def step():
        out = lossf(m(x), t); out['tot_loss'].backward(); opt.clip_grad_norm_(10.0); opt.step(); opt.zero_grad(); ema.update(m)
    for _ in range(5): step()
    torch.cuda.synchronize()
    hs = []
    t0 = time.perf_counter()
    for _ in range(10):
        h0 = time.perf_counter(); step(); hs.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 10
    print(f"B={B}: step wall {tot*1e3:.2f} ms, host enqueue per step {sum(hs)/len(hs)*1e3:.2f} ms (min {min(hs)*1e3:.2f})")
