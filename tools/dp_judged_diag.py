#!/usr/bin/env python3
"""Diagnostic of tests/test_gpu_dist.py::test_overlapped_bucket_allreduce_two_ranks_judged_shape: two ranks on one GPU over gloo, YOLOv5l at
64 x 640 x 640; per trial: the exchanged gradient against the mean of the local ones, and WHERE they differ (parameter names).
usage: dp_judged_diag.py [trials] [small|large] [batch] [img]"""
import os
import socket
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, q, size, B, img, trials):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.utils import FlatSGD
    from yoloseries_amd.utils.dist import DataParallelGrads
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    import bench
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = {"small": models.YOLOV5Small, "large": models.YOLOV5Large}[size](3, 80).to(dev).train()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=0.0, nesterov=True)
    dp = DataParallelGrads(model)
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(10 + rank)).to(dev)
    t = torch.from_numpy(synth_targets(B, img, 80, 6, seed=20 + rank)).to(dev)

    def fresh_loss():
        return YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
    names, offs, o = [], [], 0
    for n, p in model.named_parameters():
        names.append(n); offs.append(o); o += p.numel()
    offs.append(o)
    import bisect
    out = []
    for trial in range(trials):
        locs, heads, losses, hgrads, bals, chk, rechk = [], [], [], [], [], [], []
        for rep in range(2):                       # two local passes: how reproducible is a pass by itself?
            with dp.no_sync():
                outs = model(x)
                heads.append([o.detach().clone() for o in outs])
                hg = [None] * len(outs)
                for hi, o in enumerate(outs):
                    o.register_hook(lambda g, hi=hi, hg=hg: hg.__setitem__(hi, g.detach().clone()))
                lf = fresh_loss()
                ls = lf(outs, t)["tot_loss"]
                losses.append(ls.detach().clone())
                ls.backward()
                hgrads.append(hg)
                bals.append(lf.balances)
                # did the head tensors change during the pass?  does the loss alone, run again on the SAME head tensors, reproduce the head gradients?
                after = [o.detach().clone() for o in outs]
                chk.append(all(torch.equal(a, b) for a, b in zip(after, heads[-1])))
                re_in = [o.detach().clone().requires_grad_(True) for o in outs]
                fresh_loss()(re_in, t)["tot_loss"].backward()
                rechk.append([bool(torch.equal(r.grad, h)) for r, h in zip(re_in, hg)])
            locs.append(model._yh_last_flat_grad.clone())
            opt.zero_grad(); dp.reset()
        head_same = all(torch.equal(a, b) for a, b in zip(heads[0], heads[1]))
        loss_same = bool(torch.equal(losses[0], losses[1]))
        hg_info = []
        for a, b in zip(hgrads[0], hgrads[1]):
            a32, b32 = a.float(), b.float()
            hg_info.append((bool(torch.equal(a, b)), float((a32 - b32).abs().max().item()), float(a32.abs().max().item()),
                            float((b32.norm() / a32.norm().clamp_min(1e-30)).item())))
        hg_info.append(("heads unchanged during pass", chk, "loss re-run on the same heads reproduces the hooked head gradients", rechk))
        fresh_loss()(model(x), t)["tot_loss"].backward()
        g_dp = model._yh_last_flat_grad.clone()
        opt.zero_grad(); dp.reset()
        gathered = [torch.zeros_like(locs[0]).cpu() for _ in range(world)]
        dist.all_gather(gathered, locs[0].cpu())
        g_mean = (sum(gathered) / world).to(dev)
        scale = g_mean.abs().max().item()
        err = (g_dp - g_mean).abs()
        self_err = (locs[0] - locs[1]).abs()
        top = torch.topk(err, 5)
        where = [(names[bisect.bisect_right(offs, int(i)) - 1], float(v) / scale) for v, i in zip(top.values, top.indices)]
        tops = torch.topk(self_err, 3)
        wheres = [(names[bisect.bisect_right(offs, int(i)) - 1], float(v) / scale) for v, i in zip(tops.values, tops.indices)]
        # parameters (forward order) whose gradient differs between the two LOCAL passes by more than 1e-4 of their own largest element
        bad = []
        for i, n in enumerate(names):
            a, b = locs[0][offs[i]:offs[i + 1]], locs[1][offs[i]:offs[i + 1]]
            m = max(a.abs().max().item(), 1e-30)
            e = (a - b).abs().max().item() / m
            if e > 1e-4:
                bad.append((i, n, e))
        out.append((trial, float(err.max().item() / scale), int((err > 1e-3 * scale).sum()), where, float(self_err.max().item() / scale), wheres,
                    head_same, loss_same, len(names), bad, hg_info))
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    size = sys.argv[2] if len(sys.argv) > 2 else "large"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    img = int(sys.argv[4]) if len(sys.argv) > 4 else 640
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = int(os.environ.get("DIAG_WORLD", "2"))
    procs = [ctx.Process(target=worker, args=(r, world, port, q, size, B, img, trials)) for r in range(world)]
    for p in procs:
        p.start()
    for _ in range(world):
        rank, out = q.get(timeout=900)
        for trial, e, n, where, se, wheres, hs, lsame, npar, bad, hg_info in out:
            print(f"rank {rank} trial {trial}: exchanged vs mean(local) {e:.2e} ({n} elements > 1e-3) | local vs local {se:.2e}; heads bit-equal {hs}, loss bit-equal {lsame}; "
                  f"{len(bad)} of {npar} parameters differ > 1e-4 (own scale)", flush=True)
            if bad or not all(h[0] for h in hg_info[:-1]):
                print("     head gradients (bit-equal, max |diff|, max |g|, norm ratio) per stage:", hg_info, flush=True)
            if bad:
                print("     last (deepest) differing:", [(i, nm, f"{e2:.1e}") for i, nm, e2 in bad[-6:]], flush=True)
                print("     first differing:", [(i, nm, f"{e2:.1e}") for i, nm, e2 in bad[:4]], flush=True)
    for p in procs:
        p.join(timeout=60)
