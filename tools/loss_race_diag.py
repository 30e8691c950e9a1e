#!/usr/bin/env python3
"""Is the YOLOv5 loss backward reproducible when another PROCESS shares the GPU?  (Round 6: it was not -- packed fp32
instructions of the -O3 build gave a wrong high half about once in 10^5 evaluations, only with a second process on the GPU;
csrc/Makefile EXACT, profiles/r06_step_experiments.txt (l).)  YOLOv5s, batch 64, 640 x 640, bf16 heads, fixed targets.
usage: loss_race_diag.py [procs] [iters] [mode]
  random          every process: fixed random heads, ITERS x (loss forward + backward), head gradients against iteration 0
  model_detached  real heads of a forward pass, the loss on detached leaves (no engine backward)
  model_full      whole training passes, head gradients caught by tensor hooks; also checks that the loss's inputs at backward
                  time are those of its forward
  bwd_repeat      rank 0: ONE forward, then 20 x ITERS loss backwards on the same inputs, histogram of distinct results;
                  the other ranks run whole passes as load (SAME_PROCESS_LOAD=1 with procs=1: matmuls of the same process
                  on a second stream instead)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.multiprocessing as mp


def worker(rank, iters, q, mode="random", done=None):
    import bench
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.layout import cell_major_view
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    dev = torch.device("cuda", 0)
    B, img = 64, 640
    g = torch.Generator().manual_seed(5)
    t = torch.from_numpy(synth_targets(B, img, 80, 20, seed=1)).to(dev)
    bufs = [(torch.randn(B, img // s, img // s, 256, generator=g) * 2).to(torch.bfloat16).to(dev) for s in (8, 16, 32)]
    busy = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    first, bad, first_saved = None, [], None
    detail = []
    model = None
    chk = {"saved": 0, "pred": 0, "n": 0}
    if mode == "model_full":
        # are the loss's inputs at BACKWARD time what they were at the end of its forward?
        from yoloseries_amd.loss import yolov5_loss as yl
        f0, b0 = yl._V5LossFn.forward, yl._V5LossFn.backward
        def fwd(ctx, owner, targets, *preds):
            r = f0(ctx, owner, targets, *preds)
            ctx.saved0 = ctx.saved.clone()
            ctx.canon0 = [c.clone() for c in ctx.canon]
            return r
        def bwd(ctx, *a):
            chk["n"] += 1
            chk["saved"] += int(not torch.equal(ctx.saved0, ctx.saved))
            chk["pred"] += sum(int(not torch.equal(c0, c)) for c0, c in zip(ctx.canon0, ctx.canon))
            return b0(ctx, *a)
        yl._V5LossFn.forward = staticmethod(fwd)
        yl._V5LossFn.backward = staticmethod(bwd)
    if mode != "random":
        from yoloseries_amd import models
        torch.manual_seed(0)
        model = models.YOLOV5Small(3, 80).to(dev).train()
        x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(10)).to(dev)
    if mode == "bwd_repeat":
        # rank 0: ONE forward, then the loss backward alone again and again on the same inputs; the other ranks: whole training passes as load
        if rank == 0:
            outs = model(x)
            preds = [o.detach().clone().requires_grad_(True) for o in outs]
            lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
            out = lf(preds, t)
            hist = {}
            spl = os.environ.get("SAME_PROCESS_LOAD", "")
            side = torch.cuda.Stream(dev) if spl and not spl.startswith("family:") else None
            ma = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16) if side else None
            stop = []
            if spl.startswith("family:"):          # a THREAD of this process launches one kernel family on its own stream meanwhile
                import threading
                import race_screen
                def _load():
                    with torch.cuda.stream(torch.cuda.Stream(dev)):
                        while not stop:
                            race_screen.screen(reps=2, verbose=False, deep={}, family=spl.split(":", 1)[1], burst=25)
                th = threading.Thread(target=_load, daemon=True)
                th.start()
                import time
                time.sleep(5)
            poison = None
            if os.environ.get("POISON"):          # every vector register of every SIMD := a pattern right before each backward (same stream)
                import ctypes as C
                PL = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libpk_victim.so"))
                PL.pk_poison_launch.argtypes = [C.c_uint, C.c_int, C.c_void_p]
                pats = [int(v, 16) for v in os.environ["POISON"].split(",")]
                poison = lambda it: PL.pk_poison_launch(pats[it % len(pats)], 4096, torch.cuda.current_stream().cuda_stream)      # noqa: E731
            for it in range(iters * 20):
                if poison is not None:
                    assert poison(it) == 0
                if side is not None and it % 2 == 0:          # MFMA-heavy kernels of THIS process beside the loss backward
                    with torch.cuda.stream(side):
                        mb = ma @ ma
                for p_ in preds:
                    p_.grad = None
                out["tot_loss"].backward(retain_graph=True)
                grads = [p_.grad.detach().clone() for p_ in preds]
                sig = tuple(int(g_.view(torch.int16).to(torch.int64).sum().item()) for g_ in grads)
                hist[sig] = hist.get(sig, 0) + 1
            stop.append(1)
            torch.cuda.synchronize()
            if done is not None:
                done.set()
            print("rank 0: %d backward passes, %d distinct results, counts %s" % (iters * 20, len(hist), sorted(hist.values(), reverse=True)[:10]), flush=True)
            q.put((rank, bad, dict(chk), detail))
            return
        kind = os.environ.get("LOAD_KIND", "train")     # what the OTHER process runs: train | matmul | copy | fp32 | sleep
        if kind != "train":
            import time
            a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
            f = torch.randn(1 << 26, device=dev)
            t0 = time.time()
            while time.time() - t0 < float(os.environ.get("LOAD_SECONDS", "25")):
                if kind == "matmul":
                    for _ in range(20):
                        a @ a
                elif kind == "copy":
                    for _ in range(20):
                        busy[:1 << 27].copy_(busy[1 << 27:])
                elif kind.startswith("family:"):          # launches of ONE kernel family of the shipped table, in bursts
                    import race_screen
                    race_screen.screen(reps=2, verbose=False, deep={}, family=kind.split(":", 1)[1], burst=25)
                elif kind in ("fwd_train", "fwd_eval", "fwd_loss"):
                    if model.training != (kind != "fwd_eval"):
                        model.train(kind != "fwd_eval")
                    if kind == "fwd_loss":
                        outs_ = model(x)
                        pr_ = [o.detach().requires_grad_(True) for o in outs_]
                        lf_ = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
                        lf_(pr_, t)["tot_loss"].backward()
                    else:
                        with torch.no_grad():
                            model(x)
                elif kind == "fp32":
                    for _ in range(20):
                        f.mul_(1.0001).add_(0.5)
                else:
                    time.sleep(0.5)
                torch.cuda.synchronize()
            q.put((rank, bad, dict(chk), detail))
            return
        mode = "model_full_load"
    import time
    it, t_start = -1, time.time()
    while True:
        it += 1
        if mode == "model_full_load":          # the load lasts as long as rank 0's loop (at least `iters` passes, at most two minutes)
            if it >= iters and (done is None or done.is_set() or time.time() - t_start > 120):
                break
        elif it >= iters:
            break
        if mode == "model_full_load":
            outs = model(x)
            preds = list(outs)
        elif mode == "model_detached":          # real heads of a forward pass, the loss on DETACHED leaves (no engine backward)
            with torch.no_grad():
                pass
            outs = model(x)
            preds = [o.detach().requires_grad_(True) for o in outs]
        elif mode == "model_full":            # the whole graph: head gradients caught by hooks, engine backward runs
            outs = model(x)
            preds = list(outs)
            hg = [None] * 3
            for hi, o in enumerate(outs):
                o.register_hook(lambda g_, hi=hi, hg=hg: hg.__setitem__(hi, g_.detach().clone()))
        else:
            preds = [cell_major_view(b, 255).detach().requires_grad_(True) for b in bufs]
        lf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
        out = lf(preds, t)
        out["tot_loss"].backward()
        busy[:1 << 27].copy_(busy[1 << 27:])          # some other traffic of this process
        if mode == "model_full_load":
            continue
        grads = hg if mode == "model_full" else [p.grad.detach().clone() for p in preds]
        torch.cuda.synchronize()
        if first is None:
            first = grads
        else:
            for s, (a, b) in enumerate(zip(first, grads)):
                if not torch.equal(a, b):
                    d = (a.float() - b.float()).abs()
                    idx = torch.nonzero(d.reshape(-1) > 0).flatten()
                    ch = (idx % d.shape[1]) if d.dim() == 4 else idx          # (B, C, h, w) view: channel index
                    nz = torch.nonzero(d > 0)
                    bad.append((it, s, int(idx.numel()), float(d.max().item()), nz[:4].tolist(), sorted(set((nz[:, 1] % 85).tolist()))[:12]))
                    if len(detail) < 14:
                      for bb, cc, yy, xx in nz[:3].tolist():
                        a0 = cc // 85 * 85
                        if any(q["cell"] == [bb, cc // 85, yy, xx] and q["it"] == it for q in detail):
                            continue
                        detail.append(dict(it=it, stage=s, cell=[bb, cc // 85, yy, xx],
                                           ref=a[bb, a0:a0 + 8, yy, xx].float().tolist(), got=b[bb, a0:a0 + 8, yy, xx].float().tolist(),
                                           logits=outs[s][bb, a0:a0 + 5, yy, xx].float().tolist() if model is not None else None))
    q.put((rank, bad, dict(chk), detail))


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    mode = sys.argv[3] if len(sys.argv) > 3 else "random"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    done = ctx.Event()
    ps = [ctx.Process(target=worker, args=(r, iters, q, mode, done)) for r in range(procs)]
    for p in ps:
        p.start()
    for _ in range(procs):
        rank, bad, chk, detail = q.get(timeout=900)
        print(f"rank {rank}: {len(bad)} (iteration, stage) pairs differ from iteration 0 of {iters}; loss inputs changed between its forward and backward: {chk}", flush=True)
        for b in bad[:8]:
            print("    iteration %d stage %d: %d elements differ, max |diff| %.3e, first at (b, c, y, x) %s, channels mod 85: %s" % b, flush=True)
        for dd in detail:
            print("    detail", dd, flush=True)
    for p in ps:
        p.join(timeout=60)
