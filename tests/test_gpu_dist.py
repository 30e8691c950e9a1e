"""Data-parallel train step with the overlapped bucket all-reduce, two ranks sharing the one GPU of the test box
(gloo backend: RCCL refuses two ranks on one device).  Checks DDP semantics (train_yolov5.py:219-220): the
gradient every rank applies is the MEAN of the per-rank gradients, and the ranks stay bit-identical replicas."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, size="small", B=2, img=128, full=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=0.0, nesterov=True)
    dp = DataParallelGrads(model)
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(10 + rank)).to(dev)
    t = torch.from_numpy(synth_targets(B, img, 80, 6, seed=20 + rank)).to(dev)
    res = {}
    # local gradient (no exchange), then the exchanged one for the same inputs
    # (the loss object is stateful — `balances` EMA, loss/yolov5_loss.py:187-193 — so each pass gets a fresh one)
    def fresh_loss():
        return YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
    with dp.no_sync():
        fresh_loss()(model(x), t)["tot_loss"].backward()
    g_local = model._yh_last_flat_grad.clone()
    opt.zero_grad()
    dp.reset()                       # that gradient was only measured: it is not part of an accumulation cycle
    fresh_loss()(model(x), t)["tot_loss"].backward()
    g_dp = model._yh_last_flat_grad.clone()
    gathered = [torch.zeros_like(g_local).cpu() for _ in range(world)]
    dist.all_gather(gathered, g_local.cpu())
    g_mean = (sum(gathered) / world).to(dev)
    scale = g_mean.abs().max().item()
    # wgrad uses fp32 atomics: run-to-run differences of a few ulp of the largest partial sums
    res["mean_err"] = float((g_dp - g_mean).abs().max().item() / scale)
    res["differs_from_local"] = float((g_dp - g_local).abs().max().item() / scale)
    opt.step()
    opt.zero_grad()
    for _ in range(2):
        lossf(model(x), t)["tot_loss"].backward()
        opt.step()
        opt.zero_grad()
    if not full:                     # the judged shape: exchange + replicas only (the accumulation cycle is covered at the small shape)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        others = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        res["replicas_identical"] = bool(all(torch.equal(others[0], o) for o in others))
        res["finite"] = bool(torch.isfinite(flat).all())
        res["accum_err"] = 0.0
        prog = next(iter(model._yh_state()['progs'].values()))
        res["buckets"] = len(prog.bwd_buckets)
        res["two_streams"] = bool(prog.two_streams)
        dist.barrier()
        q.put((rank, res))
        dist.destroy_process_group()
        return
    # gradient accumulation over two micro-batches (train_yolov5.py:327-337; the stock config has accumulate = 2): the first
    # backward runs under no_sync, the boundary one synchronises; every rank must step with mean_r(g1_r + g2_r)
    x2 = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(30 + rank)).to(dev)
    t2 = torch.from_numpy(synth_targets(B, img, 80, 6, seed=40 + rank)).to(dev)
    locs = []
    for xx, tt in ((x, t), (x2, t2)):
        with dp.no_sync():
            fresh_loss()(model(xx), tt)["tot_loss"].backward()
        locs.append(model._yh_last_flat_grad.clone())
        opt.zero_grad()
        dp.reset()
    with dp.no_sync():
        fresh_loss()(model(x), t)["tot_loss"].backward()
    fresh_loss()(model(x2), t2)["tot_loss"].backward()
    g_acc = opt._grad().clone()
    tot_local = (locs[0] + locs[1]).cpu()
    gathered = [torch.zeros_like(tot_local) for _ in range(world)]
    dist.all_gather(gathered, tot_local)
    g_want = (sum(gathered) / world).to(dev)
    res["accum_err"] = float((g_acc - g_want).abs().max().item() / g_want.abs().max().item())
    opt.step()
    opt.zero_grad()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    others = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    res["replicas_identical"] = bool(all(torch.equal(others[0], o) for o in others))
    res["finite"] = bool(torch.isfinite(flat).all())
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def _collect(q, procs, n, limit=400):
    """the workers' results; a worker that died without reporting fails the test at once (a crashed rank would otherwise leave its
    peer in a collective and this process waiting on the queue)"""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > limit:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f"worker exit codes {dead or 'none'} after {time.time() - t0:.0f} s without a result")
    return out


@pytest.mark.parametrize("size", ["small", "large"])
def test_overlapped_bucket_allreduce_two_ranks(dev, size):
    """`large`: BASELINE config #4's model (YOLOv5l, models/normal/yolov5l.py:16-44: 177 MiB of fp32 gradients in the packed arena,
    cut into the same four buckets) through the same data-parallel step"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, size)) for r in range(2)]
    for p in procs:
        p.start()
    out = _collect(q, procs, 2)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in out:
        assert res["finite"] and res["replicas_identical"], (rank, res)
        assert res["mean_err"] < 1e-3, (rank, res)              # averaged gradient == mean of the local ones
        assert res["differs_from_local"] > 1e-2, (rank, res)    # ... and is not just the local gradient
        assert res["accum_err"] < 1e-3, (rank, res)             # accumulation boundary: mean of the ACCUMULATED gradients


def _rccl_worker(q):
    """one rank, backend nccl (= RCCL): the communicator, the bucket hook issued from the side stream's context, the
    finishers and no_sync run exactly as in the multi-GPU job (YH_FORCE_DP keeps the collectives for world size 1)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", YH_FORCE_DP="1")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.utils import FlatSGD
    from yoloseries_amd.utils.dist import DataParallelGrads
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    import bench
    res = {}
    torch.manual_seed(0)
    B, img = 2, 128
    model = models.YOLOV5Small(3, 80).to(dev).train()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=0.0, nesterov=True)
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(10)).to(dev)
    t = torch.from_numpy(synth_targets(B, img, 80, 6, seed=20)).to(dev)

    def fresh_loss():
        return YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), bench.make_hyp(dev, img, B))
    fresh_loss()(model(x), t)["tot_loss"].backward()
    g_plain = model._yh_last_flat_grad.clone()
    opt.zero_grad()
    calls = {"n": 0}
    for dtype in (None, torch.bfloat16):
        dp = DataParallelGrads(model, bucket_dtype=dtype)
        inner = dp._bucket

        def counting(part, inner=inner):
            f = inner(part)
            calls["n"] += f is not None
            return f
        model._yh_bucket_hook = counting
        fresh_loss()(model(x), t)["tot_loss"].backward()
        g = model._yh_last_flat_grad.clone()
        opt.zero_grad()
        scale = g_plain.abs().max().item()
        res["err_" + ("fp32" if dtype is None else "bf16")] = float((g - g_plain).abs().max().item() / scale)
        with dp.no_sync():
            before = calls["n"]
            fresh_loss()(model(x), t)["tot_loss"].backward()
            res["nosync_" + ("fp32" if dtype is None else "bf16")] = calls["n"] == before
        opt.zero_grad()
        dp.reset()
    res["bucket_collectives"] = calls["n"]
    from yoloseries_amd.utils.dist import all_reduce_norm
    all_reduce_norm(model)
    torch.cuda.synchronize()
    res["backend"] = dist.get_backend()
    # the streams the step ran on really run beside the compute stream although a process group exists (yoloseries_amd/streams.py)
    from yoloseries_amd import streams
    big, tiny = torch.empty(1 << 30, dtype=torch.uint8, device=dev), torch.empty(256, dtype=torch.uint8, device=dev)
    main, side, comm = torch.cuda.default_stream(dev), streams.side_stream(dev), streams.comm_stream(dev)
    res["side_beside_main"] = streams._runs_beside(side, [main], big, tiny)
    res["comm_beside_both"] = streams._runs_beside(comm, [main, side], big, tiny)
    res["main_beside_main"] = streams._runs_beside(main, [main], big, tiny)          # the probe itself: a stream is not beside itself
    q.put(res)
    dist.destroy_process_group()


def test_rccl_single_rank_bucket_path(dev):
    """the `nccl` (RCCL) branch on the one GPU of the test box: world size 1 communicator, overlapped buckets"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res["backend"] == "nccl"
    assert res["bucket_collectives"] >= 4, res               # >= 2 buckets per backward, two exchanged backwards
    assert res["err_fp32"] < 1e-3, res                       # sum over one rank / 1 == the plain gradient (atomics noise only)
    assert res["err_bf16"] < 2e-2, res                       # bf16 buckets: one rounding of every element
    assert res["nosync_fp32"] and res["nosync_bf16"], res
    assert res["side_beside_main"] and res["comm_beside_both"] and not res["main_beside_main"], res


def test_bench_gpus2_real_step_two_ranks_one_gpu(dev):
    """the judged entry point as a plain command: `python bench.py --gpus 2` starts its two ranks itself (utils/launch.py, spawn
    before any GPU call) and runs the real data-parallel train step — two ranks sharing this box's GPU over gloo (RCCL refuses
    two ranks on one device) at a small batch; rank 0 alone prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["YH_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "4", "--img", "256", "--steps", "2",
                        "--warmup", "1", "--no-roofline", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 8
    assert j["value"] > 0 and j["final_loss"] == j["final_loss"]


def test_overlapped_bucket_allreduce_two_ranks_judged_shape(dev):
    """the same data-parallel step AT THE JUDGED SHAPE — YOLOv5l (BASELINE config #4's model), 64 images of 640 x 640 per rank, the
    shipped launch parameters, two ranks sharing the GPU over gloo: bucket marks of the packed 177 MiB gradient arena, the gz ring,
    the side stream's hand-overs and the finishers run where the bench runs them (the small test runs B = 2 at
    128 x 128).  Averaged gradient == mean of the local ones, replicas bit-identical after three steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "large", 64, 640, False)) for r in range(2)]
    for p in procs:
        p.start()
    out = _collect(q, procs, 2, limit=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in out:
        assert res["finite"] and res["replicas_identical"], (rank, res)
        assert res["mean_err"] < 1e-3, (rank, res)
        assert res["differs_from_local"] > 1e-2, (rank, res)
        assert res["buckets"] >= 3 and res["two_streams"], (rank, res)


def test_loss_backward_reproducible_beside_second_process():
    """400 YOLOv5 loss backwards on fixed inputs (batch 64, 640 x 640) while ANOTHER PROCESS trains on the same GPU: one distinct
    result.  Round 6 found 15 % of such passes with a wrong cell or more (y / h gradients of positives): packed fp32 instructions
    of the -O3 build, only with a second process on the card; the exact sources compile without the vectorizers since
    (csrc/Makefile EXACT).  The children own the GPU; this process only reads their report."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "loss_race_diag.py"), "2", "20", "bwd_repeat"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    m = re.search(r"rank 0: (\d+) backward passes, (\d+) distinct results", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-2000:]
    assert int(m.group(1)) == 400 and int(m.group(2)) == 1, m.group(0)
