"""world_size-2 and world_size-8 gloo tests (CPU) of the data-parallel exchange used by bench.py --gpus N: the flat gradient is
averaged over ranks (DDP semantics, train_yolov5.py:219-220), no_sync skips it, BN states are averaged before eval.  The
8-rank variants rehearse the rank-count plumbing of the driver's 8-GPU run (bucket plan, accumulation boundary, launch,
rendezvous, max-over-ranks timing) — RCCL itself with N > 1 ranks has never executed anywhere."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yoloseries_amd.utils.dist import DataParallelGrads, all_reduce_norm, allreduce_flat_mean, get_rank, get_world_size, synchronize
    assert get_rank() == rank and get_world_size() == world
    res = {}
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_flat_mean(g, chunks=3)
    mr1 = (world + 1) / 2.0            # mean over ranks of (rank + 1)
    mr0 = (world - 1) / 2.0            # mean over ranks of rank
    res["mean_ok"] = bool(torch.allclose(g, torch.arange(1000, dtype=torch.float32) * mr1))

    class Fake:          # stands in for a HIP model: the engine calls model._yh_grad_hook(flat_grad) after backward
        pass
    fm = Fake()
    dp = DataParallelGrads(fm, chunks=2)
    g2 = torch.full((257,), float(rank))
    fm._yh_grad_hook(g2)
    res["hook_ok"] = bool(torch.allclose(g2, torch.full((257,), mr0)))
    g3 = torch.full((5,), float(rank))
    with dp.no_sync():
        fm._yh_grad_hook(g3)
    res["nosync_ok"] = bool(torch.allclose(g3, torch.full((5,), float(rank))))
    # gradient accumulation (train_yolov5.py:327-337): the backward after un-exchanged ones is the boundary; what is
    # averaged is the ACCUMULATED gradient (DDP semantics) and the bucket hooks stand back for that backward
    res["boundary_skips_buckets"] = fm._yh_bucket_hook(torch.zeros(4)) is None and not dp.buckets_active
    g4 = torch.full((5,), 10.0 * (rank + 1))
    fm._yh_grad_hook(g4, bucketed=True)
    # local sums: rank r: r + 10 (r + 1) -> mean (world 2: 15.5); handed on = mean - local un-exchanged part
    res["accum_ok"] = bool(torch.allclose(g3 + g4, torch.full((5,), mr0 + 10.0 * mr1))) and dp.buckets_active
    # overlapped exchange: the engine hands over contiguous slices of the packed gradient arena in backward order
    from yoloseries_amd.engine import plan_grad_buckets
    arena = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    marks = [(5, 900), (9, 700), (14, 650), (20, 300), (22, 280), (30, 0)]
    buckets = plan_grad_buckets(marks, 1000, 4)
    fins = [fm._yh_bucket_hook(arena[lo:hi]) for _, lo, hi in buckets]
    for f in fins:
        f()
    res["bucket_ok"] = bool(torch.allclose(arena, torch.arange(1000, dtype=torch.float32) * mr1))
    covered = sorted((lo, hi) for _, lo, hi in buckets)
    res["bucket_cover_ok"] = covered[0][0] == 0 and covered[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    with dp.no_sync():
        res["bucket_nosync_ok"] = fm._yh_bucket_hook(arena[:10]) is None
    bn = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
    with torch.no_grad():
        bn[1].running_mean.fill_(float(rank)); bn[1].weight.fill_(1.0 + rank)
    all_reduce_norm(bn)
    res["bn_ok"] = bool(torch.allclose(bn[1].running_mean, torch.full((4,), mr0)) and torch.allclose(bn[1].weight, torch.full((4,), 1.0 + mr0)))
    synchronize()
    q.put((rank, res))
    dist.destroy_process_group()


def _exchange(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in out) == list(range(world))
    for rank, res in out:
        assert all(res.values()), (rank, res)


def test_data_parallel_exchange_world2():
    _exchange(2)


def test_data_parallel_exchange_world8():
    """the same exchange semantics with the rank count of the driver's 8-GPU run"""
    _exchange(8)


def test_bench_gpus2_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torch.distributed.run around it) must start two ranks itself — before any GPU call —
    and report n_gpus == 2 / dp2, as the reference's driver does (train_yolov5.py:858-870, utils/launch.py:39-110).  No GPU
    here: --launch-check runs the same launch / rendezvous / timing / JSON plumbing with an empty step over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["YH_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout              # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["config"]["parallelism"] == "dp2"
    # the rank count of the driver's 8-GPU run through the same launch / rendezvous / barrier / max-over-ranks / JSON plumbing
    env["YH_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--launch-check", "--steps", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["config"]["parallelism"] == "dp8" and j["scaling"] == "weak"
    # more ranks than devices over RCCL is refused with a clear message, before anything is started
    env["YH_DIST_BACKEND"] = "nccl"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
