#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/sec of one full YOLOv5s training step at 640x640
— forward (HIP conv graph, bf16) + YOLOv5 loss + backward + clip + SGD-nesterov + EMA —
batch 64 per GPU on synthetic COCO-shaped data resident in HBM, random-init weights.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL all-reduce of the flat gradient)

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     — dominant kernel family, algorithmic conv FLOPs / its measured duration (HIP events on the
                 launch stream, separate instrumented steps after the timed region)
  cpu_baseline — the oracle (torch-CPU fp32 port of the same train step) on a bounded sample, rank 0, N=1
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # bf16 dense MFMA peak, MI355X (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
TRAIN_GFLOP_PER_IMG = 49.30    # YOLOv5s@640: 3 x 16.434 GFLOP conv fwd
# conv GFLOP per image at 640x640, forward only (v5m / v5x from the same census of this package's graph)
FWD_GFLOP_640 = {"small": 16.434, "middle": 48.87, "large": 108.99, "xlarge": 218.56, "yolox_s": 24.14}


def make_hyp(dev, img, batch):
    return dict(device=dev, num_class=80, input_img_size=[img, img], batch_size=batch,
                use_focal_loss=True, focal_loss_gamma=1.5, focal_loss_alpha=0.25,
                iou_loss_scale=0.05, cls_loss_scale=0.5, cof_loss_scale=1.0, anchor_match_thr=4.0,
                class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0,
                loss_items_on_device=True)


def _host_threads():
    """the threads this process may actually run on (a 1-GPU box shares its host: affinity / cgroup share, not the socket's core
    count — 256 OpenMP threads on a 16-CPU share ran 40x slower than 16), capped at 16"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                cores = min(cores, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, 16))


def _cpu_train_leg(batch, img, budget_s, max_steps, min_steps=3):
    """(median seconds per step, steps timed) of the torch-CPU fp32 oracle's train step at `batch`: 1 warm-up, then `min_steps` timed
    steps whatever they take (a median needs more than one sample) and further ones, up to `max_steps`, while
    the budget lasts"""
    from oracle.v5loss import V5LossOracle
    from oracle.v5net import V5NetOracle
    from yoloseries_amd import models
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets
    torch.manual_seed(0)
    net = V5NetOracle(models.YOLOV5Small(3, 80).state_dict(), train=True)
    lossf = V5LossOracle(COCO_ANCHORS, make_hyp("cpu", img, batch))
    x = torch.rand(batch, 3, img, img, generator=torch.Generator().manual_seed(0))
    t = synth_targets(batch, img, 80, 20, seed=1)
    times = []
    t_begin = time.time()
    for it in range(max_steps + 1):
        t0 = time.time()
        out = lossf(net(x), t)
        out["tot_loss"].backward()
        net.sgd_step(0.01)
        times.append(time.time() - t0)
        if it >= min_steps and time.time() - t_begin > budget_s:
            break
    steady = times[1:] if len(times) > 1 else times
    return float(np.median(steady)), len(steady)


def cpu_baseline(budget_s=14.0, batch=64, img=640):
    """torch-CPU / NumPy fp32 port (oracle/) of the same path on a bounded sample, both legs of BASELINE.md section 4:
    (fwd + loss + bwd + SGD) at the judged batch (64; 1 warm-up + 3 timed steps, ~1 minute of host time) and at BASELINE config #1's
    batch 4 (`batch4`; 1 warm-up + 3..5 steps), and
    (decode + candidate filter + class-aware NMS on the synthetic NMS stress heads), each the median of its timed runs."""
    from oracle import postproc as opp
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_nms_heads
    cores = _host_threads()
    torch.set_num_threads(cores)
    sec4, nst4 = _cpu_train_leg(4, img, 0.25 * budget_s, 5)
    sec, nst = _cpu_train_leg(batch, img, 0.75 * budget_s, 3) if batch != 4 else (sec4, nst4)
    nbatch = 4
    # decode + filter + NMS (NumPy fp32, one thread: the reference's evaluator is a per-image Python / numba loop)
    heads = synth_nms_heads(nbatch, img, 80, 3, seed=2, wh_shift=1.2)
    ntimes, kept = [], 0
    t_begin = time.time()
    for it in range(6):
        t0 = time.time()
        dec = opp.decode_v5(heads, COCO_ANCHORS, (8, 16, 32))
        res = opp.postprocess_v5(dec, 0.001, 0.001, 0.65)
        ntimes.append(time.time() - t0)
        kept = sum(0 if r is None else len(r) for r in res)
        if it >= 1 and time.time() - t_begin > 0.4 * budget_s:
            break
    nsteady = ntimes[1:] if len(ntimes) > 1 else ntimes
    return {"value": round(batch / sec, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"median of {nst} train steps (fwd+loss+bwd+SGD) of YOLOv5s at batch {batch}, {img}x{img}, "
            f"torch-CPU fp32 oracle, after 1 warm-up",
            "batch4": {"value": round(4 / sec4, 3), "unit": "images/sec", "sample": f"the same step at batch 4 (BASELINE configs[0]), median of {nst4}"},
            "decode_nms": {"value": round(nbatch / float(np.median(nsteady)), 3), "unit": "images/sec", "cores": 1,
                           "sample": f"median of {len(nsteady)} runs of decode + filter(conf 0.001) + class-aware NMS(0.65) on {nbatch} synthetic "
                           f"head sets at {img}x{img} (~1 % of the anchors are candidates, {kept // nbatch} boxes kept per image), NumPy fp32 oracle"}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the train step once into a hipGraph and replay it (utils/graph.py). "
                    "Off by default: measured on MI355X / ROCm 7.2 the replay is no faster than the eager two-stream schedule "
                    "(the step is bound by the ~700 dependent kernel boundaries, not by host launch time; DESIGN.md section 7)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--model", default="small", choices=["small", "middle", "large", "xlarge"])
    ap.add_argument("--workload", default="train", choices=["train", "yolox", "infer"],
                    help="train: YOLOv5 train step (BASELINE configs[1], the default and the judged metric); yolox: YOLOXs + SimOTA "
                         "train step (configs[2]); infer: YOLOv5 eval forward + decode + class-aware NMS (configs[4]: --model xlarge --img 1280)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rehearse the multi-rank launch only (rendezvous, barrier, max-over-ranks timing, rank 0's JSON line) with an empty "
                         "step: runs without a GPU under YH_DIST_BACKEND=gloo; `value` is null")
    return ap.parse_args(argv)


def visible_gpus():
    """GPUs this process may use, WITHOUT touching the HIP runtime (torch.cuda.device_count() can fall back to hipGetDeviceCount,
    which initialises it in the launcher parent): the visibility variables if set, else the KFD topology (a GPU node has a non-zero
    simd_count).  None when neither says anything: every rank then checks its own device (worker())."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += 1 if int(props.get("simd_count", "0")) > 0 else 0
        return n
    except (OSError, ValueError):
        return None


def main():
    """`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself, the way the reference's
    driver does (train_yolov5.py:858-870 -> utils/launch.py:39-110), BEFORE anything here touches the GPU — the parent only
    spawns fresh interpreters (multiprocessing `spawn`, never an exec of a process that has initialised HIP), waits for them, and
    fails if any rank fails; rank 0 prints the JSON line.  Under torch.distributed.run (WORLD_SIZE set) the process is one rank."""
    args = parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    backend = os.environ.get("YH_DIST_BACKEND", "nccl")      # "gloo" only to rehearse N>1 without N GPUs
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
        return worker(args)
    if args.gpus == 1:
        return worker(args)
    ndev = visible_gpus()                                     # from the environment / sysfs: the parent never calls into HIP
    if backend == "nccl" and ndev is not None and args.gpus > ndev:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible (one rank per GPU over RCCL)")
    from yoloseries_amd.utils.launch import launch
    launch(worker, args.gpus, num_machines=1, machine_rank=0, backend=backend, dist_url="auto", args=(args,))


def worker(args):
    global PMC_WORKLOAD
    PMC_WORKLOAD = f"{args.workload}:{args.model}:{args.batch}:{args.img}"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("YH_DIST_BACKEND", "nccl")
    import torch.distributed as dist
    if args.launch_check:
        return launch_check(args, world, rank, backend)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and local_rank >= ndev:
        sys.exit(f"bench.py: rank {local_rank} has no GPU ({ndev} visible)")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    # YH_FORCE_DP=1: run the data-parallel path (RCCL communicator, overlapped gradient buckets) on a single rank too
    force_dp = world == 1 and os.environ.get("YH_FORCE_DP") == "1"
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    own_group = (world > 1 or force_dp) and not dist.is_initialized()     # launch() initialises the group of self-started ranks
    if own_group:
        from yoloseries_amd.utils.dist import quiet_stdout
        with quiet_stdout():               # RCCL prints its version banner on stdout: stdout is for the one JSON line
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
                dist.barrier()
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    from yoloseries_amd import models
    from yoloseries_amd.loss import YOLOV5Loss
    from yoloseries_amd.trainer import ExponentialMovingAverageModel
    from yoloseries_amd.utils import FlatSGD
    from yoloseries_amd.utils.dist import DataParallelGrads
    from yoloseries_amd.utils.synth import COCO_ANCHORS, synth_targets

    B, img = args.batch, args.img
    torch.manual_seed(0)
    cls = {"small": models.YOLOV5Small, "middle": models.YOLOV5Middle, "large": models.YOLOV5Large, "xlarge": models.YOLOV5XLarge}[args.model]
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(rank), dtype=torch.float32).to(dev)
    t = torch.from_numpy(synth_targets(B, img, 80, 20, seed=1 + rank)).to(dev)
    dp = None
    extra = {}
    if args.workload == "infer":
        # eval forward (BN folded, SiLU in the conv epilogue) + fused decode / filter / class-aware NMS; only kept rows cross PCIe
        from yoloseries_amd.trainer import YOLOV5Evaluator
        from yoloseries_amd.utils.synth import synth_nms_heads
        model = cls(3, 80).to(dev).eval()
        ehyp = dict(device=dev, num_class=80, input_img_size=[img, img], iou_threshold=0.2, conf_threshold=0.3, cls_threshold=0.3,
                    max_predictions_per_img=300, iou_type="iou", mutil_label=False, agnostic=True, postprocess_bbox=True, wfb=False,
                    use_tta=False, half=False, compute_metric_conf_threshold=0.001, compute_metric_iou_threshold=0.65,
                    compute_metric_cls_threshold=0.001)
        ev = YOLOV5Evaluator(model, torch.from_numpy(COCO_ANCHORS).to(dev), ehyp, compute_metric=True)
        gflop_img = FWD_GFLOP_640[args.model] * (img / 640.0) ** 2
        metric = f"images/sec ({img}x{img}) YOLOv5{args.model[0]} inference (forward + decode + NMS)"
        workload = (f"YOLOv5{args.model[0]} bf16 inference: eval forward + decode + filter(conf 0.001) + class-aware NMS, batch {B}/GPU x "
                    f"{img}x{img} synthetic, random-init (BASELINE.json configs[4] shape)")

        def step():
            return {"tot_loss": torch.zeros((), device=dev), "dets": ev(x)}

        # a random-init net yields ~no detections: NMS throughput is measured on synthetic head tensors:
        # ~1 % of the anchors pass conf 0.001, 50 clusters of overlapping boxes per image
        # all B images of the batch, synthesised 16 at a time with a seed per chunk to bound the host memory
        nb = B
        heads = None
        for c0 in range(0, nb, 16):
            part = [torch.from_numpy(h).to(dev) for h in synth_nms_heads(min(16, nb - c0), img, 80, 3, seed=2 + c0 // 16, wh_shift=1.2)]
            heads = part if heads is None else [torch.cat([a, b]) for a, b in zip(heads, part)]
        for _ in range(2):
            res = ev._nms_from_heads(heads)
        torch.cuda.synchronize()
        t0n = time.perf_counter()
        reps = 10
        for _ in range(reps):
            res = ev._nms_from_heads(heads)
        torch.cuda.synchronize()
        dtn = (time.perf_counter() - t0n) / reps
        kept = sum(0 if r is None else len(r) for r in res)
        n_anchor = sum(3 * (img // s) ** 2 for s in (8, 16, 32))
        ncand = sum(ev.last_ncand)
        # the fused decode + candidate filter alone, against its algorithmic read (every head logit once: anchors x 85 x 2 B)
        import ctypes as C
        from yoloseries_amd._lib import check, lib, stream_ptr
        dd, canon, ptrs = ev._desc(heads)
        capf = ((n_anchor + 3) // 4) * 4
        candf = torch.empty(nb, capf, 6, dtype=torch.float32, device=dev)
        ncf = torch.zeros(nb, dtype=torch.int32, device=dev)
        dws = torch.empty(int(lib().yh_decode_filter_ws_bytes(C.byref(dd))), dtype=torch.uint8, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            check(lib().yh_decode_filter(C.byref(dd), ptrs, float(ev.conf_threshold), float(ev.cls_threshold), candf.data_ptr(),
                                         ncf.data_ptr(), capf, dws.data_ptr(), stream_ptr()), "yh_decode_filter")
        e1.record()
        e1.synchronize()
        df_ms = e0.elapsed_time(e1) / reps
        df_bytes = nb * n_anchor * 85 * heads[0].element_size()
        extra["decode_filter"] = {"bound": "hbm", "achieved": round(df_bytes / df_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(df_bytes / df_ms / 1e6 / HBM_PEAK_GBS, 4), "ms": round(df_ms, 3),
                                  "bytes_per_image": n_anchor * 85 * heads[0].element_size(), "images": nb}
        extra["nms_synthetic"] = {"images": nb, "anchors_per_image": n_anchor, "candidates_per_image": round(ncand / nb, 1),
                                  "kept_boxes_per_image": round(kept / nb, 1), "images_per_s": round(nb / dtn, 1),
                                  "anchors_per_s": round(nb * n_anchor / dtn), "candidate_boxes_per_s": round(ncand / dtn),
                                  "kept_boxes_per_s": round(kept / dtn)}
    else:
        if args.workload == "yolox":
            from yoloseries_amd.loss import YOLOXLoss
            model = models.YOLOXSmall(1, 3, 80).to(dev).train()
            hyp = dict(device=dev, num_class=80, input_img_size=[img, img], batch_size=B, use_focal_loss=False, focal_loss_gamma=1.5,
                       focal_loss_alpha=0.25, iou_loss_scale=5.0, use_l1=True, l1_loss_scale=1.0, cls_loss_scale=1.0, cof_loss_scale=1.0,
                       class_smooth_factor=1.0, cls_pos_weight=1.0, cof_pos_weight=1.0, num_anchors=1, iou_type="ciou", topk=13,
                       center_radius=3, num_stage=3, loss_items_on_device=True)
            lossf = YOLOXLoss(hyp)
            gflop_img = 3 * FWD_GFLOP_640["yolox_s"] * (img / 640.0) ** 2
            metric = f"images/sec ({img}x{img}) YOLOXs train-step"
            workload = (f"YOLOXs bf16 train step (fwd+SimOTA loss+bwd+clip+SGD+EMA), batch {B}/GPU x {img}x{img} synthetic COCO-80, "
                        f"random-init (BASELINE.json configs[2])")
        else:
            model = cls(3, 80).to(dev).train()
            hyp = make_hyp(dev, img, B)
            lossf = YOLOV5Loss(torch.from_numpy(COCO_ANCHORS).to(dev), hyp)
            gflop_img = 3 * FWD_GFLOP_640[args.model] * (img / 640.0) ** 2
            metric = f"images/sec ({img}x{img}) YOLOv5{args.model[0]} train-step"
            workload = (f"YOLOv5{args.model[0]} bf16 train step (fwd+loss+bwd+clip+SGD+EMA), batch {B}/GPU x {img}x{img} synthetic COCO-80, "
                        f"random-init (BASELINE.json configs[1])")
        lr = 0.000625 * B                       # basic_lr_per_img x per-rank batch (train_yolov5.py:184)
        opt = FlatSGD(model, lr=lr, momentum=0.937, weight_decay=1e-4, nesterov=True)
        ema = ExponentialMovingAverageModel(model)
        dp = DataParallelGrads(model, overlap=os.environ.get("YH_DP_OVERLAP", "1") != "0") if (world > 1 or force_dp) else None

        def step():
            # YOLOXLoss converts the target boxes to xywh IN PLACE like the reference (loss/yolox_loss.py:70-75): every step
            # gets a fresh copy, as a data loader would deliver
            out = lossf(model(x), t.clone() if args.workload == "yolox" else t)
            out["tot_loss"].backward()
            opt.clip_grad_norm_(10.0)
            opt.step()
            opt.zero_grad()
            ema.update(model)
            return out

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the train step is captured once into a hipGraph (forward, loss, backward incl. its side-stream branch, clip, SGD, EMA)
    # and replayed: one launch per step instead of ~700; the warm-up steps run eagerly (they build and tune the programs)
    eager_step = step
    stepper = None
    if args.workload != "infer" and dp is None and (args.graph or os.environ.get("YH_GRAPH", "0") == "1"):
        from yoloseries_amd.utils.graph import GraphedStep
        stepper = GraphedStep(eager_step, pre_replay=[opt.graph_pre_replay, ema.graph_pre_replay], warmup=max(2, min(args.warmup, 3)))
        step = stepper
    for _ in range(max(args.warmup, 4 if stepper is not None else 0)):
        out = step()
    sync_all()
    t0 = time.perf_counter()
    w0 = time.time()
    for _ in range(args.steps):
        out = step()
    sync_all()
    dt = time.perf_counter() - t0
    if os.environ.get("YH_BENCH_STAMP"):          # wall-clock window of the timed region (concurrency experiments)
        print(f"# timed region {w0:.4f} .. {time.time():.4f}", file=sys.stderr)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss_val = float(out["tot_loss"].item())
    ips = world * B * args.steps / dt

    launch_mode = stepper.mode if stepper is not None else "eager"
    if stepper is not None and stepper.failed:
        print(f"# hipGraph capture failed, ran eagerly: {stepper.failed}", file=sys.stderr)
    step = eager_step                        # the instrumented roofline steps below time individual launches
    roof = None
    if rank == 0 and not args.no_roofline:
        # rank 0 alone instruments a few extra steps: no gradient exchange in them (the other ranks wait below)
        if dp is not None:
            with dp.no_sync():
                roof = measure_roofline(model, step, B)
        else:
            roof = measure_roofline(model, step, B)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "train" and args.model == "small":
        try:
            cpu = cpu_baseline()
        except Exception as e:      # the oracle is test infrastructure; never let it fail the measurement
            cpu = {"error": repr(e)}
    if rank == 0:
        res = {
            "metric": metric, "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "global_batch": B * world, "parallelism": f"dp{world}" + (" (forced RCCL path)" if force_dp else "")},
            "train_tflops": round(ips * gflop_img / 1000.0, 2),
            "mfma_frac_step": round(ips * gflop_img / 1000.0 / (MFMA_PEAK_TFLOPS * world), 4),
            "final_loss": round(loss_val, 4), "launch": launch_mode, "tuning": _tuning(),
            "hbm_frac_step": (round(roof["hbm_bytes_per_step"] / (dt / args.steps) / (HBM_PEAK_GBS * 1e9), 4)
                              if roof and roof.get("hbm_bytes_per_step") else None), **extra,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(res))
    if world > 1 or force_dp:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()


def launch_check(args, world, rank, backend):
    """the multi-rank plumbing of the bench without the GPU work: group (from launch() or the environment), the barrier-bracketed
    timed region with an empty step, MAX over ranks, one JSON line on rank 0.  CPU-runnable (gloo): tests/test_dist_gloo.py."""
    import torch.distributed as dist
    own_group = world > 1 and not dist.is_initialized()
    if own_group:
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ranks = torch.ones(1, dtype=torch.float64)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks)
        dt = tt.item()
    if rank == 0:
        print(json.dumps({"metric": "launch check (no GPU work)", "value": None, "unit": "images/sec", "n_gpus": world, "ranks_seen": int(ranks.item()),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / max(args.steps, 1), 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "launch check", "global_batch": args.batch * world, "parallelism": f"dp{world}"},
                          "backend": backend}), flush=True)
    if world > 1:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()


def _tuning():
    """where the per-layer launch parameters came from: the table shipped with the package (yoloseries_amd/tune_defaults.json, built
    on an MI355X by tools/make_tune_defaults.sh and selected by its score on this very bench line), this machine's cache, or timed now"""
    from yoloseries_amd.engine import tuning_source
    src = tuning_source()
    kind = "shipped table" if src["timed_now"] == 0 and src["local_cache"] == 0 else "shipped table + timed on this machine"
    return {"kind": kind, **src}


def _instrument(prog, step, nsteps):
    """HIP-event pairs around every engine launch (recorded on the stream the kernel is launched on) over `nsteps` steps"""
    prog.profile = {}
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    prof, prog.profile = prog.profile, None
    fam, per_op = {}, []
    for key, recs in prof.items():
        name, flops, nbytes, opname = key
        ms = sum(s.elapsed_time(e) for s, e in recs)
        per_op.append((ms / nsteps, name, opname, flops * len(recs) / nsteps, nbytes * len(recs) / nsteps))
        f = fam.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        f["ms"] += ms
        f["flops"] += flops * len(recs)
        f["bytes"] += nbytes * len(recs)
        f["launches"] += len(recs)
    return fam, per_op


def _lib_sha16():
    import hashlib
    from yoloseries_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


PMC_WORKLOAD = None      # "<workload>:<model>:<batch>:<img>" of this run, set by main()


def _pmc_file(fname):
    """a committed PMC summary (profiles/<fname>) if it was collected with THIS library on THIS workload, else None"""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", fname)) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None
    if j.get("lib_sha16") != _lib_sha16() or j.get("workload", "train:small:64:640") != PMC_WORKLOAD:
        return None
    return j


def _pmc_mfma_util(pmc, name):
    """matrix-unit utilisation of an engine kernel family from the MFMA PMC pass (tools/pmc_mfma.py)"""
    if pmc is None:
        return None
    for n in _PMC_ALIAS.get(name, (name.replace("yh_", "") + "_kernel", name)):
        if n in pmc["kernels"]:
            return pmc["kernels"][n]["mfma_util"]
    return None


def _pmc_traffic():
    """per-kernel HBM bytes from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py): counters cannot be read from
    inside the process.  The file is stamped with the hash of the libyolohip.so it was collected with and with the workload
    it was collected on; a different library or workload means the numbers describe other kernels / other shapes, and they
    are dropped (null) instead of being reported stale."""
    return _pmc_file(_pmc_name("pmc_traffic"))


def _pmc_name(stem):
    """profiles/pmc_traffic.json belongs to the judged line; the other workloads keep theirs next to it"""
    if PMC_WORKLOAD in (None, "train:small:64:640"):
        return stem + ".json"
    return stem + "_" + PMC_WORKLOAD.replace(":", "_") + ".json"


_PMC_ALIAS = {"yh_bn_silu_bwd_reduce": ("col_reduce_kernel<0>",), "yh_colsum": ("col_reduce_kernel<1>", "colsum_finalize_kernel"),
              "yh_bn_fold": ("bn_fold_batch_kernel",)}


def _pmc_bytes(pmc, name):
    """HBM bytes per launch of an engine kernel family from the PMC file (profiler kernel names)"""
    if pmc is None:
        return None
    names = _PMC_ALIAS.get(name, (name.replace("yh_", "") + "_kernel", name))
    vals = [pmc["kernels"][n]["hbm_bytes_per_launch"] for n in names if n in pmc["kernels"]]
    return sum(vals) if vals else None


def _limiter(name, f_mfma, f_hbm, traffic, algo_bytes):
    """what binds a kernel family that sits under 30 % of BOTH roofs (measured in DESIGN.md section 5, not inferred here)"""
    if max(f_mfma, f_hbm) >= 0.3:
        return "mfma" if f_mfma >= f_hbm else "hbm"
    if name.startswith("conv_wgs"):
        return "LDS-DMA feed (512 B per MFMA through the CU's vector memory path: the loop is issue-bound at ~62 % of the MFMA rate) + one 64 KB tile of fp32 atomics per workgroup"
    if name.startswith("conv_wgrad"):
        return "split-M epilogue: fp32 atomics / partial tiles per block (work per block too small to amortise a 64 KB tile)"
    if name.startswith(("yh_bn_finalize", "yh_bn_bwd_finalize", "yh_colsum")):
        return "launch latency (a few microseconds of work per launch)"
    if traffic and algo_bytes and traffic > 1.5 * algo_bytes:
        return "hbm re-reads (traffic well above the algorithmic bytes)"
    if name.startswith(("conv_v2", "conv_v3", "conv_halo", "conv_dg2", "conv_stem", "conv_igemm")):
        return "issue / barrier per k-step and per-tile fixed costs (ablation builds, DESIGN.md section 5)"
    return "latency"


def _family_roofline(name, d, pmc, pmc_mfma=None):
    """roofline entry of one kernel family from its algorithmic work and measured duration: the bound is the roof the family sits
    closer to (conv tiles with little reuse — 1x1 layers, the stage-1 layers — are HBM-bound, K-heavy ones MFMA-bound)"""
    sec = d["ms"] * 1e-3
    tf = d["flops"] / sec / 1e12
    gbs = d["bytes"] / sec / 1e9
    f_mfma, f_hbm = tf / MFMA_PEAK_TFLOPS, gbs / HBM_PEAK_GBS
    traffic = _pmc_bytes(pmc, name)
    common = {"kernel": name, "traffic": traffic, "avg_launch_us": round(1000 * d["ms"] / d["launches"], 2),
              "flops_per_launch": round(d["flops"] / d["launches"]), "bytes_per_launch": round(d["bytes"] / d["launches"]),
              "mfma_frac": round(f_mfma, 4), "hbm_frac": round(f_hbm, 4), "mfma_util": _pmc_mfma_util(pmc_mfma, name),
              "limiter": _limiter(name, f_mfma, f_hbm, traffic, d["bytes"] / d["launches"])}
    if f_mfma >= f_hbm:
        return {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(f_mfma, 4), **common}
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(f_hbm, 4), **common}


def _group_of(name):
    """the five groups of the step's engine kernels (the per-template `dominant kernel` hides the largest group)"""
    if name.startswith(("conv_wgrad", "conv_wgs", "conv_wgp")):
        return "wgrad"
    if name.startswith("conv_"):
        return "conv"                                   # forward convolutions and data gradients
    if name.startswith(("yh_bn_finalize", "yh_bn_bwd_finalize", "yh_bn_frozen")):
        return "finalize"
    if name.startswith(("yh_bn_silu", "yh_bn_")):
        return "bn_silu"
    return "other"                                      # pools, upsample backward, column sums, input conversion


def _groups(fam, nsteps, pmc):
    """per group: time, launches, aggregate TFLOP/s and GB/s on the ALGORITHMIC work (the same flops / bytes the per-family
    entries use), the fractions of both roofs, and PMC HBM traffic over algorithmic bytes where the committed counter file covers
    the group's kernels (`traffic_coverage` = share of the group's launches it covers)"""
    out = {}
    for name, v in fam.items():
        g = out.setdefault(_group_of(name), {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "traffic": 0.0, "tbytes": 0.0, "tl": 0})
        g["ms"] += v["ms"]; g["launches"] += v["launches"]; g["flops"] += v["flops"]; g["bytes"] += v["bytes"]
        t = _pmc_bytes(pmc, name)
        if t is not None and v["launches"] > 0:
            g["traffic"] += t * v["launches"]; g["tbytes"] += v["bytes"]; g["tl"] += v["launches"]
    res = {}
    for k, g in sorted(out.items(), key=lambda kv: -kv[1]["ms"]):
        sec = g["ms"] * 1e-3
        if sec <= 0:
            continue
        tf, gbs = g["flops"] / sec / 1e12, g["bytes"] / sec / 1e9
        res[k] = {"ms_per_step": round(g["ms"] / nsteps, 3), "launches_per_step": g["launches"] // nsteps,
                  "tflops": round(tf, 1), "gbs": round(gbs, 1), "mfma_frac": round(tf / MFMA_PEAK_TFLOPS, 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                  "traffic_over_algorithmic": round(g["traffic"] / g["tbytes"], 3) if g["tbytes"] > 0 else None,
                  "traffic_coverage": round(g["tl"] / g["launches"], 3) if g["launches"] else None}
    return res


def _foreign_launches(step, nsteps=2):
    """kernels of a steady-state step that are NOT this library's (torch fills / copies / elementwise kernels, runtime copy kernels):
    launches and device time per step from a torch.profiler (roctracer) trace of `nsteps` extra steps.  Every kernel of
    libyolohip.so lives in an anonymous namespace, which is how the two are told apart (tools/foreign_launches.py does the same on a
    rocprofv3 trace).  None when the profiler is unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(nsteps):
                step()
            torch.cuda.synchronize()
        ours = n_all = 0
        foreign = {}
        for ev in prof.events():
            if str(getattr(ev, "device_type", "")).endswith("CUDA") is False:
                continue
            name = ev.name
            n_all += 1
            if "(anonymous namespace)::" in name:
                ours += 1
                continue
            f = foreign.setdefault(name.split("(")[0][:80], [0, 0.0])
            f[0] += 1
            f[1] += float(getattr(ev, "device_time", 0.0) or getattr(ev, "cuda_time", 0.0) or 0.0)
        if ours == 0:
            return None
        return {"launches_per_step": round(sum(v[0] for v in foreign.values()) / nsteps, 1),
                "ms_per_step": round(sum(v[1] for v in foreign.values()) / nsteps / 1000.0, 4),
                "library_launches_per_step": round(ours / nsteps, 1),
                "kernels": {k: round(v[0] / nsteps, 1) for k, v in sorted(foreign.items(), key=lambda kv: -kv[1][1])[:6]}}
    except Exception as e:       # diagnostics only: never fail the measurement
        return {"error": repr(e)[:200]}


def measure_roofline(model, step, B, nsteps=3):
    """Roofline of the dominant kernel family (template instantiation, profiler spelling: the one with the largest total time),
    plus the same entry for the conv family with the largest time among the MFMA-bound ones (`conv`), per-family times, and the
    step's HBM traffic from the committed PMC passes.

    The timed region runs the weight gradients on a side stream next to the dgrad / BatchNorm chain; a kernel's duration
    then includes whatever ran beside it.  The roofline numbers are therefore taken with the kernels back to back
    (`two_streams` off == `YH_BWD_STREAMS=0`; profiles/r02_kernel_stats_serial.csv is the rocprofv3 summary of that
    run) — a statement about the kernel, not about the overlap; `overlapped` repeats the same kernel as it runs in the
    timed configuration."""
    prog = next(iter(model._yh_state()['progs'].values()))
    two = getattr(prog, "two_streams", False)
    prog.two_streams = False
    fam, per_op = _instrument(prog, step, nsteps)
    prog.two_streams = two
    if os.environ.get("YH_BENCH_LAYERS"):
        for ms, name, opname, flops, nbytes in sorted(per_op, reverse=True)[:int(os.environ["YH_BENCH_LAYERS"])]:
            print(f"# {ms:8.3f} ms/step  {name:36s} {opname:44s} {flops / 1e9 / max(ms, 1e-9):8.1f} TFLOP/s {nbytes / 1e6 / max(ms, 1e-9):8.1f} GB/s",
                  file=sys.stderr)
    work = {k: v for k, v in fam.items() if v["ms"] > 0 and (v["flops"] > 0 or v["bytes"] > 0)}
    if not work:
        return None
    pmc = _pmc_traffic()
    pmc_mfma = _pmc_file(_pmc_name("pmc_mfma"))
    dom = max(work, key=lambda k: work[k]["ms"])
    roof = _family_roofline(dom, work[dom], pmc, pmc_mfma)
    mfma_fams = {k: v for k, v in work.items() if v["flops"] > 0 and
                 v["flops"] / MFMA_PEAK_TFLOPS / 1e12 >= v["bytes"] / HBM_PEAK_GBS / 1e9}
    conv_dom = max(mfma_fams, key=lambda k: mfma_fams[k]["ms"]) if mfma_fams else None
    total_ms = sum(v["ms"] for v in fam.values()) / nsteps
    overlapped = None
    if two:
        fam2, _ = _instrument(prog, step, nsteps)
        d2 = fam2.get(dom)
        if d2 and d2["ms"] > 0:
            o = _family_roofline(dom, d2, None)
            overlapped = {"achieved": o["achieved"], "frac": o["frac"], "avg_launch_us": o["avg_launch_us"],
                          "engine_kernel_ms_per_step": round(sum(v["ms"] for v in fam2.values()) / nsteps, 3)}
    roof.update({"mode": "kernels back to back (side stream off)", "launches_per_step": work[dom]["launches"] // nsteps,
                 "overlapped": overlapped})
    if conv_dom is not None:
        roof["conv"] = _family_roofline(conv_dom, mfma_fams[conv_dom], pmc, pmc_mfma)
        roof["conv"]["ms_per_step"] = round(mfma_fams[conv_dom]["ms"] / nsteps, 3)
    # all MFMA work of the step against the time its kernels take (weighted mean over the conv / wgrad families)
    cf = sum(v["flops"] for v in work.values())
    cms = sum(v["ms"] for v in work.values() if v["flops"] > 0)
    roof["conv_kernels_tflops"] = round(cf / (cms * 1e-3) / 1e12, 2) if cms > 0 else None
    roof["algorithmic_hbm_bytes_per_step"] = round(sum(v["bytes"] for v in work.values()) / nsteps)
    # whole step (engine kernels + loss + optimizer + packing) as the PMC passes saw it; null when the file is stale or absent
    roof["hbm_bytes_per_step"] = round(pmc["hbm_bytes_per_step"]) if pmc is not None and pmc.get("hbm_bytes_per_step") else None
    roof["mfma_util_step"] = pmc_mfma.get("mfma_util_all_kernels") if pmc_mfma else None
    roof["family_ms_per_step"] = {k: round(v["ms"] / nsteps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
    roof["groups"] = _groups(fam, nsteps, pmc)
    if os.environ.get("YH_BENCH_FOREIGN", "1") != "0":
        roof["groups"]["foreign"] = _foreign_launches(step)
    roof["engine_kernel_ms_per_step"] = round(total_ms, 3)
    return roof


if __name__ == "__main__":
    main()
