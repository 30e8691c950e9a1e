"""Distributed helpers — mirror of the reference's utils/dist.py / utils/allreduce_norm.py surface that the
drivers import (get_rank, get_world_size, get_local_rank, synchronize, all_reduce_norm, get_num_devices),
ROCm/CPU safe (no nvidia-smi), plus the data-parallel gradient exchange of the HIP models.

One process per GPU; torch.distributed backend "nccl" is RCCL on ROCm (xGMI inside a node); the CPU tests
use "gloo".  The reference wraps the model in DDP (train_yolov5.py:219-220): gradients are AVERAGED over
ranks, BatchNorm statistics stay per-rank during training and are averaged only before evaluation
(utils/allreduce_norm.py:56-98).  Here the engine hands over ONE flat fp32 gradient buffer per backward,
so the exchange is a few large all-reduces of contiguous slices, issued while the backward is still running
(buckets follow the backward order of the layers), instead of DDP's per-parameter bucket hooks.
"""
import os

import torch
import torch.distributed as dist

__all__ = ['get_rank', 'get_world_size', 'get_local_rank', 'get_local_size', 'get_num_devices', 'synchronize', 'is_main_process',
           'all_reduce_norm', 'DataParallelGrads', 'allreduce_flat_mean', 'quiet_stdout']


def _on():
    return dist.is_available() and dist.is_initialized()


class quiet_stdout:
    """`with quiet_stdout():` sends file descriptor 1 to stderr for the duration: RCCL prints a version banner on stdout when a
    process's first communicator comes up, and a driver that prints a machine-readable line (bench.py) wants stdout to itself"""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        import sys
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def get_world_size():
    return dist.get_world_size() if _on() else 1


def get_rank():
    return dist.get_rank() if _on() else 0


_LOCAL_PROCESS_GROUP = None      # set by utils.launch for the ranks of one machine (utils/dist.py:32, utils/launch.py:121-128)


def get_local_rank():
    """rank inside this machine: the local process group when launch() made one, else torchrun's LOCAL_RANK"""
    if _LOCAL_PROCESS_GROUP is not None and _on():
        return dist.get_rank(group=_LOCAL_PROCESS_GROUP)
    return int(os.environ.get("LOCAL_RANK", "0"))


def get_local_size():
    if _LOCAL_PROCESS_GROUP is not None and _on():
        return dist.get_world_size(group=_LOCAL_PROCESS_GROUP)
    return int(os.environ.get("LOCAL_WORLD_SIZE", "1"))


def is_main_process():
    return get_rank() == 0


def get_num_devices():
    """utils/dist.py:34-41 without nvidia-smi: visible devices from the environment or the runtime"""
    vis = os.environ.get("CUDA_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES")
    if vis:
        return len([v for v in vis.split(",") if v != ""])
    return torch.cuda.device_count()


def synchronize():
    """barrier across ranks (utils/dist.py:66-79)"""
    if _on() and dist.get_world_size() > 1:
        dist.barrier()


def allreduce_flat_mean(flat, group=None, chunks=1):
    """in-place mean over ranks of one flat tensor; `chunks` > 1 issues several async collectives so that
    the tail of the buffer can overlap with whatever the caller still computes"""
    world = dist.get_world_size(group) if _on() else 1
    if world == 1 and not (_on() and os.environ.get("YH_FORCE_DP")):     # YH_FORCE_DP: run the 1-rank communicator too
        return []
    n = flat.numel()
    step = (n + chunks - 1) // chunks
    works = []
    for o in range(0, n, step):
        part = flat[o:o + step]
        works.append((dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group, async_op=True), part))
    for w, part in works:
        w.wait()
        part.div_(world)
    return works


class DataParallelGrads:
    """Averages the engine's flat gradient over the ranks during / right after each backward (DDP semantics:
    mean of per-rank gradients; no_sync() skips the exchange on accumulation steps, train_yolov5.py:327).

    Gradient accumulation (train_yolov5.py:327-337: non-boundary micro-steps run under no_sync, the boundary step
    synchronises) follows DDP: what is averaged at the boundary is the ACCUMULATED gradient.  The engine produces one
    fresh flat gradient per backward, so the un-exchanged ones are summed here (`_local_acc`); at the boundary the
    bucket collectives are skipped, the total (local sum + this backward) is mean-all-reduced once, and the flat
    gradient handed on to the optimizer / autograd is replaced by  mean(total) - local sum,  so that whatever
    accumulates downstream (FlatSGD._on_grad or p.grad +=) ends at mean(total) on every rank.

    bucket_dtype=torch.bfloat16 exchanges the buckets in bf16 (half the xGMI bytes; the sum of `world` bf16 values
    is rounded once per hop — use for bandwidth-bound models such as YOLOv5l/x, keep fp32 for parity runs)."""

    def __init__(self, model, group=None, chunks=2, overlap=True, bucket_dtype=None):
        self.model, self.group, self.chunks = model, group, chunks
        self.enabled = True
        self.overlap = overlap
        self.bucket_dtype = bucket_dtype
        self._local_acc = None
        model._yh_grad_hook = self._post
        if overlap:
            # the engine calls this per finished gradient bucket DURING the backward (engine.Program.backward):
            # the collective runs on RCCL's stream while the remaining dgrad/wgrad kernels keep the CUs busy
            model._yh_bucket_hook = self._bucket

    def _world(self):
        return dist.get_world_size(self.group) if _on() else 1

    def _post(self, flat_g, bucketed=False):
        """called by the engine with the complete flat gradient of one backward (after the bucket finishers)"""
        if not self.enabled:
            self._local_acc = flat_g.clone() if self._local_acc is None else self._local_acc.add_(flat_g)
            return
        if self._local_acc is not None:
            total = self._local_acc + flat_g
            allreduce_flat_mean(total, self.group, self.chunks)
            opt = getattr(getattr(self.model, "_yh_grad_hook_opt", None), "__self__", None)
            if opt is not None and hasattr(opt, "replace_accumulated"):
                # flat-arena optimizer: its accumulator (== _local_acc) is REPLACED by the averaged total, so every rank
                # steps with bit-identical gradients (replicas stay identical, as under DDP)
                flat_g.copy_(total)
                opt.replace_accumulated()
            else:
                # autograd accumulates into p.grad: hand on  mean(total) - local sum  (equal up to one rounding)
                torch.sub(total, self._local_acc, out=flat_g)
            self._local_acc = None
            return
        if not bucketed:
            allreduce_flat_mean(flat_g, self.group, self.chunks)

    def _bucket(self, part):
        """async all-reduce of one contiguous slice of the packed gradient arena; returns the finisher that
        makes the compute stream wait for it and turns the sum into the mean"""
        world = self._world()
        single = world == 1 and not os.environ.get("YH_FORCE_DP")
        if (not self.enabled) or (self._local_acc is not None) or single or part.numel() == 0:
            return None
        if part.is_cuda and dist.get_backend(self.group) == "nccl" and os.environ.get("YH_DP_COMM_STREAM", "1") != "0":
            return self._bucket_on_comm_stream(part)
        if self.bucket_dtype is not None and self.bucket_dtype != part.dtype:
            low = part.to(self.bucket_dtype)
            work = dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

            def finish_low():
                work.wait()
                # `low` was allocated in the caller's (side-stream) context and is consumed on the current stream
                if low.is_cuda:
                    low.record_stream(torch.cuda.current_stream())
                part.copy_(low)
                part.div_(world)
            return finish_low
        work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        def finish():
            work.wait()
            part.div_(world)
        return finish

    def _bucket_on_comm_stream(self, part):
        """RCCL: the bucket's all-reduce (ncclAvg) issued as a SYNCHRONOUS collective on a stream of our own choosing — torch runs
        a collective with async_op=False on the current stream — instead of ProcessGroupNCCL's internal stream.  Which hardware
        queue that internal stream shares is not ours to pick: on the compute stream's queue the collective, which waits for
        the weight-gradient stream, would stall the whole backward behind it (yoloseries_amd/streams.py).  `comm_stream` is probed to
        run beside both; it waits for the caller's stream (the engine calls from the weight-gradient stream's context, which
        it has ordered behind the compute stream), the finisher makes the compute stream wait for the exchange."""
        from ..streams import comm_stream
        dev = part.device
        comm = comm_stream(dev)
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        comm.wait_event(ready)
        with torch.cuda.stream(comm):
            if self.bucket_dtype is not None and self.bucket_dtype != part.dtype:
                low = part.to(self.bucket_dtype)
                dist.all_reduce(low, op=dist.ReduceOp.AVG, group=self.group)
                part.copy_(low)
            else:
                dist.all_reduce(part, op=dist.ReduceOp.AVG, group=self.group)
            done.record(comm)

        def finish():
            torch.cuda.current_stream(dev).wait_event(done)
        return finish

    def reset(self):
        """forget un-exchanged accumulation steps (the caller dropped their gradients, e.g. optimizer.zero_grad() mid-cycle)"""
        self._local_acc = None

    @property
    def buckets_active(self):
        """True when the next backward exchanges its gradient bucket by bucket; the engine's autograd node asks and otherwise runs
        the backward without bucket segmentation (no_sync and accumulation-boundary steps: every hook would return None)"""
        return self.overlap and self.enabled and self._local_acc is None

    class _NoSync:
        def __init__(self, dp):
            self.dp = dp

        def __enter__(self):
            self.prev, self.dp.enabled = self.dp.enabled, False

        def __exit__(self, *a):
            self.dp.enabled = self.prev

    def no_sync(self):
        return DataParallelGrads._NoSync(self)


def all_reduce_norm(module):
    """average every BatchNorm state (weight, bias, running_mean, running_var) over the ranks before
    evaluation — utils/allreduce_norm.py:56-98.  With the engine's arenas this is one collective on the
    float-buffer arena plus one on the gathered affine parameters.  The reference also averages `num_batches_tracked`
    (utils/allreduce_norm.py:32-38 walks the whole state_dict): every rank counts the same forwards, so its mean is the
    value each rank already holds and it is left alone here."""
    if get_world_size() == 1 and not (_on() and os.environ.get("YH_FORCE_DP")):
        return
    states = []
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.InstanceNorm2d)):
            states += [m.weight.data, m.bias.data, m.running_mean.data, m.running_var.data]
    if not states:
        return
    flat = torch.cat([s.reshape(-1).float() for s in states])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= get_world_size()
    o = 0
    for s in states:
        s.copy_(flat[o:o + s.numel()].view(s.shape))
        o += s.numel()
