"""HIP streams that really run BESIDE the compute stream.

The HIP runtime multiplexes a process's streams onto a handful of hardware queues (four by default); a new stream joins the
least-used queue, which may be the compute stream's own — then the "side" stream's kernels simply queue up between the compute
stream's, and a packet that waits for another queue stalls everything behind it.  Which queue a stream lands on depends on how
many other streams the process has touched before (torch's pools, RCCL's internals): measured on an MI355X, the engine's
weight-gradient stream shared the compute stream's queue as soon as a process group had been initialised, and the train step
lost its two-stream overlap (11.6 -> 13.8 ms, profiles/r05_dp_path.txt).  So a stream is not taken on trust: candidates are
probed — a long fill on every stream it must run beside, a tiny one on the candidate, device timestamps compared — until one
finishes its tiny kernel while the long ones are still running.
"""
import os
import warnings

import torch

_PROBE_BYTES = 128 << 20      # filled _PROBE_FILLS times per probe: ~1.5 ms of device time against 128 MiB of memory
_PROBE_FILLS = 24
_cache = {}


def _runs_beside(cand, others, big, tiny):
    """does a kernel on `cand` finish while long kernels on each of `others` are still running?"""
    dev = big.device
    torch.cuda.synchronize(dev)
    for other in others:
        e0, e1, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(other):
            e0.record()
            for _ in range(_PROBE_FILLS):
                big.zero_()
            e1.record()
        with torch.cuda.stream(cand):
            tiny.zero_()
            c1.record()
        torch.cuda.synchronize(dev)
        if e0.elapsed_time(c1) > 0.5 * e0.elapsed_time(e1):
            return False
    return True


def concurrent_stream(dev, beside, priority=0, tries=8):
    """a stream of `dev` whose kernels run concurrently with those of every stream in `beside` (never during a graph capture:
    the probe synchronises the device).  Tries `tries` streams of the asked priority, then of the other priorities."""
    dev = torch.device(dev)
    if os.environ.get("YH_STREAM_PROBE", "1") == "0":
        return torch.cuda.Stream(device=dev, priority=priority)
    if torch.cuda.is_current_stream_capturing():
        # the probe synchronises the device: not inside a capture.  A fresh stream, unprobed (side_stream() does not cache it).
        warnings.warn("yoloseries_amd: stream probe skipped inside a graph capture; run one eager step before capturing",
                      RuntimeWarning, stacklevel=2)
        return torch.cuda.Stream(device=dev, priority=priority)
    try:
        return _probe(dev, beside, priority, tries)
    except (torch.cuda.OutOfMemoryError, RuntimeError) as e:
        # a memory-tight configuration (the probe buffer) or a HIP error of the probe itself must not end a training run
        warnings.warn(f"yoloseries_amd: stream probe failed ({type(e).__name__}: {str(e)[:120]}); using an unprobed stream",
                      RuntimeWarning, stacklevel=2)
        return torch.cuda.Stream(device=dev, priority=priority)


def _probe(dev, beside, priority, tries):
    big = torch.empty(_PROBE_BYTES, dtype=torch.uint8, device=dev)
    tiny = torch.empty(256, dtype=torch.uint8, device=dev)
    keep, found = [], None            # rejected candidates stay referenced until the choice is made, so the pool moves on
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    prios = [priority] + [p for p in range(min(lo, hi), max(lo, hi) + 1) if p != priority]
    for prio in prios:
        for _ in range(tries):
            try:
                cand = torch.cuda.Stream(device=dev, priority=prio)
            except (RuntimeError, ValueError):
                break
            if all(cand.cuda_stream != s.cuda_stream for s in list(beside) + keep) and _runs_beside(cand, beside, big, tiny):
                found = cand
                break
            keep.append(cand)
        if found is not None:
            break
    del big, tiny
    if found is None:
        warnings.warn("yoloseries_amd: no HIP stream runs beside the compute stream in this process (every candidate shares its "
                      "hardware queue): the two-stream schedule will run as one queue; GPU_MAX_HW_QUEUES=8 in the environment "
                      "gives the runtime more queues", RuntimeWarning, stacklevel=2)
        # every kept candidate was SEEN to share the compute queue: a fresh stream is at least unknown
        found = torch.cuda.Stream(device=dev, priority=priority)
    return found


def _indexed(dev):
    dev = torch.device(dev)
    return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())


def _compute_streams(dev):
    """the device's default stream and — when the caller runs under another one — the current stream"""
    d, c = torch.cuda.default_stream(dev), torch.cuda.current_stream(dev)
    return [d] if c.cuda_stream == d.cuda_stream else [d, c]


def side_stream(dev):
    """the engine's weight-gradient stream of a device (one per process and device; YH_SIDE_PRIO: preferred HIP priority)"""
    dev = _indexed(dev)
    key = ("side", dev.index)
    if key not in _cache:
        capturing = torch.cuda.is_current_stream_capturing()
        st = concurrent_stream(dev, _compute_streams(dev), priority=int(os.environ.get("YH_SIDE_PRIO", "0")))
        if capturing:
            return st
        _cache[key] = st
    return _cache[key]


def comm_stream(dev):
    """the stream gradient buckets are exchanged on: beside the compute stream AND the weight-gradient stream"""
    dev = _indexed(dev)
    key = ("comm", dev.index)
    if key not in _cache:
        side = side_stream(dev)
        _cache[key] = concurrent_stream(dev, [s for s in _compute_streams(dev) if s.cuda_stream != side.cuda_stream] + [side])
    return _cache[key]
