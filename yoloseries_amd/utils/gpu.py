"""Device-memory helpers — mirror of utils/gpu.py:14-62 without nvidia-smi: totals come from the HIP runtime
(torch.cuda.mem_get_info), so they are per-process-visible numbers of THIS rank's MI355X."""
import random
import time

import numpy as np
import torch

__all__ = ["get_total_and_free_memory_in_Mb", "occupy_mem", "gpu_mem_usage", "init_seed"]


def init_seed(seed, cuda_deterministic=True):
    """utils/gpu.py:14-23 (MIOpen is not on the hot path here; the flags are set for API compatibility)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.backends.cudnn.deterministic = bool(cuda_deterministic)
    torch.backends.cudnn.benchmark = not cuda_deterministic


def get_total_and_free_memory_in_Mb(cuda_device):
    """(total MiB, used MiB) of device `cuda_device` — the reference's return order (utils/gpu.py:26-33 returns total, used)"""
    free, total = torch.cuda.mem_get_info(cuda_device)
    return int(total // 2 ** 20), int((total - free) // 2 ** 20)


def occupy_mem(cuda_device, mem_ratio=0.9):
    """grab `mem_ratio` of the device once so the caching allocator owns one large segment (utils/gpu.py:36-47).  On a
    288 GB part the engine allocates every buffer once per input shape, so this is optional; kept for the drivers."""
    total, used = get_total_and_free_memory_in_Mb(cuda_device)
    block_mem = int(total * mem_ratio) - used
    if block_mem <= 0:
        return
    x = torch.empty(block_mem * 2 ** 20, dtype=torch.uint8, device=torch.device("cuda", int(cuda_device)))
    del x
    time.sleep(0.1)


def gpu_mem_usage():
    """peak allocated MiB on the current device (utils/gpu.py:55-62)"""
    return torch.cuda.max_memory_allocated() / (1024 * 1024)
