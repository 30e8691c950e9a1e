"""launch() — mirror of utils/launch.py:39-139: start one training process per GPU of this machine and run
`main_func(*args)` in each with torch.distributed initialised (backend "nccl" = RCCL over xGMI on MI355X).

The children are created with the `spawn` start method BEFORE anything in the parent touches the GPU, and the parent never
replaces itself with another program (a process that has initialised HIP must not exec): the parent only waits."""
import os
import socket
from datetime import timedelta

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from . import dist as comm

__all__ = ["launch"]

DEFAULT_TIMEOUT = timedelta(minutes=30)


def _find_free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, backend="nccl", dist_url=None, args=(),
           timeout=DEFAULT_TIMEOUT):
    """run main_func(*args) on num_machines x num_gpus_per_machine ranks; a single rank runs in this process"""
    world_size = num_machines * num_gpus_per_machine
    if world_size <= 1:
        main_func(*args)
        return
    if dist_url == "auto" or dist_url is None:
        assert num_machines == 1, "dist_url=auto cannot work with distributed training."
        dist_url = f"tcp://127.0.0.1:{_find_free_port()}"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.start_processes(_distributed_worker, nprocs=num_gpus_per_machine,
                       args=(main_func, world_size, num_gpus_per_machine, machine_rank, backend, dist_url, args, timeout),
                       daemon=False, start_method="spawn")


def _distributed_worker(local_rank, main_func, world_size, num_gpus_per_machine, machine_rank, backend, dist_url, args,
                        timeout=DEFAULT_TIMEOUT):
    global_rank = machine_rank * num_gpus_per_machine + local_rank
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["RANK"] = str(global_rank)
    os.environ["WORLD_SIZE"] = str(world_size)
    kwargs = {}
    if backend == "nccl":
        assert torch.cuda.is_available(), "no MI355X device visible to this rank"
        assert num_gpus_per_machine <= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        kwargs["device_id"] = torch.device("cuda", local_rank)
    try:
        with comm.quiet_stdout():          # RCCL's version banner goes to stderr
            dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=global_rank, timeout=timeout, **kwargs)
    except Exception as err:
        print(f"Process group URL: {dist_url} {err}")
        raise
    # the ranks of one machine form the local process group (get_local_rank / get_local_size)
    assert comm._LOCAL_PROCESS_GROUP is None
    for i in range(world_size // num_gpus_per_machine):
        pg = dist.new_group(list(range(i * num_gpus_per_machine, (i + 1) * num_gpus_per_machine)))
        if i == machine_rank:
            comm._LOCAL_PROCESS_GROUP = pg
    comm.synchronize()
    try:
        main_func(*args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
