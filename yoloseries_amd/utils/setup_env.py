"""Process-environment set-up — mirror of utils/setup_env.py:16-74 (configure_nccl / configure_omp / configure_module)
for a one-node MI355X job: backend "nccl" is RCCL on ROCm and the ranks talk over xGMI, so the InfiniBand probing of the
reference (a shell pipeline over /sys/class/infiniband) is not run; the variables it sets that RCCL also reads are kept."""
import os

from .dist import get_world_size, is_main_process

__all__ = ["configure_nccl", "configure_module", "configure_omp"]


def configure_nccl():
    """utils/setup_env.py:16-28: loop-back rendezvous sockets, no IB transport, parallel launch mode"""
    os.environ["NCCL_SOCKET_IFNAME"] = "lo"
    os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    os.environ["NCCL_IB_DISABLE"] = "1"
    os.environ["NCCL_LAUNCH_MODE"] = "PARALLEL"
    # the host driver of this pool supports dmabuf IPC only: RCCL / device-tensor sharing need it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def configure_omp(num_threads=1):
    """one OpenMP thread per rank unless the user chose otherwise (utils/setup_env.py:30-50)"""
    if "OMP_NUM_THREADS" not in os.environ and get_world_size() > 1:
        os.environ["OMP_NUM_THREADS"] = str(num_threads)
        if is_main_process():
            print(f"OMP_NUM_THREADS set to {num_threads} for each process; tune it for the data loader if needed")


def configure_module(ulimit_value=8192):
    """raise the open-file limit (many loader workers) and keep OpenCV single-threaded when it is installed
    (utils/setup_env.py:53-74)"""
    try:
        import resource
        soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        resource.setrlimit(resource.RLIMIT_NOFILE, (min(ulimit_value, hard), hard))
    except Exception:
        pass
    os.environ["OPENCV_OPENCL_RUNTIME"] = "disabled"
    try:
        import cv2
        cv2.setNumThreads(0)
        cv2.ocl.setUseOpenCL(False)
    except Exception:
        pass
