"""One process per GPU: bring up torch.distributed from the launcher's environment for the command-line entry points.

`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m <package>.A_score.compute ...` (likewise
C_score.pck_train, C_score.pck_train_two, C_score.extract_feature, llava.feature.extract) exports RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*; the score drivers shard their images / pairs over the ranks when a process group exists (no data-path collective, one
small all-reduce of the counters at the end).  Backend: nccl (= RCCL over xGMI) on GPU boxes, gloo where there is no GPU (tests).
Launched without a launcher this is a no-op and the entry points run single-process.
"""
import os


def init_from_env() -> bool:
    """Returns True when THIS call created the process group (the caller then owns its shutdown: finalize())."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or dist.is_initialized():
        return False
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver stack
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    return True


def finalize(owned: bool) -> None:
    import torch.distributed as dist
    if owned and dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
