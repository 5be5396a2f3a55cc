"""Data parallelism for the training step: one process per GPU, identical replicas, each rank works on its
own shard of the batch; the ONLY exchange is a sum all-reduce of the flat gradient buffer (25 661 floats =
103 KB) between the slab reduction and Adam, which reads it scaled by 1/world.  The reference has no
distributed code; this follows SURVEY.md section 8(e).  Backend "nccl" is RCCL on ROCm (xGMI inside a node);
"gloo" is used by the CPU tests."""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend: Optional[str] = None):
    """Initialises the default process group when WORLD_SIZE > 1; returns it (or None for a single process)."""
    rank, local, world = env_world()
    if world <= 1:
        return None
    # dmabuf IPC only on this driver.  The HIP runtime reads this at initialisation: launchers should export it themselves
    # (bench.py sets it before importing torch); setting it here only helps when nothing has touched the GPU yet.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return dist.group.WORLD


def shard_slice(n_total: int, rank: int, world: int) -> slice:
    """Contiguous, equal shards (the step's losses are batch means, so equal shards make the mean of the
    per-rank gradients equal the full-batch gradient)."""
    if n_total % world:
        raise ValueError(f"batch {n_total} does not split evenly over {world} ranks")
    per = n_total // world
    return slice(rank * per, (rank + 1) * per)


def _host_staged(flat: torch.Tensor, group) -> bool:
    """gloo rehearsals (tests: several ranks on one GPU, or CPU only) move device buffers through the host;
    the production backend (nccl == RCCL) reduces device memory in place over xGMI."""
    return flat.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum all-reduce of one flat bucket (single collective: latency-bound at this size)."""
    if group is not None:
        if _host_staged(flat, group):
            tmp = flat.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(tmp)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_params_(flat: torch.Tensor, group=None, src: int = 0) -> torch.Tensor:
    """Makes every replica start from rank ``src``'s parameters."""
    if group is not None and dist.get_world_size(group) > 1:
        if _host_staged(flat, group):
            tmp = flat.cpu()
            dist.broadcast(tmp, src=src, group=group)
            flat.copy_(tmp)
        else:
            dist.broadcast(flat, src=src, group=group)
    return flat


def collective_capturable(group, dev):
    """(ok, note): can a gradient all-reduce on `group` be recorded into a HIP graph?  True for the RCCL backend on device memory when a
    trial capture + replay of a small all-reduce on this very group succeeds (gloo rehearsals stage through the host)."""
    if dist.get_backend(group) != "nccl":
        return False, f"backend {dist.get_backend(group)} reduces through the host"
    world = dist.get_world_size(group)
    try:
        t = torch.ones(64, device=dev)
        dist.all_reduce(t, group=group)                    # communicator + channels set up outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            dist.all_reduce(t, group=group)
        g.replay()
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(t).all().item()) and float(t[0].item()) == float(world) ** 2
        return ok, ("trial capture + replay ok" if ok else f"trial replay gave {float(t[0].item())}, expected {world ** 2}")
    except Exception as e:       # noqa: BLE001 -- any failure means: keep the collective outside the graphs
        return False, f"trial capture failed: {type(e).__name__}: {e}"

