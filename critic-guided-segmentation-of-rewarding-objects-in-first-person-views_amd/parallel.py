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


def resolve_dp_graph(dp_graph, world: int) -> bool:
    """The default launch form of the data-parallel step.  dp_graph None = "the safe default": the all-reduce is recorded in the step's HIP
    graph on a ONE-rank group (the rehearsal the GPU tests and `bench.py --force-pg` run) and stays OUTSIDE the graphs (step graph -> eager
    all-reduce -> Adam graph, +~17 us per step) when world > 1, because no N > 1 RCCL run has exercised the captured form yet (there is no
    second GPU on the build's boxes).  CGS_DP_GRAPH=1 / dp_graph=True opt in at world > 1 (the decision is then taken collectively by
    collective_capturable); dp_graph=False forces the eager form everywhere."""
    if dp_graph is None:
        env = os.environ.get("CGS_DP_GRAPH")
        if env is not None:
            return env not in ("", "0")
        return world <= 1
    return bool(dp_graph)


def agree_all(ok: bool, group, dev) -> bool:
    """Logical AND of a per-rank flag over the group (one eager MIN all-reduce; through the host for gloo): every rank gets the same
    answer.  Must be called by all ranks of the group at the same point of the program."""
    f = torch.tensor([1.0 if ok else 0.0], device="cpu" if dist.get_backend(group) == "gloo" else dev)
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
    return bool(f.item() == 1.0)


_agree_all = agree_all


def checksum64(t: torch.Tensor) -> torch.Tensor:
    """int64[2] checksum of a tensor's BITS (fp32 / int32 words, or int64): (sum of the words, position-weighted sum), wrap-around
    int64 arithmetic -- equal on two replicas iff (up to a 2^-64 collision) every bit is equal; reorderings change the second word."""
    t = t.detach().contiguous().reshape(-1)
    w = (t if t.dtype == torch.int64 else t.view(torch.int32).to(torch.int64))
    idx = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64)
    return torch.stack([w.sum(), (w * idx).sum()])


def replica_checksums(tensors, group, dev):
    """(identical, per-rank list of hex strings): all-gathers checksum64 of each tensor of `tensors` over the group and says whether every
    rank holds the same bits -- the data-parallel contract here is BIT-identical replicas (one all-reduced gradient, the same Adam
    arithmetic on every rank).  One collective; all ranks must call it together."""
    mine = torch.cat([checksum64(t) for t in tensors])
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        mine = mine.cpu()
    else:
        mine = mine.to(dev)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    rows = [[int(v) & 0xFFFFFFFFFFFFFFFF for v in c.tolist()] for c in allc]
    return all(r == rows[0] for r in rows), ["".join(f"{v:016x}" for v in r) for r in rows]


def collective_capturable(group, dev):
    """(ok, note): can a gradient all-reduce on `group` be recorded into a HIP graph?  The answer is COLLECTIVE -- every rank of the group
    returns the same `ok`, so the ranks can never end up in different launch forms (one replaying a captured collective while its peer issues
    an eager one).  Three phases, each closed by a MIN all-reduce of the per-rank flag:
      1. an eager all-reduce (communicator + channels set up outside any capture);
      2. CAPTURE of a trial all-reduce -- nothing is communicated while capturing, so a rank whose capture throws has no partner waiting;
         only if every rank captured,
      3. every rank REPLAYS its trial graph once and checks the sum.
    False for gloo (rehearsals stage through the host).  Must be called by all ranks of the group at the same point of the program."""
    if dist.get_backend(group) != "nccl":
        return False, f"backend {dist.get_backend(group)} reduces through the host"
    world = dist.get_world_size(group)
    t = torch.ones(64, device=dev)
    dist.all_reduce(t, group=group)                        # (1) -- a failure here is a broken job, not a capture question: let it raise
    torch.cuda.synchronize()
    g, note = None, None
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            dist.all_reduce(t, group=group)
    except Exception as e:       # noqa: BLE001 -- any failure means: keep the collective outside the graphs (on EVERY rank, below)
        g, note = None, f"trial capture failed on this rank: {type(e).__name__}: {e}"
    if not _agree_all(g is not None, group, dev):          # (2)
        return False, note or "trial capture failed on another rank"
    t.fill_(float(world))
    g.replay()
    torch.cuda.synchronize()
    got = float(t[0].item())
    good = bool(torch.isfinite(t).all().item()) and got == float(world) ** 2
    if not _agree_all(good, group, dev):                   # (3)
        return False, (f"trial replay gave {got}, expected {world ** 2}" if not good else "trial replay gave a wrong sum on another rank")
    return True, f"trial capture + replay ok on all {world} rank(s)"
