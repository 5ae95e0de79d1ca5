"""Distributed bootstrap for sampling: one process per GPU, ranks from the torchrun environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), backend "nccl" (= RCCL on ROCm) or "gloo" on CPU.
Replaces the reference's mpi4py bootstrap (jukebox/utils/dist_utils.py:42-101).

Sampling shards `n_samples` across ranks: samples are independent given their labels, so the only collectives
are a broadcast of the conditioning (labels / primed codes) from rank 0 and an all_gather of the generated
codes per level -- both outside the token loop."""
import os

import torch
import torch.distributed as dist

from . import dist_adapter


def print_once(msg):
    if dist_adapter.get_rank() == 0:
        print(msg)


def print_all(msg):
    if dist_adapter.get_rank() % 8 == 0:
        print(f"{dist_adapter.get_rank() // 8}: {msg}")


def setup_dist_from_env(backend=None):
    """Returns (rank, local_rank, device).  No-op single process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    device = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("JB_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
        dist.init_process_group(backend, init_method="env://", rank=rank, world_size=world)
    return rank, local_rank, device


def shard_range(n_samples, rank=None, world=None):
    """Contiguous slice [lo, hi) of the global sample indices owned by `rank` (the first n % world ranks get one
    extra sample)."""
    rank = dist_adapter.get_rank() if rank is None else rank
    world = dist_adapter.get_world_size() if world is None else world
    base, rem = divmod(n_samples, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged():
    """gloo cannot move GPU tensors: stage through the host (tests on a single-GPU box; RCCL moves them directly)."""
    return dist.get_backend() == "gloo"


def broadcast_tensor(x, src=0):
    """Broadcast a tensor whose shape/dtype the other ranks may not know yet (rank `src` passes the tensor, the
    others pass None).  Used for labels `y` and primed codes `zs`."""
    if dist_adapter.get_world_size() == 1:
        return x
    rank = dist_adapter.get_rank()
    dev = x.device if x is not None else (torch.device("cuda", torch.cuda.current_device())
                                          if torch.cuda.is_available() else torch.device("cpu"))
    meta = torch.zeros(8, dtype=torch.int64, device="cpu" if _host_staged() else dev)
    if rank == src:
        assert x.dtype in (torch.int64, torch.float32)
        meta[0], meta[1] = x.dim(), 0 if x.dtype == torch.int64 else 1
        meta[2:2 + x.dim()] = torch.tensor(list(x.shape), dtype=torch.int64)
    dist.broadcast(meta, src)
    nd, code = int(meta[0]), int(meta[1])
    shape = [int(v) for v in meta[2:2 + nd]]
    if rank != src:
        x = torch.empty(shape, dtype=torch.int64 if code == 0 else torch.float32, device=dev)
    x = x.contiguous()
    if _host_staged() and x.is_cuda:
        h = x.cpu()
        dist.broadcast(h, src)
        return h.to(x.device)
    dist.broadcast(x, src)
    return x


def gather_shards(x_local, n_samples):
    """all_gather of per-rank sample shards (dim 0, possibly uneven) -> the full (n_samples, ...) tensor on every
    rank.  Codes are a few MB per level, so one all_gather per level is enough."""
    world = dist_adapter.get_world_size()
    if world == 1:
        return x_local
    sizes = [shard_range(n_samples, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    dev = x_local.device
    staged = _host_staged() and x_local.is_cuda
    pad = torch.zeros((mx, *x_local.shape[1:]), dtype=x_local.dtype, device="cpu" if staged else dev)
    pad[: x_local.shape[0]] = x_local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0).to(dev)
