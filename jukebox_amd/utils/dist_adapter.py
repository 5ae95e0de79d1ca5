"""Process-group shim with the reference's call surface (jukebox/utils/dist_adapter.py) that degrades to
rank 0 / world 1 when no group is initialised (the reference raises in that case)."""
import torch.distributed as dist


def _on():
    return dist.is_available() and dist.is_initialized()


def is_available():
    return dist.is_available()


def get_rank():
    return dist.get_rank() if _on() else 0


def get_world_size():
    return dist.get_world_size() if _on() else 1


def barrier():
    if _on():
        dist.barrier()


def broadcast(tensor, src):
    if _on():
        dist.broadcast(tensor, src)


def all_gather(tensor_list, tensor):
    if _on():
        dist.all_gather(tensor_list, tensor)
    else:
        tensor_list[0] = tensor
