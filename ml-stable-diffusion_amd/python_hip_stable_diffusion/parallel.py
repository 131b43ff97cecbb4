"""Multi-GPU plumbing: one process per GPU, independent prompts sharded over ranks.

The reference has no parallelism at all; the path shards naturally
because every prompt (and each CFG half) is an independent UNet evaluation - GroupNorm /
LayerNorm are per-sample - so there is NO data-path collective.  RCCL (torch.distributed backend
"nccl" on ROCm; "gloo" in the CPU tests) only moves the prompt embeddings out at the start
(broadcast, 315 KB per prompt) and the results back at the end (all_gather, 64 KB of latents
per prompt).  Per-step inter-GPU traffic is zero.
"""
import numpy as np


def shard_prompts(n_prompts, world_size):
    """Contiguous balanced partition: rank r owns prompts shard_prompts(n, w)[r]."""
    base, extra = divmod(n_prompts, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(list(range(start, start + n)))
        start += n
    return out


def _device(dist, local_rank):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", local_rank)
    return torch.device("cpu")


def broadcast_array(arr, shape, dtype, dist, local_rank=0, src=0):
    """Rank `src` passes the array, the others pass None; everyone gets a numpy copy."""
    if dist is None:
        return np.asarray(arr, dtype=dtype).reshape(shape)
    import torch
    dev = _device(dist, local_rank)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype).reshape(shape)).to(dev)
    else:
        t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_arrays(arr, dist, local_rank=0):
    """all_gather equal-shaped per-rank arrays along axis 0 (rank order)."""
    if dist is None:
        return np.asarray(arr)
    import torch
    dev = _device(dist, local_rank)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return torch.cat(outs, dim=0).cpu().numpy()
