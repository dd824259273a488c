"""Multi-GPU layout of the path: images are independent through every stage, so a batch shards
across ranks with NO data-path collective; the only exchange is one all-reduce (sum) of a few float64 accumulators
(rate, squared error, dead maps, image count), or an all-gather of per-image values when a bit-identical
`numpy.mean` over images is wanted (reconstructing_eae_kodak.py:810-815 uses numpy's pairwise summation).

Works with any initialised `torch.distributed` backend: "nccl" (= RCCL over xGMI) with device tensors on the GPU box,
"gloo" with host tensors in the CPU tests.
"""
import numpy
import torch
import torch.distributed as dist


def shard_bounds(nb_images, rank, world_size):
    """Contiguous block [start, stop) of `nb_images` owned by `rank`; the first `nb_images % world_size` ranks get one
    more image (ragged batches allowed)."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError('`rank` must belong to [0, `world_size`).')
    if nb_images < 0:
        raise ValueError('`nb_images` is negative.')
    (q, r) = divmod(nb_images, world_size)
    start = rank*q + min(rank, r)
    return (start, start + q + (1 if rank < r else 0))


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reduce_statistics(local_statistics, device=None):
    """Sum of a 1D float64 vector over all ranks (the path's single collective). Returns a numpy float64 array."""
    local = numpy.ascontiguousarray(local_statistics, dtype=numpy.float64)
    if _world() == 1:
        return local.copy()
    t = torch.from_numpy(local.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_per_image(local_values, nb_images_total, device=None):
    """All-gather of per-image float64 values in global image order (exact-parity mode): every rank returns the full
    (nb_images_total, k) array, so `numpy.mean(..., axis=0)` is bit-identical to a single-process run."""
    local = numpy.ascontiguousarray(local_values, dtype=numpy.float64)
    if local.ndim == 1:
        local = local[:, None]
    world = _world()
    if world == 1:
        return local.copy()
    k = local.shape[1]
    biggest = -(-nb_images_total//world)
    padded = numpy.zeros((biggest, k), dtype=numpy.float64)
    padded[:local.shape[0]] = local
    t = torch.from_numpy(padded)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = []
    for (r, part) in enumerate(parts):
        (start, stop) = shard_bounds(nb_images_total, r, world)
        out.append(part.cpu().numpy()[:stop - start])
    return numpy.concatenate(out, axis=0)
