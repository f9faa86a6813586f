"""Multi-GPU partitioning of LERC work.

A band blob is ONE sequential block stream with one global zMin / zMax and one checksum, so a single raster does
not split across GPUs.  What shards is the unit above it: independent rasters / mosaic tiles, each its own blob.
Ranks take contiguous tile ranges and never exchange pixel or blob data; the only collectives are metadata:
the per-tile blob sizes (4 bytes per tile) that turn into the offsets of a mosaic container, and the
max-over-ranks of the elapsed time that bench.py reports.  One process per GPU, torch.distributed ("nccl" is RCCL
on ROCm; the CPU tests run the same code over "gloo").
"""
import torch
import torch.distributed as dist


def tile_range(rank, world, n_tiles):
    """Contiguous share of `n_tiles` for `rank`: (first, count).  The first n_tiles % world ranks take one more."""
    if world <= 0 or not 0 <= rank < world or n_tiles < 0:
        raise ValueError((rank, world, n_tiles))
    base, extra = divmod(n_tiles, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def max_over_ranks(seconds, device="cpu"):
    """The slowest rank's time: what a whole-job throughput has to be computed from."""
    if _world() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_manifest(local_sizes, n_tiles, device="cpu"):
    """All ranks learn every tile's blob size.  local_sizes: this rank's sizes in tile order (its tile_range).
    Returns (sizes[n_tiles], offsets[n_tiles + 1]) as int64 CPU tensors: tile t lives at offsets[t] of the mosaic."""
    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    first, count = tile_range(rank, world, n_tiles)
    if len(local_sizes) != count:
        raise ValueError(f"rank {rank} owns {count} tiles, got {len(local_sizes)} sizes")
    sizes = torch.zeros(n_tiles, dtype=torch.int64, device=device)
    sizes[first:first + count] = torch.as_tensor(list(local_sizes), dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)    # disjoint ranges: the sum is the concatenation
    sizes = sizes.cpu()
    offsets = torch.zeros(n_tiles + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(sizes, 0)
    return sizes, offsets
