"""Static sharding of a batch of independent images over GPUs (no collective on
the data path).  The reference parallelises the same way, one image per pool
thread over the sorted file list (src/main.cpp:538-544, 624-655); here block g of
the sorted list goes to GPU g, and inside a GPU images round-robin over streams."""


def static_blocks(n_items, world):
    """Contiguous blocks of ceil(n/world) items: [(begin, end)] per rank."""
    if world < 1:
        raise ValueError("world must be >= 1")
    per = -(-n_items // world) if n_items > 0 else 0
    return [(min(r * per, n_items), min((r + 1) * per, n_items)) for r in range(world)]


def my_block(n_items, world, rank):
    return static_blocks(n_items, world)[rank]


def stream_of(local_index, n_streams):
    return local_index % n_streams


def max_over_ranks(seconds, dist=None, device=None, always=False):
    """Whole-job time = slowest rank (bench.py contract).  `always`: run the reduction on a world-size-1 group too
    (bench.py --force-dist: the one RCCL collective a one-GPU box can execute)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return seconds
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
