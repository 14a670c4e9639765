"""CPU, world_size 2 and 8 over gloo: the N > 1 path of bench.py — static image
sharding with no data-path collective, and the max-over-ranks timing reduce."""
import importlib
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sharding = importlib.import_module("image-lens-reproject_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = sharding.my_block(n_items, world, rank)
    mine = list(range(b, e))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)  # test-only: proves the shards partition the batch
    t = sharding.max_over_ranks(1.0 + rank, dist)
    q.put((rank, gathered, t))
    dist.barrier()
    dist.destroy_process_group()


def test_static_blocks_partition():
    sharding = importlib.import_module("image-lens-reproject_amd.sharding")
    for n in (0, 1, 7, 8, 255, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            blocks = sharding.static_blocks(n, world)
            flat = [i for b, e in blocks for i in range(b, e)]
            assert flat == list(range(n))
            assert max(e - b for b, e in blocks) <= -(-n // world) if n else True
    assert sharding.static_blocks(256, 8) == [(32 * i, 32 * (i + 1)) for i in range(8)]
    assert [sharding.stream_of(i, 3) for i in range(5)] == [0, 1, 2, 0, 1]


def test_two_ranks_gloo():
    world, n_items = 2, 257
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, t in results:
        flat = [i for part in gathered for i in part]
        assert flat == list(range(n_items))
        assert t == 2.0  # max over ranks of (1.0, 2.0)


def test_eight_ranks_gloo_uneven_shards():
    """The world size the driver's scaling run ends at (8 GPUs): 8 gloo ranks, a batch that does not divide (40 images:
    blocks of 5) and one that leaves the last ranks short (43: 6 x 7 + 1 + 0): the blocks partition the batch in order,
    and the whole-job time is the slowest rank's."""
    world = 8
    for n_items in (40, 43):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + (os.getpid() + n_items) % 2000
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert sorted(r for r, _g, _t in results) == list(range(world))
        for rank, gathered, t in results:
            assert [i for part in gathered for i in part] == list(range(n_items))
            assert max(len(part) for part in gathered) == -(-n_items // world)
            assert t == 8.0  # max over ranks of 1.0 .. 8.0
