"""world_size-2 gloo tests of the N>1 path (runs on CPU): unit sharding and the one-time weight broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def test_shard_range_covers_everything_once():
    from film_hip.sharding import shard_range, tiles_of_rank
    for n in (0, 1, 7, 16, 63, 64):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = shard_range(n, world, r)
                assert 0 <= b <= e <= n and e - b in (n // world, n // world + 1)
                seen += list(range(b, e))
            assert seen == list(range(n))
    assert tiles_of_rank([4, 4], 8, 3) == [6, 7]          # config 5: 16 tiles over 8 GPUs
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    from film_hip.sharding import broadcast_weights, shard_range
    eng = FilmEngine(TINY, device=-1)          # plan-only handles: packing + import/export, no compute
    if rank == 0:
        eng.set_weights(W.make_synthetic_weights(TINY, seed=0))
    broadcast_weights(eng, dist, src=0)
    blob = eng.export_packed()
    digest = torch.tensor([float(np.abs(blob).sum()), float(blob[::97].sum())], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    # each rank also reports its share of a 5-pair job
    b, e = shard_range(5, world, rank)
    mine = torch.tensor([e - b], dtype=torch.int64)
    dist.all_reduce(mine)
    ok = all(torch.equal(g, gathered[0]) for g in gathered) and int(mine.item()) == 5 and float(digest[0]) > 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
