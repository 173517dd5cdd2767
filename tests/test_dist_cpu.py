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


def test_bench_launcher_world2_plan_only():
    """`python bench.py --gpus 2` end to end on CPU: bench.py re-executes itself under torch.distributed.run with two
    ranks (127.0.0.1), gloo rendezvous, rank 0 packs + broadcasts the weights, both ranks shard the job, build their
    plan, barrier, max-over-ranks timing, ONE JSON line from rank 0 with n_gpus = the ranks that reported."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--plan-only', '--tiny-net',
                          '--steps', '2', '--warmup', '0', '--pairs', '5'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['plan_only'] is True and r['value'] is None
    assert r['config']['pairs_sharded'] == 5 and r['config']['weights_identical_on_all_ranks'] is True
    assert r['config']['plan_ops'] > 20


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a box with fewer GPUs exits non-zero with a clear message instead of printing
    a 1-GPU number (this container has none)."""
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with fewer than 2 GPUs')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 2
    assert 'refusing' in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith('{')]
    # a torchrun world that disagrees with --gpus is refused as well
    env['WORLD_SIZE'] = '4'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--plan-only'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 2 and 'WORLD_SIZE=4' in out.stderr
