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


# ---------------------------------------------------------------------------------------------------------------------
# one frame pair, tiles sharded over the ranks (film_hip.sharding.TileShardedRecursion) - SURVEY 8(e)
# ---------------------------------------------------------------------------------------------------------------------
_TS_H, _TS_W, _TS_BLOCK, _TS_T, _TS_ALIGN = 44, 56, [2, 2], 2, 8     # tiles of 22x28, each padded to 24x32


def _oracle_batch_fn(weights):
    """batch_fn of the driver with the CPU oracle (TINY net) as the model: per element pad to align, forward, crop -
    Interpolator.interpolate (eval/interpolator.py:152-176), what the reference's tile loop calls per patch."""
    from conftest import oracle_options
    from film_hip.options import TINY
    from oracle import film_oracle as fo
    it = fo.OracleInterpolator(weights, align=_TS_ALIGN, block_shape=None, opt=oracle_options(TINY))
    dt = np.full((1,), 0.5, np.float32)

    def fn(a, b):
        out = [it.interpolate(a[i:i + 1].numpy(), b[i:i + 1].numpy(), dt) for i in range(a.shape[0])]
        return torch.from_numpy(np.concatenate(out, axis=0))
    return fn


def _tile_worker(rank, world, port, q):
    import sys
    for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from film_hip import weights as W
    from film_hip.options import TINY
    from film_hip.sharding import TileShardedRecursion
    import inputs
    x0, x1 = inputs.frame_pair(1, _TS_H, _TS_W, seed=23)
    drv = TileShardedRecursion(_oracle_batch_fn(W.make_synthetic_weights(TINY, seed=0)), _TS_BLOCK, dist)
    seq = drv.run(torch.from_numpy(x0[0]), torch.from_numpy(x1[0]), _TS_T)
    q.put((rank, drv.tiles, None if seq is None else seq.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_tile_sharded_pair_world_n_gloo(world, tiny_weights):
    """ONE frame pair, its 2x2 tiles dealt over `world` gloo ranks for the whole T = 2 recursion tree: the tile sets are
    disjoint and complete (world 3: shares of 2, 1, 1 - the padded gather), only rank 0 gets frames, and they are
    byte-identical to (a) the same driver on one rank and (b) the REFERENCE-ORDER depth-first recursion
    (eval/util.py:62-91) over the tiled OracleInterpolator (eval/interpolator.py:192-206)."""
    import inputs
    from conftest import oracle_options
    from eval import util
    from film_hip.options import TINY
    from film_hip.sharding import TileShardedRecursion
    from oracle import film_oracle as fo
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tiles = [t for _, ts, _ in res for t in ts]
    assert tiles == [0, 1, 2, 3]
    assert all(seq is None for r, _, seq in res if r != 0)
    got = res[0][2]
    assert got.shape == (2 ** _TS_T + 1, _TS_H, _TS_W, 3)
    x0, x1 = inputs.frame_pair(1, _TS_H, _TS_W, seed=23)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)      # as the workers: oneDNN's blocking (and with it the fp32 rounding) follows the thread count
    one = TileShardedRecursion(_oracle_batch_fn(tiny_weights), _TS_BLOCK, None).run(
        torch.from_numpy(x0[0]), torch.from_numpy(x1[0]), _TS_T).numpy()
    assert np.abs(got - one).max() < 1e-5, np.abs(got - one).max()
    assert np.array_equal(got, one)
    it = fo.OracleInterpolator(tiny_weights, align=_TS_ALIGN, block_shape=_TS_BLOCK, opt=oracle_options(TINY))
    want = list(util._recursive_generator(x0[0], x1[0], _TS_T, it)) + [x1[0]]
    torch.set_num_threads(nthreads)
    assert np.array_equal(got, np.stack(want))


def test_bench_strong_scaling_launcher_world2_plan_only():
    """`python bench.py --gpus 2 --scaling strong` on CPU (gloo, plan-only engines): the tiles of ONE pair sharded over
    two ranks through the real driver + gather, stitched result identical to one rank, "scaling": "strong" in the line."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--plan-only', '--tiny-net', '--steps', '1',
                          '--warmup', '0', '--scaling', 'strong', '--workload', '1080p_2x2_T3'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert r['n_gpus'] == 2 and r['scaling'] == 'strong' and r['config']['tiles_sharded'] == 4
    assert r['config']['stitched_identical_to_one_rank'] is True and r['config']['weights_identical_on_all_ranks'] is True


def test_bench_strong_scaling_world8_4k_t6_plan_only():
    """First contact with the driver's 8-GPU tier, rehearsed on CPU: `bench.py --gpus 8 --scaling strong --workload 4k_4x4_T6`
    (BASELINE configs[4]: 16 tiles over 8 ranks, the whole T = 6 tree tile-local, one gather) through the real launcher with gloo
    and plan-only handles - including the once-per-job autotune sequence in front of the timed loop, in which only rank 0 warms up
    and must not enter a collective (round-4 ADVICE: it did, and deadlocked).  Eight ranks answer, every tile is owned exactly once
    by contiguous pairs, the stitched 65-frame sequence equals the one-rank result."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--plan-only', '--tiny-net', '--steps', '1',
                          '--warmup', '0', '--scaling', 'strong', '--workload', '4k_4x4_T6'],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert r['n_gpus'] == 8 and r['scaling'] == 'strong' and r['config']['tiles_sharded'] == 16
    assert r['ranks']['communicator_size'] == 8
    assert r['ranks']['units_by_rank'] == [[2 * g, 2 * g + 1] for g in range(8)]
    assert r['config']['stitched_identical_to_one_rank'] is True and r['config']['weights_identical_on_all_ranks'] is True


def _failing_worker(rank, world, port, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    from film_hip.sharding import TileShardedRecursion, share_tune

    def batch_fn(a, c):
        if rank == 1:
            raise ValueError('rank 1 breaks inside its recursion')
        return (a + c) * 0.5
    g = torch.Generator().manual_seed(3)
    f1, f2 = torch.rand((32, 48, 3), generator=g), torch.rand((32, 48, 3), generator=g)
    drv = TileShardedRecursion(batch_fn, [2, 2], dist)
    try:
        drv.run(f1, f2, 2)
        what = 'no error'
    except ValueError as e:
        what = 'own: ' + str(e)
    except RuntimeError as e:
        what = 'peer: ' + str(e)
    # the job is still usable: a tune-cache broadcast (text) behind the failed run
    eng = FilmEngine(TINY, device=-1)
    share_tune(eng, dist, src=0)
    q.put((rank, what, eng.export_tune().startswith('# film_hip tune cache')))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_aborts_the_tile_gather_on_every_rank():
    """Round-3 ADVICE: a rank that raises inside its share of the recursion used to leave the others in the gather until the
    collective timeout.  Now every rank all-reduces an ok flag first: the failing rank re-raises its own error, the others raise
    too, nobody hangs - and the process group is still usable afterwards (share_tune broadcast)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1].startswith('peer: ') and 'a rank failed' in res[0][1] and res[0][2]
    assert res[1][1] == 'own: rank 1 breaks inside its recursion' and res[1][2]
