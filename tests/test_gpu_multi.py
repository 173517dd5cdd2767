"""First contact with more than one GPU (round-5 verdict, next-round item 7): armed, not yet run - the boxes of rounds 1-6
have one MI355X, where these tests SKIP; on the first multi-GPU box they run by themselves with the driver's `pytest -m gpu`.

Both go through bench.py's own launcher (`python bench.py --gpus 2` re-executes itself under torch.distributed.run, one rank per GPU,
RCCL over xGMI): the weak mode (every rank its own frame pairs, weights broadcast once, no collective in the hot path) and the strong
mode (TileShardedRecursion: the tiles of ONE 1080p pair sharded over the ranks for the whole T = 3 recursion tree, one gather).
The two-rank result must carry `ranks.communicator_size == 2` on the nccl (= RCCL) backend and the SAME output bits as one rank.
The CPU rehearsal of the same launcher (gloo, plan-only handles) is tests/test_dist_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _ngpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_ngpus() < 2, reason='needs two MI355X in one node (RCCL); one-GPU boxes skip')]


def _bench(*args, timeout=900):
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-split', *args],
                         capture_output=True, text=True, env=env, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])


def test_two_ranks_weak_scaling_over_rccl_match_one_rank():
    """`bench.py --gpus 2`, headline workload: rank r interpolates its own 1080p pair (seeds independent of the world size), weights
    broadcast once over RCCL; rank 0's timed output has the bits of the one-GPU run, the golden-pair parity check ran on both ranks."""
    one = _bench('--gpus', '1', '--steps', '2', '--warmup', '1')
    two = _bench('--gpus', '2', '--steps', '2', '--warmup', '1')
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak'
    assert two['ranks']['communicator_size'] == 2 and two['ranks']['backend'] == 'nccl' and len(two['ranks']['ms_per_step_by_rank']) == 2
    assert two['ranks']['weight_broadcast_ms'] is not None
    assert two['timed_output_bit_identical_to_first_call'] is True
    assert two['output_crc32'] == one['output_crc32'] and one['output_crc32'] is not None
    assert two['value'] > one['value']          # two frames per step instead of one, in max-over-ranks time


def test_two_ranks_strong_scaling_tile_sharded_recursion_matches_one_rank():
    """`bench.py --gpus 2 --scaling strong --workload 1080p_2x2_T3`: the four tiles of ONE pair sharded over two ranks (two tiles each)
    for the whole T = 3 tree (7 generated frames), one gather per step; the gathered frames have the bits of the same driver on one rank."""
    one = _bench('--gpus', '1', '--scaling', 'strong', '--workload', '1080p_2x2_T3', '--steps', '2', '--warmup', '1')
    two = _bench('--gpus', '2', '--scaling', 'strong', '--workload', '1080p_2x2_T3', '--steps', '2', '--warmup', '1')
    assert two['n_gpus'] == 2 and two['scaling'] == 'strong'
    assert two['ranks']['communicator_size'] == 2 and two['ranks']['backend'] == 'nccl'
    assert two['ranks']['gather_ms_per_step'] is not None
    assert two['output_crc32'] == one['output_crc32'] and one['output_crc32'] is not None


def _bcast_rank(rank, world, uid_path, q):
    import ctypes
    import time
    import zlib
    import numpy as np
    import torch
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    torch.cuda.set_device(rank)
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))

    class UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_char * 128)]
    uid = UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        with open(uid_path + '.tmp', 'wb') as f:
            f.write(bytes(uid))
        os.replace(uid_path + '.tmp', uid_path)
    else:
        for _ in range(600):
            if os.path.exists(uid_path):
                break
            time.sleep(0.1)
        ctypes.memmove(ctypes.byref(uid), open(uid_path, 'rb').read(), 128)
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    eng = FilmEngine(TINY, device=rank)
    if rank == 0:
        eng.set_weights(W.make_synthetic_weights(TINY, seed=0))
    eng.bcast_weights(comm.value, root=0, rank=rank)
    q.put((rank, zlib.crc32(np.ascontiguousarray(eng.export_packed()).tobytes())))
    eng.close()


def test_weight_broadcast_behind_the_c_abi_two_ranks(tmp_path):
    """film_bcast_weights with NO torch.distributed: two processes, one GPU each, an RCCL communicator made by hand (ncclGetUniqueId on
    rank 0, handed over through a file); rank 1 ends up with rank 0's parameter blob."""
    import multiprocessing as mp
    import zlib
    import numpy as np
    from film_hip import weights as W
    from film_hip.engine import FilmEngine
    from film_hip.options import TINY
    ref = FilmEngine(TINY, device=-1)
    ref.set_weights(W.make_synthetic_weights(TINY, seed=0))
    want = zlib.crc32(np.ascontiguousarray(ref.export_packed()).tobytes())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_rank, args=(r, 2, str(tmp_path / 'uid'), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: want, 1: want}

