"""Multi-GPU layout of the hot path: one process per GPU, independent units, no data-path collective.

Frame pairs (BASELINE config 4) and the tiles of a tiled frame (configs 3/5; each tile's whole recursion
tree depends only on that tile, eval/interpolator.py:194-206 + eval/util.py:82-91) are independent, so
units are dealt out contiguously and the only collective is the one-time broadcast of the packed
weight blob (RCCL over xGMI on GPUs - torch.distributed backend "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `n_units` for `rank`; the first n % world ranks get one extra unit."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError('bad world/rank')
    base, extra = divmod(n_units, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def tiles_of_rank(block_shape: List[int], world: int, rank: int) -> List[int]:
    """Row-major tile indices (image_to_patches order) owned by `rank`."""
    b, e = shard_range(block_shape[0] * block_shape[1], world, rank)
    return list(range(b, e))


def broadcast_weights(engine, dist, src: int = 0, device=None) -> None:
    """Rank `src` has called engine.set_weights(); every other rank receives the flat parameter blob (137.7 MB for
    the published net) and imports it; each rank builds its kernel layouts itself (multi-threaded, < 1 s).  `device` = torch device of the blob (cuda for RCCL, None/cpu for
    gloo with plan-only engines)."""
    import torch
    n = engine.packed_size()
    rank = dist.get_rank()
    if device is not None and getattr(device, 'type', 'cpu') == 'cuda':
        blob = torch.empty(n, dtype=torch.float32, device=device)
        if rank == src:
            engine.export_packed_device(blob.data_ptr(), n)
        dist.broadcast(blob, src=src)
        torch.cuda.synchronize(device)
        if rank != src:
            engine.import_packed_device(blob.data_ptr(), n)
    else:
        blob = torch.from_numpy(engine.export_packed()) if rank == src else torch.empty(n, dtype=torch.float32)
        dist.broadcast(blob, src=src)
        if rank != src:
            engine.import_packed(blob.numpy())


def sharded_interpolator(model_path, align, block_shape, dist, local_rank: int, precision: int = 0, options=None):
    """eval.interpolator.Interpolator of this rank's GPU: rank 0 reads + packs the weights of `model_path`, every
    other rank receives the packed blob by broadcast (RCCL over xGMI) instead of reading and repacking."""
    import torch
    from eval.interpolator import Interpolator
    from .engine import FilmEngine
    from .options import PUBLISHED
    from . import weights as W
    opt = options or PUBLISHED
    rank = dist.get_rank()
    if rank == 0:
        it = Interpolator(model_path, align, block_shape, device=local_rank, options=opt, precision=precision)
        engine = it.engine
    else:
        engine = FilmEngine(opt, device=local_rank)
    broadcast_weights(engine, dist, src=0, device=torch.device('cuda', local_rank))
    if rank != 0:
        it = Interpolator.__new__(Interpolator)
        it._options, it._engine = opt, engine
        if precision:
            engine.set_option('precision', int(precision))
        it._align, it._block_shape = align or None, block_shape or None
    return it
