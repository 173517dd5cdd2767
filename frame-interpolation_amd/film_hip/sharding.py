"""Multi-GPU layout of the hot path: one process per GPU, independent units, no data-path collective.

Frame pairs (BASELINE config 4) and the tiles of a tiled frame (configs 3/5; each tile's whole recursion
tree depends only on that tile, eval/interpolator.py:194-206 + eval/util.py:82-91) are independent, so
units are dealt out contiguously and the only collective is the one-time broadcast of the packed
weight blob (RCCL over xGMI on GPUs - torch.distributed backend "nccl"; "gloo" in the CPU tests).

ONE frame pair on several GPUs (configs 2 / 4: a 1080p pair has 4 tiles, a 4K pair 16): TileShardedRecursion below.
Rank g owns the tiles tiles_of_rank(block_shape, world, g) of BOTH input frames for the whole 2^T - 1 frame recursion
tree - the reference re-tiles every mid-frame on the same grid, so a tile of a generated frame depends on that tile of
its two parents only - and runs the breadth-first recursion on them as a batch of small frames through the UNTILED
entry point (per tile: pad to `align`, model, crop = exactly what the reference's tile loop does per patch,
eval/interpolator.py:199-206).  Nothing is exchanged until the end, when the generated tiles are gathered to rank 0
(one collective per pair, after the last model invocation) and stitched with patches_to_image's layout.
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


def share_tune(engine, dist, src: int = 0) -> None:
    """Every rank adopts rank `src`'s autotune choices (film_export_tune text: per conv shape the fastest tile of its kernel
    family - they cannot change a result, only the speed): one measurement for the job instead of one per rank."""
    objs = [engine.export_tune() if dist.get_rank() == src else None]
    dist.broadcast_object_list(objs, src=src)
    if dist.get_rank() != src and objs[0]:
        engine.import_tune(objs[0])


def sharded_interpolator(model_path, align, block_shape, dist, local_rank: int, precision: int = 0, options=None):
    """eval.interpolator.Interpolator of this rank's GPU: rank 0 reads + packs the weights of `model_path`, every
    other rank receives the packed blob by broadcast (RCCL over xGMI) instead of reading and repacking."""
    import torch
    from eval.interpolator import Interpolator
    from .engine import FilmEngine
    from .options import PUBLISHED
    opt = options or PUBLISHED
    rank = dist.get_rank()
    dev = torch.device('cuda', local_rank)
    it, err = None, None
    if rank == 0:
        try:
            it = Interpolator(model_path, align, block_shape, device=local_rank, options=opt, precision=precision)
        except Exception as e:   # noqa: BLE001 - the other ranks must not be left waiting in the broadcast
            err = e
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev)
    dist.broadcast(ok, src=0)
    if int(ok.item()) == 0:
        if err is not None:
            raise err
        raise RuntimeError('rank 0 could not load the model (see its traceback)')
    engine = it.engine if rank == 0 else FilmEngine(opt, device=local_rank)
    broadcast_weights(engine, dist, src=0, device=dev)
    if rank != 0:
        it = Interpolator('', align, block_shape, options=opt, precision=precision, engine=engine)
    return it

# ------------------------------------------------------------------------------------------------------------------
# one frame pair, tiles sharded over the ranks
# ------------------------------------------------------------------------------------------------------------------
def extract_tiles(frame, block_shape: List[int], tiles: List[int]):
    """[H,W,3] tensor -> [len(tiles), H/bh, W/bw, 3]: the row-major tiles `tiles` of image_to_patches
    (eval/interpolator.py:66-99; same divisibility asserts)."""
    import torch
    bh, bw = block_shape
    h, w, c = frame.shape
    ph, pw = h // bh, w // bw
    assert h == ph * bh, 'block_height=%d should evenly divide height=%d.' % (bh, h)
    assert w == pw * bw, 'block_width=%d should evenly divide width=%d.' % (bw, w)
    if not tiles:
        return torch.empty((0, ph, pw, c), dtype=frame.dtype, device=frame.device)
    return torch.stack([frame[(t // bw) * ph:(t // bw + 1) * ph, (t % bw) * pw:(t % bw + 1) * pw] for t in tiles]).contiguous()


def stitch_tiles(patches, block_shape: List[int]):
    """[F, bh*bw, ph, pw, 3] -> [F, bh*ph, bw*pw, 3] (patches_to_image, eval/interpolator.py:102-126, per frame)."""
    bh, bw = block_shape
    f, n, ph, pw, c = patches.shape
    assert n == bh * bw
    return patches.reshape(f, bh, bw, ph, pw, c).permute(0, 1, 3, 2, 4, 5).reshape(f, bh * ph, bw * pw, c).contiguous()


def recurse_tiles(t0, t1, times_to_interpolate: int, batch_fn):
    """Breadth-first mid-point recursion (eval/util.py:62-91, one batched call per depth as film_hip/recursive.py) on a
    set of independent tiles: t0, t1 [n, ph, pw, 3] -> [2^T + 1, n, ph, pw, 3] in temporal order (inputs included).
    batch_fn(x0, x1) maps two [K, ph, pw, 3] batches to the K mid-frames (pad / model / crop per element)."""
    import torch
    frames = torch.stack([t0, t1])
    n = t0.shape[0]
    for _ in range(times_to_interpolate):
        k = frames.shape[0] - 1
        out = torch.empty((2 * k + 1,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
        if n:     # a rank without tiles (more ranks than tiles) only takes part in the gather
            a = frames[:-1].reshape((k * n,) + tuple(frames.shape[2:])).contiguous()
            b = frames[1:].reshape((k * n,) + tuple(frames.shape[2:])).contiguous()
            out[0::2] = frames
            out[1::2] = batch_fn(a, b).reshape((k, n) + tuple(frames.shape[2:]))
        frames = out
    return frames


class TileShardedRecursion:
    """One frame pair, T recursions, tiles of block_shape dealt over the ranks of `dist` (None = one rank).

    run(frame1, frame2, T) -> on rank `dst` the [2^T + 1, H, W, 3] sequence (frame1, generated frames, frame2), None on
    the other ranks.  batch_fn: see recurse_tiles (DeviceInterpolator(engine, align).batch on GPUs).  No collective
    before the final gather; with one rank this is the tiled Interpolator path run tile-wise (bit-identical, test)."""

    def __init__(self, batch_fn, block_shape: List[int], dist=None, dst: int = 0, collective_at_world1: bool = False):
        self.batch_fn = batch_fn
        self.block_shape = [int(block_shape[0]), int(block_shape[1])]
        # collective_at_world1: tests drive the gather through the process group even when it has a single rank
        self.dist = dist if (dist is not None and dist.is_initialized() and
                             (dist.get_world_size() > 1 or collective_at_world1)) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.dst = dst
        self.ntiles = self.block_shape[0] * self.block_shape[1]
        self.tiles = tiles_of_rank(self.block_shape, self.world, self.rank)
        self.counts = [len(tiles_of_rank(self.block_shape, self.world, r)) for r in range(self.world)]
        self._recv, self._recv_key = None, None
        self._gather_ms, self._events, self.runs = 0.0, [], 0     # time spent in the gather collective (property gather_ms) / number of run() calls

    def local(self, frame1, frame2, times_to_interpolate: int):
        """This rank's share: [2^T + 1, n_own, ph, pw, 3]."""
        return recurse_tiles(extract_tiles(frame1, self.block_shape, self.tiles),
                             extract_tiles(frame2, self.block_shape, self.tiles), times_to_interpolate, self.batch_fn)

    def gather(self, local_mids):
        """[F, n_own, ph, pw, 3] of every rank -> the list of per-rank receive buffers [F, nmax, ph, pw, 3] on rank dst (None
        elsewhere; one rank: [local_mids]).  The buffers are REUSED: they are valid until the next gather() / run() of this object
        (run() copies the frames out of them before it returns).  One gather of equally sized (padded to the largest share) buffers; rank r's tiles are
        the contiguous range shard_range gives.  Before the gather every rank contributes an "ok" flag to an all-reduce: a rank
        whose recursion failed makes ALL ranks raise instead of leaving the others in the collective until its timeout.  The
        receive buffers are allocated once per shape and reused (round-3 ADVICE); gather_ms / runs time the collective."""
        import time
        import torch
        if self.dist is None:
            return [local_mids]
        failed = local_mids is None
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=self._device(local_mids))
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if int(flag.item()):
            raise RuntimeError(f'rank {self.rank}: a rank failed before the gather' + (' (this one)' if failed else ''))
        nmax = max(self.counts)
        f = local_mids.shape[0]
        send = local_mids
        if local_mids.shape[1] != nmax:
            send = torch.zeros((f, nmax) + tuple(local_mids.shape[2:]), dtype=local_mids.dtype, device=local_mids.device)
            send[:, :local_mids.shape[1]] = local_mids
        send = send.contiguous()
        recv = None
        if self.rank == self.dst:
            key = (tuple(send.shape), send.dtype, send.device)
            if self._recv_key != key:
                self._recv = [torch.empty_like(send) for _ in range(self.world)]
                self._recv_key = key
            recv = self._recv
        if send.is_cuda:
            # device events on the current stream (the collective is ordered against it): the time of the gather itself, without
            # draining the compute in front of it into the measurement and without a host synchronisation per run (round-4 ADVICE);
            # gather_ms is read lazily, see the property
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.dist.gather(send, recv, dst=self.dst)
            e1.record()
            self._events.append((e0, e1))
        else:
            t0 = time.perf_counter()
            self.dist.gather(send, recv, dst=self.dst)
            self._gather_ms += (time.perf_counter() - t0) * 1e3
        self.runs += 1
        return recv if self.rank == self.dst else None

    @property
    def gather_ms(self) -> float:
        """Milliseconds spent in the gather collectives so far (synchronises on the recorded events when first read)."""
        for e0, e1 in self._events:
            e1.synchronize()
            self._gather_ms += e0.elapsed_time(e1)
        self._events = []
        return self._gather_ms

    def _device(self, t):
        import torch
        if t is not None:
            return t.device
        return torch.device('cuda', torch.cuda.current_device()) if self.dist.get_backend() == 'nccl' else torch.device('cpu')

    def run(self, frame1, frame2, times_to_interpolate: int):
        """(rank dst) [2^T + 1, H, W, 3]: frame1, the generated frames stitched straight from the receive buffers into ONE output
        tensor, frame2 - no concatenated copies of the whole sequence in between (4K, T = 6: 1.6 GB each)."""
        import torch
        err = None
        try:
            seq = self.local(frame1, frame2, times_to_interpolate)
            mine = seq[1:-1].contiguous()
        except Exception as e:   # noqa: BLE001 - the other ranks must not be left waiting in the gather
            err, mine = e, None
            if self.dist is None:
                raise
        try:
            recv = self.gather(mine)
        except RuntimeError:
            if err is not None:
                raise err
            raise
        if recv is None:
            return None
        bh, bw = self.block_shape
        f = recv[0].shape[0]
        ph, pw, c = recv[0].shape[2:]
        out = torch.empty((f + 2, bh * ph, bw * pw, c), dtype=frame1.dtype, device=frame1.device)
        out[0] = frame1
        out[-1] = frame2
        t = 0
        for r in range(self.world):
            for k in range(self.counts[r]):
                out[1:-1, (t // bw) * ph:(t // bw + 1) * ph, (t % bw) * pw:(t % bw + 1) * pw] = recv[r][:, k]
                t += 1
        return out
