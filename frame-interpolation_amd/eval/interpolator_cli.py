r"""Recursive in-betweening over directories of frames - CLI twin of the reference's
eval/interpolator_cli.py (same flags and outputs; plain Python loop instead of the in-process apache-beam
DirectRunner, sorted with a natural-order key instead of natsort, mp4 via the ffmpeg binary if present).

  cd frame-interpolation_amd
  python -m eval.interpolator_cli --model_path <model dir> --pattern "<root>/*" --times_to_interpolate 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         -m eval.interpolator_cli --model_path <model dir> --pattern "<root>/*" --times_to_interpolate 3

Multi-GPU (the reference maps directories over beam workers, eval/interpolator_cli.py:180-187): under torchrun
every rank owns one GPU (LOCAL_RANK) and a contiguous share of the work - whole directories when there are at
least as many as ranks, otherwise consecutive input pairs of each directory (a pair's recursion tree is
independent of every other pair; output frame indices are global, so ranks write disjoint files), and for a
directory with fewer pairs than ranks under --block_height / --block_width tiling the TILES of every pair
(film_hip.sharding.TileShardedRecursion: rank g owns tiles_of_rank(block, world, g) for the whole recursion tree,
one gather of the generated tiles per pair, rank 0 writes the frames - BASELINE configs[4], a 4K pair with 4x4
tiles on 8 GPUs).  No collective on the data path; weights are read from --model_path on rank 0 and broadcast
once (RCCL).

For every directory matching --pattern: frames *.png, *.jpg, *.jpeg (each group naturally sorted, groups
concatenated in that order, as upstream) are expanded 2^T-fold and written to
<dir>/interpolated_frames/frame_%03d.png; with --output_video also <dir>/interpolated.mp4.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
from typing import List

import numpy as np

from . import interpolator as interpolator_lib
from . import util

_INPUT_EXT = ['png', 'jpg', 'jpeg']


def build_parser() -> argparse.ArgumentParser:
    """Flags of reference eval/interpolator_cli.py:85-121."""
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--pattern', required=True, help='The pattern to determine the directories with the input frames.')
    ap.add_argument('--model_path', default=None, help='The path of the saved model to use.')
    ap.add_argument('--times_to_interpolate', type=int, default=5,
                    help='The number of times to run recursive midpoint interpolation. '
                         'The number of output frames will be 2^times_to_interpolate+1.')
    ap.add_argument('--fps', type=int, default=30, help='Frames per second to play interpolated videos in slow motion.')
    ap.add_argument('--align', type=int, default=64,
                    help='If >1, pad the input size so it is evenly divisible by this value.')
    ap.add_argument('--block_height', type=int, default=1, help='Number of patches along height.')
    ap.add_argument('--block_width', type=int, default=1, help='Number of patches along width.')
    ap.add_argument('--precision', type=int, default=0, choices=[0, 1, 2],
                    help='(extension) engine precision mode: 0 fp32 MFMA, 1 bf16x6, 2 bf16x3.')
    ap.add_argument('--output_video', action='store_true', default=False,
                    help='If true, creates a video of the frames in the interpolated_frames/ subdirectory')
    return ap


def natural_key(s: str):
    """Natural sort key (digits compare numerically), the ordering natsort.natsorted gives for frame names."""
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', s)]


def list_input_frames(directory: str) -> List[str]:
    """reference ProcessDirectory.process, eval/interpolator_cli.py:165-169."""
    out: List[str] = []
    for ext in _INPUT_EXT:
        out += sorted(glob.glob(f'{directory}/*.{ext}'), key=natural_key)
    return out


def output_frames(frames: List[np.ndarray], frames_dir: str) -> None:
    """reference _output_frames, eval/interpolator_cli.py:127-149: stale frame_*.png are removed first."""
    if os.path.isdir(frames_dir):
        for old in glob.glob(f'{frames_dir}/frame_*.png'):
            os.remove(old)
    else:
        os.makedirs(frames_dir)
    for idx, frame in enumerate(frames):
        util.write_image(f'{frames_dir}/frame_{idx:03d}.png', frame)


def write_video(path: str, frames: List[np.ndarray], fps: int) -> None:
    ffmpeg = util.get_ffmpeg_path()
    h, w = frames[0].shape[:2]
    cmd = [ffmpeg, '-y', '-f', 'rawvideo', '-pix_fmt', 'rgb24', '-s', f'{w}x{h}', '-r', str(fps), '-i', '-',
           '-pix_fmt', 'yuv420p', path]
    with subprocess.Popen(cmd, stdin=subprocess.PIPE) as p:
        for f in frames:
            p.stdin.write(util.to_uint8(f).tobytes())
        p.stdin.close()
        p.wait()


def write_video_uint8(path: str, frames: List[np.ndarray], fps: int) -> None:
    ffmpeg = util.get_ffmpeg_path()
    h, w = frames[0].shape[:2]
    cmd = [ffmpeg, '-y', '-f', 'rawvideo', '-pix_fmt', 'rgb24', '-s', f'{w}x{h}', '-r', str(fps), '-i', '-',
           '-pix_fmt', 'yuv420p', path]
    with subprocess.Popen(cmd, stdin=subprocess.PIPE) as p:
        for f in frames:
            p.stdin.write(f.tobytes())
        p.stdin.close()
        p.wait()


def process_directory(directory: str, it, args) -> int:
    inputs = list_input_frames(directory)
    if len(inputs) < 2:
        return 0
    frames_dir = f'{directory}/interpolated_frames'
    output_frames([], frames_dir)      # (creates the directory / removes stale frame_*.png, as the reference does before writing)
    fast = util.interpolate_pairs_to_files(inputs, 0, len(inputs) - 1, len(inputs) - 1, args.times_to_interpolate, it, frames_dir,
                                           keep=bool(args.output_video))
    if fast is not None:               # the device pipeline (HIP-backed Interpolator): same files, encoded while the GPU works
        n, kept = fast
        if args.output_video:
            write_video_uint8(f'{directory}/interpolated.mp4', kept, args.fps)
        return n
    frames = list(util.interpolate_recursively_from_files(inputs, args.times_to_interpolate, it))
    output_frames(frames, f'{directory}/interpolated_frames')
    if args.output_video:
        write_video(f'{directory}/interpolated.mp4', frames, args.fps)
    return len(frames)


def plan_work(directories: List[str], world: int, rank: int):
    """Work of `rank`: (whole_directories, [(directory, first_pair, end_pair, n_pairs)]).  With at least `world`
    directories each rank takes whole directories (shard_range over the directory list, as the reference's ParDo
    over directories); with fewer, the consecutive input pairs of every directory are dealt out instead."""
    from film_hip.sharding import shard_range
    pairs = [(d, max(0, len(list_input_frames(d)) - 1)) for d in directories]
    pairs = [(d, n) for d, n in pairs if n > 0]
    if world == 1 or len(pairs) >= world:
        b, e = shard_range(len(pairs), world, rank)
        return True, [(d, 0, n, n) for d, n in pairs[b:e]]
    out = []
    for d, n in pairs:
        b, e = shard_range(n, world, rank)
        if e > b:
            out.append((d, b, e, n))
    return False, out


def tile_mode(n_pairs: int, world: int, ntiles: int) -> bool:
    """A directory's pairs cannot occupy every rank but its tiles can: shard the tiles of each pair instead."""
    return world > 1 and ntiles > 1 and 0 < n_pairs < world


def process_directory_tile_sharded(directory: str, driver, args, rank: int) -> int:
    """Every rank walks every input pair of the directory with its own tiles (driver = TileShardedRecursion bound to the
    rank's GPU); rank 0 receives the stitched frames and writes them with the reference's naming
    (eval/interpolator_cli.py:127-149; frame (pair p, step k) -> index p * 2^T + k, last input frame at the end)."""
    import torch
    inputs = list_input_frames(directory)
    if len(inputs) < 2:
        return 0
    frames_dir = f'{directory}/interpolated_frames'
    if rank == 0:
        output_frames([], frames_dir)
    step = 2 ** args.times_to_interpolate
    n = 0
    kept = []
    for p in range(len(inputs) - 1):
        f1, f2 = util.read_image(inputs[p]), util.read_image(inputs[p + 1])
        seq = driver.run(driver.to_device(f1), driver.to_device(f2), args.times_to_interpolate)
        if seq is None:
            continue
        seq = seq.cpu().numpy()
        last = p == len(inputs) - 2
        for k in range(step + (1 if last else 0)):
            # the two ends are the input frames themselves, as the reference yields them (eval/util.py:79-80,122-123)
            frame = f1 if k == 0 else f2 if k == step else seq[k]
            util.write_image(f'{frames_dir}/frame_{p * step + k:03d}.png', frame)
            if args.output_video:
                kept.append(frame)
            n += 1
    if rank == 0 and args.output_video and kept:
        write_video(f'{directory}/interpolated.mp4', kept, args.fps)
    return n


def process_pair_range(directory: str, first: int, end: int, n_pairs: int, it, args, clear: bool) -> int:
    """Input pairs [first, end) of a directory: frame (pair p, step k) -> frame_%03d with index p * 2^T + k; the
    rank that owns the last pair also writes the final input frame (eval/util.py:122-123)."""
    inputs = list_input_frames(directory)
    frames_dir = f'{directory}/interpolated_frames'
    if clear:
        output_frames([], frames_dir)
    else:
        os.makedirs(frames_dir, exist_ok=True)
    step = 2 ** args.times_to_interpolate
    fast = util.interpolate_pairs_to_files(inputs, first, end, n_pairs, args.times_to_interpolate, it, frames_dir)
    if fast is not None:
        return fast[0]
    n = 0
    for p in range(first, end):
        seq = list(util.interpolate_recursively_from_files(inputs[p:p + 2], args.times_to_interpolate, it))
        keep = seq if p == n_pairs - 1 else seq[:-1]
        for k, frame in enumerate(keep):
            util.write_image(f'{frames_dir}/frame_{p * step + k:03d}.png', frame)
        n += len(keep)
    return n


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.output_video:
        util.get_ffmpeg_path()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    block = [args.block_height, args.block_width]
    directories = [d for d in sorted(glob.glob(args.pattern)) if os.path.isdir(d)]
    if world == 1:
        it = interpolator_lib.Interpolator(args.model_path, args.align, block, precision=args.precision)
        for directory in directories:
            n = process_directory(directory, it, args)
            print(f'{directory}: {n} frames')
        return
    import datetime
    import torch
    import torch.distributed as dist
    from film_hip.sharding import TileShardedRecursion, sharded_interpolator
    from film_hip.torch_io import DeviceInterpolator
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'rank {rank} needs GPU {local_rank}, only {torch.cuda.device_count()} visible')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # directories differ in size: a rank that finishes early waits at the final barrier for as long as the slowest
    # rank works, so the collective timeout is a day, not the 10-minute default of the NCCL watchdog
    dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev,
                            timeout=datetime.timedelta(hours=24))
    it = sharded_interpolator(args.model_path, args.align, block, dist, local_rank, precision=args.precision)
    whole, work = plan_work(directories, world, rank)
    ntiles = max(1, args.block_height) * max(1, args.block_width)
    pairs_of = {d: max(0, len(list_input_frames(d)) - 1) for d in directories}
    tiled_dirs = [] if whole else [d for d in directories if tile_mode(pairs_of[d], world, ntiles)]
    # stale frames are removed by rank 0 before anybody writes (a directory's pairs may be spread over ranks) - only in
    # directories that are in the work list (the single-rank path leaves directories with < 2 inputs untouched)
    if rank == 0 and not whole:
        for d in directories:
            if pairs_of[d] > 0 and d not in tiled_dirs:
                output_frames([], f'{d}/interpolated_frames')
    dist.barrier()
    if tiled_dirs:
        driver = TileShardedRecursion(DeviceInterpolator(it.engine, align=it.align).batch, block, dist)
        driver.to_device = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        for d in tiled_dirs:     # every rank, same order: the gather inside is a collective
            n = process_directory_tile_sharded(d, driver, args, rank)
            print(f'[rank {rank}] {d}: tiles {driver.tiles} of {ntiles} for {pairs_of[d]} pair(s), {n} frames written')
        it.engine.save_tune_cache()
    for directory, b, e, n_pairs in work:
        if directory in tiled_dirs:
            continue
        if whole:
            n = process_directory(directory, it, args)
        else:
            n = process_pair_range(directory, b, e, n_pairs, it, args, clear=False)
        print(f'[rank {rank}] {directory}: pairs [{b},{e}) of {n_pairs}, {n} frames')
    dist.barrier()
    if rank == 0 and args.output_video and not whole:
        for d in directories:
            if d in tiled_dirs:
                continue
            files = sorted(glob.glob(f'{d}/interpolated_frames/frame_*.png'), key=natural_key)
            if files:
                write_video(f'{d}/interpolated.mp4', [util.read_image(f) for f in files], args.fps)
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1:])
