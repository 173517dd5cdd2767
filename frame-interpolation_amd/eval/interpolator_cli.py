r"""Recursive in-betweening over directories of frames - CLI twin of the reference's
eval/interpolator_cli.py (same flags and outputs; plain Python loop instead of the in-process apache-beam
DirectRunner, sorted with a natural-order key instead of natsort, mp4 via the ffmpeg binary if present).

  cd frame-interpolation_amd
  python -m eval.interpolator_cli --model_path <model dir> --pattern "<root>/*" --times_to_interpolate 3

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


def process_directory(directory: str, it, args) -> int:
    inputs = list_input_frames(directory)
    if len(inputs) < 2:
        return 0
    frames = list(util.interpolate_recursively_from_files(inputs, args.times_to_interpolate, it))
    output_frames(frames, f'{directory}/interpolated_frames')
    if args.output_video:
        write_video(f'{directory}/interpolated.mp4', frames, args.fps)
    return len(frames)


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.output_video:
        util.get_ffmpeg_path()
    it = interpolator_lib.Interpolator(args.model_path, args.align, [args.block_height, args.block_width], precision=args.precision)
    for directory in sorted(glob.glob(args.pattern)):
        if os.path.isdir(directory):
            n = process_directory(directory, it, args)
            print(f'{directory}: {n} frames')


if __name__ == '__main__':
    main(sys.argv[1:])
