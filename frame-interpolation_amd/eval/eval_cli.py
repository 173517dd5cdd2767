r"""Benchmark evaluation of a frame interpolation model over triplet folders - the image-folder twin of the
reference's eval/eval_cli.py (which reads TFRecords from gs://, eval/config/*.gin).

  python -m eval.eval_cli --model_path <saved model dir> --triplet_dir <root> --output_dir <dir> \
      [--metrics l1 l2 ssim psnr] [--max_examples -1] [--output_frames] [--align 64] \
      [--block_height 1 --block_width 1]

`triplet_dir` is searched recursively for folders that hold a triplet: Vimeo-90K style `im1.png im2.png im3.png`
(eval/config/vimeo_90K.gin) or `frame_0 / frame_1(middle) / frame_2` style names; more generally any folder with
exactly three images, in natural order (first, ground-truth middle, last).

Same outputs as the reference loop (eval/eval_cli.py:88-178): `readme.txt`, `results.csv` with a title row
`key, <metric>, ...`, one row per example, a final `mean, ...` row; predictions are clipped to [0,1] before the
metrics (:165); with --output_frames the inputs, ground truth and prediction are written as `<key>_<name>.png`.
"""
import argparse
import os
import re
import sys
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import interpolator as interpolator_lib
from . import metrics as metrics_lib
from . import util

_IMAGE_EXT = ('.png', '.jpg', '.jpeg')


def _natural(s: str):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', s)]


def find_triplets(root: str) -> List[Tuple[str, Tuple[str, str, str]]]:
    """[(key, (first, middle, last))], key = folder path relative to root with '/' -> '_'."""
    out = []
    for d, _sub, files in sorted(os.walk(root)):
        imgs = sorted((f for f in files if f.lower().endswith(_IMAGE_EXT)), key=_natural)
        if len(imgs) == 3:
            rel = os.path.relpath(d, root)
            key = 'root' if rel == '.' else rel.replace(os.sep, '_')
            out.append((key, tuple(os.path.join(d, f) for f in imgs)))
    return out


def run_evaluation(interpolator, triplets: Sequence[Tuple[str, Tuple[str, str, str]]], output_dir: str,
                   max_examples: int = -1, metrics: Sequence[str] = ('l1', 'l2', 'ssim', 'psnr'),
                   output_frames: bool = False, model_path: str = '', source: str = '') -> dict:
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, 'readme.txt'), 'w') as f:
        print('Results for:', file=f)
        print(f' model:   {model_path}', file=f)
        print(f' triplets: {source}', file=f)
    fns = metrics_lib.test_losses(list(metrics))
    all_losses = {n: [] for n, _ in fns}
    dt = np.full((1,), 0.5, np.float32)
    with open(os.path.join(output_dir, 'results.csv'), 'w') as csv_file:
        print(', '.join(['key'] + [n for n, _ in fns]), file=csv_file)
        for i, (key, (f0, fy, f1)) in enumerate(triplets):
            if 0 <= max_examples <= i:
                break
            x0, y, x1 = util.read_image(f0), util.read_image(fy), util.read_image(f1)
            image = interpolator(x0[None], x1[None], dt)
            if output_frames:
                for name, img in (('x0', x0), ('x1', x1), ('y', y), ('image', image[0])):
                    util.write_image(os.path.join(output_dir, f'{key}_{name}.png'), img)
            image = np.clip(image, 0.0, 1.0)   # eval/eval_cli.py:162-165
            values = [fn(image, y[None]) for _n, fn in fns]
            for (n, _fn), v in zip(fns, values):
                all_losses[n].append(v)
            print(f'{key}, {str(values)[1:-1]}', file=csv_file)
        totals = {n: float(np.mean(v)) for n, v in all_losses.items() if v}
        if totals:
            print(f'mean, {str([totals[n] for n, _ in fns])[1:-1]}', file=csv_file)
    return totals


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--model_path', required=True, help='The path of the saved model to use (SavedModel dir or film_weights.npz dir).')
    ap.add_argument('--triplet_dir', required=True, help='Root folder searched for image triplets.')
    ap.add_argument('--output_dir', required=True, help='Directory to store the results into.')
    ap.add_argument('--metrics', nargs='+', default=['l1', 'l2', 'ssim', 'psnr'], help='evaluation.metrics of the gin config.')
    ap.add_argument('--max_examples', type=int, default=-1, help='Maximum examples to evaluate (-1: all).')
    ap.add_argument('--output_frames', action='store_true', help='If true, saves the inputs, ground-truth and interpolated frames.')
    ap.add_argument('--align', type=int, default=64, help='If >1, pad the input size so it is evenly divisible by this value.')
    ap.add_argument('--block_height', type=int, default=1)
    ap.add_argument('--block_width', type=int, default=1)
    ap.add_argument('--device', type=int, default=0, help='HIP device ordinal.')
    ap.add_argument('--precision', type=int, default=0, choices=[0, 1, 2], help='engine precision mode: 0 fp32 MFMA, 1 bf16x6, 2 bf16x3.')
    args = ap.parse_args(argv)
    triplets = find_triplets(args.triplet_dir)
    if not triplets:
        print(f'no image triplets under {args.triplet_dir}', file=sys.stderr)
        return 1
    it = interpolator_lib.Interpolator(args.model_path, args.align, [args.block_height, args.block_width], device=args.device, precision=args.precision)
    totals = run_evaluation(it, triplets, args.output_dir, args.max_examples, args.metrics, args.output_frames,
                            args.model_path, args.triplet_dir)
    print('mean,', totals)
    return 0


if __name__ == '__main__':
    sys.exit(main())
