r"""One mid-frame from two image files on the MI355X engine - CLI twin of the reference's
eval/interpolator_test.py (same flag names and defaults, argparse instead of absl which is not
installed here).

  cd frame-interpolation_amd
  python -m eval.interpolator_test --frame1 photos/one.png --frame2 photos/two.png \
      --model_path <dir with film_weights.npz or a TF2 SavedModel> --output_frame out.png
"""
import argparse
import os
import sys

import numpy as np

from . import interpolator as interpolator_lib
from . import util


def build_parser() -> argparse.ArgumentParser:
    """Flags of reference eval/interpolator_test.py:39-70."""
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--frame1', required=True, help='The filepath of the first input frame.')
    ap.add_argument('--frame2', required=True, help='The filepath of the second input frame.')
    ap.add_argument('--model_path', default=None, help='The path of the saved model to use.')
    ap.add_argument('--output_frame', default=None, help='The output filepath of the interpolated mid-frame.')
    ap.add_argument('--align', type=int, default=64,
                    help='If >1, pad the input size so it is evenly divisible by this value.')
    ap.add_argument('--block_height', type=int, default=1,
                    help='An int >= 1, number of patches along height, patch_height = height//block_height, '
                         'should be evenly divisible.')
    ap.add_argument('--precision', type=int, default=0, choices=[0, 1, 2],
                    help='(extension) engine precision mode: 0 fp32 MFMA, 1 bf16x6, 2 bf16x3.')
    ap.add_argument('--block_width', type=int, default=1,
                    help='An int >= 1, number of patches along width, patch_width = width//block_width, '
                         'should be evenly divisible.')
    return ap


def run(args) -> str:
    """reference _run_interpolator (eval/interpolator_test.py:73-99)."""
    it = interpolator_lib.Interpolator(model_path=args.model_path, align=args.align,
                                       block_shape=[args.block_height, args.block_width], precision=args.precision)
    first = util.read_image(args.frame1)[np.newaxis]
    second = util.read_image(args.frame2)[np.newaxis]
    half = np.full(shape=(1,), fill_value=0.5, dtype=np.float32)
    mid = it(first, second, half)[0]
    out = args.output_frame or f'{os.path.dirname(args.frame1)}/output_frame.png'
    util.write_image(out, mid)
    return out


def main(argv=None) -> None:
    print('wrote', run(build_parser().parse_args(argv)))


if __name__ == '__main__':
    main(sys.argv[1:])
