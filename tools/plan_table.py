#!/usr/bin/env python
"""Layer -> kernel family table of a plan (no GPU needed): python tools/plan_table.py [B H W] [--precision p]
Prints one markdown row per convolution launch: tag, level size, Cin -> Cout, kernel family, split-K factor, GFLOP."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'frame-interpolation_amd'))
from film_hip.engine import FilmEngine  # noqa: E402
from film_hip.options import PUBLISHED  # noqa: E402


def family(op):
    t = op['tile']
    if op['kind'] != 'conv_mfma':
        return op['kind']
    if t & 8192:
        return 'conv_wino2d F(4,3)x x F(2,3)y'
    if t & 16384:
        return 'conv_fold4 (upsample + 2x2, difference form)'
    if t & 1024:
        return 'conv_foldx3'
    if t & 256:
        return 'conv_winox3' if t & 512 else 'conv_wino43 F(4,3)' if t & 2048 else 'conv_wino F(2,3)'
    if t & 128:
        return 'conv_halo_split x3' if t & 512 else 'conv_halo_split x6'
    if t & 64:
        return 'conv_halo'
    if t & 32:
        return 'conv_c3 (3-channel, direct)' if (t & 15) == 7 else 'conv_igemm (3-channel)'
    return 'conv_buf' + (' (4 phases)' if op.get('fold') else '')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('shape', nargs='*', type=int, default=[4, 576, 960])
    ap.add_argument('--precision', type=int, default=0)
    args = ap.parse_args()
    eng = FilmEngine(PUBLISHED, device=-1)
    if args.precision:
        eng.set_option('precision', args.precision)
    plan = eng.plan(*args.shape)
    print('| op | batch x H x W | K -> Cout | kernel | split-K | GFLOP |')
    print('|---|---|---|---|---|---|')
    tot = {}
    for op in plan['ops']:
        if not op['kind'].startswith('conv'):
            continue
        fam = family(op)
        tot[fam] = tot.get(fam, 0.0) + op['flops']
        if op['kind'] == 'conv_mfma':
            print(f"| {op['tag']} | {op['NB']}x{op['H']}x{op['W']} | {op['ksize']}x{op['ksize']}x{op['Ctot']} -> {op['Cout']} | {fam} | "
                  f"{op.get('ksplit', 1)} | {op['flops'] / 1e9:.2f} |")
    print()
    all_f = sum(tot.values())
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f'{k}: {v / 1e9:.1f} GFLOP ({100 * v / all_f:.1f} %)')


if __name__ == '__main__':
    main()
