#!/usr/bin/env python
"""Summarises a rocprofv3 run of bench.py (rocpd sqlite .db from `rocprofv3 --kernel-trace --stats`, or the
PMC variant) into the markdown tables committed under profiles/.

  python tools/rocprof_summary.py gpurun_out/rocprof/bench_results.db --forwards 5 > profiles/r01_kernel_stats.md

Only the steady-state forwards are counted: the engine autotunes tile shapes when a plan is created, which
launches every candidate once; those trial launches are excluded by keeping the LAST `forwards` x
(launches per forward) dispatches of the engine's kernels."""
import argparse
import re
import sqlite3
import sys
from collections import defaultdict

ENGINE = re.compile(r'conv_buf_kernel|conv_halo_kernel|conv_wino43_kernel|conv_wino2d_kernel|conv_fold4_kernel|conv_wino_kernel|conv_winox3_kernel|conv_foldx3_kernel|conv_halo_split_kernel|conv_igemm_kernel|conv_c3_kernel|conv_splitk_reduce_kernel|conv_pw_kernel|flow_head_kernel|warp_vec_kernel|warp_c3_kernel|'
                    r'pool_vec_kernel|pool_c3_kernel|flow_up_kernel|flow_add_kernel|pack_flow_kernel|frame_to_tiles_kernel|'
                    r'tiles_to_frame_kernel')


def short(name):
    name = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    if name.startswith('void '):
        name = name[5:]
    return re.sub(r'\(.*\)$', '', name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--forwards', type=int, required=True, help='forward passes in the run (warmup + steps + 1 profiled)')
    ap.add_argument('--launches-per-forward', type=int, default=0, help='0 = infer from pack_flow/conv_pw<3> counts')
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    rows = [(n, s, e) for n, s, e in cur.execute('select name, start, end from kernels order by start') if ENGINE.search(n)]
    # the last launch of a forward: tiles_to_frame (film_interpolate: once per forward, never in autotune), else the RGB
    # head conv_pw<3> (film_forward with fuse bit 16 off; with it on the head is part of the last decoder convolution)
    last = 'tiles_to_frame_kernel' if any('tiles_to_frame_kernel' in r[0] for r in rows) else 'conv_pw_kernel<3>'
    heads = [i for i, r in enumerate(rows) if last in r[0]]
    if len(heads) < args.forwards:
        sys.exit(f'found {len(heads)} forwards, expected {args.forwards}')
    per_fwd = args.launches_per_forward or (heads[-1] - heads[-2])
    first = heads[-args.forwards] - per_fwd + 1
    steady = rows[first:heads[-1] + 1]
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for n, s, e in steady:
        a = agg[short(n)]
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f'steady-state forwards: {args.forwards}, launches per forward: {per_fwd}, kernel time per forward: '
          f'{total / args.forwards / 1e3:.3f} ms\n')
    print('| kernel | calls/forward | total ms/forward | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'| `{k}` | {a[0] / args.forwards:.1f} | {a[1] / args.forwards / 1e3:.3f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | '
              f'{a[3]:.1f} | {100 * a[1] / total:.2f} |')
    conv = sum(a[1] for k, a in agg.items() if k.startswith(('conv_buf', 'conv_halo', 'conv_wino', 'conv_fold', 'conv_igemm', 'conv_c3')))
    nconv = sum(a[0] for k, a in agg.items() if k.startswith(('conv_buf', 'conv_halo', 'conv_wino', 'conv_fold', 'conv_igemm', 'conv_c3')))
    print(f'\nMFMA conv kernels (conv_wino2d / conv_fold4 / conv_wino43 / conv_wino / conv_winox3 / conv_halo / conv_halo_split / conv_foldx3 / conv_buf, all tile shapes, + the first-layer conv_c3_kernel / conv_igemm_kernel): {nconv / args.forwards:.0f} launches/forward, '
          f'{conv / args.forwards / 1e3:.3f} ms/forward, average launch {conv / nconv:.1f} us')


if __name__ == '__main__':
    main()
