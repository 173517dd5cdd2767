#!/usr/bin/env python
"""Where the idle matrix-pipe cycles of conv_wino2d_kernel go, per layer class of the 1080p 2x2-tiled plan.

  python tools/w2d_idle_budget.py --run  [--reps 5] [--out gpurun_out/w2d_budget.jsonl]     (GPU box: times every 3x3 layer of the plan)
  python tools/w2d_idle_budget.py --table gpurun_out/w2d_budget.jsonl > profiles/r06_w2d_idle_budget.md

Every conv_wino2d layer shape of the plan (plan-only handle: no GPU needed for the list) runs in tools/bin/w2d_bench with the
W2D_DBG_TIME instantiation of the kernel (wave 0 of every workgroup stamps s_memtime at entry / first MFMA / last MFMA / exit and sums
the cycles it sat in the K loop's `s_waitcnt` + barrier pairs), both tile widths; the faster one is the row of the table (what the
autotuner picks).  Accounting, per WAVE SLOT (a CU holds 8 = two per SIMD; a slot's fair share of the matrix pipe is one
v_mfma_f32_32x32x2_f32 = 64 cycles every 128):

  span of the launch T  =  mfma (128 cycles x MFMAs of a wave x workgroups per slot)
                         + prologue + K-loop waits + K-loop issue gaps (loop - waits - 128 x MFMAs) + epilogue       (x workgroups per slot)
                         + CU idle (T - the summed lifetimes of the slot's workgroups: tail round, imbalance, launch ramp)

which sums to T by construction.  T = hipEvent time of the instrumented launch x the shader clock measured inside it (every workgroup also
reads the 100 MHz s_memrealtime at entry and exit: cycles per tick).  s_memtime stamps only compare within one CU, so the launch span is
cross-checked per CU (HW_ID / XCC_ID of every workgroup): `longest CU span / T` must be ~1, and `instrumented / plain` says what the stamps
cost.  Masked MFMA columns (tiles hanging over the level's edge) are inside `mfma`; their share is its own column.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frame-interpolation_amd'))


def plan_shapes(B, H, W):
    from film_hip.engine import FilmEngine
    from film_hip.options import PUBLISHED
    eng = FilmEngine(PUBLISHED, device=-1)
    ops = [o for o in eng.plan(B, H, W)['ops'] if o['kind'] == 'conv_mfma' and (o['tile'] & 8192) and o.get('ksplit', 1) <= 1]
    shapes = {}
    for o in ops:
        key = (o['NB'], o['H'], o['W'], o['Ctot'], o['Cout'])
        shapes.setdefault(key, []).append(o['tag'])
    return shapes


def run(args):
    shapes = plan_shapes(*args.shape)
    exe = os.path.join(ROOT, 'tools', 'bin', 'w2d_bench')
    with open(args.out, 'w') as out:
        for (NB, H, W, C, Cout), tags in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4]):
            cmd = [exe, str(args.reps), '-2', '=w2d 64 time,=w2d 32 time,=w2d 32 ns2 time,=w2d 64,=w2d 32,=w2d 32 ns2 plain', str(NB), str(H), str(W), str(C), str(Cout)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            plain = {}
            for line in r.stdout.splitlines():
                s = line.strip()
                if (s.startswith('w2d 64 ') or s.startswith('w2d 32 ')) and ' min ' in s:
                    plain[s.split(' min ')[0].strip()] = float(s.split('min')[1].split('ms')[0])
            for line in r.stdout.splitlines():
                if '[time-json]' in line:
                    d = json.loads(line.split('[time-json]')[1])
                    d['tags'] = tags
                    # the un-instrumented twin of the timed variant: 'w2d 64 time' -> 'w2d 64', 'w2d 32 ns2 time' -> 'w2d 32 ns2 plain' (two DMA stages)
                    twin = d.get('variant', f'w2d {d["bn"]} time').replace(' time', '')
                    d['plain_ms'] = plain.get(twin + ' plain' if 'ns2' in twin else twin)
                    out.write(json.dumps(d) + '\n')
                    out.flush()
            print(NB, H, W, C, Cout, plain, flush=True)


def budget(d):
    bn, nwg = d['bn'], d['nwg']
    slots = 256 * (2 if bn == 32 else 1)           # workgroup slots of the chip (238-256 VGPRs: two waves per SIMD)
    T = d['ms'] * 1e6 * d['ghz']                    # the instrumented launch in shader cycles (event time x measured clock)
    per_slot = nwg / slots                          # workgroups a slot runs
    n_mfma = (d['C'] // 8) * 24                     # per wave
    mfma = 128.0 * n_mfma * per_slot
    pro, wait, epi = d['prologue'] * per_slot, d['wait'] * per_slot, d['epilogue'] * per_slot
    gap = (d['loop'] - d['wait']) * per_slot - mfma
    idle = T - d['life'] * per_slot
    tiles = nwg // (d['Cout'] // bn)
    masked = 1.0 - (d['NB'] * d['H'] * d['W']) / (tiles * 256.0)
    return {'T': T, 'mfma': mfma, 'prologue': pro, 'wait': wait, 'gap': gap, 'epilogue': epi, 'idle': idle, 'masked': masked,
            'rounds': per_slot, 'ghz': d['ghz'], 'span_ratio': d['span_max'] / T, 'instr_ratio': d['ms'] / d['plain_ms'] if d.get('plain_ms') else None}


def table(args):
    rows = [json.loads(l) for l in open(args.table)]
    best = {}
    for d in rows:
        key = (d['NB'], d['H'], d['W'], d['C'], d['Cout'])
        if key not in best or (d.get('plain_ms') or d['ms']) < (best[key].get('plain_ms') or best[key]['ms']):
            best[key] = d
    classes = ((0, 64), (65, 128), (129, 256), (257, 528), (529, 100000))
    print('# conv_wino2d_kernel: where the idle matrix-pipe cycles go (1080p 2x2-tiled plan, every 3x3 layer on this kernel)\n')
    print('Accounting: see `tools/w2d_idle_budget.py` (docstring).  All columns are shares of the launch span T; '
          '`mfma` = the matrix pipe busy (fair share 128 cycles per MFMA per wave slot), the rest is idle pipe by cause.\n')
    hdr = '| layer shape (batch x H x W, K -> Cout) | launches | tile | ms (plain) | rounds | mfma | of it masked | prologue | K-loop waits | K-loop issue gaps | epilogue | CU idle | clock GHz | longest CU span / T | instrumented / plain |'
    print(hdr)
    print('|' + '---|' * (hdr.count('|') - 1))
    agg = {c: dict(ms=0.0, mfma=0.0, prologue=0.0, wait=0.0, gap=0.0, epilogue=0.0, idle=0.0, masked=0.0, n=0) for c in classes}
    tot = dict(ms=0.0, mfma=0.0, prologue=0.0, wait=0.0, gap=0.0, epilogue=0.0, idle=0.0, masked=0.0, n=0)
    for key, d in sorted(best.items(), key=lambda kv: (kv[0][3], -kv[0][1])):
        b = budget(d)
        ms = d.get('plain_ms') or d['ms']
        n = len(d['tags'])
        sh = {k: b[k] / b['T'] for k in ('mfma', 'prologue', 'wait', 'gap', 'epilogue', 'idle')}
        print(f"| {key[0]}x{key[1]}x{key[2]}, {key[3]} -> {key[4]} | {n} | 8x32x{d['bn']}{' (2 stages)' if 'ns2' in d.get('variant', '') else ''} | {ms:.3f} | {b['rounds']:.2f} | {sh['mfma']:.3f} | {b['masked']:.3f} | "
              f"{sh['prologue']:.3f} | {sh['wait']:.3f} | {sh['gap']:.3f} | {sh['epilogue']:.3f} | {sh['idle']:.3f} | {b['ghz']:.2f} | {b['span_ratio']:.3f} | "
              f"{b['instr_ratio'] or float('nan'):.3f} |")
        for c in classes:
            if c[0] <= key[3] <= c[1]:
                for a in (agg[c], tot):
                    a['ms'] += ms * n
                    a['n'] += n
                    for k in sh:
                        a[k] += sh[k] * ms * n
                    a['masked'] += b['masked'] * sh['mfma'] * ms * n
    print('\n## By input-channel class (time-weighted; ms = sum over the launches of one forward)\n')
    print('| K | launches | ms | mfma | masked (share of the launch time) | prologue | K-loop waits | K-loop issue gaps | epilogue | CU idle |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for c, a in list(agg.items()) + [(('all',), tot)]:
        if not a['n']:
            continue
        name = 'all' if c == ('all',) else (f'{c[0]}..{c[1]}' if c[1] < 100000 else f'{c[0]}..')
        print(f"| {name} | {a['n']} | {a['ms']:.2f} | " + ' | '.join(f"{a[k] / a['ms']:.3f}" for k in ('mfma', 'masked', 'prologue', 'wait', 'gap', 'epilogue', 'idle')) + ' |')
    print('\n## What each idle term is worth (ms of one forward if it went to zero with everything else unchanged)\n')
    print('| term | ms |')
    print('|---|---|')
    for k, label in (('masked', 'masked MFMA columns'), ('prologue', 'prologue'), ('wait', 'K-loop s_waitcnt + barrier'), ('gap', 'K-loop issue gaps'),
                     ('epilogue', 'epilogue'), ('idle', 'CU idle (tail round, imbalance, ramp)')):
        print(f'| {label} | {tot[k]:.2f} |')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--run', action='store_true')
    ap.add_argument('--table')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--out', default='gpurun_out/w2d_budget.jsonl')
    ap.add_argument('--shape', nargs=3, type=int, default=[4, 576, 960])
    args = ap.parse_args()
    if args.run:
        run(args)
    if args.table:
        table(args)


if __name__ == '__main__':
    main()
