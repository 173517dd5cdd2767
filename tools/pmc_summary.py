#!/usr/bin/env python
"""Summarises the rocprofv3 PMC passes of tools/gpu_pmc.sh (csv) for the LAST forward of each run: per kernel
class the summed counters, derived MFMA-pipe busy / effective clock / L2 hit rate / HBM-side bytes, as markdown +
the JSON bench.py reads for roofline.traffic.

  python tools/pmc_summary.py gpurun_out/pmc --md profiles/rNN_pmc_summary.md --json profiles/rNN_pmc_conv.json
"""
import argparse
import collections
import csv
import json
import os
import re


def short(name):
    m = re.search(r'(conv_buf_kernel|conv_halo_kernel|conv_winox3_kernel|conv_wino43_kernel|conv_wino2d_kernel|conv_fold4_kernel|conv_wino_kernel|conv_foldx3_kernel|conv_splitk_reduce_kernel|conv_halo_split_kernel|conv_igemm_kernel|conv_c3_kernel|conv_pw_kernel|flow_head_kernel|warp_vec_kernel|warp_c3_kernel|'
                  r'pool_vec_kernel|pool_c3_kernel|flow_up_kernel|flow_add_kernel|pack_flow_kernel|frame_to_tiles_kernel|'
                  r'tiles_to_frame_kernel)', name)
    return m.group(1) if m else None


def klass(k):
    return 'conv_mfma' if k in ('conv_buf_kernel', 'conv_halo_kernel', 'conv_wino_kernel', 'conv_wino43_kernel', 'conv_wino2d_kernel', 'conv_fold4_kernel', 'conv_winox3_kernel', 'conv_foldx3_kernel', 'conv_splitk_reduce_kernel',
                              'conv_halo_split_kernel', 'conv_igemm_kernel', 'conv_c3_kernel') else k.replace('_kernel', '')


def last_forward(d):
    rows = list(csv.DictReader(open(os.path.join(d, 'pmc_counter_collection.csv'))))
    kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(os.path.join(d, 'pmc_kernel_trace.csv')))}
    disp = collections.OrderedDict()
    for r in rows:
        k = short(r['Kernel_Name'])
        if k is None:
            continue
        e = disp.setdefault(int(r['Dispatch_Id']), {'k': k})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(disp)
    ends = [i for i in ids if disp[i]['k'] == 'tiles_to_frame_kernel']
    if len(ends) < 2:
        raise SystemExit(f'{d}: fewer than two forwards found')
    sel = [i for i in ids if ends[-2] < i <= ends[-1]]
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for i in sel:
        e = disp[i]
        t = kt[str(i)]
        # the dominant kernel also gets a row of its own (it is what roofline.frac in the bench line is about)
        for key in ([klass(e['k'])] + (['conv_wino43_kernel (dominant kernel, part of conv_mfma)'] if e['k'] == 'conv_wino43_kernel' else []) +
                    (['conv_wino2d_kernel (nested Winograd, part of conv_mfma)'] if e['k'] == 'conv_wino2d_kernel' else [])):
            c = out[key]
            c['launches'] += 1
            c['us'] += (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
            for n, v in e.items():
                if n != 'k':
                    c[n] += v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--md')
    ap.add_argument('--json')
    ap.add_argument('--command', default='tools/gpu_pmc.sh')
    args = ap.parse_args()
    passes = {p: last_forward(os.path.join(args.dir, p)) for p in ('sq1', 'sq2', 'tcc1', 'fetch', 'write')
              if os.path.isdir(os.path.join(args.dir, p))}
    lines = [f'# PMC passes ({args.command}: one rocprofv3 process per pass, `--kernel-trace --pmc ...` only), last forward of each run',
             '', 'Units: SQ_* counters are summed over all SEs/XCDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for',
             'cycles); SQ_VALU_MFMA_BUSY_CYCLES = 64 x MFMA wave-instructions for v_mfma_f32_32x32x2_f32 (1024 SIMDs);',
             'FETCH_SIZE / WRITE_SIZE are KB at the L2 <-> fabric interface (Infinity-Cache hits included), FETCH_SIZE is doubled',
             'per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of wide coalesced reads at 64 B).', '']
    js = {}
    classes = sorted({c for p in passes.values() for c in p}, key=lambda c: -passes.get('sq1', passes[next(iter(passes))])[c]['us'])
    for c in classes:
        lines.append(f'## {c}')
        lines.append('| counter | value | reading |')
        lines.append('|---|---|---|')
        sq1 = passes.get('sq1', {}).get(c)
        if sq1:
            gui = sq1['GRBM_GUI_ACTIVE'] / 8
            clk = gui / sq1['us'] / 1e3
            lines.append(f"| launches / kernel time | {sq1['launches']:.0f} / {sq1['us'] / 1e3:.3f} ms | (under the profiler) |")
            # launches of a few microseconds: GRBM_GUI_ACTIVE includes dispatch ramp-up outside the kernel-trace timestamps and
            # the quotient is not a clock (round 1 printed 4.7-5.5 "GHz" here) - only quoted for classes averaging >= 50 us
            long_enough = sq1['us'] / sq1['launches'] >= 50
            clk_txt = f'effective clock {clk:.2f} GHz' if long_enough else 'launches too short for a clock estimate'
            lines.append(f"| GRBM_GUI_ACTIVE | {sq1['GRBM_GUI_ACTIVE']:.4g} | {gui:.4g} cycles -> {clk_txt} |")
            if sq1.get('SQ_VALU_MFMA_BUSY_CYCLES'):
                busy = sq1['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / gui
                lines.append(f"| SQ_VALU_MFMA_BUSY_CYCLES | {sq1['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} | **MFMA pipe busy {busy * 100:.1f} %** of the active cycles |")
                js.setdefault(c, {})['mfma_busy'] = busy
            wc = sq1['SQ_WAVE_CYCLES']
            lines.append(f"| SQ_WAVE_CYCLES / WAIT_INST_ANY / WAIT_ANY / ACTIVE_INST_ANY | {wc:.4g} / {sq1['SQ_WAIT_INST_ANY']:.4g} / {sq1['SQ_WAIT_ANY']:.4g} / {sq1['SQ_ACTIVE_INST_ANY']:.4g} | issue-stall {sq1['SQ_WAIT_INST_ANY'] / wc * 100:.0f} %, parked at waitcnt/barrier {sq1['SQ_WAIT_ANY'] / wc * 100:.0f} %, issuing {sq1['SQ_ACTIVE_INST_ANY'] / wc * 100:.0f} % |")
            lines.append(f"| SQ_WAVES | {sq1['SQ_WAVES']:.4g} | |")
            js.setdefault(c, {}).update(clock_ghz=clk if long_enough else None, launches=sq1['launches'])
        sq2 = passes.get('sq2', {}).get(c)
        if sq2:
            lines.append(f"| SQ_INSTS_VALU / LDS / VMEM | {sq2['SQ_INSTS_VALU']:.4g} / {sq2['SQ_INSTS_LDS']:.4g} / {sq2['SQ_INSTS_VMEM']:.4g} | |")
            if sq2.get('SQ_LDS_IDX_ACTIVE'):
                lines.append(f"| SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT | {sq2['SQ_LDS_IDX_ACTIVE']:.4g} / {sq2['SQ_LDS_BANK_CONFLICT']:.4g} | bank-conflict cycles {sq2['SQ_LDS_BANK_CONFLICT'] / sq2['SQ_LDS_IDX_ACTIVE'] * 100:.1f} % of LDS-active cycles |")
        tcc = passes.get('tcc1', {}).get(c)
        if tcc and tcc.get('TCC_REQ_sum'):
            lines.append(f"| TCC_REQ / HIT / MISS | {tcc['TCC_REQ_sum']:.4g} / {tcc['TCC_HIT_sum']:.4g} / {tcc['TCC_MISS_sum']:.4g} | L2 hit rate {tcc['TCC_HIT_sum'] / (tcc['TCC_HIT_sum'] + tcc['TCC_MISS_sum']) * 100:.0f} % |")
        f, w = passes.get('fetch', {}).get(c), passes.get('write', {}).get(c)
        if f and w:
            fb, wb = f['FETCH_SIZE'] * 1024 * 2, w['WRITE_SIZE'] * 1024
            lines.append(f"| FETCH_SIZE / WRITE_SIZE (KB) | {f['FETCH_SIZE']:.4g} / {w['WRITE_SIZE']:.4g} | read {fb / 1e9:.2f} GB (corrected x2), written {wb / 1e9:.2f} GB per forward; {(fb + wb) / f['launches'] / 1e6:.1f} MB per launch |")
            js.setdefault(c, {}).update(fetch_bytes=fb, write_bytes=wb, hbm_bytes_per_launch=(fb + wb) / f['launches'])
        lines.append('')
    if args.md:
        open(args.md, 'w').write('\n'.join(lines) + '\n')
    else:
        print('\n'.join(lines))
    if args.json and 'conv_mfma' in js:
        # which build of the kernels the counters belong to: the bench line every pass printed carries film_version() (a hash of
        # csrc/ + the public header); bench.py replays roofline.traffic from this file ONLY when its own build string is the same
        build = None
        for pname in ('fetch', 'write', 'sq1', 'sq2', 'tcc1'):
            log = os.path.join(args.dir, pname + '.log')
            if os.path.isfile(log):
                for line in open(log, errors='replace'):
                    if line.startswith('{"metric"'):
                        try:
                            b = json.loads(line).get('build')
                        except ValueError:
                            continue
                        if build is not None and b != build:
                            raise SystemExit(f'{log}: build {b} differs from the other passes ({build})')
                        build = b
        out = {'build': build, 'source': f'{args.md or args.dir} ({args.command}, last forward)', 'kernel': 'conv class: conv_wino2d (dominant) / conv_wino43 / conv_buf (+ split-K reduce) + first-layer conv_c3',
               'fetch_correction': 2.0, 'classes': js, 'hbm_bytes_per_launch': js['conv_mfma'].get('hbm_bytes_per_launch'),
               'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md; Infinity-Cache hits are included in these counters'}
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
