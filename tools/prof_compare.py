#!/usr/bin/env python
"""Per-layer table of one or two engine profiles (bench.py --profile-out): ms, TFLOP/s, time lost vs 150 TFLOP/s.
  python tools/prof_compare.py new.json [old.json]"""
import json, sys
new = json.load(open(sys.argv[1]))
old = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
oldm = {o['tag']: o for o in old['ops']} if old else {}
rows = []
for o in new['ops']:
    if o['kind'] != 'conv_mfma':
        continue
    ideal = o['flops'] / 150e12 * 1e3
    rows.append((o['ms'] - ideal, o))
rows.sort(key=lambda r: -r[0])
tot = sum(o['ms'] for _, o in rows)
fl = sum(o['flops'] for _, o in rows)
print(f'conv total {tot:.2f} ms, {fl / tot / 1e9:.1f} TFLOP/s')
cum = 0
for lost, o in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    cum += lost
    tf = o['flops'] / (o['ms'] * 1e-3) / 1e12
    extra = ''
    if o['tag'] in oldm:
        q = oldm[o['tag']]
        extra = f"  was {q['ms']:6.3f} ms tile {q['tile']:3d}"
    print(f"lost {lost:6.3f} cum {cum:6.2f}  {o['ms']:6.3f} ms {tf:6.1f} TF tile {o['tile']:3d}{extra}  {o['tag']}")
