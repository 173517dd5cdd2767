"""Summarise an alternating w2d_bench A/B log ('## <binary>' lines between runs): min of min / mean of avg per (shape, variant)."""
import re, sys, collections
cur = None; shape = None
d = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open(sys.argv[1]):
    if line.startswith('## '): cur = line[3:].strip(); continue
    if line.startswith('=='): shape = line[3:40].strip(); continue
    m = re.match(r'\s+(w2d[^m]*?)\s+min\s+([\d.]+) ms\s+avg\s+([\d.]+)', line)
    if m: d[(shape, m.group(1).strip())][cur].append((float(m.group(2)), float(m.group(3))))
base, new = sys.argv[2], sys.argv[3]
for k, v in d.items():
    b = v[base]; n = v[new]
    if not b or not n: continue
    bm = min(x[0] for x in b); nm = min(x[0] for x in n)
    ba = sum(x[1] for x in b) / len(b); na = sum(x[1] for x in n) / len(n)
    print(f"{k[0]:38s} {k[1]:18s} min {bm:.3f} -> {nm:.3f} ({(nm / bm - 1) * 100:+.1f}%)  avg {ba:.3f} -> {na:.3f} ({(na / ba - 1) * 100:+.1f}%)")
