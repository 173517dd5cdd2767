#!/usr/bin/env python
"""Every counter of a rocprofv3 --pmc pass per dispatch, normalised by GRBM_GUI_ACTIVE / 8 (cycles) where that helps.
  python tools/pmc_raw_table.py <pass dir> [name-substring]"""
import csv, sys, collections
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else 'conv_'
rows = list(csv.DictReader(open(f'{d}/pmc_counter_collection.csv')))
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(f'{d}/pmc_kernel_trace.csv'))}
disp = collections.OrderedDict()
for r in rows:
    disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
for k, v in disp.items():
    if flt not in v['name']:
        continue
    t = kt[k]
    dur = (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) / 1e3
    nm = v['name'].split('conv_')[-1][:60]
    print(f"{nm:62s} {dur:8.1f} us  " + '  '.join(f"{c}={x:.4g}" for c, x in v.items() if c != 'name'))
