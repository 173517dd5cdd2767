#!/usr/bin/env python
"""Per-dispatch table of a rocprofv3 --pmc pass (csv): duration, effective clock, MFMA-pipe busy, wave stall split.
  python tools/pmc_table.py gpurun_out/pmc_conv/sq1 [name-substring]"""
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
    gui = v.get('GRBM_GUI_ACTIVE', 0) / 8
    mf = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024
    wc = max(v.get('SQ_WAVE_CYCLES', 1), 1)
    nm = v['name'].split('conv_')[-1][:48]
    print(f"{nm:50s} {dur:8.1f} us  clk {gui / dur / 1e3:5.2f} GHz  mfma_busy {mf / max(gui, 1) * 100:5.1f}%  "
          f"parked {v.get('SQ_WAIT_ANY', 0) / wc * 100:4.1f}%  issue-stall {v.get('SQ_WAIT_INST_ANY', 0) / wc * 100:4.1f}%  "
          f"active {v.get('SQ_ACTIVE_INST_ANY', 0) / wc * 100:4.1f}%  waves {v.get('SQ_WAVES', 0):.0f}")
