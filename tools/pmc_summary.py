#!/usr/bin/env python3
"""summarise a rocprofv3 --pmc csv (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...) per kernel+grid"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48]
    key = (k, r['Grid_Size'])
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':50s} {'grid':>9s} {'n':>4s} {'us':>9s} {'GHz':>6s} {'MfmaUtil%':>9s}")
for key, c in agg.items():
    if 'conv' not in key[0]: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[key]) / len(dur[key])
    gui = m['GRBM_GUI_ACTIVE'] / 8.0          # summed over 8 XCDs
    util = 100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024)
    print(f"{key[0]:50s} {key[1]:>9s} {len(dur[key]):4d} {us:9.1f} {gui/us/1e3:6.2f} {util:9.1f}")
