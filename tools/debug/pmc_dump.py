"""Average of every counter per kernel from rocprofv3 --pmc output directories:  python tools/debug/pmc_dump.py DIR [DIR ...] [substr]"""
import collections
import csv
import glob
import os
import sys

dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
sub = [a for a in sys.argv[1:] if not os.path.isdir(a)]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if sub and not any(s in k for s in sub):
                continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    print(k[:110])
    for n, v in sorted(c.items()):
        print('   %-28s launches %3d  mean %16.0f' % (n, len(v), sum(v) / len(v)))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        busy = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
        gui = sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'])
        print('   MfmaUtil %.1f %%   (gui per XCD %.0f cycles)' % (100 * busy / (1024 * gui / 8), gui / 8))
