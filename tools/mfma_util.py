#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE.

SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the shader cycles an MFMA occupies its pipe (32 per v_mfma_f32_32x32x16_*:
MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is the launch's duration in shader cycles AT THE CLOCK ACTUALLY HELD.
    MfmaUtil = MFMA_BUSY / (SIMDs x GUI_ACTIVE / XCDS),  SIMDs = 256 CUs x 4
(rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs: the sum divided by the launch time comes out at 8 x ~2 GHz)
so it is a fraction of the MFMA peak at the sustained clock (the chip sits at its 1400 W cap under these kernels and clocks
1.7-1.9 GHz, not the 2.4 GHz the 2.5 PFLOP/s figure assumes).

    python tools/mfma_util.py <dir with *_counter_collection.csv>
"""
import collections
import csv
import glob
import os
import sys

SIMDS = 256 * 4
XCDS = 8


def main():
    agg = collections.OrderedDict()
    for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            per[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] = float(r['Counter_Value'])
        for (_, k), c in per.items():
            if not k.startswith('void laff::') and not k.startswith('laff::'):
                continue
            agg.setdefault(k, []).append(c)
    print('# MFMA utilisation per launch (averages over the launches of `bench.py --steps 5 --warmup 2 --no-cpu-baseline`)')
    print('%-92s %8s %14s %14s %9s %10s' % ('kernel', 'launches', 'mfma_busy_cyc', 'gui_active_sum', 'MfmaUtil', 'lds_confl'))
    for k, rows in agg.items():
        n = len(rows)
        busy = sum(r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for r in rows) / n
        gui = sum(r.get('GRBM_GUI_ACTIVE', 0.0) for r in rows) / n
        conf = sum(r.get('SQ_LDS_BANK_CONFLICT', 0.0) for r in rows) / n
        util = busy / (SIMDS * gui / XCDS) if gui else 0.0
        name = k if len(k) <= 92 else k[:89] + '...'
        print('%-92s %8d %14.0f %14.0f %8.1f%% %10.0f' % (name, n, busy, gui, 100 * util, conf))


if __name__ == '__main__':
    main()
