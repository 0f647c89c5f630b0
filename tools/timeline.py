#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: for the LAST complete step (a step starts at the first
kernel after the longest idle gap pattern) print every kernel's start offset, duration and the idle gap before it.

    python tools/timeline.py <dir with *_kernel_trace.csv> [kernels per step]
"""
import csv
import glob
import os
import sys


def main():
    f = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    # steps end with the metrics kernel; take the last full step
    ends = [i for i, r in enumerate(rows) if 'rank_metrics' in r['Kernel_Name']]
    if len(ends) < 2:
        print('no two steps found')
        return
    # the last step that is a plain replay of the timed graph (no laff_stamp launches of bench.py's instrumented capture in it)
    step = None
    for k in range(len(ends) - 1, 0, -1):
        cand = rows[ends[k - 1] + 1: ends[k] + 1]
        if not any('stamp_kernel' in r['Kernel_Name'] for r in cand):
            step, ends = cand, ends[:k + 1]
            break
    if step is None:
        step = rows[ends[-2] + 1: ends[-1] + 1]
    t0 = int(step[0]['Start_Timestamp'])
    prev_end = t0
    busy = 0.0
    print('%9s %9s %8s  %s' % ('start_us', 'dur_us', 'gap_us', 'kernel'))
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%9.1f %9.1f %8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:110]))
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    prev_step_end = int(rows[ends[-2]]['End_Timestamp'])
    print('step span %.1f us, kernel busy %.1f us, idle inside the step %.1f us, gap from the previous step %.1f us, %d kernels'
          % (span, busy, span - busy, (t0 - prev_step_end) / 1e3, len(step)))


if __name__ == '__main__':
    main()
