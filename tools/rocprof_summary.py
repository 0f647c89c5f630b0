#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/.

    python tools/rocprof_summary.py <dir with *_kernel_trace.csv / *_counter_collection.csv> [--out profiles/NAME.txt]

Kernel trace  -> per-kernel count / avg / min / max / total duration (us) and share of GPU time.
Counter files -> per-kernel average of every collected counter; FETCH_SIZE / WRITE_SIZE (reported in KiB) are also
shown as bytes per launch, FETCH_SIZE additionally x2 (MI355X_MICROARCH.md: on gfx950 rocprofv3 counts a wide
coalesced read stream at half its bytes).
"""
import collections
import csv
import glob
import os
import sys


def short(name, n=96):
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    d = sys.argv[1]
    out = sys.stdout
    if '--out' in sys.argv:
        out = open(sys.argv[sys.argv.index('--out') + 1], 'w')
    for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
            a = agg.setdefault(r['Kernel_Name'], [0, 0.0, 1e30, 0.0, r.get('VGPR_Count', ''), r.get('LDS_Block_Size', ''), r.get('Grid_Size', ''), []])
            a[7].append(dur)
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
        tot = sum(a[1] for a in agg.values())
        print('# kernel trace: %s  (total GPU kernel time %.1f us)' % (os.path.basename(f), tot), file=out)
        print('%-98s %6s %10s %10s %10s %10s %11s %6s %5s %7s' % ('kernel', 'calls', 'avg_us', 'median_us', 'min_us', 'max_us', 'total_us', 'pct', 'vgpr', 'lds'), file=out)
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            med = sorted(a[7])[len(a[7]) // 2]
            print('%-98s %6d %10.2f %10.2f %10.2f %10.2f %11.1f %5.1f%% %5s %7s' % (short(k), a[0], a[1] / a[0], med, a[2], a[3], a[1], 100 * a[1] / tot, a[4], a[5]), file=out)
        print(file=out)
    for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            agg.setdefault((r['Kernel_Name'], r['Counter_Name']), []).append(float(r['Counter_Value']))
        print('# counters: %s' % os.path.basename(f), file=out)
        print('%-98s %-28s %8s %18s  %s' % ('kernel', 'counter', 'launches', 'avg per launch', 'note'), file=out)
        for (k, c), v in agg.items():
            avg = sum(v) / len(v)
            note = ''
            if c == 'FETCH_SIZE':
                note = '= %.1f MB as reported; x2 (gfx950 wide-read correction) = %.1f MB' % (avg * 1024 / 1e6, 2 * avg * 1024 / 1e6)
            elif c == 'WRITE_SIZE':
                note = '= %.1f MB' % (avg * 1024 / 1e6)
            print('%-98s %-28s %8d %18.1f  %s' % (short(k), c, len(v), avg, note), file=out)
        print(file=out)


if __name__ == '__main__':
    main()
