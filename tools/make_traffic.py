#!/usr/bin/env python3
"""Per-launch HBM traffic of the hot-path kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KiB).
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (a wide coalesced read stream is counted at half its bytes).

    python tools/make_traffic.py <fetch dir> <write dir> <out.json>
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laff_amd.build import source_hash  # noqa: E402

ENTRY = [('sim_strip_kernel', 'sim_gemm'), ('fc_strip_kernel', 'fc_act_bn'), ('fc_strip_pack_kernel', 'fc_strip_pack'), ('rank_export_kernel', 'rank_export'), ('gemm_nt_x3_fused_grouped_kernel', 'fc_act_bn'), ('gemm_nt_x3_grouped_kernel', 'fc_act_bn'), ('gemm_nt_grouped_kernel', 'fc_act_bn'), ('gemm_nt_x3_kernel', 'sim_gemm'),
         ('gemm_nt_kernel', 'sim_gemm'), ('split_rows_kernel', 'split_rows'), ('fuse_reg_kernel', 'fuse'), ('fuse_stream_kernel', 'fuse'),
         ('frame_fuse_kernel', 'frame_fuse'), ('row_dot_gt_kernel', 'row_dot_gt'), ('fc_gather_kernel', 'fc_gather'),
         ('rank_metrics_kernel', 'rank_metrics'), ('pack_rows_kernel', 'pack_rows'), ('rank_prepare_kernel', 'rank_prepare'),
         ('rank_resolve_kernel', 'rank_resolve'), ('split_rows_kernel', 'row_scales')]


def collect(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            for pat, name in ENTRY:
                if pat in r['Kernel_Name']:
                    out[name].append(float(r['Counter_Value']) * 1024)
                    break
    return out


def main():
    fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / max(1, len(fetch[k]))
        w = sum(write[k]) / max(1, len(write[k]))
        kernels[k] = {'launches_seen': len(fetch[k]), 'fetch_reported_MB': round(f / 1e6, 1), 'fetch_corrected_MB': round(2 * f / 1e6, 1),
                      'write_MB': round(w / 1e6, 1)}
    json.dump({'_comment': 'average bytes per launch over every launch of the kernel in `python3 bench.py --steps 20 --warmup 5 --profile-steps 5 '
                           '--no-cpu-baseline --no-extra-modes` (warm-up, graph replays and the eager profiling pass alike); FETCH_SIZE x2 per '
                           'MI355X_MICROARCH.md; made by tools/make_traffic.py from separate --pmc passes',
               'workload': 'c4_40kx10k', 'precision': 'fp16', 'fc_precision': 'fp16x3', 'src_sha': source_hash(), 'kernels': kernels},
              open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == '__main__':
    main()
