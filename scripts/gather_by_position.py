#!/usr/bin/env python
"""Per-launch time of the LDS-staged gather BY POSITION in the step, from a rocprofv3 --kernel-trace CSV (VERDICT r05 "next" #3: the
kernel-stats summary shows 0.145 - 0.292 ms for one kernel on one workload; which launches are the slow ones?).  Every gather launch
is classified by the kernel that ran right before it on the stream (the first step's head from NCHW = k_layer MODE 7, the fused
last-layer + tail + next head = MODE 6, a plain layer = MODE 0, the standalone projections = MODE 3) and by its layer index within the step.

  python scripts/gather_by_position.py gpurun_out/<tag>/prof/**/ddp_kernel_trace.csv [--skip-calls 1]"""
import collections
import csv
import json
import re
import sys


def mode_of(name):
    m = re.search(r'k_layer<(\d+), (\d+)', name)
    if m:
        return 'k_layer MODE ' + m.group(2)
    for k in ('k_bev_q', 'k_bev_u_update', 'k_depth_update', 'k_gemm', 'k_msda_gather_lds'):
        if k in name:
            return k
    return name.split('(')[0][-40:]


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index('--skip-calls') + 1]) if '--skip-calls' in sys.argv else 1
    rows = [r for r in csv.DictReader(open(path)) if r['Kind'] == 'KERNEL_DISPATCH']
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    by_prev = collections.defaultdict(list)
    by_layer = collections.defaultdict(list)
    prev, layer, calls = None, 0, 0
    for r in rows:
        name = r['Kernel_Name']
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3          # us
        if 'k_msda_gather_lds' in name:
            gap = (int(r['Start_Timestamp']) - int(prev['End_Timestamp'])) / 1e3 if prev is not None else 0.0
            if calls > skip:                                                         # the first sample() calls of a process are warm-up
                by_prev[mode_of(prev['Kernel_Name']) if prev is not None else '-'].append((dur, gap))
                by_layer[layer].append(dur)
            layer += 1
        else:
            m = mode_of(name)
            if m in ('k_layer MODE 7', 'k_layer MODE 6', 'k_layer MODE 3', 'k_layer MODE 2', 'k_layer MODE 4', 'k_layer MODE 8', 'k_layer MODE 9'):
                layer = 0                                                            # a step head: the next gather is layer 0's
            if m in ('k_layer MODE 7',) or 'nchw_to_sb' in name:
                calls += 1 if (prev is None or 'nchw_to_sb' not in prev['Kernel_Name']) else 0
        prev = r

    def stat(v):
        v = sorted(v)
        return dict(n=len(v), mean_us=round(sum(v) / len(v), 1), median_us=round(v[len(v) // 2], 1), min_us=round(v[0], 1), max_us=round(v[-1], 1))
    out = {'by_previous_kernel': {k: dict(stat([d for d, _ in v]), gap_after_previous_us=round(sum(g for _, g in v) / len(v), 1))
                                  for k, v in sorted(by_prev.items())},
           'by_layer_index_in_step': {str(k): stat(v) for k, v in sorted(by_layer.items())}}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
