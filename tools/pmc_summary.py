#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic per launch.

Usage:  python tools/pmc_summary.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> [out.json]

Follows /opt/skills/guides/MI355X_MICROARCH.md section "HBM": FETCH_SIZE / WRITE_SIZE are collected in
separate passes (TCC slots), values are KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads, so
the read side is calibrated on a kernel of this same run whose byte count is known exactly
(k_transpose_in reads batch*m*8 bytes and writes m*bpad*8 bytes, 8 B per lane as every kernel here).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    files = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no counter_collection.csv under ' + dirname)
    per = defaultdict(lambda: [0.0, 0])
    disp = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            name = r['Kernel_Name']
            key = (r.get('Dispatch_Id'), name)
            disp[key] = disp.get(key, 0.0) + float(r['Counter_Value'])
    for (d, name), v in disp.items():
        per[name][0] += v
        per[name][1] += 1
    return per


def short(name):
    import re
    m = re.search(r'(k_[a-z_0-9]+(?:<\d+>)?)', name)
    if m:
        return m.group(1)
    return name.split('(')[0].split('::')[-1].strip() or name[:40]


def main():
    fetch = load(sys.argv[1], 'FETCH_SIZE')
    write = load(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, [0.0, 0])
        w, nw = write.get(name, [0.0, 0])
        out[short(name)] = {'launches': max(nf, nw),
                            'fetch_KiB_per_launch': f / nf if nf else None,
                            'write_KiB_per_launch': w / nw if nw else None}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
