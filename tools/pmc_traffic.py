#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/pmc_traffic.json (HBM bytes per launch and per phase).

Usage: python tools/pmc_traffic.py <fetch pass dir> <write pass dir> <raw_entries> <n> <batch> <out.json> [note]

Method (MI355X_MICROARCH.md, section HBM): the two counters are collected in separate passes; values are KiB;
WRITE_SIZE is exact; FETCH_SIZE under-reports wide coalesced reads on gfx950, so the read side is calibrated on
k_transpose_in of the same run, whose reads are known exactly (batch * m * 8 bytes per launch, m = raw_entries for
the value transposition and n for the right-hand-side one; one of each per step).
"""
import json
import re
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmc_summary import load, short   # noqa: E402
from parapint_amd._native import kernel_source_sha1   # noqa: E402

PHASE_OF = [
    (r'^k_gather_level|^k_gather_flat|^k_gather_tiles|^k_chain_front', 'factor_levels'),
    (r'^k_scale_level|^k_front_invert|^k_scale_wide', 'factor_levels'),
    (r'^k_count_codes|^k_schur_tiles|^k_schur_mfma|^k_schur_reduce|^k_scatter_schur', 'schur_tiles'),
    (r'^k_bcr_fwd|^k_bcr_bwd|^k_bcr_rhs', 'coupling_solve'),
    (r'^k_add_q|^k_ldl_|^k_bk_factor|^k_write_tail|^k_publish_status|^k_dense_|^k_bcr_|^k_btd_|^k_corner_add', 'dense_S'),
    (r'^k_fwd_level|^k_chain_fwd', 'fwd_levels'), (r'^k_fwd_coupling|^k_rs_reduce', 'fwd_coupling'),
    (r'^k_coupling_solve', 'coupling_solve'), (r'^k_bwd_level|^k_chain_bwd|^k_transpose_out', 'bwd_levels'),
    (r'^k_residual|^k_corner_rc', 'solution_check'),
]


def main():
    fetch = load(sys.argv[1], 'FETCH_SIZE')
    write = load(sys.argv[2], 'WRITE_SIZE')
    raw_entries, n, batch = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    out_path = sys.argv[6]
    note = sys.argv[7] if len(sys.argv) > 7 else ''
    per = {}
    for name in set(fetch) | set(write):
        s = short(name)
        f, nf = fetch.get(name, [0.0, 0])
        w, nw = write.get(name, [0.0, 0])
        e = per.setdefault(s, {'fetch_KiB': 0.0, 'write_KiB': 0.0, 'launches': 0})
        e['fetch_KiB'] += f
        e['write_KiB'] += w
        e['launches'] += max(nf, nw)
    steps = next(per[k] for k in ('k_bk_factor', 'k_btd_finish', 'k_publish_status') if k in per)['launches']   # one per numeric factorisation
    if 'k_transpose_in' in per:
        tr = per['k_transpose_in']
        known_read_KiB = steps * batch * (raw_entries + n) * 8 / 1024.0
        calib = known_read_KiB / tr['fetch_KiB']
    else:
        # the interface path on native device vectors launches no transposition any more: the read-side factor measured
        # on k_transpose_in in every earlier pass of this code base (profiles/r01_*/, r02_interface/: 1.99) is used
        calib = float(sys.argv[8]) if len(sys.argv) > 8 else 1.99
        note += ' (no k_transpose_in in this run: FETCH_SIZE calibration %.2f taken from profiles/r02_interface)' % calib
    kernels, phases = {}, {}
    for s, e in sorted(per.items()):
        if not s.startswith('k_'):
            continue
        rd = e['fetch_KiB'] * calib * 1024.0
        wr = e['write_KiB'] * 1024.0
        kernels[s] = {'launches_per_step': e['launches'] / steps,
                      'hbm_read_bytes_per_launch': rd / e['launches'], 'hbm_write_bytes_per_launch': wr / e['launches']}
        ph = next((p for pat, p in PHASE_OF if re.search(pat, s)), None)
        if s == 'k_transpose_in':
            # one launch per step belongs to the value upload (assemble), one to the forward solve
            share = raw_entries / float(raw_entries + n)
            for p, sh in (('assemble', share), ('fwd_levels', 1.0 - share)):
                q = phases.setdefault(p, {'hbm_bytes_per_step': 0.0, 'launches_per_step': 0.0})
                q['hbm_bytes_per_step'] += (rd + wr) * sh / steps
                q['launches_per_step'] += 1.0
            continue
        if ph:
            q = phases.setdefault(ph, {'hbm_bytes_per_step': 0.0, 'launches_per_step': 0.0})
            q['hbm_bytes_per_step'] += (rd + wr) / steps
            q['launches_per_step'] += e['launches'] / steps
    total = sum(p['hbm_bytes_per_step'] for p in phases.values())
    out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of bench.py '
                     '(C3 unless the note names another workload) on one MI355X; read side calibrated on k_transpose_in (known byte count) as '
                     'MI355X_MICROARCH.md section HBM prescribes. ' + note,
           'steps_in_pass': steps, 'fetch_calibration': calib, 'hbm_bytes_per_step_total': total,
           'kernel_source_sha1': kernel_source_sha1(),      # bench.py reports these counters only for the build they were taken on
           'kernels': kernels, 'phases': phases}
    json.dump(out, open(out_path, 'w'), indent=1)
    print('steps', steps, 'calibration', round(calib, 4), 'total GB/step', round(total / 1e9, 3))
    for p, v in sorted(phases.items()):
        print('  %-16s %8.3f GB/step  %5.1f launches' % (p, v['hbm_bytes_per_step'] / 1e9, v['launches_per_step']))


if __name__ == '__main__':
    main()
