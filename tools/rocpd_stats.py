"""Per-kernel statistics and (optionally) the launch timeline of a window from a rocprofv3 sqlite result (rocpd):
python tools/rocpd_stats.py results.db [--timeline first_index count] [--match substring]"""
import re
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = cur.execute('select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, s.arch_vgpr_count, s.sgpr_count '
                       'from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()
    return rows


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    name = name.replace('(anonymous namespace)::', '')
    return name[:70]


def main():
    rows = load(sys.argv[1])
    if '--timeline' in sys.argv:
        i = sys.argv.index('--timeline')
        a, n = int(sys.argv[i + 1]), int(sys.argv[i + 2])
        t0 = rows[a][1]
        prev_end = t0
        for r in rows[a:a + n]:
            print('%9.1f us  +%6.1f gap  %7.1f us  grid %8d wg %4d  %s' % ((r[1] - t0) / 1e3, (r[1] - prev_end) / 1e3,
                                                                          (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), r[4], short(r[0])))
            prev_end = r[2]
        return
    match = sys.argv[sys.argv.index('--match') + 1] if '--match' in sys.argv else None
    agg = {}
    for r in rows:
        k = short(r[0])
        if match and match not in k:
            continue
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, r[5], r[6]])
        d = (r[2] - r[1]) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('%-72s %7s %10s %8s %8s %8s %5s %5s' % ('kernel', 'calls', 'total us', 'avg', 'min', 'max', 'vgpr', 'sgpr'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-72s %7d %10.1f %8.2f %8.2f %8.2f %5d %5d' % (k, a[0], a[1], a[1] / a[0], a[2], a[3], a[4], a[5]))
    print('total %.1f us over %d launches' % (tot, sum(a[0] for a in agg.values())))


if __name__ == '__main__':
    main()
