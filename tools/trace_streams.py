"""Per-queue timeline of the last full step of a bench run from a rocprofv3 kernel-trace csv (diagnostic): for workloads
whose pattern groups run on streams of their own (C4) -- which queue holds the critical path, and which kernels fill it.
usage: trace_streams.py <kernel_trace.csv> [max lines per queue [step index]]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
short = lambda n: re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))
names = [short(r['Kernel_Name']) for r in rows]
# a step starts at the first factor launch (gather) that follows a launch of the end of a back-solve (backward sweep, hand-over copy, check) in start order
starts = [i for i in range(1, len(rows)) if names[i].startswith('k_gather') and not names[i].startswith('k_gather_xc') and (names[i - 1].startswith(('k_bwd', 'k_residual', '__amd_rocclr_copy', 'k_transpose_out', 'k_copy')))]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2       # (third argument: which step of the run, in start order; default the last full one)
a, b = (starts[which], starts[which + 1] if which + 1 != 0 and which + 1 < len(starts) else starts[-1]) if len(starts) > 1 else (0, len(rows))
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
qkey = 'Queue_Id' if 'Queue_Id' in step[0] else 'Stream_Id'
byq = collections.defaultdict(list)
for r in step:
    byq[r.get('Stream_Id', '?') + '/' + r.get(qkey, '?')].append(r)
print('step: %d launches, %.1f us from first start to last end' % (len(step), (max(int(r['End_Timestamp']) for r in step) - t0) / 1e3))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print('\n== stream/queue %s: %d launches, busy %.1f us, from %.1f to %.1f us' % (
        q, len(rs), busy / 1e3, (int(rs[0]['Start_Timestamp']) - t0) / 1e3, (int(rs[-1]['End_Timestamp']) - t0) / 1e3))
    agg = collections.OrderedDict()
    for r in rs:
        k = short(r['Kernel_Name'])
        d = agg.setdefault(k, [0, 0, 1 << 62, 0])
        d[0] += 1
        d[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        d[2] = min(d[2], int(r['Start_Timestamp']) - t0)
        d[3] = max(d[3], int(r['End_Timestamp']) - t0)
    for k, d in agg.items():
        print('   %-44s x%4d  sum %8.1f us  avg %6.1f  window %8.1f .. %8.1f' % (k[:44], d[0], d[1] / 1e3, d[1] / 1e3 / d[0], d[2] / 1e3, d[3] / 1e3))
    prev = int(rs[0]['Start_Timestamp'])
    for r in rs[:limit]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%9.1f  gap %6.1f  dur %6.1f  grid %-7s wg %-5s %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', '?'),
                                                                    r.get('Workgroup_Size_X', '?'), short(r['Kernel_Name'])[:60]))
        prev = e
