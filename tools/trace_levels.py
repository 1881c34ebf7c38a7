"""Per-launch durations of one iteration from a rocprofv3 kernel trace csv (diagnostic)."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')) for r in rows]
# find the last full iteration: from the last k_transpose_in whose successor is a gather kernel
starts = [i for i, n in enumerate(names[:-1]) if n.startswith('k_transpose_in') and names[i + 1].startswith('k_gather')]
if len(starts) < 2:     # device-resident sources: an iteration starts at the first bottom-level gather after a back solve
    starts = [i for i, n in enumerate(names[:-1]) if n.startswith('k_gather_level_lean') and not names[i - 1].startswith('k_gather')]
i0 = starts[-2]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for i in range(i0, min(len(rows), i0 + int(sys.argv[2]) if len(sys.argv) > 2 else i0 + 140)):
    r = rows[i]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f us  gap %6.1f  dur %7.1f  grid %-8s %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
          r.get('Grid_Size_X', '?') + 'x' + r.get('Grid_Size_Y', '?'), names[i][:70]))
    prev_end = e
