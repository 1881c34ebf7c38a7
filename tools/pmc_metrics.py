"""Per-(kernel, grid size) averages of the counters collected by tools/pmc_metrics.sh (diagnostic)."""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
order = {}
for f in sorted(glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
        if not name.startswith('k_'):
            continue
        key = (name, int(r['Grid_Size']))
        order.setdefault(key, len(order))
        a = acc[key][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
counters = sorted({c for v in acc.values() for c in v})
print('\t'.join(['kernel', 'grid'] + counters))
for key in sorted(acc, key=lambda k: order[k]):
    row = acc[key]
    print('\t'.join([key[0], str(key[1])] + ['%.5g' % (row[c][0] / row[c][1]) if c in row else '-' for c in counters]))
