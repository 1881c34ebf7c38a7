#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace stats of the default bench; prints the per-kernel table.
# usage: tools/kstats.sh <tag> [bench flags...]
set -o pipefail
tag=${1:-kstats}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-boundary "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
cp $f $out/kernel_stats.csv
find $out -name '*kernel_trace.csv' -size +8M -delete
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print('%-60s calls %6s  avg %9.1f us  total %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3, r['Percentage']))
PY
