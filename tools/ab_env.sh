#!/bin/bash
# A/B of an environment switch by per-kernel rocprof durations (run on the GPU box): tools/ab_env.sh VAR [bench args]
var=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in off on off on; do
  if [ $mode = on ]; then export $var=1; else unset $var; fi
  rm -rf gpurun_out/ab_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$mode -- python3 bench.py --steps 20 --warmup 3 --profile-steps 1 --no-cpu-baseline --no-boundary "$@" > gpurun_out/ab_$mode.json 2>/dev/null
  echo "== $var $mode"
  python3 - gpurun_out/ab_$mode <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_gather_level' in n or 'k_scale_level' in n:
        print('  %-58s calls %4s avg %8.2f us' % (n.replace('(anonymous namespace)::', '')[:58], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
