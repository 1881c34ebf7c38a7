#!/bin/bash
# Run ON THE GPU BOX: average duration of the kernels whose name matches PATTERN for several builds of the kernel library.
# usage: tools/kstat_kernels.sh "base v1 v2" PATTERN [bench flags...]      ("base" = product build, else PP_LIB_VARIANT)
variants=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $variants; do
  if [ $v = base ]; then unset PP_LIB_VARIANT; else export PP_LIB_VARIANT=$v; fi
  rm -rf gpurun_out/ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$v -- python3 bench.py --steps 10 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-boundary "$@" > gpurun_out/ks_$v.json 2>/dev/null
  echo "== $v"
  python3 - gpurun_out/ks_$v "$pat" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Name']):
        print('  %-60s calls %5s avg %8.2f us' % (r['Name'].replace('(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  find gpurun_out/ks_$v -name '*kernel_trace.csv' -delete
done
