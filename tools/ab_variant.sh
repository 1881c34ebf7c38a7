#!/bin/bash
# A/B of alternative builds of the kernel library (csrc/libparapint_hip_<name>.so, selected with PP_LIB_VARIANT) by
# per-kernel rocprof durations; run on the GPU box: tools/ab_variant.sh "base v1 v2" [bench args]   ("base" = product build)
variants=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in $variants; do
  if [ $v = base ]; then unset PP_LIB_VARIANT; else export PP_LIB_VARIANT=$v; fi
  rm -rf gpurun_out/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -- python3 bench.py --steps 20 --warmup 3 --profile-steps 1 --no-cpu-baseline --no-boundary "$@" > gpurun_out/ab_$v.json 2>/dev/null
  echo "== $v (round $round)"
  python3 - gpurun_out/ab_$v gpurun_out/ab_$v.json <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_gather' in n or 'k_scale_level' in n or 'k_panel' in n:
        tot += float(r['TotalDurationNs'])
        print('  %-58s calls %4s avg %8.2f us  total %9.1f us' % (n.replace('(anonymous namespace)::', '')[:58], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print('  factor kernels total %.1f us; bench %.1f it/s, factor_levels %.4f ms' % (tot / 1e3, d['value'], d['phases']['factor_levels']['ms_per_step']))
except Exception as e:
    print('  (no bench line: %s)' % e)
PY
done
done
