#!/bin/bash
# Run ON THE GPU BOX: the default bench under a list of PP_PLAN_TUNE settings (plan.hpp:PlanOptions), one line each.
# usage: tools/tune_sweep.sh "k=v,k=v" "k=v" ... ("" = defaults)    [BENCH_ARGS env: extra bench flags]
for t in "$@"; do
  if [ -z "$t" ]; then unset PP_PLAN_TUNE; else export PP_PLAN_TUNE="$t"; fi
  python3 bench.py --no-cpu-baseline --no-boundary --steps 30 --warmup 5 $BENCH_ARGS > gpurun_out/tune.json 2> gpurun_out/tune.err || { echo "FAILED $t"; tail -2 gpurun_out/tune.err; continue; }
  python3 - "$t" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/tune.json').read().strip().splitlines()[-1])
ph = d['phases']
print('%-48s %8.1f it/s %7.4f ms  factor %.4f (%d launches) fwd %.4f bwd %.4f  correct %s' % (
    sys.argv[1] or '(defaults)', d['value'], d['ms_per_step'], ph['factor_levels']['ms_per_step'], ph['factor_levels']['launches_per_step'],
    ph['fwd_levels']['ms_per_step'], ph['bwd_levels']['ms_per_step'], d['correct']))
PY
done
