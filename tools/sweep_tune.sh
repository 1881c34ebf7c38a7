#!/bin/bash
# usage: tools/sweep_tune.sh "k=v,k=v" "k=v" ...   (one bench run per PP_PLAN_TUNE setting)
for t in "$@"; do
  out=$(PP_PLAN_TUNE="$t" timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-boundary 2>/dev/null)
  python - "$t" "$out" <<'PY'
import sys, json
t, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    print('%-60s %7.1f it/s ok=%s' % (t, d['value'], d['correct']), {k: round(v['ms_per_step'], 3) for k, v in d['phases'].items() if k in ('factor_levels', 'fwd_levels', 'bwd_levels')})
except Exception as e:
    print(t, 'FAILED', out[-200:])
PY
done
