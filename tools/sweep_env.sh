#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   (one bench run per value of the environment variable VAR)
var=$1; shift
for t in "$@"; do
  out=$(env $var="$t" timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-boundary 2>/dev/null)
  python - "$var=$t" "$out" <<'PY'
import sys, json
t, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    print('%-40s %7.1f it/s ok=%s' % (t, d['value'], d['correct']), {k: round(v['ms_per_step'], 3) for k, v in d['phases'].items()})
except Exception as e:
    print(t, 'FAILED', out[-200:])
PY
done
