#!/bin/bash
# usage: tools/sweep_blocks.sh b1 b2 ...   (one bench run per block count: the per-rank share at P = 1024 / b ranks)
for b in "$@"; do
  out=$(timeout -k 10 120 python bench.py --blocks $b --steps 30 --warmup 5 --no-cpu-baseline --no-boundary 2>/dev/null)
  python - "blocks=$b" "$out" <<'PY'
import sys, json
t, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    print('%-14s %7.1f it/s %6.3f ms ok=%s' % (t, d['value'], d['ms_per_step'], d['correct']), {k: round(v['ms_per_step'], 3) for k, v in d['phases'].items()})
except Exception as e:
    print(t, 'FAILED', out[-200:])
PY
done
